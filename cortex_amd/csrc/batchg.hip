// batchg.hip — batched exact search for the row widths batch2_kernel (batch.hip) has no instance for.
//
// VectorIndex::search_batch (vector/index.rs:390-410) is dimension-agnostic and the reference ships 1024-d embeddings
// (BGE-large, embedding.rs:43-50: BASELINE config 5's width).  batch2_kernel keeps 64 queries, split into bf16 hi + lo,
// in the consumer waves' REGISTERS — dim/4 VGPRs per wave: 192 at 768-d, 256 at 1024-d, the whole budget of a wave at two
// waves per SIMD — so round 1 sent every other width to one scan per query: B times the HBM traffic.  Here the roles of
// the operands swap: the queries (pre-split once per call into the MFMA B-operand layout, batchg_split_queries_kernel)
// are staged K-block by K-block through LDS, and the ROWS go straight from HBM into registers, are split there
// (hi = bf16(a), lo = bf16(a - hi): the same three-product scheme as batch.hip, |cos error| <= 1e-6) and multiplied
// as the MFMA A operand.  Any dim % 128 == 0 up to 4096; no split store, no shadow: the f32 rows are read once per
// 64 queries.
//
//   block = 4 waves, TWO blocks per CU (each runs into its own barriers), row tile = 128 rows (wave w: rows 32 w .. + 31 =
//     two 16-row A fragments: a query fragment read from LDS feeds six MFMAs — LDS bandwidth, 48 KiB per 8 KiB of rows in
//     the first version of this kernel, was what its arithmetic side was bound by)
//   K-block = 64 k = 2 MFMA steps of 32; per K-block and wave: 8 x global_load_dwordx4 of one KiB each (4 rows x 256 B),
//     handed through the wave's own 8 KiB LDS region into MFMA operand layout, split there, 4 query groups x 2 fragments
//     x 3 mfma_f32_16x16x32_bf16 per step
//   queries: K-block kb+1 is fetched into registers while K-block kb is computed, written to the other LDS buffer at
//     the top of the next iteration; ONE raw barrier per K-block
//   the main loop holds no vector memory operation but the row / query fetches, and every one of them is unconditional:
//     vmcnt is one in-order queue, so a load the epilogue waits for (a norm, |q|^2) or a fetch the compiler cannot count
//     (inside an `if`) turns its waits into vmcnt(0) — a drain of the two K-blocks of rows in flight
//   epilogue, dense mode: cosines [query][row] to HBM, launch_dense_topk (scan.hip) then takes the top k per query.  Those
//     4 bytes per row and query are 6 % of the traffic at 1024-d and cost the row stream a FIFTH of its rate (64-byte
//     pieces into 64 distant streams; profiles/r02/tuning.md section 4), and they are read back once more
//   epilogue, filter mode (the default from 262,144 rows; a row filter is applied to the sample and to the candidates): a first dense pass over 1 row tile in
//     64 (32 for k > 32) gives every query a bound — the k-th best score of the sample; the k-th best of all rows can only
//     be higher — and the pass over all rows writes only the (key, cosine) of the rows that reach it, ~64 k per query, into
//     per-block lists (slots from LDS counters: a returning global atomic would be one more load in the vmcnt queue) that
//     the radix merge folds.  A pair is divided out only if dot >= (bound - 1e-4) |q| |row|.  A block list that runs over
//     sets a flag; the dense pass, its top-k and merge are launched behind it with run_if = that flag and return at
//     once when it is 0 — exact whatever the data, no host round trip
// Bound: HBM.  Algorithmic bytes per launch = n_rows * dim * 4.
#include <algorithm>

#include "kernels.hpp"
#include "topk.hpp"

namespace cx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

namespace bg {
constexpr int WAVES = 4;            // per block; two blocks per CU
constexpr int TILE_ROWS = 128;      // rows per block tile of the f32 instance: 4 waves x 32 (the bf16 instance: 4 x 64)
constexpr int KB = 64;              // k per K-block
constexpr int STEPS = KB / 32;      // MFMA steps per K-block
constexpr int NQ = 64;              // queries per pass
constexpr int STEP_BYTES = 2 * NQ * 4 * 16;   // one step's query image: hi [64][4 kq][8 bf16] | lo = 8 KiB
constexpr int KB_BYTES = STEPS * STEP_BYTES;  // 16 KiB
constexpr int SUB_BYTES = 32 * KB * 4;        // a wave's 32 rows x one K-block f32 = 8 KiB
constexpr int LDS_BYTES = 2 * KB_BYTES + WAVES * SUB_BYTES + 256;   // query images double buffered + one row region per wave + 64 counters = 64.25 KiB

__device__ inline void split4g(const f32x4 v, bf16x4_t &hi, bf16x4_t &lo) {   // as batch.hip's split4
    const uint32_t p01 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v.x, v.y}, bf16x2_t));
    const uint32_t p23 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v.z, v.w}, bf16x2_t));
    const f32x2 b01 = {__uint_as_float(p01 << 16), __uint_as_float(p01 & 0xFFFF0000u)};
    const f32x2 b23 = {__uint_as_float(p23 << 16), __uint_as_float(p23 & 0xFFFF0000u)};
    const f32x2 d01 = (f32x2){v.x, v.y} - b01, d23 = (f32x2){v.z, v.w} - b23;
    const uint32_t q01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(d01, bf16x2_t));
    const uint32_t q23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(d23, bf16x2_t));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(bf16x4_t, (u32x2){p01, p23});
    lo = __builtin_bit_cast(bf16x4_t, (u32x2){q01, q23});
}
}  // namespace bg

// queries [nq][dim] f32 -> qimg: per MFMA step s (32 k): hi image [64 queries][4 kq][8 bf16] | lo image, 8 KiB a step.
// Lane kq of the MFMA holds the k values {32 s + 4 kq .. + 3} and {32 s + 16 + 4 kq .. + 3} (the row side loads the
// same two float4 per lane: 64 contiguous bytes per row per load instruction); queries beyond nq are zero.
// Also |q|^2 per query.
// perm16 (bf16 row stores): lane kq of the MFMA holds the 8 CONSECUTIVE k values 32 s + 8 kq .. + 7 — one 16-byte piece of a bf16 row.
__global__ __launch_bounds__(256) void batchg_split_queries_kernel(const float *queries, uint32_t nq, uint32_t dim, char *qimg, float *qq, int perm16) {
    const uint32_t n_steps = dim / 32u;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_steps * bg::NQ * 4u; t += gridDim.x * blockDim.x) {
        const uint32_t s = t / (bg::NQ * 4u), j = (t / 4u) % bg::NQ, kq = t % 4u;
        f32x4 v0 = {0.0f, 0.0f, 0.0f, 0.0f}, v1 = v0;
        if (j < nq) {
            const f32x4 *q4 = reinterpret_cast<const f32x4 *>(queries + (size_t)j * dim + 32u * s);
            v0 = perm16 ? q4[2u * kq] : q4[kq];
            v1 = perm16 ? q4[2u * kq + 1u] : q4[4u + kq];
        }
        bf16x4_t h0, l0, h1, l1;
        bg::split4g(v0, h0, l0);
        bg::split4g(v1, h1, l1);
        char *base = qimg + (size_t)s * bg::STEP_BYTES + (j * 4u + kq) * 16u;
        *reinterpret_cast<bf16x8_t *>(base) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8_t *>(base + bg::STEP_BYTES / 2) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    // |q|^2: one wave per query (blocks of 4 waves stride over the queries)
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t j = wave; j < bg::NQ; j += n_waves) {
        float sacc = 0.0f;
        if (j < nq)
            for (uint32_t c = lane; c < dim; c += 64u) { const float x = queries[(size_t)j * dim + c]; sacc = fmaf(x, x, sacc); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
        if (lane == 0) qq[j] = sacc;
    }
}

struct BatchGArgs {
    const float *rows;     // [n_rows (+ 256 readable)][dim]; null for a bf16 store
    const uint16_t *rows16;   // bf16 store: the rows ARE the A operand (no split, two products per pair)
    const float *norms;    // |row|^2 (cx_index::d_norms)
    const char *qimg;      // split query images, dim / 32 steps of 8 KiB
    const float *qq;       // [64] |q|^2
    float *dense;          // dense mode: [64][stride] cosines out
    uint32_t n_rows, dim, nq, stride;
    uint32_t tile_step;    // dense mode over a sample: block tile t reads row tile t * tile_step, writes dense column tile t
    uint32_t n_tiles;      // tiles this launch walks
    // filter mode (tau_ord != null): instead of 4 bytes per row and query, only the (key, cosine) of the rows whose score
    // reaches the query's bound, into this block's list of the query
    const uint32_t *tau_ord;   // [64] score_ord of the bound (0: everything passes)
    uint64_t *cand_keys;       // [64][gridDim.x][cb]
    float *cand_sims;
    uint32_t *cand_counts;     // [64][gridDim.x] out: entries written to each list
    uint32_t *overflow;        // [1] set when a block's list of some query was too short
    uint32_t cb;
    DevFilter flt;             // filter mode: checked for the rows that reach the bound only
    const uint32_t *run_if;    // non-null: the launch does nothing unless *run_if != 0 (the exact fallback)
};

// PROBE: 0 = the product; 1 = loads only (no LDS transpose, no split, no MFMA); 2 = no row loads (MFMAs on stale data);
// 3 = rows only (no barriers, no query staging)
template <int PROBE, bool FILTER, bool R16>
__global__ __launch_bounds__(256, 2) void batchg_kernel(const BatchGArgs a) {
    using namespace bg;
    // rows per wave: an f32 K-block of a row is 256 B, a bf16 one 128 B — the bf16 instance gives every wave twice the rows
    // (four A fragments), so that a K-block is the same 8 KiB of rows per wave and a query fragment feeds eight MFMAs
    constexpr int NF = R16 ? 4 : 2;                 // 16-row A fragments per wave
    constexpr uint32_t RPW = 16u * NF;              // rows per wave
    constexpr uint32_t TR = WAVES * RPW;            // rows per block tile: 128 (f32) / 256 (bf16)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (a.run_if && *a.run_if == 0u) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t j = lane & 15u, kq = lane >> 4;
    const uint32_t n_tiles = a.n_tiles;
    const uint32_t n_kb = a.dim / KB;
    if (blockIdx.x >= n_tiles) return;
    const uint32_t my_tiles = (n_tiles - 1u - blockIdx.x) / gridDim.x + 1u;
    const uint32_t total_kb = my_tiles * n_kb;   // K-blocks this block walks: the query images cycle once per tile
    char *Qs = smem;                                              // [2][KB_BYTES] query images, double buffered
    char *Rw = smem + 2 * KB_BYTES + wave * SUB_BYTES;            // this wave's 32 rows x 64 k f32, pieces swizzled
    uint32_t *Cnt = reinterpret_cast<uint32_t *>(smem + 2 * KB_BYTES + WAVES * SUB_BYTES);   // filter mode: entries in this block's list of each query
    if (FILTER && tid < 64u) Cnt[tid] = 0u;

    auto tile_barrier = [&]() {   // raw barrier: __syncthreads() would drain the rows in flight (vmcnt(0))
        if constexpr (PROBE == 3) return;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // query staging: a K-block image is 16 KiB = 4 x 16 B per thread
    f32x4 qr[4];
    auto q_fetch = [&](uint32_t it_) {   // it_: index in this block's K-block sequence
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.qimg + (size_t)(it_ % n_kb) * KB_BYTES) + tid;
        if constexpr (PROBE == 3) return;
#pragma unroll
        for (int e = 0; e < 4; e++) qr[e] = src[e * 256];
    };
    auto q_store = [&](uint32_t buf) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(Qs + buf * KB_BYTES) + tid;
        if constexpr (PROBE == 3) return;
#pragma unroll
        for (int e = 0; e < 4; e++) dst[e * 256] = qr[e];
    };

    // Row side.  A wave owns 32 rows (two MFMA A fragments: every query fragment read from LDS feeds six MFMAs).  A K-block
    // of them = 32 rows x 256 B = 8 load instructions of ONE KiB each: lane l reads piece l % 16 (16 B) of row
    // 4 i + l / 16.  The registers go to the wave's own 8 KiB LDS region and come back in MFMA layout — same wave, LDS
    // executes a wave's instructions in order: no barrier.  Piece p of row r sits at p ^ (r & 15): the 16 lanes of a
    // read group (one piece index, 16 rows) then hit 16 different 16-byte bank groups; a write group is the 16 pieces
    // of one row, permuted inside its 256 bytes.
    const uint32_t lrow = lane >> 4, lpiece = lane & 15u;
    f32x4 acc[NF][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int f = 0; f < NF; f++)
#pragma unroll
            for (int g = 0; g < 4; g++) acc[f][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    const uint32_t lane_off = lrow * a.dim * 4u + lpiece * 16u;
    f32x4 xa[8], xb[8];   // two K-blocks of this wave's rows in flight (16 KiB per wave, 128 KiB per CU)
    // bf16 store: a K-block of a row is 128 B = 8 pieces of 8 elements; lane l reads piece l % 8 of row 8 i + l / 8, eight
    // load instructions (64 rows) per K-block; piece p of row r sits at p ^ ((r >> 1) & 7) (rows are 128 B apart: two rows per 256 B
    // of banks)
    const uint32_t lrow16 = lane >> 3, lpiece16 = lane & 7u;
    const uint32_t lane_off16 = lrow16 * a.dim * 2u + lpiece16 * 16u;
    auto r_fetch = [&](f32x4 (&dst)[8], uint32_t tile, uint32_t kb) {   // kb: K-block inside the row
        if constexpr (R16) {
            const char *base = reinterpret_cast<const char *>(a.rows16) + ((size_t)tile * a.tile_step * TR + wave * RPW) * a.dim * 2u + (size_t)kb * (KB * 2u);
#pragma unroll
            for (int i = 0; i < 8; i++) dst[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + (size_t)i * 8u * a.dim * 2u + lane_off16));
            return;
        }
        const char *base = reinterpret_cast<const char *>(a.rows) + ((size_t)tile * a.tile_step * TR + wave * RPW) * a.dim * 4u + (size_t)kb * (KB * 4u);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if constexpr (PROBE == 2) { asm volatile("" : "+v"(dst[i])); continue; }
            dst[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + (size_t)i * 4u * a.dim * 4u + lane_off));
        }
    };
    auto lds_put = [&](const f32x4 (&src)[8]) {
        if constexpr (R16) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t r = 8u * (uint32_t)i + lrow16;
                *reinterpret_cast<f32x4 *>(Rw + r * 128u + ((lpiece16 ^ ((r >> 1) & 7u)) << 4)) = src[i];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t r = 4u * (uint32_t)i + lrow;
            *reinterpret_cast<f32x4 *>(Rw + r * 256u + ((lpiece ^ (r & 15u)) << 4)) = src[i];
        }
    };
    // step s (of the K-block's two): lane (j, kq) needs k = 32 s + 4 kq .. + 3 and 32 s + 16 + 4 kq .. + 3 of rows j
    // and 16 + j: pieces 8 s + kq and 8 s + 4 + kq
    auto compute_kb = [&](uint32_t buf) {
        const char *Q = Qs + buf * KB_BYTES + (j * 4u + kq) * 16u;
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            s16x8 ah[NF], al[2];
            if constexpr (R16) {
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    const uint32_t row = 16u * f + j;
                    ah[f] = *reinterpret_cast<const s16x8 *>(Rw + row * 128u + (((4u * s + kq) ^ ((row >> 1) & 7u)) << 4));
                }
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const s16x8 qh = *reinterpret_cast<const s16x8 *>(Q + s * STEP_BYTES + g * 1024);
                    const s16x8 ql = *reinterpret_cast<const s16x8 *>(Q + s * STEP_BYTES + STEP_BYTES / 2 + g * 1024);
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], ql, acc[f][g], 0, 0, 0);   // small term first
                        acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], qh, acc[f][g], 0, 0, 0);
                    }
                }
                continue;
            }
#pragma unroll
            for (int f = 0; f < 2; f++) {
                const char *R = Rw + (16u * f + j) * 256u;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(R + (((8u * s + kq) ^ j) << 4));
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(R + (((8u * s + 4u + kq) ^ j) << 4));
                bf16x4_t h0, l0, h1, l1;
                split4g(v0, h0, l0);
                split4g(v1, h1, l1);
                ah[f] = __builtin_bit_cast(s16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                al[f] = __builtin_bit_cast(s16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const s16x8 qh = *reinterpret_cast<const s16x8 *>(Q + s * STEP_BYTES + g * 1024);
                const s16x8 ql = *reinterpret_cast<const s16x8 *>(Q + s * STEP_BYTES + STEP_BYTES / 2 + g * 1024);
#pragma unroll
                for (int f = 0; f < 2; f++) {
                    acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[f], qh, acc[f][g], 0, 0, 0);   // small terms first
                    acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], ql, acc[f][g], 0, 0, 0);
                    acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], qh, acc[f][g], 0, 0, 0);
                }
            }
        }
    };
    auto consume = [&](const f32x4 (&src)[8], uint32_t buf) {
        if constexpr (PROBE == 1 || PROBE == 3) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[0][i & 3] += src[i];
            return;
        }
        lds_put(src);
        compute_kb(buf);
    };
    // C layout: lane (j, kq) holds rows 4 kq + e (e = 0..3) of fragment f for query j of group g: four consecutive rows.
    // Nothing here may be a vector load: vmcnt is in order, so waiting for one would first drain the row loads of the next
    // two K-blocks — the whole prefetch, once per tile.  |q|^2 and the bounds sit in registers from the start, |row|^2
    // comes by scalar loads (a fragment's 16 rows are a uniform address; d_norms has a row tile (256 floats) of readable padding).
    // The square roots are hoisted (|q| once per kernel, |row| once per row instead of once per pair): the same IEEE values
    // as cosine_from_sums.  Filter mode divides only where the pair can reach the bound: score ~ dot / (|q| |row|) within
    // a few ulp, so dot < (bound - 1e-4) |q| |row| rules a pair out with two multiplies (a NaN or zero norm fails the
    // comparison and takes the exact path) — 99.9 % of the pairs at k = 10.
    float nqv[4], thrv[4];
    uint32_t tauv[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t q = (uint32_t)g * 16u + j;
        nqv[g] = sqrtf(q < a.nq ? a.qq[q] : 1.0f);
        tauv[g] = (FILTER && q < a.nq) ? a.tau_ord[q] : 0u;
        // no bound (0), NaN (1) or a sampled k-th best of 0.0 (2: the clamp of a non-positive cosine): every pair is a candidate.
        // With a bound of 0.0 a cut at dot < -1e-4 |q||r| dropped rows whose score also clamps to 0.0 — equal to the bound,
        // legitimate under (score desc, row asc) — so ties at 0 could resolve differently from the scan paths (round-2 ADVICE)
        thrv[g] = tauv[g] > 2u ? __uint_as_float(tauv[g] - 2u) - 1e-4f : -__builtin_inff();
    }
    const __attribute__((address_space(4))) float *norms_c = (const __attribute__((address_space(4))) float *)a.norms;
    auto epilogue = [&](uint32_t tile) {
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const uint32_t in_tile = wave * RPW + 16u * f + 4u * kq;                 // first of this lane's four rows, inside the tile
            const uint32_t w0 = tile * a.tile_step * TR + wave * RPW + 16u * f, r0 = w0 + 4u * kq;   // ... in the store
            float tn[16];
#pragma unroll
            for (int e = 0; e < 16; e++) tn[e] = norms_c[(size_t)w0 + e];
            f32x4 rr = {tn[0], tn[1], tn[2], tn[3]};
#pragma unroll
            for (int c = 1; c < 4; c++)
                if (kq == (uint32_t)c) rr = f32x4{tn[4 * c], tn[4 * c + 1], tn[4 * c + 2], tn[4 * c + 3]};
            if (r0 >= a.n_rows) continue;
            f32x4 nr;
#pragma unroll
            for (int e = 0; e < 4; e++) nr[e] = sqrtf(rr[e]);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t q = (uint32_t)g * 16u + j;
                if (q >= a.nq) continue;
                if constexpr (FILTER) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (acc[f][g][e] < thrv[g] * (nqv[g] * nr[e])) continue;
                        const float sim = cosine_from_norms(acc[f][g][e], nqv[g], nr[e]);
                        const float score = score_of(distance_of(sim));
                        if (score_ord(score) >= tauv[g] && r0 + (uint32_t)e < a.n_rows && row_passes(a.flt, r0 + (uint32_t)e)) {
                            const uint32_t slot = atomicAdd(&Cnt[q], 1u);   // LDS: a returning GLOBAL atomic would drain the row prefetch
                            if (slot < a.cb) {
                                const size_t at = ((size_t)q * gridDim.x + blockIdx.x) * a.cb + slot;
                                a.cand_keys[at] = make_key(score, r0 + (uint32_t)e);
                                a.cand_sims[at] = sim;
                            }
                        }
                    }
                } else {
                    f32x4 c;
#pragma unroll
                    for (int e = 0; e < 4; e++) c[e] = cosine_from_norms(acc[f][g][e], nqv[g], nr[e]);
                    float *dst = a.dense + (size_t)q * a.stride + ((size_t)tile * TR + in_tile);   // dense column: the tile as this launch counts it
                    if (r0 + 3u < a.n_rows) *reinterpret_cast<f32x4 *>(dst) = c;
                    else { for (uint32_t e = 0; e < 4u; e++) if (r0 + e < a.n_rows) dst[e] = c[e]; }
                }
            }
        }
    };

    // Pipeline over the block's K-blocks, tile by tile, two K-blocks per loop iteration (dim % 256 == 0).  K-block `it`
    // reads its query image from LDS buffer it & 1; the image of it + 1 (fetched during it - 1) is written to the other
    // buffer at the top of the K-block and the image of it + 2 requested.  Rows: two register sets = two K-blocks; a set
    // is re-requested for the K-block TWO ahead (which may belong to the next tile) the moment it has been handed to LDS.
    // One barrier per K-block, for the query hand-over only.
    // Every fetch below is UNCONDITIONAL (past the end the address is clamped to data already read): with a load that may
    // or may not have been issued the compiler cannot count vmcnt and falls back to vmcnt(0) — a drain of the prefetch.
    uint32_t tile = blockIdx.x, it = 0;
    zero_acc();
    q_fetch(0);
    q_store(0);
    q_fetch(total_kb > 1u ? 1u : 0u);
    r_fetch(xa, tile, 0);
    r_fetch(xb, tile, 1);
    tile_barrier();
    // where K-block kbl + 2 of the current tile lives: (tile, kbl + 2) or the next tile's (kbl + 2 - n_kb); past the end: (tile, kbl)
    auto ahead = [&](uint32_t t, uint32_t kbl, uint32_t &tl, uint32_t &kb2) {
        kb2 = kbl + 2u;
        tl = tile;
        if (kb2 >= n_kb) {
            if (t + 1u < my_tiles) { kb2 -= n_kb; tl = tile + gridDim.x; }
            else kb2 = kbl;
        }
    };
    const uint32_t last_kb = total_kb - 1u;
    for (uint32_t t = 0; t < my_tiles; t++) {
        for (uint32_t kbl = 0; kbl < n_kb; kbl += 2u, it += 2u) {
            uint32_t tl, kb2;
            // even K-block (buffer 0): xa
            q_store(1);
            q_fetch(std::min(it + 2u, last_kb));
            ahead(t, kbl, tl, kb2);
            consume(xa, 0);
            r_fetch(xa, tl, kb2);
            tile_barrier();
            // odd K-block (buffer 1): xb
            q_store(0);
            q_fetch(std::min(it + 3u, last_kb));
            ahead(t, kbl + 1u, tl, kb2);
            consume(xb, 1);
            r_fetch(xb, tl, kb2);
            if (kbl + 2u == n_kb) {
                epilogue(tile);
                tile += gridDim.x;
                zero_acc();
            }
            tile_barrier();
        }
    }
    if constexpr (FILTER) {
        if (tid < 64u) {   // the last tile_barrier() ordered the counters
            const uint32_t c = Cnt[tid];
            a.cand_counts[(size_t)tid * gridDim.x + blockIdx.x] = c < a.cb ? c : a.cb;
            if (c > a.cb) *a.overflow = 1u;
        }
    }
}

bool batchg_supported(uint32_t dim, uint32_t k) { return dim % (2 * bg::KB) == 0 && dim <= 4096 && k >= 1 && k <= TOPK_MAX; }
size_t batchg_qimg_bytes(uint32_t dim) { return (size_t)(dim / 32u) * bg::STEP_BYTES; }

uint32_t batchg_tile_rows(bool rows16) { return rows16 ? 2u * bg::TILE_ROWS : (uint32_t)bg::TILE_ROWS; }
uint32_t batchg_grid(uint32_t n_rows, bool rows16) {
    const uint32_t tr = batchg_tile_rows(rows16), n_tiles = (n_rows + tr - 1) / tr;
    return std::min<uint32_t>(n_tiles, 2u * device_cus());
}
uint32_t batchg_sample_rows(uint32_t n_rows, uint32_t tile_step, uint32_t *n_tiles_out, bool rows16) {
    if (n_tiles_out) *n_tiles_out = 0;
    if (!n_rows || !tile_step) return 0;
    const uint32_t tr = batchg_tile_rows(rows16);
    const uint32_t n_tiles = (n_rows + tr - 1) / tr, ns = (n_tiles + tile_step - 1) / tile_step;
    const uint32_t last_phys = (ns - 1u) * tile_step * tr;
    if (n_tiles_out) *n_tiles_out = ns;
    return (ns - 1u) * tr + std::min<uint32_t>(tr, n_rows - last_phys);
}

int launch_batchg_split(const float *d_queries, uint32_t nq, uint32_t dim, char *d_qimg, float *d_qq, hipStream_t stream, bool rows16) {
    if (!batchg_supported(dim, 1) || nq == 0 || nq > bg::NQ) return set_err(CX_ERR_VALIDATION, "batchg: dim %u / %u queries not supported", dim, nq);
    hipLaunchKernelGGL(batchg_split_queries_kernel, dim3(16), dim3(256), 0, stream, d_queries, nq, dim, d_qimg, d_qq, rows16 ? 1 : 0);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// nq <= 64 queries (already split: launch_batchg_split) against the rows.
//   dense mode (f == null): cosines [64][stride]; tile_step > 1 walks every tile_step-th row tile only and packs the
//     columns (the bound-finding sample of the filter mode); run_if: see BatchGArgs
//   filter mode: candidates of each query into cand_keys / cand_sims [64][batchg_grid(n_rows)][cb]
int launch_batchg_pass(const float *rows, const float *norms, uint32_t n_rows, uint32_t dim, uint32_t nq, const char *d_qimg, const float *d_qq,
                       float *d_dense, uint32_t stride, uint32_t tile_step, const BatchGFilter *f, const uint32_t *run_if, hipStream_t stream,
                       const uint16_t *rows16) {
    using namespace bg;
    if (!batchg_supported(dim, 1) || nq == 0 || nq > NQ || tile_step == 0) return set_err(CX_ERR_VALIDATION, "batchg: dim %u / %u queries not supported", dim, nq);
    if (!n_rows) return CX_OK;
    static std::atomic<uint64_t> attr_devices{0};
    if (first_use_on_device(attr_devices)) {
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchg_kernel<0, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchg_kernel<0, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchg_kernel<1, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchg_kernel<2, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchg_kernel<3, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchg_kernel<0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchg_kernel<0, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    }
    BatchGArgs a;
    memset(&a, 0, sizeof a);
    a.flt.trivial = 1;
    a.rows = rows; a.rows16 = rows16; a.norms = norms; a.qimg = d_qimg; a.qq = d_qq; a.dense = d_dense;
    a.n_rows = n_rows; a.dim = dim; a.nq = nq; a.stride = stride; a.tile_step = tile_step; a.run_if = run_if;
    const uint32_t tr = batchg_tile_rows(rows16 != nullptr), n_tiles = (n_rows + tr - 1) / tr;
    a.n_tiles = (n_tiles + tile_step - 1) / tile_step;
    uint32_t grid = std::min<uint32_t>(a.n_tiles, 2u * device_cus());
    if (f) {
        a.tau_ord = f->tau_ord; a.cand_keys = f->cand_keys; a.cand_sims = f->cand_sims; a.cand_counts = f->cand_counts; a.overflow = f->overflow; a.cb = f->cb; a.flt = f->flt;
        grid = batchg_grid(n_rows, rows16 != nullptr);   // the candidate lists are laid out for exactly this grid
        if (rows16) hipLaunchKernelGGL((batchg_kernel<0, true, true>), dim3(grid), dim3(256), LDS_BYTES, stream, a);
        else hipLaunchKernelGGL((batchg_kernel<0, true, false>), dim3(grid), dim3(256), LDS_BYTES, stream, a);
        CX_HIP(hipGetLastError());
        return CX_OK;
    }
    if (rows16) {
        hipLaunchKernelGGL((batchg_kernel<0, false, true>), dim3(grid), dim3(256), LDS_BYTES, stream, a);
        CX_HIP(hipGetLastError());
        return CX_OK;
    }
    static const int probe = getenv("CX_BATCHG_PROBE") ? atoi(getenv("CX_BATCHG_PROBE")) : 0;   // measurement arms, results invalid
    if (probe == 1) hipLaunchKernelGGL((batchg_kernel<1, false, false>), dim3(grid), dim3(256), LDS_BYTES, stream, a);        // loads only
    else if (probe == 2) hipLaunchKernelGGL((batchg_kernel<2, false, false>), dim3(grid), dim3(256), LDS_BYTES, stream, a);   // no row loads
    else if (probe == 3) hipLaunchKernelGGL((batchg_kernel<3, false, false>), dim3(grid), dim3(256), LDS_BYTES, stream, a);   // rows only: no barriers, no query staging
    else hipLaunchKernelGGL((batchg_kernel<0, false, false>), dim3(grid), dim3(256), LDS_BYTES, stream, a);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// the round-2 entry: split + one dense pass over every row
int launch_batchg_scores(const float *rows, const float *norms, uint32_t n_rows, uint32_t dim, const float *d_queries, uint32_t nq,
                         char *d_qimg, float *d_qq, float *d_dense, uint32_t stride, hipStream_t stream) {
    if (int rc = launch_batchg_split(d_queries, nq, dim, d_qimg, d_qq, stream, false)) return rc;
    return launch_batchg_pass(rows, norms, n_rows, dim, nq, d_qimg, d_qq, d_dense, stride, 1, nullptr, nullptr, stream, nullptr);
}

}  // namespace cx
