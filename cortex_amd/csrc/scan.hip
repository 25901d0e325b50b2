// scan.hip — the single-query cosine top-k scan (BASELINE config 2 / headline).
//
// Replaces HnswIndex::brute_force_search (vector/index.rs:259-294) and the
// per-pair EmbeddingPoint::distance (:169-179).  HBM-bound: the only traffic
// that matters is one pass over the f32 row store, N*dim*4 bytes per query.
//
// Shape of the kernel (gfx950, wave64):
//  - G lanes cooperate on one row (G = 64 for dim % 256 == 0, 32 for
//    dim % 128 == 0, 16 for dim % 64 == 0); every lane issues 16-byte loads
//    and a wave-level load instruction covers 64/G rows x G*16 contiguous
//    bytes each, so every 128-byte line is fetched whole and once.
//  - R row groups are loaded back to back before any arithmetic: R*dim/(4G)
//    independent 16-byte loads in flight per lane; with 8-16 waves per CU
//    that is far more than the ~32 KiB per CU the HBM latency needs.
//  - the query slice a lane needs never changes, so it sits in registers
//    (dim/(4G) float4s), read once from HBM; the row's sum of squares is
//    accumulated next to the dot product (the reference recomputes it per
//    pair too), so no norm array is read.
//  - dot / norms are reduced over the G lanes by a butterfly, the
//    reference's epilogue is applied one IEEE op at a time, and the score is
//    offered to the wave's register top-k list (topk.hpp).
//  - rows beyond 256 MiB are streamed with non-temporal loads: they cannot
//    stay in the 256 MiB Infinity Cache between queries anyway.
//  - per-block lists are merged by a second, tiny kernel (merge_kernel).
#include <algorithm>

#include "kernels.hpp"
#include "select.hpp"
#include "topk.hpp"

namespace cx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ inline f32x4 ld4(const f32x4 *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <int G>
__device__ inline float group_sum(float v) {
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Shared tail: merge the block's 4 wave lists into wave 0 and write the
// block's partial list.
template <int KS>
__device__ inline void block_merge_store(WaveTopK<KS> &top, uint32_t k, uint64_t *part_keys, float *part_sims) {
    __shared__ uint64_t sk[3][64 * KS];
    __shared__ float ss[3][64 * KS];
    const int wave = (int)(threadIdx.x >> 6);
    if (wave > 0) top.store(sk[wave - 1], ss[wave - 1]);
    __syncthreads();
    if (wave == 0) {
        const uint32_t lane = (uint32_t)lane_id();
        for (int w = 0; w < 3; w++) {
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const uint32_t i = (uint32_t)s * 64u + lane;
                const uint64_t kg = i < k ? sk[w][i] : 0ull;
                const float sg = i < k ? ss[w][i] : 0.0f;
                top.offer_lanes(kg, sg, [](uint32_t) { return true; });
            }
        }
        top.store(part_keys + (size_t)blockIdx.x * k, part_sims + (size_t)blockIdx.x * k);
    }
}

// MODE 0: top-k partial lists.  MODE 1: dense keys.
template <int D, int G, int R, int KS, bool NT, int MODE>
__global__ __launch_bounds__(256) void scan_kernel(const ScanArgs a) {
    constexpr int GPW = 64 / G;        // rows per wave-level load instruction
    constexpr int NJ = D / (4 * G);    // 16-byte loads per lane per row
    constexpr int RPW = R * GPW;       // rows per wave iteration
    static_assert(D % (4 * G) == 0, "dim must be a multiple of 4*G");
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int lig = lane % G;
    const int grp = lane / G;

    // query slice -> registers; |q|^2 with the same lane split
    f32x4 q[NJ];
    const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.query);
    float qq = 0.0f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        q[j] = q4[j * G + lig];
        qq += q[j].x * q[j].x + q[j].y * q[j].y + q[j].z * q[j].z + q[j].w * q[j].w;
    }
    qq = group_sum<G>(qq) + a.q_tail_sumsq;

    WaveTopK<KS> top;
    if constexpr (MODE == 0) top.init(a.k);

    const uint32_t n_rows = a.n_rows;
    const uint32_t n_tiles = (n_rows + RPW - 1) / RPW;
    const uint32_t stride = gridDim.x * 4u;
    for (uint32_t t = blockIdx.x * 4u + (uint32_t)wave; t < n_tiles; t += stride) {
        const uint32_t base = t * RPW;
        f32x4 v[R][NJ];
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint32_t row = base + (uint32_t)(r * GPW + grp);
            row = row < n_rows ? row : n_rows - 1;  // clamp: tail lanes re-read the last row
            const f32x4 *p = reinterpret_cast<const f32x4 *>(a.rows + (size_t)row * D) + lig;
#pragma unroll
            for (int j = 0; j < NJ; j++) v[r][j] = ld4<NT>(p + j * G);
        }
        float dot[R], rr[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            float d0 = 0.0f, n0 = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                d0 += v[r][j].x * q[j].x + v[r][j].y * q[j].y + v[r][j].z * q[j].z + v[r][j].w * q[j].w;
                n0 += v[r][j].x * v[r][j].x + v[r][j].y * v[r][j].y + v[r][j].z * v[r][j].z + v[r][j].w * v[r][j].w;
            }
            dot[r] = d0;
            rr[r] = n0;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            dot[r] = group_sum<G>(dot[r]);
            rr[r] = group_sum<G>(rr[r]);
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t row = base + (uint32_t)(r * GPW + grp);
            const float sim = cosine_from_sums(dot[r], qq, rr[r]);
            const float score = score_of(distance_of(sim));
            if constexpr (MODE == 0) {
                // one candidate per row group: lanes with lig == 0 carry it
                const uint64_t key = (row < n_rows && lig == 0) ? make_key(score, row) : 0ull;
                const DevFilter &f = a.flt;
                top.offer_lanes(key, sim, [&f](uint32_t rw) { return row_passes(f, rw); });
            } else {
                if (row < n_rows && lig == 0) {
                    bool ok = row_passes(a.flt, row);
                    if (a.has_threshold) ok = ok && (score >= a.threshold);
                    a.dense_keys[row] = ok ? make_key(score, row) : 0ull;
                    a.dense_sims[row] = sim;
                }
            }
        }
    }
    if constexpr (MODE == 0) block_merge_store<KS>(top, a.k, a.part_keys, a.part_sims);
}

// The same scan over a bf16 row store (cx_create_ex, CX_DTYPE_BF16): a 16-byte load is 8 elements, G lanes x 16 B cover a
// row piece, the products of bf16 values are exact in f32 and everything from there on is the f32 kernel's arithmetic —
// the reference's distance on the bf16-rounded rows.  Half the bytes per row: the roofline is n_rows * dim * 2.
template <int D, int G, int R, int KS, bool NT, int MODE>
__global__ __launch_bounds__(256) void scan16_kernel(const ScanArgs a) {
    constexpr int GPW = 64 / G;        // rows per wave-level load instruction
    constexpr int NJ = D / (8 * G);    // 16-byte loads per lane per row
    constexpr int RPW = R * GPW;
    static_assert(D % (8 * G) == 0, "dim must be a multiple of 8*G");
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int lig = lane % G;
    const int grp = lane / G;
    f32x4 q[NJ][2];
    const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.query);
    float qq = 0.0f;
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            q[j][h] = q4[(j * G + lig) * 2 + h];
            qq += q[j][h].x * q[j][h].x + q[j][h].y * q[j][h].y + q[j][h].z * q[j][h].z + q[j][h].w * q[j][h].w;
        }
    qq = group_sum<G>(qq) + a.q_tail_sumsq;

    WaveTopK<KS> top;
    if constexpr (MODE == 0) top.init(a.k);
    const uint32_t n_rows = a.n_rows;
    const uint32_t n_tiles = (n_rows + RPW - 1) / RPW;
    const uint32_t stride = gridDim.x * 4u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (uint32_t t = blockIdx.x * 4u + (uint32_t)wave; t < n_tiles; t += stride) {
        const uint32_t base = t * RPW;
        f32x4 v[R][NJ];
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint32_t row = base + (uint32_t)(r * GPW + grp);
            row = row < n_rows ? row : n_rows - 1;
            const f32x4 *p = reinterpret_cast<const f32x4 *>(a.rows16 + (size_t)row * D) + lig;
#pragma unroll
            for (int j = 0; j < NJ; j++) v[r][j] = ld4<NT>(p + j * G);
        }
        float dot[R], rr[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            float d0 = 0.0f, n0 = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const u32x4 u = __builtin_bit_cast(u32x4, v[r][j]);
                // element 2 i is the low half of word i, element 2 i + 1 the high half
                const f32x4 lo = {__uint_as_float(u.x << 16), __uint_as_float(u.y << 16), __uint_as_float(u.z << 16), __uint_as_float(u.w << 16)};
                const f32x4 hi = {__uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y & 0xFFFF0000u), __uint_as_float(u.z & 0xFFFF0000u), __uint_as_float(u.w & 0xFFFF0000u)};
                // elements 0..7 = lo.x hi.x lo.y hi.y lo.z hi.z lo.w hi.w against q[j][0] = elements 0..3, q[j][1] = 4..7
                d0 += lo.x * q[j][0].x + hi.x * q[j][0].y + lo.y * q[j][0].z + hi.y * q[j][0].w;
                d0 += lo.z * q[j][1].x + hi.z * q[j][1].y + lo.w * q[j][1].z + hi.w * q[j][1].w;
                n0 += lo.x * lo.x + hi.x * hi.x + lo.y * lo.y + hi.y * hi.y;
                n0 += lo.z * lo.z + hi.z * hi.z + lo.w * lo.w + hi.w * hi.w;
            }
            dot[r] = d0;
            rr[r] = n0;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            dot[r] = group_sum<G>(dot[r]);
            rr[r] = group_sum<G>(rr[r]);
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t row = base + (uint32_t)(r * GPW + grp);
            const float sim = cosine_from_sums(dot[r], qq, rr[r]);
            const float score = score_of(distance_of(sim));
            if constexpr (MODE == 0) {
                const uint64_t key = (row < n_rows && lig == 0) ? make_key(score, row) : 0ull;
                const DevFilter &f = a.flt;
                top.offer_lanes(key, sim, [&f](uint32_t rw) { return row_passes(f, rw); });
            } else {
                if (row < n_rows && lig == 0) {
                    bool ok = row_passes(a.flt, row);
                    if (a.has_threshold) ok = ok && (score >= a.threshold);
                    a.dense_keys[row] = ok ? make_key(score, row) : 0ull;
                    a.dense_sims[row] = sim;
                }
            }
        }
    }
    if constexpr (MODE == 0) block_merge_store<KS>(top, a.k, a.part_keys, a.part_sims);
}

// Any dimension (the reference's own tests use dim = 3): one wave per row,
// lane l takes elements l, l+64, ...; scalar loads.  Correctness path only.
template <int KS, int MODE>
__global__ __launch_bounds__(256) void scan_generic_kernel(const ScanArgs a) {
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const uint32_t dim = a.dim;
    float qq = 0.0f;
    for (uint32_t j = (uint32_t)lane; j < dim; j += 64u) qq += a.query[j] * a.query[j];
    qq = group_sum<64>(qq) + a.q_tail_sumsq;

    WaveTopK<KS> top;
    if constexpr (MODE == 0) top.init(a.k);
    const uint32_t stride = gridDim.x * 4u;
    for (uint32_t row = blockIdx.x * 4u + (uint32_t)wave; row < a.n_rows; row += stride) {
        const float *p = a.rows + (size_t)row * dim;
        const uint16_t *p16 = a.rows16 + (size_t)row * dim;
        float d0 = 0.0f, n0 = 0.0f;
        for (uint32_t j = (uint32_t)lane; j < dim; j += 64u) {
            const float x = a.rows16 ? bf16_bits_to_f32(p16[j]) : p[j];
            d0 += x * a.query[j];
            n0 += x * x;
        }
        d0 = group_sum<64>(d0);
        n0 = group_sum<64>(n0);
        const float sim = cosine_from_sums(d0, qq, n0);
        const float score = score_of(distance_of(sim));
        if constexpr (MODE == 0) {
            const uint64_t key = lane == 0 ? make_key(score, row) : 0ull;
            const DevFilter &f = a.flt;
            top.offer_lanes(key, sim, [&f](uint32_t rw) { return row_passes(f, rw); });
        } else if (lane == 0) {
            bool ok = row_passes(a.flt, row);
            if (a.has_threshold) ok = ok && (score >= a.threshold);
            a.dense_keys[row] = ok ? make_key(score, row) : 0ull;
            a.dense_sims[row] = sim;
        }
    }
    if constexpr (MODE == 0) block_merge_store<KS>(top, a.k, a.part_keys, a.part_sims);
}

// Second stage: one block folds [n_lists][k] partial lists into the final,
// sorted top-k.  16 waves each sweep a strided share with coalesced loads,
// then wave 0 folds the 15 other lists and ranks the k survivors.
template <int KS>
__global__ __launch_bounds__(1024) void merge_kernel(const MergeArgs m0) {
    if (m0.run_if && *m0.run_if == 0u) return;
    MergeArgs m = m0;  // one block per query
    m.part_keys += (size_t)blockIdx.x * m0.n_lists * m0.k;
    m.part_sims += (size_t)blockIdx.x * m0.n_lists * m0.k;
    m.out_rows += (size_t)blockIdx.x * m0.k;
    m.out_scores += (size_t)blockIdx.x * m0.k;
    m.out_dists += (size_t)blockIdx.x * m0.k;
    m.out_count += blockIdx.x;
    __shared__ uint64_t sk[16][64 * KS];
    __shared__ float ss[16][64 * KS];
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t k = m.k;
    const uint32_t total = m.n_lists * k;
    WaveTopK<KS> top;
    top.init(k);
    for (uint32_t i0 = wave * 64u; i0 < total; i0 += 16u * 64u) {
        const uint32_t i = i0 + lane;
        const uint64_t kg = i < total ? m.part_keys[i] : 0ull;
        const float sg = i < total ? m.part_sims[i] : 0.0f;
        top.offer_lanes(kg, sg, [](uint32_t) { return true; });
    }
    if (wave > 0) top.store(sk[wave], ss[wave]);
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 16; w++) {
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const uint32_t i = (uint32_t)s * 64u + lane;
                const uint64_t kg = i < k ? sk[w][i] : 0ull;
                const float sg = i < k ? ss[w][i] : 0.0f;
                top.offer_lanes(kg, sg, [](uint32_t) { return true; });
            }
        }
        // the list is sorted best-first: rank = position
        uint32_t count = 0;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const uint32_t i = (uint32_t)s * 64u + lane;
            const uint64_t ki = top.key[s];
            const bool valid = i < k && ki != 0ull;
            if (valid) {
                const float dist = distance_of(top.sim[s]);
                m.out_rows[i] = key_row(ki);
                m.out_dists[i] = dist;
                m.out_scores[i] = score_of(dist);
            }
            count += (uint32_t)__popcll(__ballot(valid));
        }
        if (lane == 0) *m.out_count = count;
    }
}

// Second stage for k <= 32 (the interactive searches: k = 5, 10, 30): no serial
// fold.  Every partial list is sorted best-first, so its head is its best key.
//  P0 heads -> LDS.
//  P1 t0 = k-th largest of a 64-head sample (rank counting inside wave 0 with
//     v_readlane broadcasts).  At least k real candidates are >= t0, so it is a
//     valid lower bound on the answer's k-th key.
//  P2 heads >= t0 are compacted (typically ~ n*k/64 of them);
//  P3 rank counting among those gives tau = the exact k-th largest head.
//     Keys are unique, so exactly k lists have a head >= tau, each holding at
//     most k entries: <= k*k <= 1024 survivors.
//  P4 every (list, slot) pair is looked at once, in parallel: entries >= tau
//     of lists whose head is >= tau are appended to LDS.
//  P5 every survivor counts the survivors that beat it; rank < k -> out[rank].
// NT = threads in the block (a multiple of 64, <= 1024); n_lists <= 2048.
constexpr uint32_t MERGE_SMALL_MAX_LISTS = 2048;
template <int NT>
__device__ inline void merge_small_body(const MergeArgs &m0) {
    // one block per query: query q's lists start at q*n_lists*k, its outputs at q*k
    if (m0.run_if && *m0.run_if == 0u) return;
    MergeArgs m = m0;
    m.part_keys += (size_t)blockIdx.x * m0.n_lists * m0.k;
    m.part_sims += (size_t)blockIdx.x * m0.n_lists * m0.k;
    m.out_rows += (size_t)blockIdx.x * m0.k;
    m.out_scores += (size_t)blockIdx.x * m0.k;
    m.out_dists += (size_t)blockIdx.x * m0.k;
    m.out_count += blockIdx.x;
    __shared__ uint64_t heads[MERGE_SMALL_MAX_LISTS];
    __shared__ uint64_t hsurv[MERGE_SMALL_MAX_LISTS];
    __shared__ uint64_t surv_k[1024];
    __shared__ float surv_s[1024];
    __shared__ uint32_t s_nh, s_n;
    __shared__ uint64_t s_t0, s_tau;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t k = m.k, n = m.n_lists;
    for (uint32_t i = tid; i < n; i += NT) heads[i] = m.part_keys[(size_t)i * k];
    if (tid == 0) { s_nh = 0; s_n = 0; s_t0 = 0ull; s_tau = 0ull; }
    __syncthreads();
    if (wave == 0) {
        const uint32_t step = n >= 64u ? n / 64u : 1u;
        const uint32_t i = lane * step;
        const uint64_t mine = i < n ? heads[i] : 0ull;
        uint32_t rank = 0;
        for (int l = 0; l < 64; l++) {
            const uint64_t o = readlane_u64(mine, l);
            rank += (o > mine || (o == mine && (uint32_t)l < lane)) ? 1u : 0u;  // zeros tie: order by lane
        }
        if (rank == k - 1u) s_t0 = mine;
    }
    __syncthreads();
    const uint64_t t0 = s_t0;
    for (uint32_t i = tid; i < n; i += NT) {
        const uint64_t h = heads[i];
        if (h != 0ull && h >= t0) hsurv[atomicAdd(&s_nh, 1u)] = h;
    }
    __syncthreads();
    const uint32_t SH = s_nh;
    for (uint32_t i = tid; i < SH; i += NT) {
        const uint64_t hi = hsurv[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < SH; j++) rank += hsurv[j] > hi ? 1u : 0u;
        if (rank == k - 1u) s_tau = hi;  // stays 0 when fewer than k lists are non-empty
    }
    __syncthreads();
    const uint64_t tau = s_tau;
    for (uint32_t e = tid; e < n * k; e += NT) {
        const uint64_t h = heads[e / k];
        if (h == 0ull || h < tau) continue;
        const uint64_t kg = m.part_keys[e];
        if (kg == 0ull || kg < tau) continue;
        const uint32_t pos = atomicAdd(&s_n, 1u);
        if (pos < 1024u) { surv_k[pos] = kg; surv_s[pos] = m.part_sims[e]; }
    }
    __syncthreads();
    const uint32_t S = s_n < 1024u ? s_n : 1024u;
    for (uint32_t i = tid; i < S; i += NT) {
        const uint64_t ki = surv_k[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < S; j++) rank += surv_k[j] > ki ? 1u : 0u;
        if (rank < k) {
            const float dist = distance_of(surv_s[i]);
            m.out_rows[rank] = key_row(ki);
            m.out_dists[rank] = dist;
            m.out_scores[rank] = score_of(dist);
            if (m.bound_out && rank == k - 1u) m.bound_out[blockIdx.x] = score_ord(score_of(dist));
        }
    }
    if (tid == 0) {
        *m.out_count = S < k ? S : k;
        if (m.bound_out && S < k) m.bound_out[blockIdx.x] = 0u;
        if (m.clear_word && blockIdx.x == 0) *m.clear_word = 0u;
    }
}

__global__ __launch_bounds__(1024) void merge_small_kernel(const MergeArgs m) { merge_small_body<1024>(m); }

// ------------------------------------------------------------------ host

static int num_cus() { return (int)device_cus(); }

// Blocks per CU of the scan grid: enough row bytes in flight per CU to cover HBM latency.  Measured optimum per
// dim at 1-4M rows (profiles/r01/tuning.md): short rows need more blocks, 1024 wants one more than 768.
static uint32_t blocks_per_cu(uint32_t dim, bool rows16 = false) {
    static int env = -1;
    if (env < 0) {
        const char *e = getenv("CX_SCAN_BLOCKS_PER_CU");
        env = e ? atoi(e) : 0;
        if (env < 0) env = 0;
        if (env > 8) env = 8;
    }
    if (env) return (uint32_t)env;
    if (rows16) return dim <= 256 ? 5 : 4;   // bf16 store, 1M x 768 / 1024: 4 blocks per CU 0.83 / 0.82 of the peak, 2: 0.80 / 0.72
    if (dim <= 128) return 5;
    if (dim <= 256) return 4;
    if (dim <= 384) return 2;
    if (dim <= 512) return 3;
    if (dim <= 768) return 2;
    if (dim <= 1024) return 3;
    return 2;
}

uint32_t scan_grid_blocks(uint32_t n_rows, uint32_t dim, bool rows16) {
    uint32_t want = (n_rows + 3u) / 4u;  // at least one row per wave
    uint32_t cap = (uint32_t)num_cus() * blocks_per_cu(dim, rows16);
    if (want < 1u) want = 1u;
    return want < cap ? want : cap;
}

template <int D, int G, int R, int MODE>
static void launch_fixed(const ScanArgs &a, uint32_t grid, int ks, bool nt, hipStream_t s) {
#define CX_LAUNCH(KS_)                                                                        \
    do {                                                                                       \
        if (nt) hipLaunchKernelGGL((scan_kernel<D, G, R, KS_, true, MODE>), dim3(grid), dim3(256), 0, s, a);  \
        else hipLaunchKernelGGL((scan_kernel<D, G, R, KS_, false, MODE>), dim3(grid), dim3(256), 0, s, a);    \
    } while (0)
    if constexpr (MODE == 1) { CX_LAUNCH(1); }
    else {
        if (ks == 1) CX_LAUNCH(1);
        else if (ks == 2) CX_LAUNCH(2);
        else CX_LAUNCH(4);
    }
#undef CX_LAUNCH
}

template <int MODE>
static void launch_generic(const ScanArgs &a, uint32_t grid, int ks, hipStream_t s) {
    if constexpr (MODE == 1) {
        hipLaunchKernelGGL((scan_generic_kernel<1, 1>), dim3(grid), dim3(256), 0, s, a);
    } else {
        if (ks == 1) hipLaunchKernelGGL((scan_generic_kernel<1, 0>), dim3(grid), dim3(256), 0, s, a);
        else if (ks == 2) hipLaunchKernelGGL((scan_generic_kernel<2, 0>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((scan_generic_kernel<4, 0>), dim3(grid), dim3(256), 0, s, a);
    }
}

template <int D, int G, int R, int MODE>
static void launch_fixed16(const ScanArgs &a, uint32_t grid, int ks, bool nt, hipStream_t s) {
#define CX_LAUNCH(KS_)                                                                        \
    do {                                                                                       \
        if (nt) hipLaunchKernelGGL((scan16_kernel<D, G, R, KS_, true, MODE>), dim3(grid), dim3(256), 0, s, a);  \
        else hipLaunchKernelGGL((scan16_kernel<D, G, R, KS_, false, MODE>), dim3(grid), dim3(256), 0, s, a);    \
    } while (0)
    if constexpr (MODE == 1) { CX_LAUNCH(1); }
    else {
        if (ks == 1) CX_LAUNCH(1);
        else if (ks == 2) CX_LAUNCH(2);
        else CX_LAUNCH(4);
    }
#undef CX_LAUNCH
}

template <int MODE>
static void dispatch_scan(const ScanArgs &a, uint32_t grid, int ks, bool nt, hipStream_t s) {
    if (a.rows16) {   // bf16 store: G lanes x 16 B = G x 8 elements per row piece
        const bool aligned16 = ((reinterpret_cast<uintptr_t>(a.rows16) | reinterpret_cast<uintptr_t>(a.query)) & 15u) == 0;
        if (aligned16) {
            switch (a.dim) {
                case 128: return launch_fixed16<128, 16, 8, MODE>(a, grid, ks, nt, s);
                case 256: return launch_fixed16<256, 32, 8, MODE>(a, grid, ks, nt, s);
                case 384: return launch_fixed16<384, 16, 4, MODE>(a, grid, ks, nt, s);
                case 512: return launch_fixed16<512, 64, 8, MODE>(a, grid, ks, nt, s);
                case 768: {
                    static const int r16 = getenv("CX_SCAN16_R") ? atoi(getenv("CX_SCAN16_R")) : 4;  // tuning knob
                    if (r16 == 2) return launch_fixed16<768, 32, 2, MODE>(a, grid, ks, nt, s);
                    if (r16 == 8) return launch_fixed16<768, 32, 8, MODE>(a, grid, ks, nt, s);
                    if (r16 == 16) return launch_fixed16<768, 16, 4, MODE>(a, grid, ks, nt, s);
                    return launch_fixed16<768, 32, 4, MODE>(a, grid, ks, nt, s);
                }
                case 1024: {
                    static const int r16 = getenv("CX_SCAN16_R") ? atoi(getenv("CX_SCAN16_R")) : 4;
                    if (r16 == 2) return launch_fixed16<1024, 64, 2, MODE>(a, grid, ks, nt, s);
                    if (r16 == 8) return launch_fixed16<1024, 64, 8, MODE>(a, grid, ks, nt, s);
                    return launch_fixed16<1024, 64, 4, MODE>(a, grid, ks, nt, s);
                }
                case 1536: return launch_fixed16<1536, 64, 2, MODE>(a, grid, ks, nt, s);
                default: break;
            }
        }
        return launch_generic<MODE>(a, grid, ks, s);
    }
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.rows) | reinterpret_cast<uintptr_t>(a.query)) & 15u) == 0;
    if (aligned) {
        switch (a.dim) {
            case 128: return launch_fixed<128, 32, 4, MODE>(a, grid, ks, nt, s);
            case 256: return launch_fixed<256, 64, 8, MODE>(a, grid, ks, nt, s);
            case 384: {
                static const int r384 = getenv("CX_SCAN_R384") ? atoi(getenv("CX_SCAN_R384")) : 4;  // tuning knob
                if (r384 == 2) return launch_fixed<384, 32, 2, MODE>(a, grid, ks, nt, s);
                if (r384 == 8) return launch_fixed<384, 32, 8, MODE>(a, grid, ks, nt, s);
                return launch_fixed<384, 32, 4, MODE>(a, grid, ks, nt, s);
            }
            case 512: return launch_fixed<512, 64, 4, MODE>(a, grid, ks, nt, s);
            case 768: {
                static const int r768 = getenv("CX_SCAN_R") ? atoi(getenv("CX_SCAN_R")) : 4;  // tuning knob
                if (r768 == 2) return launch_fixed<768, 64, 2, MODE>(a, grid, ks, nt, s);
                if (r768 == 8) return launch_fixed<768, 64, 8, MODE>(a, grid, ks, nt, s);
                return launch_fixed<768, 64, 4, MODE>(a, grid, ks, nt, s);
            }
            case 1024: return launch_fixed<1024, 64, 2, MODE>(a, grid, ks, nt, s);
            case 1536: return launch_fixed<1536, 64, 2, MODE>(a, grid, ks, nt, s);
            default: break;
        }
    }
    launch_generic<MODE>(a, grid, ks, s);
}

__global__ void merge_radix_kernel(const MergeArgs m0);   // defined below

static bool use_old_merge() {   // CX_MERGE_WAVETOPK=1: the previous second stage, for A/B runs
    static const int v = getenv("CX_MERGE_WAVETOPK") ? atoi(getenv("CX_MERGE_WAVETOPK")) : 0;
    return v != 0;
}

int launch_scan_topk(const ScanArgs &a, const MergeArgs &m, bool nontemporal, hipStream_t stream, hipEvent_t ev0,
                     hipEvent_t ev1) {
    if (a.k > TOPK_MAX) return set_err(CX_ERR_VALIDATION, "launch_scan_topk: k=%u exceeds %u", a.k, TOPK_MAX);
    const int ks = a.k <= 64 ? 1 : (a.k <= 128 ? 2 : 4);
    const uint32_t grid = scan_grid_blocks(a.n_rows, a.dim, a.rows16 != nullptr);
    if (m.n_lists != grid) return set_err(CX_ERR_VALIDATION, "launch_scan_topk: scratch sized for %u lists, grid is %u", m.n_lists, grid);
    if (ev0) CX_HIP(hipEventRecord(ev0, stream));
    dispatch_scan<0>(a, grid, ks, nontemporal, stream);
    if (ev1) CX_HIP(hipEventRecord(ev1, stream));
    if (a.k <= 32 && grid <= MERGE_SMALL_MAX_LISTS) hipLaunchKernelGGL(merge_small_kernel, dim3(1), dim3(1024), 0, stream, m);
    else if (use_old_merge()) {
        if (ks == 1) hipLaunchKernelGGL((merge_kernel<1>), dim3(1), dim3(1024), 0, stream, m);
        else if (ks == 2) hipLaunchKernelGGL((merge_kernel<2>), dim3(1), dim3(1024), 0, stream, m);
        else hipLaunchKernelGGL((merge_kernel<4>), dim3(1), dim3(1024), 0, stream, m);
    } else hipLaunchKernelGGL(merge_radix_kernel, dim3(1), dim3(1024), 0, stream, m);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

bool merge_batch_writes_bound(uint32_t k, uint32_t n_lists) { return k >= 1 && k <= 32 && n_lists <= MERGE_SMALL_MAX_LISTS && !use_old_merge(); }

int launch_merge_batch(const MergeArgs &m, uint32_t nq, hipStream_t stream, bool sorted_lists) {
    if (!nq || !m.k) return CX_OK;
    if (!sorted_lists) hipLaunchKernelGGL(merge_radix_kernel, dim3(nq), dim3(1024), 0, stream, m);   // a selection over all entries
    else if (m.k <= 32 && m.n_lists <= MERGE_SMALL_MAX_LISTS) hipLaunchKernelGGL(merge_small_kernel, dim3(nq), dim3(1024), 0, stream, m);
    else if (use_old_merge()) {
        if (m.k <= 64) hipLaunchKernelGGL((merge_kernel<1>), dim3(nq), dim3(1024), 0, stream, m);
        else if (m.k <= 128) hipLaunchKernelGGL((merge_kernel<2>), dim3(nq), dim3(1024), 0, stream, m);
        else hipLaunchKernelGGL((merge_kernel<4>), dim3(nq), dim3(1024), 0, stream, m);
    } else hipLaunchKernelGGL(merge_radix_kernel, dim3(nq), dim3(1024), 0, stream, m);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// Top-k of nq dense cosine rows (batchg.hip's output, [nq][stride]): the second half of the single-query scan — keys
// from (score, row), the row filter for the rows that beat the running bound, the wave's register list, the block's
// partial list — on 4 B per row and query instead of the row itself.  grid = (chunks, nq).
template <int KS>
__global__ __launch_bounds__(256) void dense_topk_kernel(const float *dense, uint32_t stride, uint32_t n_rows, uint32_t k, const DevFilter flt,
                                                         uint64_t *part_keys, float *part_sims, const uint32_t *run_if) {
    if (run_if && *run_if == 0u) return;
    const uint32_t q = blockIdx.y, lane = (uint32_t)lane_id(), wave = threadIdx.x >> 6;
    const float *d = dense + (size_t)q * stride;
    WaveTopK<KS> top;
    top.init(k);
    const uint32_t per = (n_rows + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per, hi = lo + per < n_rows ? lo + per : n_rows;
    for (uint32_t r0 = lo + wave * 64u; r0 < hi; r0 += 256u) {
        const uint32_t row = r0 + lane;
        uint64_t key = 0ull;
        float sim = 0.0f;
        if (row < hi) {
            sim = d[row];
            key = make_key(score_of(distance_of(sim)), row);
        }
        top.offer_lanes(key, sim, [&flt](uint32_t rw) { return row_passes(flt, rw); });
    }
    const size_t base = (size_t)q * gridDim.x * k;
    block_merge_store<KS>(top, k, part_keys + base, part_sims + base);
}

// tau_ord[q] = score_ord of the k-th largest score among the cosines dense[q][0 .. n) of rows that pass the filter (0 when
// fewer than k do): a block-wide radix
// select over the 32-bit ordinals, one block per query, most significant byte first — a bound needs no list.  (The
// wave-list top-k + merge this replaces took 0.15 + 0.04 ms on a 39k-row sample at k = 100: more than a fifth of the
// pass it prepares.)
__global__ __launch_bounds__(1024) void bound_select_kernel(const float *dense, uint32_t stride, uint32_t n, uint32_t k, uint32_t *tau_ord,
                                                           const DevFilter flt, uint32_t tile_rows, uint32_t tile_step) {
    __shared__ uint32_t hist[256], wtot[4];
    __shared__ uint32_t s_prefix, s_mask, s_need, s_found_all, s_found;
    const uint32_t tid = threadIdx.x, NT = 1024;
    if (n < k) { if (tid == 0) tau_ord[blockIdx.x] = 0u; return; }
    const float *d = dense + (size_t)blockIdx.x * stride;
    if (tid == 0) { s_prefix = 0u; s_mask = 0u; s_need = k; s_found_all = 1u; s_found = 0u; }
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix, mask = s_mask;
        for (uint32_t e0 = tid; e0 < n; e0 += 8u * NT) {
            float v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) { const uint32_t e = e0 + u * NT; v[u] = e < n ? d[e] : 0.0f; }
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
                const uint32_t o = score_ord(score_of(distance_of(v[u])));
                const uint32_t e = e0 + u * NT;   // column e of the sample = row (e / tile_rows) * tile_step * tile_rows + e % tile_rows
                bool act = e < n && (o & mask) == prefix;
                if (act && !flt.trivial) act = row_passes(flt, (e / tile_rows) * tile_step * tile_rows + e % tile_rows);
                const uint32_t bin = (o >> shift) & 255u;
                const uint64_t am = __ballot(act);
                if (am == 0ull) continue;
                // scores share their leading bytes: a wave that lands in one bin adds its count once (64 same-address
                // LDS atomics serialise)
                const int first = __ffsll((unsigned long long)am) - 1;
                const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, first);
                if (__ballot(act && bin == b0) == am) {
                    if ((int)(tid & 63u) == first) atomicAdd(&hist[b0], (uint32_t)__popcll(am));
                } else if (act) {
                    atomicAdd(&hist[bin], 1u);
                }
            }
        }
        __syncthreads();
        {   // the bin that holds the need-th largest of the values still in play
            uint32_t mine;
            const uint32_t above = bins_above(hist, tid, wtot, mine);
            const uint32_t need = s_need;
            if (tid < 256 && above < need && need <= above + mine) {
                s_prefix = prefix | (tid << shift);
                s_mask = mask | (0xFFu << shift);
                s_need = need - above;
                s_found = 1u;
            }
        }
        __syncthreads();
        if (tid == 0) { if (!s_found) s_found_all = 0u; s_found = 0u; }   // fewer than k rows pass the filter: no bound
        __syncthreads();
    }
    if (tid == 0) tau_ord[blockIdx.x] = s_found_all ? s_prefix : 0u;
}
// The same for samples of up to NV * 1024 columns: every thread keeps its NV ordinals in registers, read once
// (bound_select_kernel re-reads the sample in each of its four passes: 36 us at 16k columns, this 10).
template <int NV>
__global__ __launch_bounds__(1024) void bound_select_reg_kernel(const float *dense, uint32_t stride, uint32_t n, uint32_t k, uint32_t *tau_ord,
                                                               const DevFilter flt, uint32_t tile_rows, uint32_t tile_step) {
    __shared__ uint32_t sh[264];
    const uint32_t tid = threadIdx.x;
    const float *d = dense + (size_t)blockIdx.x * stride;
    uint32_t key[NV];   // 32-bit ordinals (>= 1 for a live score, 0 = empty): 64 per thread fit the 128 registers of a 1024-thread block
    constexpr int CHK = NV < 16 ? NV : 16;   // loads in flight per thread (all NV at once would need 2 NV registers)
#pragma unroll
    for (int c = 0; c < NV; c += CHK) {
        float v[CHK];
#pragma unroll
        for (int u = 0; u < CHK; u++) { const uint32_t e = tid + (uint32_t)(c + u) * 1024u; v[u] = e < n ? d[e] : 0.0f; }
#pragma unroll
        for (int u = 0; u < CHK; u++) {
            const uint32_t e = tid + (uint32_t)(c + u) * 1024u;
            bool live = e < n;
            if (live && !flt.trivial) live = row_passes(flt, (e / tile_rows) * tile_step * tile_rows + e % tile_rows);
            key[c + u] = live ? score_ord(score_of(distance_of(v[u]))) : 0u;
        }
    }
    if constexpr (NV <= 8) {
        const uint32_t t = block_select_kth<NV, uint32_t>(key, k, 24, 0, sh);
        if (tid == 0) tau_ord[blockIdx.x] = t;
    } else {
        // Two stages.  The radix walk costs four passes over every register of every thread (59 us for 39k scores at
        // k = 100: ballots and scalar bookkeeping per value, not the LDS atomics).  A first walk over 4 values per thread —
        // a uniform 4,096-column sub-sample — finds a score t0 that about 2.5 k of ALL the scores should reach; one compare
        // per value collects those into LDS; a second 4-values-per-thread walk over the survivors finds the exact k-th.
        // Too few survivors (< k): t0 is taken further down the sub-sample; more than the list holds: the full walk.
        __shared__ uint32_t surv[4096];
        __shared__ uint32_t s_cnt;
        const uint32_t sub[4] = {key[0], key[1], key[2], key[3]};
        uint32_t ksub = (uint32_t)(((uint64_t)k * 4096u * 5u / 2u + n - 1u) / n) + 2u;
        uint32_t t = 0u;
        bool done = false;
        for (int round = 0; round < 4 && !done; round++, ksub *= 2u) {
            const uint32_t t0 = ksub <= 4096u ? block_select_kth<4, uint32_t>(sub, ksub, 24, 0, sh) : 0u;   // 0: every live score survives
            __syncthreads();
            if (tid == 0) s_cnt = 0u;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NV; u++)
                if (key[u] != 0u && key[u] >= t0) {
                    const uint32_t slot = atomicAdd(&s_cnt, 1u);
                    if (slot < 4096u) surv[slot] = key[u];
                }
            __syncthreads();
            const uint32_t cnt = s_cnt;   // uniform
            if (cnt > 4096u) break;       // (a sample full of equal scores): the full walk below
            if (cnt >= k || t0 == 0u) {
                uint32_t r[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const uint32_t e = tid + (uint32_t)u * 1024u; r[u] = e < cnt ? surv[e] : 0u; }
                t = block_select_kth<4, uint32_t>(r, k, 24, 0, sh);   // 0 when fewer than k scores are live at all
                done = true;
            }
            __syncthreads();
        }
        if (!done) t = block_select_kth<NV, uint32_t>(key, k, 24, 0, sh);
        if (tid == 0) tau_ord[blockIdx.x] = t;
    }
}

int launch_bound_select(const float *d_dense, uint32_t stride, uint32_t n, uint32_t nq, uint32_t k, uint32_t *tau_ord, const DevFilter &flt,
                        uint32_t tile_rows, uint32_t tile_step, hipStream_t stream) {
    if (!nq || !k) return CX_OK;
    if (n <= 16u * 1024u) hipLaunchKernelGGL(bound_select_reg_kernel<16>, dim3(nq), dim3(1024), 0, stream, d_dense, stride, n, k, tau_ord, flt, tile_rows, tile_step);
    else if (n <= 32u * 1024u) hipLaunchKernelGGL(bound_select_reg_kernel<32>, dim3(nq), dim3(1024), 0, stream, d_dense, stride, n, k, tau_ord, flt, tile_rows, tile_step);
    else if (n <= 64u * 1024u) hipLaunchKernelGGL(bound_select_reg_kernel<64>, dim3(nq), dim3(1024), 0, stream, d_dense, stride, n, k, tau_ord, flt, tile_rows, tile_step);   // k = 100 at 1.25M x 768: 39k sampled scores, 63 us through the kernel below
    else hipLaunchKernelGGL(bound_select_kernel, dim3(nq), dim3(1024), 0, stream, d_dense, stride, n, k, tau_ord, flt, tile_rows, tile_step);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

uint32_t dense_topk_chunks(uint32_t n_rows) {
    const uint32_t c = (n_rows + 16383u) / 16384u;   // >= 16k rows per block: the running bound means something
    return c < 1u ? 1u : (c > 256u ? 256u : c);
}

int launch_dense_topk(const float *d_dense, uint32_t stride, uint32_t n_rows, uint32_t nq, uint32_t k, const DevFilter &flt,
                      uint64_t *part_keys, float *part_sims, uint32_t chunks, hipStream_t stream, const uint32_t *run_if) {
    if (!nq || !n_rows || !k) return CX_OK;
    if (k > TOPK_MAX) return set_err(CX_ERR_VALIDATION, "dense top-k: k=%u exceeds %u", k, TOPK_MAX);
    const dim3 grid(chunks, nq);
    if (k <= 64) hipLaunchKernelGGL((dense_topk_kernel<1>), grid, dim3(256), 0, stream, d_dense, stride, n_rows, k, flt, part_keys, part_sims, run_if);
    else if (k <= 128) hipLaunchKernelGGL((dense_topk_kernel<2>), grid, dim3(256), 0, stream, d_dense, stride, n_rows, k, flt, part_keys, part_sims, run_if);
    else hipLaunchKernelGGL((dense_topk_kernel<4>), grid, dim3(256), 0, stream, d_dense, stride, n_rows, k, flt, part_keys, part_sims, run_if);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

int launch_scan_dense(const ScanArgs &a, bool nontemporal, hipStream_t stream) {
    const uint32_t grid = scan_grid_blocks(a.n_rows, a.dim, a.rows16 != nullptr);
    dispatch_scan<1>(a, grid, 1, nontemporal, stream);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// ------------------------------------------------ cross-shard merge (§8e)

// One block per query: fold n_parts lists of k (already sorted per shard, but
// order is not relied on) into the global top-k.  Keys are rebuilt with the
// global row so ties resolve by global insertion order.
// Second stage for k > 32 (or very many lists): block-wide RADIX SELECT of the k-th largest key over all
// n_lists * k candidates, most significant byte first (256-bin histogram in LDS per pass; candidates stay in
// L2).  As soon as "keys above the selected bin + keys in it" fit the survivor buffer the walk stops, those
// keys are collected and ranked against each other (keys are unique: the row is part of the key).  Typically 3-4
// passes of ~2 us; the WaveTopK merge it replaces inserted candidates one at a time (0.23 ms at k = 100 and
// 0.9 ms at k = 256 for 512 lists — as long as the scan itself).
constexpr uint32_t MERGE_RADIX_CAP = 2048;
__global__ __launch_bounds__(1024) void merge_radix_kernel(const MergeArgs m0) {
    if (m0.run_if && *m0.run_if == 0u) return;
    MergeArgs m = m0;  // one block per query
    m.part_keys += (size_t)blockIdx.x * m0.n_lists * m0.k;
    m.part_sims += (size_t)blockIdx.x * m0.n_lists * m0.k;
    m.out_rows += (size_t)blockIdx.x * m0.k;
    m.out_scores += (size_t)blockIdx.x * m0.k;
    m.out_dists += (size_t)blockIdx.x * m0.k;
    m.out_count += blockIdx.x;
    __shared__ uint32_t hist[256], wtot[4];
    __shared__ uint64_t surv_k[MERGE_RADIX_CAP];
    __shared__ float surv_s[MERGE_RADIX_CAP];
    __shared__ uint64_t s_prefix, s_mask;
    __shared__ uint32_t s_need, s_bin_cnt, s_n, s_found;
    const uint32_t tid = threadIdx.x, NT = 1024;
    const uint32_t k = m.k, total = m.n_lists * k;
    // segmented input (batchg's candidate lists): only the first seg_counts[segment] entries of each seg_len entries are
    // set — the rest is never read, so nobody has to zero it
    const uint32_t *seg_counts = m.seg_counts ? m.seg_counts + (size_t)blockIdx.x * (total / m.seg_len) : nullptr;
    auto live = [&](uint32_t e) { return e < total && (!seg_counts || e % m.seg_len < seg_counts[e / m.seg_len]); };
    if (tid == 0) { s_prefix = 0ull; s_mask = 0ull; s_need = k; s_bin_cnt = 0; s_n = 0; s_found = 0; }
    uint64_t low = 1ull;   // collect every key >= low (1 = every non-empty key)
    for (int shift = 56; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint64_t prefix = s_prefix, mask = s_mask;
        // eight keys in flight per thread: one block has to pull the whole candidate set through one CU, and
        // with a single load outstanding per thread a pass is latency-bound (15 us for 51k keys)
        for (uint32_t e0 = tid; e0 < total; e0 += 8u * NT) {
            uint64_t key[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) { const uint32_t e = e0 + u * NT; key[u] = live(e) ? m.part_keys[e] : 0ull; }
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
                const bool act = key[u] != 0ull && (key[u] & mask) == prefix;
                const uint32_t bin = (uint32_t)(key[u] >> shift) & 255u;
                const uint64_t am = __ballot(act);
                if (am == 0ull) continue;
                // the leading bytes of score keys are nearly constant: when the whole wave lands in one bin, one
                // lane adds the count (64 same-address LDS atomics serialise: 25 us for such a pass).  Peeling
                // off several popular bins per wave measured no better than this.
                const int first = __ffsll((unsigned long long)am) - 1;
                const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, first);
                if (__ballot(act && bin == b0) == am) {
                    if ((int)(tid & 63u) == first) atomicAdd(&hist[b0], (uint32_t)__popcll(am));
                } else if (act) {
                    atomicAdd(&hist[bin], 1u);
                }
            }
        }
        __syncthreads();
        {   // the bin that holds the need-th largest of the keys still in play
            uint32_t mine;
            const uint32_t above = bins_above(hist, tid, wtot, mine);
            const uint32_t need = s_need;
            if (tid < 256 && above < need && need <= above + mine) {
                s_prefix = prefix | ((uint64_t)tid << shift);
                s_mask = mask | (0xFFull << shift);
                s_need = need - above;
                s_bin_cnt = mine;
                s_found = 1;
            }
        }
        __syncthreads();
        if (!s_found) break;           // fewer than k non-empty keys in all: everything is a result (low stays 1)
        low = s_prefix;                // keys >= low: the (k - need) above the bin + the bin itself
        if ((k - s_need) + s_bin_cnt <= MERGE_RADIX_CAP) break;
        __syncthreads();
        if (tid == 0) s_found = 0;
    }
    __syncthreads();
    for (uint32_t e0 = tid; e0 < total; e0 += 8u * NT) {
        uint64_t key[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) { const uint32_t e = e0 + u * NT; key[u] = live(e) ? m.part_keys[e] : 0ull; }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++)
            if (key[u] != 0ull && key[u] >= low) {
                const uint32_t pos = atomicAdd(&s_n, 1u);
                if (pos < MERGE_RADIX_CAP) { surv_k[pos] = key[u]; surv_s[pos] = m.part_sims[e0 + u * NT]; }
            }
    }
    __syncthreads();
    const uint32_t S = s_n < MERGE_RADIX_CAP ? s_n : MERGE_RADIX_CAP;
    for (uint32_t i = tid; i < S; i += NT) {
        const uint64_t ki = surv_k[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < S; j++) rank += surv_k[j] > ki ? 1u : 0u;
        if (rank < k) {
            const float dist = distance_of(surv_s[i]);
            m.out_rows[rank] = key_row(ki);
            m.out_dists[rank] = dist;
            m.out_scores[rank] = score_of(dist);
        }
    }
    if (tid == 0) *m.out_count = S < k ? S : k;
}

// batchg's candidate lists ([segments][seg_len] slots per query, the first seg_counts[segment] of each set) -> the final
// top k, one block per query.  The live entries (a few hundred to a few thousand) go into registers through a prefix sum of
// the counts, block_select_kth finds the k-th key exactly, and the <= k survivors are ranked against each other — one
// read of the lists instead of the radix merge's pass-by-pass re-reads (29 us at k = 10, 40 at k = 100: now ~10).
// More than NV * 1024 live entries (a weak bound and a wide list): *redo is set — the caller passes batchg's overflow flag,
// so the exact dense pass queued behind does the batch again.
template <int NV>
__global__ __launch_bounds__(1024) void cand_select_kernel(const MergeArgs m0, uint32_t *redo) {
    __shared__ uint32_t offs[1025];
    __shared__ uint32_t sh[264];
    __shared__ uint64_t surv_k[256 + 64];
    __shared__ float surv_s[256 + 64];
    __shared__ uint32_t s_n;
    const uint32_t tid = threadIdx.x, q = blockIdx.x;
    const uint32_t k = m0.k, slots = m0.n_lists * k, n_seg = slots / m0.seg_len;   // n_seg <= 1024 (checked by the launcher)
    const uint32_t *counts = m0.seg_counts + (size_t)q * n_seg;
    const uint64_t *keys = m0.part_keys + (size_t)q * slots;
    const float *sims = m0.part_sims + (size_t)q * slots;
    // exclusive prefix sum of the segment counts (Hillis-Steele over <= 1024 values)
    uint32_t c = tid < n_seg ? counts[tid] : 0u;
    if (c > m0.seg_len) c = m0.seg_len;
    offs[tid + 1u > 1024u ? 1024u : tid + 1u] = c;
    if (tid == 0) { offs[0] = 0u; s_n = 0u; }
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t v = (tid + 1u > d) ? offs[tid + 1u - d] : 0u;
        __syncthreads();
        if (tid + 1u > d) offs[tid + 1u] += v;
        __syncthreads();
    }
    const uint32_t total = offs[n_seg < 1024u ? n_seg : 1024u];
    if (total > (uint32_t)NV * 1024u) {   // uniform
        if (tid == 0) *redo = 1u;
        return;
    }
    uint64_t key[NV];
    float sim[NV];
#pragma unroll
    for (int u = 0; u < NV; u++) {
        const uint32_t ci = tid + (uint32_t)u * 1024u;
        key[u] = 0ull;
        sim[u] = 0.0f;
        if (ci < total) {
            uint32_t lo = 0, hi = n_seg;   // the segment with offs[seg] <= ci < offs[seg + 1]
            while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (offs[mid] <= ci) lo = mid; else hi = mid; }
            const size_t at = (size_t)lo * m0.seg_len + (ci - offs[lo]);
            key[u] = keys[at];
            sim[u] = sims[at];
        }
    }
    const uint64_t t = block_select_kth<NV>(key, k, 56, 0, sh);   // 0: fewer than k entries — all of them are results
    const uint64_t low = t ? t : 1ull;
#pragma unroll
    for (int u = 0; u < NV; u++)
        if (key[u] >= low && key[u] != 0ull) {   // keys are unique (the row is part of the key): exactly min(k, total) survivors
            const uint32_t pos = atomicAdd(&s_n, 1u);
            if (pos < 256u + 64u) { surv_k[pos] = key[u]; surv_s[pos] = sim[u]; }
        }
    __syncthreads();
    const uint32_t S = s_n < 256u + 64u ? s_n : 256u + 64u;
    uint32_t *out_rows = m0.out_rows + (size_t)q * k;
    float *out_scores = m0.out_scores + (size_t)q * k, *out_dists = m0.out_dists + (size_t)q * k;
    for (uint32_t i = tid; i < S; i += 1024u) {
        const uint64_t ki = surv_k[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < S; j++) rank += surv_k[j] > ki ? 1u : 0u;
        if (rank < k) {
            const float dist = distance_of(surv_s[i]);
            out_rows[rank] = key_row(ki);
            out_dists[rank] = dist;
            out_scores[rank] = score_of(dist);
        }
    }
    if (tid == 0) m0.out_count[q] = S < k ? S : k;
}

// the candidate lists of nq queries -> top k each; *d_redo (not cleared here) is set when a query has more live entries
// than the register path holds: the caller redoes the batch some other way
int launch_cand_select(const MergeArgs &m, uint32_t nq, uint32_t *d_redo, hipStream_t stream) {
    if (!nq || !m.k) return CX_OK;
    if (!m.seg_counts || !m.seg_len || !d_redo) return set_err(CX_ERR_VALIDATION, "cand_select: segmented lists expected");
    const uint32_t n_seg = m.n_lists * m.k / m.seg_len;
    if (n_seg > 1024u || m.k > 256u) return launch_merge_batch(m, nq, stream, false);
    // expected live entries per query: k x the sample step (64 up to k = 32, 32 beyond): 2k / 3.2k (k = 100) / 8k (k = 256)
    if (m.k <= 32) hipLaunchKernelGGL(cand_select_kernel<4>, dim3(nq), dim3(1024), 0, stream, m, d_redo);
    else if (m.k <= 128) hipLaunchKernelGGL(cand_select_kernel<8>, dim3(nq), dim3(1024), 0, stream, m, d_redo);
    else hipLaunchKernelGGL(cand_select_kernel<16>, dim3(nq), dim3(1024), 0, stream, m, d_redo);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// SEQ = false: shard p's rows are local, part_base[p] makes them global and ties resolve by (part, slot) — parts are
// ordered by row range (the multi-process path, sharded.py).  SEQ = true (the single-process sharded index,
// sharded.cpp): rows already ARE global insertion sequence numbers (< 2^32, unique), shards interleave, so the key's
// row field holds the sequence number itself — ties resolve by insertion order exactly as in one index — and the
// payload carries the candidate's index instead of its distance; out_rows is then u32.
template <int KS, bool SEQ>
__global__ __launch_bounds__(256) void merge_parts_kernel(uint32_t n_parts, uint32_t nq, uint32_t k, uint64_t lstride, uint64_t cstride,
                                                          const PartBase part_base, const uint32_t *rows,
                                                          const float *scores, const float *dists,
                                                          const uint32_t *counts, void *out_rows_v,
                                                          float *out_scores, float *out_dists,
                                                          uint32_t *out_counts) {
    // a 64-bit global row does not fit the 32-bit row field of a key, so the
    // key's row field holds the candidate's index (part*k + slot): parts are
    // ordered by row range and slots by row within equal scores, hence the
    // index order equals the global row order among ties.
    __shared__ uint64_t sk[4][64 * KS];
    __shared__ float ss[4][64 * KS];
    const uint32_t qi = blockIdx.x;
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t total = n_parts * k;
    WaveTopK<KS> top;
    top.init(k);
    for (uint32_t i0 = wave * 64u; i0 < total; i0 += 4u * 64u) {
        const uint32_t i = i0 + lane;
        uint64_t kg = 0ull;
        float sg = 0.0f;
        if (i < total) {
            const uint32_t p = i / k, slot = i % k;
            if (slot < counts[(size_t)p * cstride + qi]) {
                const size_t src = (size_t)p * lstride + (size_t)qi * k + slot;
                if (SEQ) {
                    kg = make_key(scores[src], rows[src]);
                    sg = __uint_as_float(i);   // payload: where the candidate sits (only ever moved, never computed with)
                } else {
                    kg = make_key(scores[src], i);
                    sg = dists[src];  // payload: the shard's distance, carried through unchanged
                }
            }
        }
        top.offer_lanes(kg, sg, [](uint32_t) { return true; });
    }
    if (wave > 0) top.store(sk[wave], ss[wave]);
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 4; w++) {
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const uint32_t i = (uint32_t)s * 64u + lane;
                const uint64_t kg = i < k ? sk[w][i] : 0ull;
                const float sg = i < k ? ss[w][i] : 0.0f;
                top.offer_lanes(kg, sg, [](uint32_t) { return true; });
            }
        }
        uint32_t count = 0;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const uint32_t i = (uint32_t)s * 64u + lane;
            const uint64_t ki = top.key[s];
            const bool valid = i < k && ki != 0ull;
            if (valid) {
                const uint32_t idx = SEQ ? __float_as_uint(top.sim[s]) : key_row(ki);
                const uint32_t p = idx / k, slot = idx % k;
                const size_t src = (size_t)p * lstride + (size_t)qi * k + slot;
                if (SEQ) static_cast<uint32_t *>(out_rows_v)[(size_t)qi * k + i] = key_row(ki);
                else static_cast<uint64_t *>(out_rows_v)[(size_t)qi * k + i] = part_base.base[p] + rows[src];
                out_scores[(size_t)qi * k + i] = scores[src];
                out_dists[(size_t)qi * k + i] = SEQ ? dists[src] : top.sim[s];
            }
            count += (uint32_t)__popcll(__ballot(valid));
        }
        if (lane == 0) out_counts[qi] = count;
    }
}

static int launch_merge_parts_impl(bool seq, uint32_t n_parts, uint32_t nq, uint32_t k, uint64_t part_stride, const PartBase &part_base,
                                   const uint32_t *d_rows, const float *d_scores, const float *d_dists,
                                   const uint32_t *d_counts, void *out_rows, float *out_scores, float *out_dists,
                                   uint32_t *out_counts, hipStream_t stream) {
    if (k > TOPK_MAX) return set_err(CX_ERR_VALIDATION, "merge: k=%u exceeds %u", k, TOPK_MAX);
    if (nq == 0 || k == 0) return CX_OK;
    const uint64_t lstride = part_stride ? part_stride : (uint64_t)nq * k;
    const uint64_t cstride = part_stride ? part_stride : (uint64_t)nq;
#define CX_MP(KS_, SEQ_) hipLaunchKernelGGL((merge_parts_kernel<KS_, SEQ_>), dim3(nq), dim3(256), 0, stream, n_parts, nq, k, lstride, cstride, \
                                            part_base, d_rows, d_scores, d_dists, d_counts, out_rows, out_scores,       \
                                            out_dists, out_counts)
    if (seq) {
        if (k <= 64) CX_MP(1, true);
        else if (k <= 128) CX_MP(2, true);
        else CX_MP(4, true);
    } else {
        if (k <= 64) CX_MP(1, false);
        else if (k <= 128) CX_MP(2, false);
        else CX_MP(4, false);
    }
#undef CX_MP
    CX_HIP(hipGetLastError());
    return CX_OK;
}

int launch_merge_parts(uint32_t n_parts, uint32_t nq, uint32_t k, uint64_t part_stride, const PartBase &part_base,
                       const uint32_t *d_rows, const float *d_scores, const float *d_dists,
                       const uint32_t *d_counts, uint64_t *out_rows, float *out_scores, float *out_dists,
                       uint32_t *out_counts, hipStream_t stream) {
    return launch_merge_parts_impl(false, n_parts, nq, k, part_stride, part_base, d_rows, d_scores, d_dists, d_counts, out_rows,
                                   out_scores, out_dists, out_counts, stream);
}

int launch_merge_parts_seq(uint32_t n_parts, uint32_t nq, uint32_t k, uint64_t part_stride, const uint32_t *d_seq_rows,
                           const float *d_scores, const float *d_dists, const uint32_t *d_counts, uint32_t *out_seq_rows,
                           float *out_scores, float *out_dists, uint32_t *out_counts, hipStream_t stream) {
    PartBase pb;
    memset(&pb, 0, sizeof pb);
    return launch_merge_parts_impl(true, n_parts, nq, k, part_stride, pb, d_seq_rows, d_scores, d_dists, d_counts, out_seq_rows,
                                   out_scores, out_dists, out_counts, stream);
}

// One shard's lists of a call -> its part of the root's gather buffer (possibly peer memory: posted xGMI writes),
// local rows translated to global insertion sequence numbers on the way.  Part layout = sharded.py's packed chunk:
// rows[nq*k] | scores[nq*k] | dists[nq*k] | counts[nq].
// The shard's lists are k_src wide (its own k_eff), the part k_dst wide (the call's k).
__global__ __launch_bounds__(256) void publish_part_kernel(const uint32_t *rows, const float *scores, const float *dists,
                                                           const uint32_t *counts, const uint32_t *gseq, uint32_t nq, uint32_t k_src,
                                                           uint32_t k_dst, uint32_t n_rows, uint32_t *dst) {
    const uint32_t n = nq * k_dst;
    uint32_t *d_rows = dst;
    float *d_scores = reinterpret_cast<float *>(dst + n), *d_dists = reinterpret_cast<float *>(dst + 2 * (size_t)n);
    uint32_t *d_counts = dst + 3 * (size_t)n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t q = i / k_dst, slot = i % k_dst;
        if (slot < k_src && slot < counts[q]) {
            const size_t src = (size_t)q * k_src + slot;
            const uint32_t r = rows[src];
            d_rows[i] = r < n_rows ? gseq[r] : 0xFFFFFFFFu;   // an impossible row stays impossible: the host check reports it
            d_scores[i] = scores[src];
            d_dists[i] = dists[src];
        }
    }
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x)
        d_counts[q] = counts[q] <= k_src ? counts[q] : 0xFFFFFFFFu;
}

int launch_publish_part(const uint32_t *rows, const float *scores, const float *dists, const uint32_t *counts,
                        const uint32_t *gseq, uint32_t nq, uint32_t k_src, uint32_t k_dst, uint32_t n_rows, uint32_t *dst,
                        hipStream_t stream) {
    if (!nq) return CX_OK;
    const uint32_t n = std::max(nq * k_dst, nq);
    const uint32_t grid = std::min<uint32_t>((n + 255u) / 256u, 1024u);
    hipLaunchKernelGGL(publish_part_kernel, dim3(grid), dim3(256), 0, stream, rows, scores, dists, counts, gseq, nq, k_src, k_dst,
                       n_rows, dst);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// rows src_rows[i] of this shard -> vector dst_pos[i] of a query block that may live on another device
template <typename S>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const S *src, float *dst, const uint32_t *src_rows,
                                                           const uint32_t *dst_pos, uint32_t n, uint32_t dim) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t r = wave; r < n; r += n_waves) {
        const S *s = src + (size_t)src_rows[r] * dim;
        float *d = dst + (size_t)dst_pos[r] * dim;
        for (uint32_t j = lane; j < dim; j += 64u) d[j] = ldf(s + j);
    }
}

int launch_scatter_rows(const float *src, float *dst, const uint32_t *d_src_rows, const uint32_t *d_dst_pos, uint32_t n,
                        uint32_t dim, hipStream_t stream) {
    if (!n || !dim) return CX_OK;
    const uint32_t blocks = std::min<uint32_t>((n + 3u) / 4u, 2048u);
    hipLaunchKernelGGL(scatter_rows_kernel<float>, dim3(blocks), dim3(256), 0, stream, src, dst, d_src_rows, d_dst_pos, n, dim);
    CX_HIP(hipGetLastError());
    return CX_OK;
}
int launch_scatter_rows(const uint16_t *src, float *dst, const uint32_t *d_src_rows, const uint32_t *d_dst_pos, uint32_t n,
                        uint32_t dim, hipStream_t stream) {
    if (!n || !dim) return CX_OK;
    const uint32_t blocks = std::min<uint32_t>((n + 3u) / 4u, 2048u);
    hipLaunchKernelGGL(scatter_rows_kernel<uint16_t>, dim3(blocks), dim3(256), 0, stream, src, dst, d_src_rows, d_dst_pos, n, dim);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// ------------------------------------------------------- row maintenance

// src_rows == null: rows 0 .. n_dst in order (a conversion of a contiguous block)
template <typename S, typename D>
__global__ __launch_bounds__(256) void gather_rows_kernel(const S *src, D *dst, const uint32_t *src_rows,
                                                          uint32_t n_dst, uint32_t dim) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t r = wave; r < n_dst; r += n_waves) {
        const S *s = src + (size_t)(src_rows ? src_rows[r] : r) * dim;
        D *d = dst + (size_t)r * dim;
        for (uint32_t j = lane; j < dim; j += 64u) stf(d + j, ldf(s + j));
    }
}

template <typename S>
__global__ __launch_bounds__(256) void row_norms_kernel(const S *rows, float *norms, uint32_t row_lo, uint32_t row_hi, uint32_t dim, uint32_t *lossy) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t r = row_lo + wave; r < row_hi; r += n_waves) {
        const S *p = rows + (size_t)r * dim;
        float s = 0.0f;
        bool nz = false;
        for (uint32_t c = lane; c < dim; c += 64u) { const float x = ldf(p + c); s = fmaf(x, x, s); nz = nz || x != 0.0f; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) norms[r] = s;
        // (s is the same in every lane now) squares that all vanished under a row that is not zero: internal.hpp, norms_lossy
        if (s == 0.0f && __ballot(nz) != 0ull && lane == 0 && lossy) atomicOr(lossy, 1u);
    }
}

int launch_row_norms(const float *rows, float *norms, uint32_t row_lo, uint32_t row_hi, uint32_t dim, uint32_t *lossy, hipStream_t stream) {
    if (row_hi <= row_lo) return CX_OK;
    const uint32_t n = row_hi - row_lo;
    const uint32_t blocks = n / 4u + 1u < 4096u ? n / 4u + 1u : 4096u;
    hipLaunchKernelGGL(row_norms_kernel<float>, dim3(blocks), dim3(256), 0, stream, rows, norms, row_lo, row_hi, dim, lossy);
    CX_HIP(hipGetLastError());
    return CX_OK;
}
int launch_row_norms(const uint16_t *rows, float *norms, uint32_t row_lo, uint32_t row_hi, uint32_t dim, uint32_t *lossy, hipStream_t stream) {
    if (row_hi <= row_lo) return CX_OK;
    const uint32_t n = row_hi - row_lo;
    const uint32_t blocks = n / 4u + 1u < 4096u ? n / 4u + 1u : 4096u;
    hipLaunchKernelGGL(row_norms_kernel<uint16_t>, dim3(blocks), dim3(256), 0, stream, rows, norms, row_lo, row_hi, dim, lossy);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

__global__ __launch_bounds__(256) void scatter_lists_kernel(const uint32_t *src_rows, const float *src_scores, const float *src_dists,
                                                            const uint32_t *src_cnt, const uint32_t *pos, uint32_t k_src,
                                                            uint32_t k_dst, uint32_t *dst_rows, float *dst_scores,
                                                            float *dst_dists, uint32_t *dst_cnt) {
    const uint32_t i = blockIdx.x, p = pos[i], c = src_cnt[i] < k_src ? src_cnt[i] : k_src;
    for (uint32_t e = threadIdx.x; e < c && e < k_dst; e += 256u) {
        dst_rows[(size_t)p * k_dst + e] = src_rows[(size_t)i * k_src + e];
        dst_scores[(size_t)p * k_dst + e] = src_scores[(size_t)i * k_src + e];
        if (dst_dists) dst_dists[(size_t)p * k_dst + e] = src_dists[(size_t)i * k_src + e];
    }
    if (threadIdx.x == 0) dst_cnt[p] = c < k_dst ? c : k_dst;
}

int launch_scatter_lists(const uint32_t *src_rows, const float *src_scores, const float *src_dists, const uint32_t *src_cnt,
                         const uint32_t *d_pos, uint32_t n, uint32_t k_src, uint32_t k_dst, uint32_t *dst_rows,
                         float *dst_scores, float *dst_dists, uint32_t *dst_cnt, hipStream_t stream) {
    if (!n) return CX_OK;
    hipLaunchKernelGGL(scatter_lists_kernel, dim3(n), dim3(256), 0, stream, src_rows, src_scores, src_dists, src_cnt, d_pos, k_src,
                       k_dst, dst_rows, dst_scores, dst_dists, dst_cnt);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

template <typename S, typename D>
static int gather_rows_t(const S *src, D *dst, const uint32_t *d_src_rows, uint32_t n_dst, uint32_t dim, hipStream_t stream) {
    if (!n_dst) return CX_OK;
    uint32_t grid = (n_dst + 3u) / 4u;
    if (grid > 4096u) grid = 4096u;
    hipLaunchKernelGGL((gather_rows_kernel<S, D>), dim3(grid), dim3(256), 0, stream, src, dst, d_src_rows, n_dst, dim);
    CX_HIP(hipGetLastError());
    return CX_OK;
}
int launch_gather_rows(const float *src, float *dst, const uint32_t *d_src_rows, uint32_t n_dst, uint32_t dim, hipStream_t stream) {
    return gather_rows_t(src, dst, d_src_rows, n_dst, dim, stream);
}
// bf16 stores: rows out as f32 (query blocks, save, copies to the caller), rows moved inside the store, f32 rows in
int launch_gather_rows(const uint16_t *src, float *dst, const uint32_t *d_src_rows, uint32_t n_dst, uint32_t dim, hipStream_t stream) {
    return gather_rows_t(src, dst, d_src_rows, n_dst, dim, stream);
}
int launch_gather_rows(const uint16_t *src, uint16_t *dst, const uint32_t *d_src_rows, uint32_t n_dst, uint32_t dim, hipStream_t stream) {
    return gather_rows_t(src, dst, d_src_rows, n_dst, dim, stream);
}
int launch_gather_rows(const float *src, uint16_t *dst, const uint32_t *d_src_rows, uint32_t n_dst, uint32_t dim, hipStream_t stream) {
    return gather_rows_t(src, dst, d_src_rows, n_dst, dim, stream);
}


// tau_ord[q] = score_ord of the k-th entry of query q's ordered list (0 = no bound: fewer than k entries)
__global__ void tau_from_lists_kernel(const float *scores, const uint32_t *counts, uint32_t nq, uint32_t k, uint32_t *tau_ord) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) tau_ord[q] = counts[q] >= k ? score_ord(scores[(size_t)q * k + (k - 1u)]) : 0u;
}
int launch_tau_from_lists(const float *d_scores, const uint32_t *d_counts, uint32_t nq, uint32_t k, uint32_t *tau_ord, hipStream_t stream) {
    if (!nq) return CX_OK;
    hipLaunchKernelGGL(tau_from_lists_kernel, dim3((nq + 63u) / 64u), dim3(64), 0, stream, d_scores, d_counts, nq, k, tau_ord);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// dst[idx[i]] = val ? val[i] : value  (a handful of rows of a per-row array: the dedup pass's dense rows)
__global__ void patch_u32_kernel(uint32_t *dst, const uint32_t *idx, const uint32_t *val, uint32_t value, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = val ? val[i] : value;
}
int launch_patch_u32(uint32_t *dst, const uint32_t *d_idx, const uint32_t *d_val, uint32_t value, uint32_t n, hipStream_t stream) {
    if (!n) return CX_OK;
    hipLaunchKernelGGL(patch_u32_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, dst, d_idx, d_val, value, n);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// segment t of (to, w) [src_off[t], src_off[t + 1]) -> out_* at dst_off[seg_pos[t]] ..., out_from = from_row[t]: one block per segment
__global__ __launch_bounds__(256) void copy_edge_segments_kernel(const uint64_t *dst_off, const uint32_t *seg_pos, const uint64_t *src_off,
                                                                 const uint32_t *from_row, const uint32_t *to, const float *w, uint32_t *out_from,
                                                                 uint32_t *out_to, float *out_w) {
    const uint32_t t = blockIdx.x;
    const uint64_t lo = src_off[t], n = src_off[t + 1] - lo, base = dst_off[seg_pos[t]];
    for (uint64_t e = threadIdx.x; e < n; e += blockDim.x) {
        out_from[base + e] = from_row[t];
        out_to[base + e] = to[lo + e];
        out_w[base + e] = w[lo + e];
    }
}
int launch_copy_edge_segments(const uint64_t *d_dst_off, const uint32_t *d_seg_pos, const uint64_t *d_src_off, const uint32_t *d_from_row,
                              const uint32_t *d_to, const float *d_w, uint32_t n_seg, uint32_t *out_from, uint32_t *out_to, float *out_w,
                              hipStream_t stream) {
    if (!n_seg) return CX_OK;
    hipLaunchKernelGGL(copy_edge_segments_kernel, dim3(n_seg), dim3(256), 0, stream, d_dst_off, d_seg_pos, d_src_off, d_from_row, d_to, d_w, out_from,
                       out_to, out_w);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

}  // namespace cx
