// allpairs.hip — the auto-linker's similarity pass as one dense contraction
// (BASELINE config 3).
//
// Replaces the per-node loop of AutoLinker::run_cycle (linker/auto_linker.rs:
// 215-264: one search(emb, 100) per scanned node, each an O(N*d) scan) and
// DedupScanner::scan (linker/dedup.rs:65-127: one search_threshold per node).
// Over n_scan x N pairs that is a Q x C^T GEMM — the one place on this path
// where MFMA is the right unit:
//
//  1. build_shadow_kernel   rows -> L2-normalised bf16 shadow [rows][dim]
//  2. pair_filter_kernel    bf16 MFMA 32x32x16 GEMM over 128x128 tiles whose
//                           epilogue never writes the score matrix: it only
//                           appends column j to row i's candidate list when
//                           approx_cos >= thr - eps.  eps bounds the bf16 error
//                           rigorously (|x^.y^ - cos| <= 2u(1+u) + f32 sum slack,
//                           u = 2^-8), so no pair with exact score >= thr is lost.
//  3. rescore_kernel        exact f32 cosine of the sparse survivors with the
//                           scan kernel's arithmetic, exact threshold, ordered
//                           top-k per scanned row (register top-k lists).
//  4. link_count/emit       the reference's walk over each ordered list: skip
//                           self, skip storage-deleted neighbours, score >= thr
//                           -> edge, stop at max_edges_per_node.
//
// MFMA-bound: 2*n_scan*N*dim flops; the bf16 shadow (154 MB at 100k x 768) sits
// in the 256 MiB Infinity Cache, tiles are ordered so that an XCD's L2 holds
// the 8+8 panels its concurrent blocks share.
#include "kernels.hpp"
#include "topk.hpp"

namespace cx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline uint16_t f32_to_bf16_rne(float x) {  // finite inputs only
    uint32_t u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// ---------------------------------------------------------------- 1. shadow

// err_max (optional): the largest rounding error of a row, || bf16(x) - x || for the normalised row x, as the bits of a
// positive f32 (atomic max): batchs.hip's screening bound uses the error the shadow really has instead of the worst case.
// irr (optional): IRREGULAR rows (kernels.hpp: bs_regular — |x|^2 zero, non-finite or outside [1e-30, 1e30] in the reference's f32
// arithmetic: it scores such a row 0, 1 or NaN, vector/index.rs:172-177) get a zero shadow row — no screening pass ever lists
// them — and go on the index's irregular list instead: irr.cnt[0] counts them (it may pass BS_IRR_CAP: the host then switches
// the screening paths off), irr.rows holds the first BS_IRR_CAP.  A row rebuilt in place (upsert of a known id) is not listed
// twice: entries [0, irr.n_before) — those of earlier, completed launches — are searched first; rows of one launch are distinct.
struct IrrList { uint32_t *cnt; uint32_t *rows; uint32_t n_before; };
template <typename S>
__global__ __launch_bounds__(256) void build_shadow_kernel(const S *rows, uint16_t *shadow, uint32_t row_lo,
                                                           uint32_t row_hi, uint32_t dim, uint32_t tiled, uint32_t *err_max, const IrrList irr) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t r = row_lo + wave; r < row_hi; r += n_waves) {
        const S *p = rows + (size_t)r * dim;
        float ss = 0.0f;
        {   // |x|^2 exactly as batchs_rescore_kernel sums it (lane-strided, separately rounded products, wave_sum): the two must
            // agree on which rows are irregular, to the bit
#pragma clang fp contract(off)
            for (uint32_t j = lane; j < dim; j += 64u) { const float x = ldf(p + j); ss += x * x; }
        }
        ss = wave_sum(ss);
        const bool regular = bs_regular(ss);
        const float inv = regular ? 1.0f / sqrtf(ss) : 0.0f;  // irregular rows -> zero shadow (never a candidate of a screening pass)
        uint16_t *o = shadow + (size_t)r * dim;
        float es = 0.0f;
        for (uint32_t j = lane; j < dim; j += 64u) {
            const float v = regular ? ldf(p + j) * inv : 0.0f;   // (|v| <= 1: every element of a regular row is finite)
            const uint16_t b = f32_to_bf16_rne(v);
            const float e = v - bf16_bits_to_f32(b);
            es += e * e;
            if (tiled) shadow[tiled_shadow_off(r, j >> 3, dim / 32u) + (j & 7u)] = b;
            else o[j] = b;
        }
        if (err_max) {
            es = wave_sum(es);
            if (lane == 0u) __hip_atomic_fetch_max(err_max, __float_as_uint(sqrtf(es) * (1.0f + 1.0e-5f)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!regular && irr.cnt) {
            bool listed = false;
            const uint32_t nb = irr.n_before < BS_IRR_CAP ? irr.n_before : BS_IRR_CAP;
            for (uint32_t i = lane; i < nb; i += 64u) listed = listed || irr.rows[i] == r;
            if (!__ballot(listed) && lane == 0u) {
                const uint32_t pos = atomicAdd(irr.cnt, 1u);
                if (pos < BS_IRR_CAP) irr.rows[pos] = r;
            }
        }
    }
}

int launch_build_shadow(const float *rows, uint16_t *shadow, uint32_t row_lo, uint32_t row_hi, uint32_t dim,
                        hipStream_t stream) {
    if (row_hi <= row_lo) return CX_OK;
    uint32_t grid = (row_hi - row_lo + 3u) / 4u;
    if (grid > 8192u) grid = 8192u;
    hipLaunchKernelGGL(build_shadow_kernel<float>, dim3(grid), dim3(256), 0, stream, rows, shadow, row_lo, row_hi, dim, 0u, nullptr, IrrList{nullptr, nullptr, 0u});
    CX_HIP(hipGetLastError());
    return CX_OK;
}
int launch_build_shadow(const uint16_t *rows16, uint16_t *shadow, uint32_t row_lo, uint32_t row_hi, uint32_t dim,
                        hipStream_t stream) {
    if (row_hi <= row_lo) return CX_OK;
    uint32_t grid = (row_hi - row_lo + 3u) / 4u;
    if (grid > 8192u) grid = 8192u;
    hipLaunchKernelGGL(build_shadow_kernel<uint16_t>, dim3(grid), dim3(256), 0, stream, rows16, shadow, row_lo, row_hi, dim, 0u, nullptr, IrrList{nullptr, nullptr, 0u});
    CX_HIP(hipGetLastError());
    return CX_OK;
}
int launch_build_shadow_index(const float *rows, const uint16_t *rows16, uint16_t *shadow, bool tiled, uint32_t row_lo, uint32_t row_hi, uint32_t dim,
                              hipStream_t stream, uint32_t *err_max, uint32_t *irr_cnt, uint32_t *irr_rows, uint32_t irr_n_before) {
    if (row_hi <= row_lo) return CX_OK;
    if (tiled && dim % 32u) return set_err(CX_ERR_VALIDATION, "tiled shadow needs dim %% 32 == 0 (got %u)", dim);
    uint32_t grid = (row_hi - row_lo + 3u) / 4u;
    if (grid > 8192u) grid = 8192u;
    const IrrList irr{irr_cnt, irr_rows, irr_n_before};
    if (rows16) hipLaunchKernelGGL(build_shadow_kernel<uint16_t>, dim3(grid), dim3(256), 0, stream, rows16, shadow, row_lo, row_hi, dim, tiled ? 1u : 0u, tiled ? err_max : nullptr, irr);
    else hipLaunchKernelGGL(build_shadow_kernel<float>, dim3(grid), dim3(256), 0, stream, rows, shadow, row_lo, row_hi, dim, tiled ? 1u : 0u, tiled ? err_max : nullptr, irr);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// ---- irregular rows and the filter (kernels.hpp: bs_regular).  Their shadow rows are zero, so no filter kernel ever returns them — as
// a neighbour or for their own scan.  The reference computes every pair with them all the same (index.rs:172-177: a row whose
// squares underflow scores 1.0 against everything it has a positive dot with), so after the filter of a pass
//  - every scanned row that IS irregular is marked as run over (its list comes from the exact path, like any row the filter
//    could not serve), and
//  - the store's irregular rows are appended to the candidates of every other scanned row, to be scored exactly with the rest.
// irr_now_kernel: which entries of the index's irregular list are irregular NOW (an upsert may have replaced the row: it would
// be a candidate twice) — the build's own sum, one wave per entry.
template <typename S>
__global__ __launch_bounds__(256) void irr_now_kernel(const S *rows, uint32_t dim, uint32_t n_rows, const uint32_t *irr_rows, uint32_t irr_n, uint32_t *irr_ok) {
    const uint32_t e = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (e >= irr_n) return;
    const uint32_t r = irr_rows[e];
    float ss = 1.0f;
    if (r < n_rows) {
#pragma clang fp contract(off)
        const S *p = rows + (size_t)r * dim;
        ss = 0.0f;
        for (uint32_t j = lane; j < dim; j += 64u) { const float x = ldf(p + j); ss += x * x; }
    }
    ss = wave_sum(ss);
    if (lane == 0u) irr_ok[e] = (r < n_rows && !bs_regular(ss)) ? 1u : 0u;
}
// irr_append_kernel: the irregular rows behind every scanned row's candidates (a list that runs over its cap with them is redone on
// the exact path like any other).  irr_mark_kernel, AFTER the rescore (which writes every row's overflow flag itself): the scanned
// vectors that are irregular THEMSELVES are flagged as run over — a flag, not a count: the entries of a list are only ever the ones a
// kernel wrote.  ext_vecs: the scanned vectors are not rows of this shard ([n_scan][dim] f32: the sharded pass's external blocks) —
// "is the scanned vector irregular" is then decided from the vector (its own shadow was built zero just the same).
__global__ __launch_bounds__(256) void irr_append_kernel(const uint32_t *irr_rows, const uint32_t *irr_ok, uint32_t irr_n, uint32_t n_scan,
                                                         uint32_t *cand_cnt, uint32_t *cand, uint32_t cap) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (i >= n_scan) return;
    const uint32_t cnt = cand_cnt[i];
    uint32_t n_ok = 0;
    for (uint32_t e0 = 0; e0 < irr_n; e0 += 64u) {
        const uint32_t e = e0 + lane;
        const bool ok = e < irr_n && irr_ok[e] != 0u;
        const uint32_t r = ok ? irr_rows[e] : 0u;
        const uint64_t m = __ballot(ok);
        const uint32_t pos = cnt + n_ok + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (ok && pos < cap) cand[(size_t)i * cap + pos] = r;   // (the row itself included: the rules drop self pairs like any other)
        n_ok += (uint32_t)__popcll(m);
    }
    if (lane == 0u && n_ok) cand_cnt[i] = cnt + n_ok;
}
__global__ __launch_bounds__(256) void irr_mark_kernel(const uint32_t *irr_rows, const uint32_t *irr_ok, uint32_t irr_n, const uint32_t *scan_rows, const float *ext_vecs,
                                                       uint32_t dim, uint32_t n_scan, uint32_t *overflow) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (i >= n_scan) return;
    bool self = false;
    if (ext_vecs) {
#pragma clang fp contract(off)
        float ss = 0.0f;
        for (uint32_t j = lane; j < dim; j += 64u) { const float x = ext_vecs[(size_t)i * dim + j]; ss += x * x; }
        self = !bs_regular(wave_sum(ss));
    } else {
        const uint32_t me = scan_rows ? scan_rows[i] : i;
        for (uint32_t e = lane; e < irr_n; e += 64u) self = self || (irr_ok[e] != 0u && irr_rows[e] == me);
    }
    const bool any_self = __ballot(self) != 0ull;
    if (lane == 0u && any_self) overflow[i] = 1u;
}
int launch_irr_append(const float *rows, const uint16_t *rows16, uint32_t dim, uint32_t n_rows, const uint32_t *irr_rows, uint32_t irr_n, uint32_t *irr_ok,
                      uint32_t n_scan, uint32_t *cand_cnt, uint32_t *cand, uint32_t cap, hipStream_t stream) {
    if (!n_scan || !irr_n) return CX_OK;
    if (rows16) hipLaunchKernelGGL(irr_now_kernel<uint16_t>, dim3((irr_n + 3u) / 4u), dim3(256), 0, stream, rows16, dim, n_rows, irr_rows, irr_n, irr_ok);
    else hipLaunchKernelGGL(irr_now_kernel<float>, dim3((irr_n + 3u) / 4u), dim3(256), 0, stream, rows, dim, n_rows, irr_rows, irr_n, irr_ok);
    hipLaunchKernelGGL(irr_append_kernel, dim3((n_scan + 3u) / 4u), dim3(256), 0, stream, irr_rows, irr_ok, irr_n, n_scan, cand_cnt, cand, cap);
    CX_HIP(hipGetLastError());
    return CX_OK;
}
int launch_irr_mark(const uint32_t *irr_rows, const uint32_t *irr_ok, uint32_t irr_n, const uint32_t *scan_rows, const float *ext_vecs, uint32_t dim, uint32_t n_scan,
                    uint32_t *overflow, hipStream_t stream) {
    if (!n_scan || (!irr_n && !ext_vecs)) return CX_OK;
    hipLaunchKernelGGL(irr_mark_kernel, dim3((n_scan + 3u) / 4u), dim3(256), 0, stream, irr_rows, irr_ok, irr_n, scan_rows, ext_vecs, dim, n_scan, overflow);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// ---------------------------------------------------------------- 2. filter GEMM

constexpr int BM = 128, BN = 128, BK = 64;          // block tile; BK bf16 = 128 B per row
constexpr int TILE_BYTES = BM * BK * 2;             // 16 KiB per operand per stage
constexpr int LDS_BYTES = 2 * 2 * TILE_BYTES;       // 2 stages x (A + B) = 64 KiB

// LDS image of an operand tile: 128 rows x 128 B; the 16-byte piece p of row r
// sits at piece position p ^ ((r >> 1) & 7), so the 16 rows a ds_read_b128 lane
// group touches at one logical piece fall on 16 different 16-byte bank groups.
// global_load_lds writes linearly (base + lane*16), so the permutation is
// applied to the per-lane SOURCE address and again on the read.
__device__ inline uint32_t lds_off(uint32_t r, uint32_t p) { return r * 128u + ((p ^ ((r >> 1) & 7u)) << 4); }

// SHAPE 32: mfma_f32_32x32x16_bf16 (2x2 tiles per wave).  SHAPE 16: mfma_f32_16x16x32_bf16 (4x4 tiles per
// wave) — same flops per LDS byte; the chip holds a higher clock on it (MI355X_MICROARCH.md, DVFS item 7).
template <int SHAPE>
__global__ __launch_bounds__(256) void pair_filter_kernel(const PairFilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t wm = wave >> 1, wn = wave & 1u;

    // tile order: XCD-contiguous chunks (blocks b, b+8, ... share an XCD), inside a
    // chunk walk 8 I-panels per J-panel so concurrent blocks share panels in L2
    const uint32_t tiles_i = (a.n_scan + BM - 1) / BM, tiles_j = (a.n_rows + BN - 1) / BN;
    const uint32_t T = a.symmetric ? a.n_tiles : tiles_i * tiles_j;
    uint32_t b = blockIdx.x;
    {
        const uint32_t q = T / 8u, r = T % 8u, xcd = b % 8u;
        b = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + b / 8u;
    }
    uint32_t ti, tj;
    if (a.symmetric) {  // cosine is symmetric: only tiles with tj >= ti exist, each emits both directions
        const uint32_t t = a.tile_list[b];
        ti = t >> 16;
        tj = t & 0xFFFFu;
    } else {
        const uint32_t GS = 8u;
        const uint32_t per_group = GS * tiles_j;
        const uint32_t group = b / per_group, first_i = group * GS;
        const uint32_t gsz = (tiles_i - first_i) < GS ? (tiles_i - first_i) : GS;
        ti = first_i + (b % per_group) % gsz;
        tj = (b % per_group) / gsz;
    }
    const uint32_t i0 = ti * BM, j0 = tj * BN;

    // loader: each wave-level LDS-DMA moves 8 rows x 128 B; 16 per operand tile, 4 per wave
    // The shard's rows come from the tiled shadow when there is one (dim % 32 == 0: the row-major copy is not kept): a
    // lane's 16 bytes of K-block kt (64 elements = two K-steps of the tiled layout) sit 2 KiB further per kt, and the 8
    // rows x 128 B of one instruction are two runs of 512 contiguous bytes instead of eight of 128.
    const uint32_t lrow = lane >> 3, lslot = lane & 7u;
    const uint16_t *srcA[4], *srcB[4];
    const bool tiled = a.shadow_t != nullptr;
    const uint32_t stepB = tiled ? 1024u : (uint32_t)BK, stepA = (tiled && !a.shadow_q) ? 1024u : (uint32_t)BK;   // elements per K-block
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t r = (wave * 4u + (uint32_t)q) * 8u + lrow;      // row inside the tile
        const uint32_t piece = lslot ^ ((r >> 1) & 7u);
        uint32_t gi = i0 + r;
        gi = gi < a.n_scan ? gi : a.n_scan - 1u;
        const uint32_t ga = a.scan_rows ? a.scan_rows[gi] : gi;
        uint32_t gb = j0 + r;
        gb = gb < a.n_rows ? gb : a.n_rows - 1u;
        if (a.shadow_q) srcA[q] = a.shadow_q + (size_t)ga * a.dim + piece * 8u;
        else if (tiled) srcA[q] = a.shadow_t + tiled_shadow_off(ga, piece, a.dim / 32u);
        else srcA[q] = a.shadow + (size_t)ga * a.dim + piece * 8u;
        srcB[q] = tiled ? a.shadow_t + tiled_shadow_off(gb, piece, a.dim / 32u) : a.shadow + (size_t)gb * a.dim + piece * 8u;
    }
    auto stage = [&](uint32_t buf, uint32_t kt) {
        char *A = smem + buf * 2 * TILE_BYTES, *B = A + TILE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t off = (wave * 4u + (uint32_t)q) * 1024u;  // wave-uniform LDS base of this DMA
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcA[q] + (size_t)kt * stepA),
                                             (__attribute__((address_space(3))) void *)(A + off), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcB[q] + (size_t)kt * stepB),
                                             (__attribute__((address_space(3))) void *)(B + off), 16, 0, 0);
        }
    };

    const uint32_t KT = a.dim / BK;
    auto emit = [&](uint32_t i, uint32_t j) {
        if (i < a.n_scan && j < a.n_rows) {
            const uint32_t slot = atomicAdd(a.cand_cnt + i, 1u);
            if (slot < a.cap) a.cand[(size_t)i * a.cap + slot] = j;
        }
    };
    const bool mirror = a.symmetric && ti != tj;

    if constexpr (SHAPE == 32) {
        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[m][n][e] = 0.0f;
        const uint32_t fr = lane & 31u, fh = lane >> 5;
        stage(0, 0);
        __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and makes it visible to all waves
        for (uint32_t kt = 0; kt < KT; kt++) {
            const uint32_t buf = kt & 1u;
            if (kt + 1 < KT) stage(buf ^ 1u, kt + 1);
            const char *A = smem + buf * 2 * TILE_BYTES, *B = A + TILE_BYTES;
#pragma unroll
            for (uint32_t ks = 0; ks < 4; ks++) {
                bf16x8 af[2], bf[2];
#pragma unroll
                for (uint32_t m = 0; m < 2; m++)
                    af[m] = *reinterpret_cast<const bf16x8 *>(A + lds_off(wm * 64u + m * 32u + fr, 2u * ks + fh));
#pragma unroll
                for (uint32_t n = 0; n < 2; n++)
                    bf[n] = *reinterpret_cast<const bf16x8 *>(B + lds_off(wn * 64u + n * 32u + fr, 2u * ks + fh));
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < 2; n++)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
            }
            __syncthreads();  // next stage landed; everyone is done reading this one
        }
        // C[row][col], col = lane & 31 (j), row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5) (i)
#pragma unroll
        for (uint32_t m = 0; m < 2; m++)
#pragma unroll
            for (uint32_t n = 0; n < 2; n++) {
                const uint32_t j = j0 + wn * 64u + n * 32u + fr;
#pragma unroll
                for (uint32_t e = 0; e < 16; e++) {
                    if (acc[m][n][e] >= a.thr_lo) {
                        const uint32_t i = i0 + wm * 64u + m * 32u + (e & 3u) + 8u * (e >> 2) + 4u * fh;
                        emit(i, j);
                        if (mirror) emit(j, i);
                    }
                }
            }
    } else {
        f32x4 acc[4][4];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int n = 0; n < 4; n++) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const uint32_t fr = lane & 15u, fq = lane >> 4;   // A/B: row fr of the 16-row block, k quarter fq (8 bf16)
        stage(0, 0);
        __syncthreads();
        for (uint32_t kt = 0; kt < KT; kt++) {
            const uint32_t buf = kt & 1u;
            if (kt + 1 < KT) stage(buf ^ 1u, kt + 1);
            const char *A = smem + buf * 2 * TILE_BYTES, *B = A + TILE_BYTES;
#pragma unroll
            for (uint32_t ks = 0; ks < 2; ks++) {       // two 32-deep k-steps per 64-deep stage
                bf16x8 af[4], bf[4];
#pragma unroll
                for (uint32_t m = 0; m < 4; m++)
                    af[m] = *reinterpret_cast<const bf16x8 *>(A + lds_off(wm * 64u + m * 16u + fr, 4u * ks + fq));
#pragma unroll
                for (uint32_t n = 0; n < 4; n++)
                    bf[n] = *reinterpret_cast<const bf16x8 *>(B + lds_off(wn * 64u + n * 16u + fr, 4u * ks + fq));
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int n = 0; n < 4; n++)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
            }
            __syncthreads();
        }
        // C[row][col], col = lane & 15 (j), row = 4*(lane >> 4) + e (i)
#pragma unroll
        for (uint32_t m = 0; m < 4; m++)
#pragma unroll
            for (uint32_t n = 0; n < 4; n++) {
                const uint32_t j = j0 + wn * 64u + n * 16u + fr;
#pragma unroll
                for (uint32_t e = 0; e < 4; e++) {
                    if (acc[m][n][e] >= a.thr_lo) {
                        const uint32_t i = i0 + wm * 64u + m * 16u + 4u * fq + e;
                        emit(i, j);
                        if (mirror) emit(j, i);
                    }
                }
            }
    }
}

void pair_filter_tile_list(uint32_t n_rows, std::vector<uint32_t> &out) {
    const uint32_t tiles = (n_rows + BM - 1) / BM, GS = 8;
    out.clear();
    out.reserve((size_t)tiles * (tiles + 1) / 2);
    for (uint32_t g0 = 0; g0 < tiles; g0 += GS) {          // 8 I-panels at a time ...
        const uint32_t g1 = g0 + GS < tiles ? g0 + GS : tiles;
        for (uint32_t tj = g0; tj < tiles; tj++)            // ... walking the J-panels to their right
            for (uint32_t ti = g0; ti < g1 && ti <= tj; ti++) out.push_back((ti << 16) | tj);
    }
}

int launch_pair_filter(const PairFilterArgs &a, hipStream_t stream) {
    if (a.symmetric && (!a.tile_list || (a.n_rows + BM - 1) / BM > 0xFFFFu))
        return set_err(CX_ERR_VALIDATION, "pair filter: symmetric pass needs a tile list and < 65536 tiles per side");
    if (a.dim % BK != 0 || a.dim == 0) return set_err(CX_ERR_VALIDATION, "pair filter needs dim %% 64 == 0 (got %u)", a.dim);
    if (!a.n_scan || !a.n_rows) return CX_OK;
    const uint64_t tiles = a.symmetric ? a.n_tiles : (uint64_t)((a.n_scan + BM - 1) / BM) * ((a.n_rows + BN - 1) / BN);
    if (tiles > 0x7FFFFFFFull) return set_err(CX_ERR_VALIDATION, "pair filter: too many tiles");
    static std::atomic<uint64_t> attr_devices{0};
    if (first_use_on_device(attr_devices)) {
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_filter_kernel<32>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_filter_kernel<16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    }
    static const int shape = getenv("CX_PAIR_MFMA") ? atoi(getenv("CX_PAIR_MFMA")) : 16;  // +5 % over 32x32x16 (profiles/r01)
    if (shape == 16) hipLaunchKernelGGL(pair_filter_kernel<16>, dim3((uint32_t)tiles), dim3(256), LDS_BYTES, stream, a);
    else hipLaunchKernelGGL(pair_filter_kernel<32>, dim3((uint32_t)tiles), dim3(256), LDS_BYTES, stream, a);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// ---------------------------------------------------------------- 3. rescore

// One wave per scanned row: exact cosine (the scan kernel's arithmetic: lane-
// strided f32 partials + butterfly, reference epilogue) of every candidate,
// exact threshold, ordered top-k.  Rows whose candidate list overflowed are
// flagged and redone by the caller on the exact scan path.
template <typename S> struct StoreOf;
template <> struct StoreOf<float> { static __device__ const float *get(const RescoreArgs &a) { return a.rows; } };
template <> struct StoreOf<uint16_t> { static __device__ const uint16_t *get(const RescoreArgs &a) { return a.rows16; } };

template <int KS, typename S>
__global__ __launch_bounds__(256) void rescore_kernel(const RescoreArgs a) {
    const S *rows = StoreOf<S>::get(a);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t dim = a.dim;
    const bool vec4 = (dim & 3u) == 0;
    for (uint32_t i = wave; i < a.n_scan; i += n_waves) {
        const uint32_t qrow = a.scan_rows ? a.scan_rows[i] : i;
        // the scanned vector: an external f32 vector, or a row of the store (either element type)
        auto q4 = [&](uint32_t j) -> f32x4 { return a.q_rows ? reinterpret_cast<const f32x4 *>(a.q_rows + (size_t)qrow * dim)[j] : row4(rows, qrow, dim, j); };
        auto q1 = [&](uint32_t j) -> float { return a.q_rows ? a.q_rows[(size_t)qrow * dim + j] : ldf(rows + (size_t)qrow * dim + j); };
        const uint32_t total = a.cand_cnt[i];
        const uint32_t cnt = total < a.cap ? total : a.cap;
        if (lane == 0) a.overflow[i] = total > a.cap ? 1u : 0u;
        float qq = 0.0f;
        if (vec4) {
            for (uint32_t j = lane; j < dim / 4u; j += 64u) {
                const f32x4 x = q4(j);
                qq = fmaf(x.w, x.w, fmaf(x.z, x.z, fmaf(x.y, x.y, fmaf(x.x, x.x, qq))));
            }
        } else {
            for (uint32_t j = lane; j < dim; j += 64u) { const float x = q1(j); qq = fmaf(x, x, qq); }
        }
        qq = wave_sum(qq);
        WaveTopK<KS> top;
        top.init(a.topk);
        // four candidates per step: their row loads and their reductions are independent, so the latency of
        // one candidate's butterfly hides under the others' (one at a time the kernel was bound by 12 dependent
        // cross-lane steps per candidate, not by the 13 TB/s of L2-served row reads)
        for (uint32_t c0 = 0; c0 < cnt; c0 += 4) {
            uint32_t jrow[4];
            float d0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, n0[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int u = 0; u < 4; u++) jrow[u] = a.cand[(size_t)i * a.cap + (c0 + u < cnt ? c0 + u : cnt - 1u)];
            if (vec4) {
                for (uint32_t j = lane; j < dim / 4u; j += 64u) {
                    const f32x4 y = q4(j);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const f32x4 x = row4(rows, jrow[u], dim, j);
                        // explicit fixed-order FMA chains: a pair's score must not depend on which of the four
                        // slots it landed in (left to the compiler, slots get different packed/scalar FMA trees)
                        d0[u] = fmaf(x.w, y.w, fmaf(x.z, y.z, fmaf(x.y, y.y, fmaf(x.x, y.x, d0[u]))));
                        n0[u] = fmaf(x.w, x.w, fmaf(x.z, x.z, fmaf(x.y, x.y, fmaf(x.x, x.x, n0[u]))));
                    }
                }
            } else {
                for (uint32_t j = lane; j < dim; j += 64u) {
                    const float y = q1(j);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float x = ldf(rows + (size_t)jrow[u] * dim + j);
                        d0[u] = fmaf(x, y, d0[u]);
                        n0[u] = fmaf(x, x, n0[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) { d0[u] = wave_sum(d0[u]); n0[u] = wave_sum(n0[u]); }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (c0 + u >= cnt) break;
                const float sim = cosine_from_sums(d0[u], qq, n0[u]);
                const float score = score_of(distance_of(sim));
                if (!(score >= a.threshold)) continue;              // rules.rs:50 / index.rs:386 (NaN fails)
                if (a.meta[jrow[u]] & META_REMOVED) continue;       // removed from the index
                const uint64_t key = make_key(score, jrow[u]);
                if (key > top.tau) top.insert(key, sim);
            }
        }
        // ordered list out
        uint32_t n_out = 0;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const uint32_t r = (uint32_t)s * 64u + lane;
            const bool valid = r < a.topk && top.key[s] != 0ull;
            if (valid) {
                a.out_rows[(size_t)i * a.topk + r] = key_row(top.key[s]);
                a.out_scores[(size_t)i * a.topk + r] = score_of(distance_of(top.sim[s]));
                if (a.out_dists) a.out_dists[(size_t)i * a.topk + r] = distance_of(top.sim[s]);
            }
            n_out += (uint32_t)__popcll(__ballot(valid));
        }
        if (lane == 0) a.out_cnt[i] = n_out;
    }
}

// ---- symmetric pass: each pair scored once ------------------------------------------------------------------
// The rescore is bound by bytes: 1.5M candidate rows x 3 KiB of scattered reads per 100k x 768 pass.  In the symmetric
// pass (i, j) is in list i exactly when (j, i) is in list j (the filter emits both from one accumulator; on diagonal
// tiles both accumulators hold the same sum), and the exact cosine is symmetric bit for bit: the same products in the
// same order, |q|^2 and |row|^2 summed by the same chain, and a commutative product of the two roots.  So row i sums
// only its entries j >= i — and those whose own list overflowed and will be redone — and writes the result into its
// own slot and into the slot list j holds for i.  A second kernel then orders each list from the stored cosines.
constexpr uint32_t PAIR_SIM_UNSET = 0xFFFFFFFFu;   // a NaN no cosine_from_sums result has: "nobody wrote this entry"

__global__ __launch_bounds__(256) void pair_sims_clear_kernel(const RescoreArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave; i < a.n_scan; i += n_waves) {
        const uint32_t total = a.cand_cnt[i], cnt = total < a.cap ? total : a.cap;
        for (uint32_t c = lane; c < cnt; c += 64u) reinterpret_cast<uint32_t *>(a.pair_sims)[(size_t)i * a.cap + c] = PAIR_SIM_UNSET;
    }
}

template <typename S>
__global__ __launch_bounds__(256) void pair_score_kernel(const RescoreArgs a) {
    const S *rows = StoreOf<S>::get(a);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t dim = a.dim;
    const bool vec4 = (dim & 3u) == 0;
    for (uint32_t i = wave; i < a.n_scan; i += n_waves) {
        const uint32_t total = a.cand_cnt[i];
        if (total > a.cap) continue;                       // overflowed: redone on the exact path, nobody waits for it
        const uint32_t cnt = total;
        float qq = 0.0f;
        if (vec4) {
            for (uint32_t j = lane; j < dim / 4u; j += 64u) {
                const f32x4 x = row4(rows, i, dim, j);
                qq = fmaf(x.w, x.w, fmaf(x.z, x.z, fmaf(x.y, x.y, fmaf(x.x, x.x, qq))));
            }
        } else {
            for (uint32_t j = lane; j < dim; j += 64u) { const float x = ldf(rows + (size_t)i * dim + j); qq = fmaf(x, x, qq); }
        }
        qq = wave_sum(qq);
        // 64 list entries at a time, one per lane; the ones this row has to sum are then taken four at a time (their
        // row reads and reductions overlap), so no step runs with idle slots
        for (uint32_t c0 = 0; c0 < cnt; c0 += 64u) {
            const uint32_t c = c0 + lane;
            const bool in = c < cnt;
            const uint32_t jl = in ? a.cand[(size_t)i * a.cap + c] : 0u;
            const bool partner_redone = in && a.cand_cnt[jl] > a.cap;
            uint64_t todo = __ballot(in && (jl >= i || partner_redone));
            const uint64_t mirrored = __ballot(in && jl > i && !partner_redone);
            while (todo) {
                uint32_t jrow[4], slot[4];
                bool need[4];
                float d0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, n0[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    need[u] = todo != 0ull;
                    const int l = need[u] ? __ffsll((unsigned long long)todo) - 1 : 0;
                    if (need[u]) todo &= todo - 1ull;
                    slot[u] = (uint32_t)l;
                    jrow[u] = (uint32_t)__builtin_amdgcn_readlane((int)jl, l);
                }
                // the same fixed-order chains as rescore_kernel: a pair scores the same through either kernel
                if (vec4) {
                    for (uint32_t j = lane; j < dim / 4u; j += 64u) {
                        const f32x4 y = row4(rows, i, dim, j);
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (!need[u]) continue;
                            const f32x4 x = row4(rows, jrow[u], dim, j);
                            d0[u] = fmaf(x.w, y.w, fmaf(x.z, y.z, fmaf(x.y, y.y, fmaf(x.x, y.x, d0[u]))));
                            n0[u] = fmaf(x.w, x.w, fmaf(x.z, x.z, fmaf(x.y, x.y, fmaf(x.x, x.x, n0[u]))));
                        }
                    }
                } else {
                    for (uint32_t j = lane; j < dim; j += 64u) {
                        const float y = ldf(rows + (size_t)i * dim + j);
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (!need[u]) continue;
                            const float x = ldf(rows + (size_t)jrow[u] * dim + j);
                            d0[u] = fmaf(x, y, d0[u]);
                            n0[u] = fmaf(x, x, n0[u]);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (!need[u]) continue;
                    const float sim = cosine_from_sums(wave_sum(d0[u]), qq, wave_sum(n0[u]));
                    if (lane == 0) a.pair_sims[(size_t)i * a.cap + c0 + slot[u]] = sim;
                    if ((mirrored >> slot[u]) & 1ull) {   // the slot list j keeps for i
                        const uint32_t cj = a.cand_cnt[jrow[u]];
                        for (uint32_t t = lane; t < cj; t += 64u)
                            if (a.cand[(size_t)jrow[u] * a.cap + t] == i) a.pair_sims[(size_t)jrow[u] * a.cap + t] = sim;
                    }
                }
            }
        }
    }
}

template <int KS>
__global__ __launch_bounds__(256) void select_lists_kernel(const RescoreArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave; i < a.n_scan; i += n_waves) {
        const uint32_t total = a.cand_cnt[i];
        const uint32_t cnt = total < a.cap ? total : a.cap;
        bool redo = total > a.cap;
        if (cnt <= 64u && !redo) {
            // the usual case, one entry per lane: order by rank counting (cnt x (2 v_readlane + compare)) instead of
            // cnt serial insertions into the register list
            const bool in = lane < cnt;
            const uint32_t jr = in ? a.cand[(size_t)i * a.cap + lane] : 0u;
            const uint32_t sb = in ? reinterpret_cast<const uint32_t *>(a.pair_sims)[(size_t)i * a.cap + lane] : 0u;
            const uint32_t mt = in ? a.meta[jr] : META_REMOVED;
            if (__ballot(in && sb == PAIR_SIM_UNSET)) {
                if (lane == 0) { a.overflow[i] = 1u; a.out_cnt[i] = 0u; }
                continue;
            }
            const float sim = __uint_as_float(sb);
            const float score = score_of(distance_of(sim));
            const bool ok = in && score >= a.threshold && !(mt & META_REMOVED);   // NaN fails the comparison
            const uint64_t key = ok ? make_key(score, jr) : 0ull;
            uint32_t rank = 0;
            const uint32_t n_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
            for (uint32_t f = 0; f < n_u; f++) rank += readlane_u64(key, (int)f) > key ? 1u : 0u;
            const bool valid = ok && rank < a.topk;
            if (valid) {
                a.out_rows[(size_t)i * a.topk + rank] = jr;
                a.out_scores[(size_t)i * a.topk + rank] = score;
                if (a.out_dists) a.out_dists[(size_t)i * a.topk + rank] = distance_of(sim);
            }
            const uint32_t n_out = (uint32_t)__popcll(__ballot(valid));
            if (lane == 0) { a.overflow[i] = 0u; a.out_cnt[i] = n_out; }
            continue;
        }
        WaveTopK<KS> top;
        top.init(a.topk);
        for (uint32_t c0 = 0; c0 < cnt && !redo; c0 += 64u) {
            const uint32_t c = c0 + lane;
            const uint32_t jr = c < cnt ? a.cand[(size_t)i * a.cap + c] : 0u;
            const uint32_t sb = c < cnt ? reinterpret_cast<const uint32_t *>(a.pair_sims)[(size_t)i * a.cap + c] : 0u;
            if (__ballot(c < cnt && sb == PAIR_SIM_UNSET)) { redo = true; break; }   // an entry nobody scored: exact path
            const uint32_t mt = c < cnt ? a.meta[jr] : META_REMOVED;
            const uint32_t m = cnt - c0 < 64u ? cnt - c0 : 64u;
            for (uint32_t l = 0; l < m; l++) {
                const float sim = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)sb, (int)l));
                const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)jr, (int)l);
                const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)mt, (int)l);
                const float score = score_of(distance_of(sim));
                if (!(score >= a.threshold)) continue;              // rules.rs:50 / index.rs:386 (NaN fails)
                if (meta & META_REMOVED) continue;                  // removed from the index
                const uint64_t key = make_key(score, row);
                if (key > top.tau) top.insert(key, sim);
            }
        }
        if (lane == 0) a.overflow[i] = redo ? 1u : 0u;
        uint32_t n_out = 0;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const uint32_t r = (uint32_t)s * 64u + lane;
            const bool valid = !redo && r < a.topk && top.key[s] != 0ull;
            if (valid) {
                a.out_rows[(size_t)i * a.topk + r] = key_row(top.key[s]);
                a.out_scores[(size_t)i * a.topk + r] = score_of(distance_of(top.sim[s]));
                if (a.out_dists) a.out_dists[(size_t)i * a.topk + r] = distance_of(top.sim[s]);
            }
            n_out += (uint32_t)__popcll(__ballot(valid));
        }
        if (lane == 0) a.out_cnt[i] = n_out;
    }
}

int launch_rescore(const RescoreArgs &a, hipStream_t stream) {
    if (!a.n_scan) return CX_OK;
    if (a.topk > TOPK_MAX) return set_err(CX_ERR_VALIDATION, "rescore: topk=%u exceeds %u", a.topk, TOPK_MAX);
    if (a.pair_sims) {
        if (a.q_rows || a.scan_rows) return set_err(CX_ERR_VALIDATION, "rescore: pair scratch is for the symmetric pass only");
        uint32_t g = (a.n_scan + 3u) / 4u;
        if (g > 16384u) g = 16384u;
        hipLaunchKernelGGL(pair_sims_clear_kernel, dim3(g), dim3(256), 0, stream, a);
        if (a.rows16) hipLaunchKernelGGL(pair_score_kernel<uint16_t>, dim3(g), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(pair_score_kernel<float>, dim3(g), dim3(256), 0, stream, a);
        if (a.topk <= 64) hipLaunchKernelGGL((select_lists_kernel<1>), dim3(g), dim3(256), 0, stream, a);
        else if (a.topk <= 128) hipLaunchKernelGGL((select_lists_kernel<2>), dim3(g), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((select_lists_kernel<4>), dim3(g), dim3(256), 0, stream, a);
        CX_HIP(hipGetLastError());
        return CX_OK;
    }
    uint32_t grid = (a.n_scan + 3u) / 4u;
    if (grid > 16384u) grid = 16384u;
    if (a.rows16) {
        if (a.topk <= 64) hipLaunchKernelGGL((rescore_kernel<1, uint16_t>), dim3(grid), dim3(256), 0, stream, a);
        else if (a.topk <= 128) hipLaunchKernelGGL((rescore_kernel<2, uint16_t>), dim3(grid), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((rescore_kernel<4, uint16_t>), dim3(grid), dim3(256), 0, stream, a);
    } else if (a.topk <= 64) hipLaunchKernelGGL((rescore_kernel<1, float>), dim3(grid), dim3(256), 0, stream, a);
    else if (a.topk <= 128) hipLaunchKernelGGL((rescore_kernel<2, float>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((rescore_kernel<4, float>), dim3(grid), dim3(256), 0, stream, a);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// ---------------------------------------------------------------- 4. link rules

// linker/auto_linker.rs:233-264 with SimilarityLinkRule (rules.rs:42-62) as the
// only rule: walk the ordered list; skip self (:235-237); skip neighbours the
// storage has tombstoned (:240-243); score >= threshold -> edge (weight =
// score) unless the node already has that edge (existing_set, :226-231,
// :249-258: dropped without counting, the walk goes on); after each neighbour
// that was not skipped, stop once max_edges_per_node edges were proposed
// (:261-263 — tested AFTER the push, so max_edges = 0 still lets the first
// neighbour's edge through, as the reference does).
// With a.dedup the same walk emits DedupScanner::scan's pairs instead (dedup.rs:65-127).
// MODE 0 counts, MODE 1 writes at the exclusive-scan offsets (positions >= max_total dropped: :284-287).
__device__ __forceinline__ bool edge_exists(const uint32_t *seg, uint32_t n, uint32_t j) {
    uint32_t lo = 0, hi = n;   // segment sorted ascending
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t v = seg[mid];
        if (v == j) return true;
        if (v < j) lo = mid + 1; else hi = mid;
    }
    return false;
}

template <int MODE>
__global__ __launch_bounds__(256) void link_rules_kernel(const LinkArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_scan) return;
    const uint32_t self = a.scan_rows ? a.scan_rows[i] : i;
    // auto_linker.rs:217-218: a node without an embedding is skipped — a row removed from the index is exactly that
    const uint32_t cnt = (a.meta && (a.meta[self] & META_REMOVED)) ? 0u : a.list_cnt[i];
    uint32_t n = 0;
    const uint64_t base = MODE == 1 ? a.offsets[i] : 0ull;
    const uint32_t *have = nullptr;
    uint32_t n_have = 0;
    if (a.existing_offsets) {
        const uint64_t lo = a.existing_offsets[i];
        have = a.existing_to + lo;
        n_have = (uint32_t)(a.existing_offsets[i + 1] - lo);
    }
    for (uint32_t r = 0; r < cnt; r++) {
        const uint32_t j = a.list_rows[(size_t)i * a.topk + r];
        if (j == self) continue;
        if (a.dedup) {
            // linker/dedup.rs:93-102: a pair is reported by whichever of its nodes is scanned first
            // (row order here); cosine is symmetric, so (i, j) was already seen iff j was scanned
            // before i, i.e. j < i and j is not storage-deleted (deleted nodes are never scanned, :72-74)
            if (j < self && !(a.deleted && a.deleted[j])) continue;
        } else if (a.deleted && a.deleted[j]) {
            continue;
        }
        const float s = a.list_scores[(size_t)i * a.topk + r];
        if (s >= a.threshold && !(n_have && edge_exists(have, n_have, j))) {
            if (MODE == 1 && base + n < a.max_total) {
                a.out_from[base + n] = self;
                a.out_to[base + n] = j;
                a.out_weight[base + n] = s;
            }
            n++;
        }
        if (n >= a.max_edges) break;
    }
    if (MODE == 0) a.counts[i] = n;
}

int launch_link_rules(const LinkArgs &a, bool emit, hipStream_t stream) {
    if (!a.n_scan) return CX_OK;
    const uint32_t grid = (a.n_scan + 255u) / 256u;
    if (emit) hipLaunchKernelGGL((link_rules_kernel<1>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((link_rules_kernel<0>), dim3(grid), dim3(256), 0, stream, a);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

}  // namespace cx
