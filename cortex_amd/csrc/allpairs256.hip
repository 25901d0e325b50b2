// allpairs256.hip — the filter GEMM of the all-pairs pass for large scan sets: 256x256 tiles on a
// 4-slot LDS ring with counted waits.
//
// Same contract as pair_filter_kernel (allpairs.hip): bf16 shadow rows in, candidate columns out, no score
// matrix.  What changes is the pipeline (cdna_hip_programming.md §5 "Pipelining across barriers", T3+T4):
//  - 256x256 block tile, BK = 32, 8 waves as 2(M) x 4(N), each wave a 128x64 sub-tile = 4x2 tiles of
//    mfma_f32_32x32x16_bf16 (128 accumulator registers), two waves per SIMD;
//  - per K-step a wave issues 4 LDS-DMA instructions (2 A + 2 B, 1 KiB each), 12 ds_read_b128 and 16 MFMAs,
//    interleaved one-for-one (step_full): twice the MFMA work per DMA and 1.3x per LDS read of the 128^2 kernel;
//  - LDS = ring of 4 slots x (A 16 KiB + B 16 KiB) = 128 KiB.  In step t a wave issues the DMA of step t+3 and
//    the fragment reads of step t+1 (double-buffered registers), then the MFMAs of step t; the wait at the end
//    of a step is `s_waitcnt vmcnt(4)` — this wave's part of step t+2 has landed, step t+3 stays in flight —
//    followed by a RAW s_barrier (a __syncthreads() would drain vmcnt(0)).  All LDS is one array and there
//    are no ordinary global loads in the loop, so hipcc adds no vmcnt(0) of its own (checked in the ISA).
//  - RAW: slot t+2 is read (in step t+1) one barrier after the wait that retired it.  WAR: slot (t+3)%4 was
//    last read in step t-2; those ds_reads retired before the MFMAs of step t-1 that consumed them.
//  - 16-byte pieces of a 64-byte row are stored at piece ^ (row >> 3 & 3): with rows at a 64-byte stride this
//    makes every ds_read_b128 lane group (MI355X_MICROARCH.md §LDS) of the 32-row x 2-piece operand pattern hit
//    16 different 16-byte bank groups (SQ_LDS_BANK_CONFLICT stays 0).
#include "kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace cx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

namespace p256 {
constexpr int BM = 256, BN = 256, BK = 32, NS = 4;
constexpr int OP_BYTES = BM * BK * 2;          // 16 KiB per operand per slot
constexpr int SLOT_BYTES = 2 * OP_BYTES;       // 32 KiB
constexpr int LDS_BYTES = NS * SLOT_BYTES;     // 128 KiB
// 16-byte piece p of a 64-byte row sits at slot p ^ (row >> 3 & 3): a ds_read_b128 lane group (16 lanes, 256 B of
// banks) of the 32-row x 2-piece MFMA operand pattern then covers all sixteen 16-byte bank groups
__device__ inline uint32_t off(uint32_t row, uint32_t piece) { return row * 64u + ((piece ^ ((row >> 3) & 3u)) << 4); }
}  // namespace p256

template <bool DIAG, bool IL>
__global__ __launch_bounds__(512) void pair_filter256_kernel(const PairFilterArgs a) {
    using namespace p256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t wm = wave >> 2, wn = wave & 3u;   // 2 x 4 waves: rows wm*128.., cols wn*64..

    const uint32_t tiles_i = (a.n_scan + BM - 1) / BM, tiles_j = (a.n_rows + BN - 1) / BN;
    const uint32_t T = a.symmetric ? a.n_tiles : tiles_i * tiles_j;
    uint32_t b = blockIdx.x;
    {
        const uint32_t q = T / 8u, r = T % 8u, xcd = b % 8u;
        b = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + b / 8u;
    }
    uint32_t ti, tj;
    if (a.symmetric) {
        const uint32_t t = a.tile_list[b];
        ti = t >> 16;
        tj = t & 0xFFFFu;
    } else {
        const uint32_t GS = 4u, per_group = GS * tiles_j;
        const uint32_t group = b / per_group, first_i = group * GS;
        const uint32_t gsz = (tiles_i - first_i) < GS ? (tiles_i - first_i) : GS;
        ti = first_i + (b % per_group) % gsz;
        tj = (b % per_group) / gsz;
    }
    const uint32_t i0 = ti * BM, j0 = tj * BN;
    unsigned long long c_vm = 0, c_bar = 0;

    // loader: one LDS-DMA = 16 rows x 64 B; 16 per operand per slot, 2 per wave
    const uint32_t lrow = lane >> 2, lpos = lane & 3u;
    const uint16_t *srcA[2], *srcB[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t r = (wave * 2u + (uint32_t)q) * 16u + lrow;
        const uint32_t piece = lpos ^ ((r >> 3) & 3u);
        uint32_t gi = i0 + r;
        gi = gi < a.n_scan ? gi : a.n_scan - 1u;
        const uint32_t ga = a.scan_rows ? a.scan_rows[gi] : gi;
        uint32_t gb = j0 + r;
        gb = gb < a.n_rows ? gb : a.n_rows - 1u;
        srcA[q] = (a.shadow_q ? a.shadow_q : a.shadow) + (size_t)ga * a.dim + piece * 8u;
        srcB[q] = a.shadow + (size_t)gb * a.dim + piece * 8u;
    }
    auto stage = [&](uint32_t slot, uint32_t kt) {
        char *A = smem + slot * SLOT_BYTES, *B = A + OP_BYTES;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t o = (wave * 2u + (uint32_t)q) * 1024u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcA[q] + kt * BK),
                                             (__attribute__((address_space(3))) void *)(A + o), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcB[q] + kt * BK),
                                             (__attribute__((address_space(3))) void *)(B + o), 16, 0, 0);
        }
    };

    // 128x64 per wave = 4 x 2 tiles of v_mfma_f32_32x32x16_bf16 (16 accumulator registers each): half the MFMA
    // instructions of the 16x16x32 form for the same LDS reads, and each leaves 24 of its 32 cycles (not 8 of
    // 16) for the step's LDS-DMA / ds_read issue
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[m][n][e] = 0.0f;

    const uint32_t KT = a.dim / BK;
    const uint32_t fr = lane & 31u, fq = lane >> 5;
    // fragment read offsets inside an operand image (loop invariant): [32-row block][k half h]: lane (fr, fq)
    // reads piece 2 h + fq of row fr
    uint32_t offA[4][2], offB[2][2];
#pragma unroll
    for (uint32_t m = 0; m < 4; m++)
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) offA[m][h] = off(wm * 128u + m * 32u + fr, 2u * h + fq);
#pragma unroll
    for (uint32_t n = 0; n < 2; n++)
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) offB[n][h] = OP_BYTES + off(wn * 64u + n * 32u + fr, 2u * h + fq);

    // Software pipeline.  Fragment registers are double buffered (statically indexed: the K loop is unrolled
    // by two): in step kt a wave first issues the DMA of step kt+3 and the ds_reads of step kt+1, then the 32
    // MFMAs of step kt, whose operands were read during step kt-1.  Both waves of a SIMD run the same program
    // in lockstep (one block per CU), so a stream that loads, then computes, then waits leaves the matrix pipe
    // idle while both load; a stream whose loads sit under its own MFMAs keeps it fed.
    bf16x8 fa0[8], fb0[4], fa1[8], fb1[4];
    auto read_frags = [&](uint32_t kt, bf16x8 *fa, bf16x8 *fb) {
        const char *S = smem + (kt & 3u) * SLOT_BYTES;
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int h = 0; h < 2; h++) fb[n * 2 + h] = *reinterpret_cast<const bf16x8 *>(S + offB[n][h]);
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int h = 0; h < 2; h++) fa[m * 2 + h] = *reinterpret_cast<const bf16x8 *>(S + offA[m][h]);
    };
    // FULL = steady state (steps kt+1 and kt+3 exist): no branches in the body, so hipcc keeps counted
    // lgkmcnt waits; with the conditions inside it joins control flow and falls back to lgkmcnt(0) right
    // after issuing the prefetch reads.
    auto mfmas = [&](const bf16x8 *fa, const bf16x8 *fb) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m * 2 + h], fb[n * 2 + h], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // Steady-state step, interleaved: after each of the 16 MFMAs ONE other vector-memory / LDS instruction of
    // the step is issued — the 4 LDS-DMAs of step kt+3 and the 12 fragment reads of step kt+1 — so that their
    // issue time (an LDS-DMA holds the wave's issue for 60-185 cycles) runs under this wave's own MFMAs instead
    // of in front of them.  As [stage][reads][MFMAs] a wave's non-MFMA issue took ~790 cycles per step against
    // the 512 its SIMD partner computes for, and the pipe idled the difference (tuning.md).
    auto step_full = [&](uint32_t kt, const bf16x8 *fa, const bf16x8 *fb, bf16x8 *na, bf16x8 *nb) {
        char *SA = smem + ((kt + 3) & 3u) * SLOT_BYTES, *SB = SA + OP_BYTES;   // slot (kt-1)&3: last read in step kt-2
        const char *S = smem + ((kt + 1) & 3u) * SLOT_BYTES;                    // landed before the barrier of step kt-1
        const uint32_t ko = (kt + 3) * BK;
        if constexpr (!IL) {   // A/B arm (CX_PAIR_INTERLEAVE=0): the same work as [stage][reads][MFMAs]
            stage((kt + 3) & 3u, kt + 3);
            read_frags(kt + 1, na, nb);
            mfmas(fa, fb);
        } else {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int idx = 0; idx < 16; idx++) {
            const int h = idx >> 3, m = (idx >> 1) & 3, n = idx & 1;
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m * 2 + h], fb[n * 2 + h], acc[m][n], 0, 0, 0);
            if ((idx & 3) == 0) {
                const int d = idx >> 2, q = d >> 1;
                const uint32_t o = (wave * 2u + (uint32_t)q) * 1024u;
                if ((d & 1) == 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcA[q] + ko),
                                                     (__attribute__((address_space(3))) void *)(SA + o), 16, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcB[q] + ko),
                                                     (__attribute__((address_space(3))) void *)(SB + o), 16, 0, 0);
            } else {
                const int r = idx - (idx >> 2) - 1;   // 0..11: B fragments first, then A
                if (r < 4) nb[r] = *reinterpret_cast<const bf16x8 *>(S + offB[r >> 1][r & 1]);
                else na[r - 4] = *reinterpret_cast<const bf16x8 *>(S + offA[(r - 4) >> 1][(r - 4) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        unsigned long long w0 = 0, w1 = 0;
        if constexpr (DIAG) { asm volatile("s_nop 0" :: "v"(acc[3][1][15])); w0 = __builtin_readcyclecounter(); }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // step kt+2 landed (read next step); kt+3 stays in flight
        if constexpr (DIAG) w1 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        if constexpr (DIAG) { const unsigned long long w2 = __builtin_readcyclecounter(); c_vm += w1 - w0; c_bar += w2 - w1; }
    };
    auto step_tail = [&](uint32_t kt, const bf16x8 *fa, const bf16x8 *fb, bf16x8 *na, bf16x8 *nb) {
        if (kt + 3 < KT) stage((kt + 3) & 3u, kt + 3);
        if (kt + 1 < KT) read_frags(kt + 1, na, nb);
        mfmas(fa, fb);
        if (kt + 3 < KT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if constexpr (DIAG) t0 = __builtin_readcyclecounter();
    // prologue: steps 0..2 in flight; 0 and 1 must land (0 is read now, 1 during step 0)
    stage(0, 0);
    if (KT > 1) stage(1, 1);
    if (KT > 2) stage(2, 2);
    if (KT > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (DIAG) t1 = __builtin_readcyclecounter();
    read_frags(0, fa0, fb0);
    uint32_t kt = 0;
    for (; kt + 5 <= KT; kt += 2) {          // both steps of the pair have their kt+3 inside the K range
        step_full(kt, fa0, fb0, fa1, fb1);
        step_full(kt + 1, fa1, fb1, fa0, fb0);
    }
    for (; kt < KT; kt += 2) {
        step_tail(kt, fa0, fb0, fa1, fb1);
        if (kt + 1 < KT) step_tail(kt + 1, fa1, fb1, fa0, fb0);
    }

    if constexpr (DIAG) { asm volatile("s_nop 0" :: "v"(acc[3][1][15])); t2 = __builtin_readcyclecounter(); }
    // epilogue: C[row][col], col = lane & 15 (j), row = 4*(lane >> 4) + e (i)
    const bool mirror = a.symmetric && ti != tj;
    // One hit position: both directions' slot atomics are issued before either result is waited for (one round
    // trip to L2 per position instead of two).
    auto emit = [&](uint32_t i, uint32_t j) {
        const bool fwd = i < a.n_scan && j < a.n_rows;
        const bool rev = mirror && fwd;   // symmetric pass: n_scan == n_rows
        uint32_t s_f = a.cap, s_r = a.cap;
        if (fwd) s_f = atomicAdd(a.cand_cnt + i, 1u);
        if (rev) s_r = atomicAdd(a.cand_cnt + j, 1u);
        if (s_f < a.cap) a.cand[(size_t)i * a.cap + s_f] = j;
        if (s_r < a.cap) a.cand[(size_t)j * a.cap + s_r] = i;
    };
    // Hits are rare (a handful per 128x64 wave tile), so the 128 accumulator values are screened 16 at a time
    // with a running maximum and ONE wave-uniform branch per 16x64 strip; only strips with a hit somewhere in
    // the wave walk their elements (one test per element took 9.6k cycles per tile, 19 % of the kernel).
    // C layout of the 32x32 tile: register e of lane l holds row 8 (e / 4) + 4 (l >> 5) + e % 4, column l & 31.
    // First all eight 32x32 tiles are screened in straight-line code (running maximum, one ballot each), then only
    // the tiles with a hit somewhere in the wave are walked.
    uint32_t strips = 0;
#pragma unroll
    for (uint32_t m = 0; m < 4; m++)
#pragma unroll
        for (uint32_t n = 0; n < 2; n++) {
            float mx = acc[m][n][0];
#pragma unroll
            for (uint32_t e = 1; e < 16; e++) mx = fmaxf(mx, acc[m][n][e]);
            strips |= __ballot(mx >= a.thr_lo) != 0ull ? 1u << (m * 2u + n) : 0u;
        }
    unsigned long long t2b = 0;
    if constexpr (DIAG) t2b = __builtin_readcyclecounter();
#pragma unroll
    for (uint32_t m = 0; m < 4; m++)
#pragma unroll
        for (uint32_t n = 0; n < 2; n++) {
            if (!((strips >> (m * 2u + n)) & 1u)) continue;
            const uint32_t j = j0 + wn * 64u + n * 32u + fr;
#pragma unroll
            for (uint32_t e = 0; e < 16; e++) {
                const bool hit = acc[m][n][e] >= a.thr_lo;
                if (__ballot(hit) == 0ull) continue;
                if (hit) {
                    const uint32_t i = i0 + wm * 128u + m * 32u + 8u * (e >> 2) + 4u * fq + (e & 3u);
                    emit(i, j);
                }
            }
        }
    if constexpr (DIAG) {
        const unsigned long long t3 = __builtin_readcyclecounter();
        if (lane == 0) { unsigned long long *o = a.diag + ((size_t)blockIdx.x * 8 + wave) * 4; o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t3 - t0;
            if (blockIdx.x % 9000 == 5017 % 9000 && wave == 5) printf("[pair256] tile %u: main-loop waits vmcnt %llu barrier %llu; epilogue: screen %llu walk %llu\n", blockIdx.x, c_vm, c_bar, t2b - t2, t3 - t2b); }
    }
}

void pair_filter256_tile_list(uint32_t n_rows, std::vector<uint32_t> &out) {
    const uint32_t tiles = (n_rows + p256::BM - 1) / p256::BM, GS = 4;
    out.clear();
    out.reserve((size_t)tiles * (tiles + 1) / 2);
    for (uint32_t g0 = 0; g0 < tiles; g0 += GS) {
        const uint32_t g1 = g0 + GS < tiles ? g0 + GS : tiles;
        for (uint32_t tj = g0; tj < tiles; tj++)
            for (uint32_t ti = g0; ti < g1 && ti <= tj; ti++) out.push_back((ti << 16) | tj);
    }
}

int launch_pair_filter256(const PairFilterArgs &a, hipStream_t stream) {
    using namespace p256;
    if (a.dim % BK != 0 || a.dim == 0) return set_err(CX_ERR_VALIDATION, "pair filter 256 needs dim %% 32 == 0 (got %u)", a.dim);
    if (!a.n_scan || !a.n_rows) return CX_OK;
    if (a.symmetric && (!a.tile_list || (a.n_rows + BM - 1) / BM > 0xFFFFu))
        return set_err(CX_ERR_VALIDATION, "pair filter 256: symmetric pass needs a tile list");
    const uint64_t tiles = a.symmetric ? a.n_tiles : (uint64_t)((a.n_scan + BM - 1) / BM) * ((a.n_rows + BN - 1) / BN);
    if (tiles > 0x7FFFFFFFull) return set_err(CX_ERR_VALIDATION, "pair filter 256: too many tiles");
    static std::atomic<uint64_t> attr_devices{0};
    if (first_use_on_device(attr_devices)) {
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_filter256_kernel<false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_filter256_kernel<true, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_filter256_kernel<false, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    }
    static const int il = getenv("CX_PAIR_INTERLEAVE") ? atoi(getenv("CX_PAIR_INTERLEAVE")) : 1;
    if (getenv("CX_PAIR_DIAG")) {   // diagnostic build: per-phase cycles per tile on stderr, results still valid
        PairFilterArgs d = a;
        const size_t n = (size_t)tiles * 8 * 4;
        CX_HIP(hipMalloc((void **)&d.diag, n * 8));
        hipLaunchKernelGGL((pair_filter256_kernel<true, true>), dim3((uint32_t)tiles), dim3(512), LDS_BYTES, stream, d);
        CX_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h(n);
        CX_HIP(hipMemcpy(h.data(), d.diag, n * 8, hipMemcpyDeviceToHost));
        CX_HIP(hipFree(d.diag));
        double s[4] = {0, 0, 0, 0};
        for (size_t w = 0; w < (size_t)tiles * 8; w++) for (int p = 0; p < 4; p++) s[p] += (double)h[w * 4 + p];
        const double nw = (double)tiles * 8;
        fprintf(stderr, "[pair256 diag] %llu tiles; cycles per tile per wave: prologue %.0f  main loop %.0f  epilogue %.0f  total %.0f\n",
                (unsigned long long)tiles, s[0] / nw, s[1] / nw, s[2] / nw, s[3] / nw);
        return CX_OK;
    }
    if (il) hipLaunchKernelGGL((pair_filter256_kernel<false, true>), dim3((uint32_t)tiles), dim3(512), LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((pair_filter256_kernel<false, false>), dim3((uint32_t)tiles), dim3(512), LDS_BYTES, stream, a);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

}  // namespace cx
