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
//  - Round 2: the shadow is read from a copy cut into the DMA's own pieces (launch_tile_shadow: one instruction = one
//    contiguous, pre-swizzled KiB = 8 whole cache lines, SGPR base + lane * 16 addressing), fragment offsets are two
//    VGPRs + immediates, and the epilogue collects hits in LDS and pays ONE round trip of slot atomics per wave.
//    Measured on the way (100k x 768, symmetric; CX_PAIR_SCHED arms): loads only (MFMAs removed) 2.88 ms, MFMAs only
//    3.87 ms at the sustained clock, the kernel 6.75 ms = their SUM — an in-order wave that is issuing an LDS-DMA
//    (100-185 cycles each: the CU's one texture-address path moves ~43 B/clk) feeds the matrix pipe nothing, and its SIMD
//    partner is in the same place.  Tried against that and rejected, all slower than the lockstep interleave: ping-pong
//    between the SIMD partners with a second barrier per K-step (7.07 ms against 6.45), the same with the load half at
//    s_setprio 1 (7.18) or no priorities (7.03), DMA issue slots staggered by wave & 3 (7.3) or between partners only
//    (7.24), a 5-slot ring (6.78), and a persistent form (one block per CU, the ring running through the tile
//    boundaries so that a tile's last three K-steps issue the next tile's first three DMAs: no prologue, no relaunch)
//    6.96 ms at 100k rows and 749 against 647 ms at 1M.  What is left is bytes per flop: a 256x256 tile moves 32 KiB per K-step through that
//    path whatever the schedule.
//  - 16-byte pieces of a 64-byte row are stored at piece ^ (row >> 3 & 3): with rows at a 64-byte stride this
//    makes every ds_read_b128 lane group (MI355X_MICROARCH.md §LDS) of the 32-row x 2-piece operand pattern hit
//    16 different 16-byte bank groups (SQ_LDS_BANK_CONFLICT stays 0).
#include "kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <type_traits>
#include <vector>

namespace cx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

namespace p256 {
constexpr int BM = 256, BN = 256, BK = 32;
constexpr int OP_BYTES = BM * BK * 2;          // 16 KiB per operand per slot
constexpr int SLOT_BYTES = 2 * OP_BYTES;       // 32 KiB
// ring of NS slots: NS - 1 K-steps of DMA in flight (NS = 4: 128 KiB; NS = 5: 160 KiB, the whole LDS of a CU)
// 16-byte piece p of a 64-byte row sits at slot p ^ (row >> 3 & 3): a ds_read_b128 lane group (16 lanes, 256 B of
// banks) of the 32-row x 2-piece MFMA operand pattern then covers all sixteen 16-byte bank groups
__device__ inline uint32_t off(uint32_t row, uint32_t piece) { return row * 64u + ((piece ^ ((row >> 3) & 3u)) << 4); }
}  // namespace p256

// TA: the A operand (scanned rows) comes from the tiled shadow too — the symmetric pass and any pass that scans the rows
// of this shard in order; otherwise (a subset / a permutation of rows, or external vectors) A is gathered row by row from
// a row-major shadow with per-lane addresses.  The B operand (this shard's rows) is always read from the tiled shadow.
template <bool DIAG, int SCHED, bool TA, int NS>
__global__ __launch_bounds__(512) void pair_filter256_kernel(const PairFilterArgs a) {
    using namespace p256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
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
    const uint32_t KT = a.dim / BK;
    constexpr uint32_t PF = NS - 1;                       // K-steps of DMA in flight
    // the counted wait: all but the (PF - 2) youngest K-steps' DMAs (4 instructions each) of this wave have landed
    auto wait_ring = [&]() {
        if constexpr (PF == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };
    static_assert(NS == 4 || NS == 5, "ring of 4 or 5 slots");

    // loader: one LDS-DMA instruction = 16 rows x 64 B = 1 KiB; 16 per operand per slot, 2 per wave (groups 2 wave, + 1).
    // Tiled source (a.shadow_t): the KiB is ONE contiguous, pre-swizzled piece — 8 whole cache lines per instruction
    // instead of 16 half lines, no line fetched twice by consecutive K-steps — addressed as a wave-uniform base (SGPRs)
    // plus lane * 16: no per-lane 64-bit pointers in the loop (the kernel sits at the 256-VGPR cap).
    const uint32_t last_blk = (a.n_rows - 1u) / 16u;
    const uint32_t a_step = a.shadow_q ? (uint32_t)BK : 512u;   // elements per K-step of a gathered A row (row-major / tiled)
    const uint32_t voff = lane * 16u;
    const char *sB[2], *sA[2];
    const uint16_t *srcA[2] = {nullptr, nullptr};   // per-lane row pointers, row-major A only
#pragma unroll
    for (int q = 0; q < 2; q++) {
        uint32_t bb = j0 / 16u + wave * 2u + (uint32_t)q;
        bb = bb < last_blk ? bb : last_blk;
        sB[q] = reinterpret_cast<const char *>(a.shadow_t) + (size_t)bb * KT * 1024u;
        if constexpr (TA) {
            uint32_t ba = i0 / 16u + wave * 2u + (uint32_t)q;
            ba = ba < last_blk ? ba : last_blk;
            sA[q] = reinterpret_cast<const char *>(a.shadow_t) + (size_t)ba * KT * 1024u;
        } else {
            const uint32_t lrow = lane >> 2, lpos = lane & 3u;
            const uint32_t r = (wave * 2u + (uint32_t)q) * 16u + lrow;
            const uint32_t piece = lpos ^ ((r >> 3) & 3u);
            uint32_t gi = i0 + r;
            gi = gi < a.n_scan ? gi : a.n_scan - 1u;
            const uint32_t ga = a.scan_rows ? a.scan_rows[gi] : gi;
            // scanned vectors that are not this shard's rows: row-major (a.shadow_q); this shard's rows in any order: the
            // tiled shadow, a K-step (32 elements) = 512 elements further
            srcA[q] = a.shadow_q ? a.shadow_q + (size_t)ga * a.dim + piece * 8u : a.shadow_t + tiled_shadow_off(ga, piece, KT);
            sA[q] = nullptr;
        }
    }
    auto dma = [&](uint32_t slot, uint32_t kt, int which) {   // which: 0 A q0, 1 B q0, 2 A q1, 3 B q1
        const int q = which >> 1;
        char *dst = smem + slot * SLOT_BYTES + ((which & 1) ? OP_BYTES : 0) + (wave * 2u + (uint32_t)q) * 1024u;
        const void *src;
        if (which & 1) src = sB[q] + (size_t)kt * 1024u + voff;
        else if constexpr (TA) src = sA[q] + (size_t)kt * 1024u + voff;
        else src = srcA[q] + (size_t)kt * a_step;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    auto stage = [&](uint32_t slot, uint32_t kt) {
#pragma unroll
        for (int w = 0; w < 4; w++) dma(slot, kt, w);
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

    const uint32_t fr = lane & 31u, fq = lane >> 5;
    // fragment read offsets: lane (fr, fq) reads piece 2 h + fq of row fr of a 32-row block.  The swizzle depends on
    // (row >> 3) & 3 only, i.e. on fr — the same for every 32-row block — so two VGPRs serve all of A and B; the block
    // (m * 2048, n * 2048) is an immediate offset of the ds_read and the slot / wave part is scalar.
    uint32_t fo[2];
#pragma unroll
    for (uint32_t h = 0; h < 2; h++) fo[h] = off(fr, 2u * h + fq);
    const uint32_t baseA = wm * 128u * 64u, baseB = OP_BYTES + wn * 64u * 64u;

    // Software pipeline.  Fragment registers are double buffered (statically indexed: the K loop is unrolled
    // by two): in step kt a wave first issues the DMA of step kt+3 and the ds_reads of step kt+1, then the 32
    // MFMAs of step kt, whose operands were read during step kt-1.  Both waves of a SIMD run the same program
    // in lockstep (one block per CU), so a stream that loads, then computes, then waits leaves the matrix pipe
    // idle while both load; a stream whose loads sit under its own MFMAs keeps it fed.
    bf16x8 fa0[8], fb0[4], fa1[8], fb1[4];
    auto rdA = [&](uint32_t slot, int m, int h) { return *reinterpret_cast<const bf16x8 *>(smem + (slot * SLOT_BYTES + baseA + fo[h]) + m * 2048); };
    auto rdB = [&](uint32_t slot, int n, int h) { return *reinterpret_cast<const bf16x8 *>(smem + (slot * SLOT_BYTES + baseB + fo[h]) + n * 2048); };
    auto read_frags = [&](uint32_t kt, bf16x8 *fa, bf16x8 *fb) {
        const uint32_t slot = kt % NS;
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int h = 0; h < 2; h++) fb[n * 2 + h] = rdB(slot, n, h);
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int h = 0; h < 2; h++) fa[m * 2 + h] = rdA(slot, m, h);
    };
    // FULL = steady state (steps kt+1 and kt+3 exist): no branches in the body, so hipcc keeps counted
    // lgkmcnt waits; with the conditions inside it joins control flow and falls back to lgkmcnt(0) right
    // after issuing the prefetch reads.
    auto mfmas = [&](const bf16x8 *fa, const bf16x8 *fb) {
        if constexpr (SCHED == 3) { asm volatile("" :: "v"(fa[0]), "v"(fa[7]), "v"(fb[0]), "v"(fb[3])); return; }
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
    // P (0..3): the MFMA slots after which this wave issues its four DMAs are P, P + 4, P + 8, P + 12.  With every wave at
    // P = 0 the eight waves of the block hit the CU's one texture-address path with 8 KiB at the same four moments of
    // a step and each issue then waits its turn (100-185 cycles of issue time per DMA, during which the in-order wave
    // feeds the matrix pipe nothing); P = wave & 3 spreads the block's 32 DMAs of a step over all sixteen MFMA slots.
    auto step_full = [&](uint32_t kt, const bf16x8 *fa, const bf16x8 *fb, bf16x8 *na, bf16x8 *nb, auto pc) {
        constexpr int P = decltype(pc)::value;
        const uint32_t dslot = (kt + PF) % NS;   // = slot of step kt-1: last read in step kt-2
        const uint32_t rslot = (kt + 1) % NS;    // landed before the barrier of step kt-1
        if constexpr (SCHED == 0 || SCHED == 3 || SCHED == 4) {   // A/B arms (CX_PAIR_SCHED): [stage][reads][MFMAs]; 3 = no MFMAs, 4 = no DMA
            if constexpr (SCHED != 4) stage(dslot, kt + PF);
            read_frags(kt + 1, na, nb);
            mfmas(fa, fb);
        } else {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int idx = 0; idx < 16; idx++) {
            const int h = idx >> 3, m = (idx >> 1) & 3, n = idx & 1;
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m * 2 + h], fb[n * 2 + h], acc[m][n], 0, 0, 0);
            if ((idx & 3) == P) {
                dma(dslot, kt + PF, idx >> 2);
            } else {
                const int r = idx - (idx >> 2) - ((idx & 3) > P ? 1 : 0);   // 0..11: B fragments first, then A
                if (r < 4) nb[r] = rdB(rslot, r >> 1, r & 1);
                else na[r - 4] = rdA(rslot, (r - 4) >> 1, (r - 4) & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        unsigned long long w0 = 0, w1 = 0;
        if constexpr (DIAG) { asm volatile("s_nop 0" :: "v"(acc[3][1][15])); w0 = __builtin_readcyclecounter(); }
        wait_ring();  // step kt+2 landed (read next step); the younger ones stay in flight
        if constexpr (DIAG) w1 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        if constexpr (DIAG) { const unsigned long long w2 = __builtin_readcyclecounter(); c_vm += w1 - w0; c_bar += w2 - w1; }
    };
    auto step_tail = [&](uint32_t kt, const bf16x8 *fa, const bf16x8 *fb, bf16x8 *na, bf16x8 *nb) {
        if (kt + PF < KT) stage((kt + PF) % NS, kt + PF);
        if (kt + 1 < KT) read_frags(kt + 1, na, nb);
        mfmas(fa, fb);
        if (kt + PF < KT) wait_ring();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if constexpr (DIAG) t0 = __builtin_readcyclecounter();
    // prologue: steps 0..2 in flight; 0 and 1 must land (0 is read now, 1 during step 0)
#pragma unroll
    for (uint32_t st = 0; st < PF; st++)
        if (st < KT) stage(st, st);
    if (KT >= PF) wait_ring();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (DIAG) t1 = __builtin_readcyclecounter();
    read_frags(0, fa0, fb0);
    uint32_t kt = 0;
    for (; kt + PF + 2 <= KT; kt += 2) {     // both steps of the pair have their kt+3 inside the K range
        step_full(kt, fa0, fb0, fa1, fb1, std::integral_constant<int, 0>{});
        step_full(kt + 1, fa1, fb1, fa0, fb0, std::integral_constant<int, 0>{});
    }
    for (; kt < KT; kt += 2) {
        step_tail(kt, fa0, fb0, fa1, fb1);
        if (kt + 1 < KT) step_tail(kt + 1, fa1, fb1, fa0, fb0);
    }

    if constexpr (DIAG) { asm volatile("s_nop 0" :: "v"(acc[3][1][15])); t2 = __builtin_readcyclecounter(); }
    // epilogue: C[row][col], col = lane & 15 (j), row = 4*(lane >> 4) + e (i)
    const bool mirror = a.symmetric && ti != tj;
    // Hits are first collected — (row, column) pairs appended to a per-wave list in LDS (the ring is free: every wave is
    // past the last K-step's barrier), positions from a ballot prefix, no atomics — and only then turned into candidate
    // slots: every lane takes entries of the list, all their slot atomics go out back to back and are waited for ONCE.
    // Before, each hit position paid its own round trip to L2 (atomicAdd -> slot -> store), 600-3000 cycles apiece and
    // one after the other: a wave with five hit positions kept its whole block (and the CU) for ~5k cycles.
    constexpr uint32_t HL_CAP = 256;                                    // pairs per wave; flushed when full
    uint32_t *hl = reinterpret_cast<uint32_t *>(smem) + wave * (2u * HL_CAP);
    uint32_t nh = 0;                                                     // wave-uniform
    auto flush = [&]() {   // the list holds (i, j) once; a mirrored tile also enters i into j's list
        uint32_t slot[HL_CAP / 64], slot2[HL_CAP / 64], ii[HL_CAP / 64], jj[HL_CAP / 64];
#pragma unroll
        for (uint32_t t = 0; t < HL_CAP / 64; t++) {
            const uint32_t idx = lane + 64u * t;
            slot[t] = a.cap;
            slot2[t] = a.cap;
            if (idx < nh) {
                ii[t] = hl[2u * idx]; jj[t] = hl[2u * idx + 1u];
                slot[t] = atomicAdd(a.cand_cnt + ii[t], 1u);
                if (mirror) slot2[t] = atomicAdd(a.cand_cnt + jj[t], 1u);
            }
        }
#pragma unroll
        for (uint32_t t = 0; t < HL_CAP / 64; t++) {
            if (slot[t] < a.cap) a.cand[(size_t)ii[t] * a.cap + slot[t]] = jj[t];
            if (slot2[t] < a.cap) a.cand[(size_t)jj[t] * a.cap + slot2[t]] = ii[t];
        }
        nh = 0;
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
    // Walk of a 32x32 tile with a hit: 16 compares into a per-lane bit mask (straight-line), then only the lanes that
    // hold a hit loop over their bits and take list positions from the wave's LDS counter.  (One ballot and one
    // wave-uniform branch per element — 16 per tile — cost ~650 cycles per tile with a hit: 3.5-5k of a tile's 38k.)
    uint32_t *wave_nh = reinterpret_cast<uint32_t *>(smem) + 8u * (2u * HL_CAP);   // [8] hits of each wave
    if (lane == 0) wave_nh[wave] = 0u;
#pragma unroll
    for (uint32_t m = 0; m < 4; m++)
#pragma unroll
        for (uint32_t n = 0; n < 2; n++) {
            if (!((strips >> (m * 2u + n)) & 1u)) continue;
            const uint32_t j = j0 + wn * 64u + n * 32u + fr;
            const uint32_t ibase = i0 + wm * 128u + m * 32u + 4u * fq;
            uint32_t mask = 0;
#pragma unroll
            for (uint32_t e = 0; e < 16; e++) {
                const uint32_t i = ibase + 8u * (e >> 2) + (e & 3u);
                mask |= (acc[m][n][e] >= a.thr_lo && i < a.n_scan && j < a.n_rows) ? (1u << e) : 0u;
            }
            while (mask) {
                const uint32_t e = (uint32_t)__builtin_ctz(mask);
                mask &= mask - 1u;
                const uint32_t i = ibase + 8u * (e >> 2) + (e & 3u);
                const uint32_t pos = atomicAdd(&wave_nh[wave], 1u);   // LDS
                if (pos < HL_CAP) {
                    hl[2u * pos] = i; hl[2u * pos + 1u] = j;
                } else {   // list full (a block of near-duplicates): this hit pays its own round trip
                    const uint32_t s_f = atomicAdd(a.cand_cnt + i, 1u);
                    uint32_t s_r = a.cap;
                    if (mirror) s_r = atomicAdd(a.cand_cnt + j, 1u);
                    if (s_f < a.cap) a.cand[(size_t)i * a.cap + s_f] = j;
                    if (s_r < a.cap) a.cand[(size_t)j * a.cap + s_r] = i;
                }
            }
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_nh[wave]);
        nh = c < HL_CAP ? c : HL_CAP;
    }
    // Hand-over: the slot atomics of the whole list go out back to back and are waited for once.  (Writing the hits with
    // plain stores into a per-tile region and assigning slots in a second kernel was no faster, 6.85 ms against 6.77: a block
    // keeps its CU until its last store is acknowledged just as long as until its last atomic returns — 2.5-5k cycles under
    // this load; only work of a next tile could hide that, and the persistent form lost more elsewhere.)
    if (nh) flush();
    if constexpr (DIAG) {
        const unsigned long long t3 = __builtin_readcyclecounter();
        if (lane == 0) { unsigned long long *o = a.diag + ((size_t)blockIdx.x * 8 + wave) * 4; o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t3 - t0;
            if (blockIdx.x % 9000 == 5017 % 9000 && wave == 5) printf("[pair256] tile %u: main-loop waits vmcnt %llu barrier %llu; epilogue: screen %llu walk %llu\n", blockIdx.x, c_vm, c_bar, t2b - t2, t3 - t2b); }
    }
}

void pair_filter256_tile_list(uint32_t n_rows, std::vector<uint32_t> &out) {
    // GS I-panels stay in the XCD's L2 while the J-panels stream past them: a tile fetches 1 / GS of a J-panel from beyond L2
    const uint32_t tiles = (n_rows + p256::BM - 1) / p256::BM, GS = getenv("CX_PAIR_GS") ? (uint32_t)std::max(1, atoi(getenv("CX_PAIR_GS"))) : 4u;
    out.clear();
    out.reserve((size_t)tiles * (tiles + 1) / 2);
    for (uint32_t g0 = 0; g0 < tiles; g0 += GS) {
        const uint32_t g1 = g0 + GS < tiles ? g0 + GS : tiles;
        for (uint32_t tj = g0; tj < tiles; tj++)
            for (uint32_t ti = g0; ti < g1 && ti <= tj; ti++) out.push_back((ti << 16) | tj);
    }
}

template <bool DIAG, int SCHED, bool TA, int NS>
static int launch256(const PairFilterArgs &a, uint32_t tiles, hipStream_t stream) {
    using namespace p256;
    constexpr int lds = NS * SLOT_BYTES;
    static std::atomic<uint64_t> attr_devices{0};
    if (first_use_on_device(attr_devices))
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_filter256_kernel<DIAG, SCHED, TA, NS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (a.ev_begin) CX_HIP(hipEventRecord((hipEvent_t)a.ev_begin, stream));
    hipLaunchKernelGGL((pair_filter256_kernel<DIAG, SCHED, TA, NS>), dim3(tiles), dim3(512), lds, stream, a);
    if (a.ev_end) CX_HIP(hipEventRecord((hipEvent_t)a.ev_end, stream));
    CX_HIP(hipGetLastError());
    return CX_OK;
}

template <bool TA, int NS>
static int launch256_sched(const PairFilterArgs &a, uint32_t tiles, hipStream_t stream) {
    // K-step schedule: 1 = every wave interleaves its DMAs and fragment reads with its own MFMAs (the product);
    // 0 / 3 / 4 = measurement arms ([DMA][reads][MFMAs]; loads only; MFMAs + reads only)
    static const int sched = getenv("CX_PAIR_SCHED") ? atoi(getenv("CX_PAIR_SCHED")) : 1;
    if (getenv("CX_PAIR_DIAG")) {   // diagnostic build: per-phase cycles per tile on stderr, results still valid
        PairFilterArgs d = a;
        const size_t n = (size_t)tiles * 8 * 4;
        CX_HIP(hipMalloc((void **)&d.diag, n * 8));
        if (int rc = launch256<true, 1, TA, NS>(d, tiles, stream)) return rc;
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
    switch (sched) {
        case 0: return launch256<false, 0, TA, NS>(a, tiles, stream);
        case 3: return launch256<false, 3, TA, NS>(a, tiles, stream);
        case 4: return launch256<false, 4, TA, NS>(a, tiles, stream);
        default: return launch256<false, 1, TA, NS>(a, tiles, stream);
    }
}

int launch_pair_filter256(const PairFilterArgs &a, hipStream_t stream) {
    using namespace p256;
    if (a.dim % BK != 0 || a.dim == 0) return set_err(CX_ERR_VALIDATION, "pair filter 256 needs dim %% 32 == 0 (got %u)", a.dim);
    if (!a.n_scan || !a.n_rows) return CX_OK;
    if (!a.shadow_t) return set_err(CX_ERR_VALIDATION, "pair filter 256 needs the tiled shadow");
    if (a.symmetric && (!a.tile_list || (a.n_rows + BM - 1) / BM > 0xFFFFu))
        return set_err(CX_ERR_VALIDATION, "pair filter 256: symmetric pass needs a tile list");
    const uint64_t tiles = a.symmetric ? a.n_tiles : (uint64_t)((a.n_scan + BM - 1) / BM) * ((a.n_rows + BN - 1) / BN);
    if (tiles > 0x7FFFFFFFull) return set_err(CX_ERR_VALIDATION, "pair filter 256: too many tiles");
    const bool tiled_a = !a.shadow_q && !a.scan_rows;   // the scanned rows are this shard's rows, in order
    static const int ns = getenv("CX_PAIR_RING") ? atoi(getenv("CX_PAIR_RING")) : 4;   // LDS ring slots: 4 or 5
    if (ns == 5) return tiled_a ? launch256_sched<true, 5>(a, (uint32_t)tiles, stream) : launch256_sched<false, 5>(a, (uint32_t)tiles, stream);
    return tiled_a ? launch256_sched<true, 4>(a, (uint32_t)tiles, stream) : launch256_sched<false, 4>(a, (uint32_t)tiles, stream);
}

}  // namespace cx
