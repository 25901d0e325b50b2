// allpairs_p.hip — the filter GEMM of the all-pairs pass as PERSISTENT blocks (round 3).
//
// Same contract as pair_filter_kernel (allpairs.hip): bf16 shadow rows in, candidate columns out, no score matrix; replaces
// the per-node search loop of AutoLinker::run_cycle (linker/auto_linker.rs:215-264) and of DedupScanner::scan
// (linker/dedup.rs:65-127).
//
// The K loop (rounds 1-2 built it as a kernel launched per tile, allpairs256.hip, retired in round 4 when row lists and
// external blocks moved here too):
//  - 256x256 block tile, BK = 32, 8 waves as 2 (M) x 4 (N), each wave a 128x64 sub-tile = 4x2 tiles of
//    mfma_f32_32x32x16_bf16 (128 accumulator registers), two waves per SIMD;
//  - per K-step a wave issues 4 LDS-DMA instructions (2 A + 2 B, 1 KiB each: one instruction = one contiguous,
//    pre-swizzled KiB of the tiled shadow, SGPR base + lane * 16), 12 ds_read_b128 and 16 MFMAs, interleaved one-for-one;
//  - LDS = ring of 4 slots x (A 16 KiB + B 16 KiB) = 128 KiB.  In step t a wave issues the DMA of step t+3 and the fragment
//    reads of step t+1 (double-buffered registers), then the MFMAs of step t; the wait at the end of a step is
//    `s_waitcnt vmcnt(4)` — this wave's part of step t+2 has landed, step t+3 stays in flight — followed by a raw
//    s_barrier (a __syncthreads() would drain vmcnt(0)).  All LDS is one array and there are no ordinary global loads in
//    the loop, so hipcc adds no vmcnt(0) of its own (checked in the ISA);
//  - RAW: slot t+2 is read (in step t+1) one barrier after the wait that retired it.  WAR: slot (t+3) % 4 was last read in
//    step t-2; those ds_reads retired before the MFMAs of step t-1 that consumed them;
//  - 16-byte pieces of a 64-byte row are stored at piece ^ (row >> 3 & 3): with rows at a 64-byte stride every
//    ds_read_b128 lane group of the 32-row x 2-piece operand pattern hits 16 different 16-byte bank groups
//    (SQ_LDS_BANK_CONFLICT stays 0);
//  - measured on the way (100k x 768, symmetric): loads only 2.88 ms, MFMAs only 3.87 ms, the per-tile kernel 6.75 ms = their
//    SUM — an in-order wave that is issuing an LDS-DMA feeds the matrix pipe nothing, and its SIMD partner is in the same
//    place; ping-pong between SIMD partners, priorities, staggered DMA slots and a 5-slot ring were all slower
//    (profiles/r02/tuning.md §2).
// What round 3 changed is everything AROUND the K loop, which at dim 768 (24 K-steps per tile) was a quarter of every tile
// (profiles/r03/tuning.md §1):
//
//  - one block per CU for the whole launch; a block walks its tiles and the LDS ring runs THROUGH the tile
//    boundaries: the last three K-steps of a tile issue the LDS-DMAs of the next tile's first three, the last one
//    reads the next tile's first fragments — no prologue, no relaunch, no cold ring;
//  - the first K-step of a tile starts its accumulators from the MFMA's inline 0 (no 128-register clear); the last
//    one runs 32x32 tile by 32x32 tile and screens each finished accumulator (running maximum, one ballot) under the
//    MFMAs of the next ones;
//  - a hit costs its wave one LDS record, not a walk: the LANES that hold a value over the threshold (a handful per
//    tile) copy their 16 accumulator values of that 32x32 tile + its coordinates into a record (80 B) behind the ring;
//    ONE wave picks the tile's records up during the next tile's second K-step — one record per lane, 16 compares,
//    (i, j) pairs appended to a list in LDS — while its SIMD partner and the other SIMDs keep computing.  Walking a
//    hit tile where it was found (16 compares + mask + list positions on all 64 lanes of the wave, the other seven
//    waves waiting at the next barrier for the one with the most hits) was 11.6 % of the kernel's cycles;
//  - the pair list leaves the CU only when it holds more than 1,024 pairs (every ~50 tiles): ONE returning atomic per
//    block for the space, coalesced 8-byte stores.  pair_scatter_kernel turns the pairs into the per-row candidate
//    lists the exact rescore reads (both directions for the mirrored tiles of the symmetric pass).  Nothing in the
//    steady state writes global memory: vmcnt is ONE in-order queue per wave (MI355X_MICROARCH.md), so a store or
//    returning atomic issued between the ring's LDS-DMAs delays the next counted wait by its acknowledgement
//    (2.5-5k cycles under this load — per tile, in the per-tile kernel);
//  - tiles are dealt in list order inside each XCD's contiguous share of the tile list, so that the blocks running
//    at any moment on an XCD work on neighbouring tiles and share panels in its L2: claimed with s_atomic_add
//    (returns through lgkmcnt, not through vmcnt; issued at K-step 1, looked at at K-step 6) or, as a measurement
//    arm, statically interleaved (block b of the XCD takes entries b, b + 32, ...).
#include "kernels.hpp"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

namespace cx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

namespace pp {
constexpr int BN = 256, BK = 32;
constexpr uint32_t REC_BYTES = 80;             // a hit lane's record: its 16 accumulator values of a 32x32 tile, i base, j
// Block shape: BM = 256 scanned rows x 256 rows, 8 waves (2 x 4 of 128 x 64), one block per CU, 4-slot ring.
// (Measured and dropped, profiles/r03/tuning.md §1.4: 128 x 256 tiles on 4-wave blocks, TWO independent blocks per CU on
// 3-slot rings, so that a block's barrier stalls only its own wave of each SIMD.  The barriers did cost less — 6 % of the
// cycles instead of 28 % — but a 128 x 256 tile moves +50 % bytes per flop from L2 to LDS and its LDS-DMAs then cost 20 %
// instead of 15 %: 6.0 ms against 5.7.  On a 3-slot ring the DMA of step t + 3 also targets the slot whose fragment reads
// were issued only one step earlier: a late ds_read can lose the race against an L2-hit DMA, which showed as a handful of
// missing hits in one test run in four.)
template <int BM_>
struct Cfg {
    static constexpr int BM = BM_;
    static constexpr int WAVES = BM_ / 32;                   // 8 | 4
    // PF = 3 K-steps of DMA ahead: what a wave issues in step t is waited for at the end of step t + 1 (the counted wait
    // leaves only the youngest step in flight) and read as fragments in step t + 2 — it is the data of step t + 3 — and
    // goes into the slot of step t - 1, whose fragments were read during step t - 2 and consumed in step t - 1.
    static constexpr int NS = 4, PF = 3;
    static constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BN * BK * 2;
    static constexpr int SLOT_BYTES = A_BYTES + B_BYTES;     // 32 | 24 KiB
    static constexpr int RING_BYTES = NS * SLOT_BYTES;       // 128 | 72 KiB
    static constexpr int NDA = (BM_ / 16) / WAVES, NDB = (BN / 16) / WAVES, ND = NDA + NDB;   // LDS-DMAs per wave and K-step: 2 + 2 | 2 + 4
    static constexpr uint32_t HL_CAP = BM_ == 256 ? 2560 : 640;      // pairs the block's list holds
    static constexpr uint32_t HL_FLUSH = BM_ == 256 ? 1024 : 256;    // written out at the next check once it holds more than this
    static constexpr uint32_t REC_PER_WAVE = BM_ == 256 ? 16 : 8;    // hit-lane records per wave and tile
    static constexpr uint32_t REC_CAP = REC_PER_WAVE * WAVES;
    static constexpr int HL_OFF = RING_BYTES;
    static constexpr int REC_OFF = HL_OFF + (int)HL_CAP * 8;
    static constexpr int CTL_OFF = REC_OFF + (int)(REC_CAP * REC_BYTES);   // [0] pairs in the list, [1] next tile, [2] flush base, [8 + w] records of wave w
    static constexpr int LDS_BYTES = CTL_OFF + 64;
    static_assert(BM_ == 256 && LDS_BYTES <= 160 * 1024, "one block per CU");
};
// the LDS image: 16-byte piece p of a 64-byte row at p ^ (row >> 3 & 3)
__device__ inline uint32_t off(uint32_t row, uint32_t piece) { return row * 64u + ((piece ^ ((row >> 3) & 3u)) << 4); }
// LDS control words, records and the pair list are touched through inline assembly: hipcc tracks every in-flight LDS-DMA
// as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front of any LDS access it cannot tell apart from the ring —
// a drain of the next tile's K-steps in every epilogue (seen in the ISA of the first version); volatile C++ accesses
// went through flat_load/flat_store for the same words.  Addresses are byte offsets in LDS.
__device__ inline uint32_t lds_read_u32(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ inline void lds_write_u32(uint32_t addr, uint32_t v) {
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v) : "memory");
}
__device__ inline void lds_write_u64(uint32_t addr, uint64_t v) { asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ inline void lds_write_f32x4(uint32_t addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ inline uint32_t lds_add_rtn_u32(uint32_t addr, uint32_t v) {
    uint32_t r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(v) : "memory");
    return r;
}
// a word of a read-only table by the scalar path (hipcc takes the vector path — and a vmcnt(0) — for any load it cannot
// prove unclobbered in a kernel that also stores)
__device__ inline uint32_t scalar_load_u32(const uint32_t *p) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
}  // namespace pp

// ARM: 0 = the product; measurement arms (results invalid): 1 = hits are found and dropped (what the hand-over costs),
// 2 = no LDS-DMA inside the K loop, 3 = no fragment reads, 4 = no barriers, 5 = no counted waits and no barriers
template <int BM, bool DYN, bool DIAG, int ARM>
__global__ __launch_bounds__(pp::Cfg<BM>::WAVES * 64, 2) void pair_filter_p_kernel(const PairFilterArgs a) {
    using namespace pp;
    using C = Cfg<BM>;
    constexpr int WAVES = C::WAVES, NS = C::NS, PF = C::PF, SLOT_BYTES = C::SLOT_BYTES, ND = C::ND, NDA = C::NDA, NDB = C::NDB;
    constexpr uint32_t HL_CAP = C::HL_CAP, HL_FLUSH = C::HL_FLUSH, REC_PER_WAVE = C::REC_PER_WAVE, REC_CAP = C::REC_CAP;
    constexpr int HL_OFF = C::HL_OFF, REC_OFF = C::REC_OFF, CTL_OFF = C::CTL_OFF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WN = 4;                             // waves along N
    constexpr int MT = 4, NT = (BN / WN) / 32;        // 32x32 tiles per wave: 4 x 2
    constexpr int NM = MT * NT * 2;                   // MFMAs per wave and K-step
    constexpr int NR = (MT + NT) * 2;                 // fragment reads per wave and K-step
    static_assert(NM == 16 && (ND == 4 || ND == 6), "the K-step's interleave");
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t wm = wave / WN, wn = wave % WN;
    const uint32_t KT = a.dim / BK;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem;   // LDS byte address of smem
    const uint32_t ctl = lds0 + CTL_OFF, hla = lds0 + HL_OFF, reca = lds0 + REC_OFF;
    const uint64_t *hl = reinterpret_cast<const uint64_t *>(smem + HL_OFF);

    // this block's share of the tile order: XCD x (blocks x, x + 8, ... share one) owns a contiguous eighth
    // scanned row i is the shard's row scan_lo + i; tiles start at the 32-row boundary at or below scan_lo — the tiled
    // shadow's pieces are 16 rows, but their 16-byte pieces are pre-swizzled with (row >> 3) & 3 of the GLOBAL row, which
    // equals the LDS tile's own row bits only from a multiple of 32 — and the rows in front of scan_lo fall out with the
    // `i < n_scan` test (unsigned wrap)
    const uint32_t scan_shift = a.scan_lo % 32u, scan_blk0 = a.scan_lo / 32u * 2u;
    const uint32_t tiles_i = (a.n_scan + scan_shift + BM - 1) / BM, tiles_j = (a.n_rows + BN - 1) / BN;
    const uint32_t T = a.symmetric ? a.n_tiles : tiles_i * tiles_j;
    const uint32_t xcd = blockIdx.x % 8u, local = blockIdx.x / 8u, per_xcd = gridDim.x / 8u;
    const uint32_t tq = T / 8u, tr = T % 8u;
    const uint32_t first = xcd < tr ? xcd * (tq + 1u) : tr * (tq + 1u) + (xcd - tr) * tq;
    const uint32_t count = tq + (xcd < tr ? 1u : 0u);

    struct Desc { uint32_t i0, j0; const char *A, *B; };
    auto make_desc = [&](uint32_t idx) {
        uint32_t ti, tj;
        if (a.symmetric) {
            const uint32_t t = scalar_load_u32(a.tile_list + idx);
            ti = t >> 16;
            tj = t & 0xFFFFu;
        } else {   // 1,024 scanned rows of I-panels per J-panel
            const uint32_t GS = 1024u / BM, per_group = GS * tiles_j;
            const uint32_t group = idx / per_group, first_i = group * GS;
            const uint32_t gsz = (tiles_i - first_i) < GS ? (tiles_i - first_i) : GS;
            ti = first_i + (idx % per_group) % gsz;
            tj = (idx % per_group) / gsz;
        }
        Desc d;
        d.i0 = ti * BM;
        d.j0 = tj * BN;
        // the tiled shadow is padded to whole 256-row tiles (ensure_shadow): no clamping of the last panel
        d.A = reinterpret_cast<const char *>(a.shadow_i ? a.shadow_i : a.shadow_t) + (size_t)(scan_blk0 + d.i0 / 16u + wave * NDA) * KT * 1024u;
        d.B = reinterpret_cast<const char *>(a.shadow_t) + (size_t)(d.j0 / 16u + wave * NDB) * KT * 1024u;
        return d;
    };

    // tile claims.  static: entries local, local + per_xcd, ... of the share.  dynamic: tickets of the XCD's counter.
    uint32_t my = local;                 // position inside the share
    if (tid < 16u) lds_write_u32(ctl + 4u * tid, 0u);
    if constexpr (DYN) {
        uint32_t v = 1;
        if (wave == 0) {
            asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(a.pair_ctl + 8u + xcd) : "memory");
            if (lane == 0) lds_write_u32(ctl + 4u, v);
        }
        __syncthreads();
        my = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u32(ctl + 4u));
    } else {
        __syncthreads();
    }
    if (my >= count) return;   // block-uniform

    const uint32_t voff = lane * 16u;
    // the wave's ND pieces of a K-step: A pieces wave * NDA + q, B pieces wave * NDB + q, B and A alternating
    auto dma = [&](uint32_t slot, uint32_t kk, const Desc &d, int which) {
        constexpr int per_a = ND / NDA;                       // every per_a-th instruction is an A piece
        const bool is_a = which % per_a == 0;
        const int q = is_a ? which / per_a : which - which / per_a - 1;
        char *dst = smem + slot * SLOT_BYTES + (is_a ? (wave * NDA + (uint32_t)q) * 1024u : C::A_BYTES + (wave * NDB + (uint32_t)q) * 1024u);
        const char *src = (is_a ? d.A : d.B) + ((size_t)q * KT + kk) * 1024u + voff;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    auto wait_ring = [&]() {   // all but this wave's youngest K-step of DMAs have landed
        if constexpr (ND == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    };

    f32x16 acc[MT][NT];
    float mxs[MT * NT];        // per lane: largest of its 16 values of each 32x32 tile (set by the tile's last K-step)
    uint32_t strips = 0;       // 32x32 tiles of this wave with a hit somewhere
    uint32_t cur_i0 = 0;       // first scanned row of the tile whose records are waiting to be picked up
    unsigned long long c_vm = 0, c_bar = 0, c_main = 0, c_epi = 0, c_hook = 0;   // DIAG: cycles in the counted waits, barriers, K loops, epilogues, hooks
    const uint32_t fr = lane & 31u, fq = lane >> 5;
    uint32_t fo[2];
#pragma unroll
    for (uint32_t h = 0; h < 2; h++) fo[h] = off(fr, 2u * h + fq);
    const uint32_t baseA = wm * 128u * 64u, baseB = C::A_BYTES + wn * (NT * 32u) * 64u;
    bf16x8 fa0[MT * 2], fb0[NT * 2], fa1[MT * 2], fb1[NT * 2];
    auto rdA = [&](uint32_t slot, int m, int h) { return *reinterpret_cast<const bf16x8 *>(smem + (slot * SLOT_BYTES + baseA + fo[h]) + m * 2048); };
    auto rdB = [&](uint32_t slot, int n, int h) { return *reinterpret_cast<const bf16x8 *>(smem + (slot * SLOT_BYTES + baseB + fo[h]) + n * 2048); };

    auto screen_tile = [&](int t) {
        const int m = t / NT, n = t % NT;
        float mx = acc[m][n][0];
#pragma unroll
        for (uint32_t e = 1; e < 16; e++) mx = fmaxf(mx, acc[m][n][e]);
        mxs[t] = mx;
        strips |= __ballot(mx >= a.thr_lo) != 0ull ? 1u << t : 0u;
    };
    // one pair into the block's list; a full list (a block of near-duplicates inside one tile) sends it straight out
    auto append_pair = [&](uint32_t i, uint32_t j) {
        const uint64_t v = (uint64_t)i | ((uint64_t)j << 32);
        const uint32_t pos = lds_add_rtn_u32(ctl, 1u);
        if (pos < HL_CAP) {
            lds_write_u64(hla + pos * 8u, v);
        } else {
            const uint32_t p = atomicAdd(a.pair_ctl, 1u);
            if (p < a.pair_cap) a.pairs[p] = v;
            else a.pair_ctl[1] = 1u;
        }
    };
    // the 16 values of one lane of a 32x32 tile: register e holds row ibase + 8 (e / 4) + e % 4 (ibase includes 4 (lane >> 5))
    auto emit_lane = [&](const float *v, uint32_t ibase, uint32_t j) {
        uint32_t mask = 0;
#pragma unroll
        for (uint32_t e = 0; e < 16; e++) mask |= v[e] >= a.thr_lo ? 1u << e : 0u;
        if (j >= a.n_rows) mask = 0;
        while (mask) {
            const uint32_t e = (uint32_t)__builtin_ctz(mask);
            mask &= mask - 1u;
            const uint32_t i = ibase + 8u * (e >> 2) + (e & 3u);
            if (i < a.n_scan) append_pair(i, j);
        }
    };

    // the list goes out: space from ONE returning atomic, coalesced 8-byte stores.  Block-uniform; drains (rare).
    auto flush = [&](uint32_t cnt) {
        if (tid == 0) lds_write_u32(ctl + 8u, atomicAdd(a.pair_ctl, cnt));
        __syncthreads();
        const uint32_t base = lds_read_u32(ctl + 8u);
        for (uint32_t e = tid; e < cnt; e += WAVES * 64u)
            if (base + e < a.pair_cap) a.pairs[base + e] = hl[e];
        if (tid == 0 && (uint64_t)base + cnt > a.pair_cap) a.pair_ctl[1] = 1u;
        __syncthreads();
        if (tid == 0) lds_write_u32(ctl, 0u);
    };

    // A 32x32 tile with a hit: the lanes that hold a value over the threshold leave a record each in the wave's own 16
    // slots — no returning LDS operation, no wait: the LDS is busy with the ring's traffic and a round trip costs 300-600
    // cycles here.  Called from INSIDE the tile's last K-step, right after the tile was screened, so that the writes drain
    // under the remaining MFMAs (issued behind the step's barrier, the next step's first `s_waitcnt lgkmcnt(0)` — the
    // compiler's wait for its fragment reads — waited for them instead: 1-2k cycles per tile).
    uint32_t rec_base = 0;     // records this wave has written for the current tile
    auto record_tile = [&](int t, const Desc &cur) {
        if constexpr (ARM == 1) { asm volatile("" :: "s"(strips)); return; }
        if (!((strips >> t) & 1u)) return;
        const int m = t / NT, n = t % NT;
        const bool mine = mxs[t] >= a.thr_lo;
        const unsigned long long bal = __ballot(mine);
        const uint32_t j = cur.j0 + wn * (NT * 32u) + (uint32_t)n * 32u + fr;
        const uint32_t ibase = cur.i0 + wm * 128u + (uint32_t)m * 32u + 4u * fq - scan_shift;   // scan position (wraps for the rows in front of scan_lo)
        if (mine) {
            const uint32_t slot = rec_base + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
            if (slot < REC_PER_WAVE) {
                const uint32_t ra = reca + (wave * REC_PER_WAVE + slot) * REC_BYTES;
                lds_write_f32x4(ra, __builtin_shufflevector(acc[m][n], acc[m][n], 0, 1, 2, 3));
                lds_write_f32x4(ra + 16u, __builtin_shufflevector(acc[m][n], acc[m][n], 4, 5, 6, 7));
                lds_write_f32x4(ra + 32u, __builtin_shufflevector(acc[m][n], acc[m][n], 8, 9, 10, 11));
                lds_write_f32x4(ra + 48u, __builtin_shufflevector(acc[m][n], acc[m][n], 12, 13, 14, 15));
                lds_write_u64(ra + 64u, (uint64_t)ibase | ((uint64_t)j << 32));
            } else {   // more hit lanes in this wave's part of the tile than records: this lane walks its values here
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; e++) v[e] = acc[m][n][e];
                emit_lane(v, ibase, j);
            }
        }
        rec_base += (uint32_t)__builtin_popcountll(bal);
    };
    auto publish_records = [&]() {   // before the last K-step's barrier: drains while the wave waits for the others
        if constexpr (ARM == 1) return;
        if (rec_base && lane == 0)
            asm volatile("ds_write_b32 %0, %1" :: "v"(ctl + 32u + 4u * wave), "v"(rec_base < REC_PER_WAVE ? rec_base : REC_PER_WAVE) : "memory");
        rec_base = 0;
    };
    // One K-step (g = steps since the block started: ring slot g & 3; kt = step inside the tile): the MFMAs of step kt
    // on the fragments read during the step before, and after each MFMA one other instruction of the step — the
    // LDS-DMAs of step kt + 3 (the NEXT tile's when kt + 3 >= KT) and the fragment reads of step kt + 1 (the next tile's
    // step 0 when kt is the last) — then the counted wait and the raw barrier (the RAW / WAR argument: head of the file).
    // KIND 1 = first step of a tile (accumulators start from the inline 0), 2 = last (tile by tile, screening under the MFMAs).
    auto step = [&](auto kind_tag, uint32_t g, uint32_t kt, const Desc &cur, const Desc &nxt, const bf16x8 *fa, const bf16x8 *fb,
                    bf16x8 *na, bf16x8 *nb) {
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr bool FIRST = KIND == 1, LAST = KIND == 2;
        const uint32_t dslot = (g + PF) % NS, rslot = (g + 1u) % NS;
        const bool in_cur = kt + PF < KT;
        Desc dd;
        dd.A = in_cur ? cur.A : nxt.A;
        dd.B = in_cur ? cur.B : nxt.B;
        const uint32_t kk = in_cur ? kt + PF : kt + PF - KT;
        __builtin_amdgcn_sched_barrier(0);
        int di = 0, ri = 0;
#pragma unroll
        for (int idx = 0; idx < NM; idx++) {
            int h = idx / (MT * NT), m = (idx / NT) % MT, n = idx % NT;
            if (LAST) {   // tile by tile: (m, n) is final after its second MFMA
                h = idx & 1;
                m = (idx >> 1) / NT;
                n = (idx >> 1) % NT;
            }
            if (FIRST && h == 0) {
                f32x16 z;
#pragma unroll
                for (int e = 0; e < 16; e++) z[e] = 0.0f;
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m * 2 + h], fb[n * 2 + h], z, 0, 0, 0);
            } else {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m * 2 + h], fb[n * 2 + h], acc[m][n], 0, 0, 0);
            }
            // the step's ND DMAs and NR fragment reads, spread evenly behind the 16 MFMAs (ND = 4: one each; ND = 6: two of the
            // MFMAs are followed by two); op k is a DMA when k * ND mod (ND + NR) < ND
            for (int k = idx * (ND + NR) / NM; k < (idx + 1) * (ND + NR) / NM; k++) {
                if ((k * ND) % (ND + NR) < ND) {
                    if constexpr (ARM != 2) dma(dslot, kk, dd, di);
                    di++;
                } else {
                    if constexpr (ARM != 3) {
                        if (ri < NT * 2) nb[ri] = rdB(rslot, ri >> 1, ri & 1);
                        else na[ri - NT * 2] = rdA(rslot, (ri - NT * 2) >> 1, (ri - NT * 2) & 1);
                    }
                    ri++;
                }
            }
            if (LAST && (idx & 1) && idx >= 3) {   // finished one pair of MFMAs ago: screened, and its hit lanes recorded, under this pair
                screen_tile((idx >> 1) - 1);
                record_tile((idx >> 1) - 1, cur);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LAST) {
            screen_tile(MT * NT - 1);
            record_tile(MT * NT - 1, cur);
            publish_records();
        }
        unsigned long long w0 = 0, w1 = 0;
        if constexpr (DIAG) { asm volatile("s_nop 0" :: "v"(acc[MT - 1][NT - 1][15])); w0 = __builtin_readcyclecounter(); }
        if constexpr (ARM != 5) wait_ring();
        if constexpr (DIAG) w1 = __builtin_readcyclecounter();
        if constexpr (ARM != 4 && ARM != 5) __builtin_amdgcn_s_barrier();
        if constexpr (DIAG) { const unsigned long long w2 = __builtin_readcyclecounter(); c_vm += w1 - w0; c_bar += w2 - w1; }
    };

    // the records of the tile before: one per lane of ONE wave (every wave has passed a barrier since the last record was
    // written), while the partner wave on its SIMD and the other SIMDs compute
    auto pick_up_records = [&](uint32_t turn) {
        if (wave != turn % WAVES) return;
        // Two LDS round trips in all (each 300-600 cycles under the ring's traffic): every lane reads the counts and BOTH of
        // its record slots at once (lane l: slots l and l + 64; a slot without a record reads stale bytes and is masked),
        // then takes the list positions of all its hits with one atomic.  The partner wave's 16 MFMAs cover about that long.
        const uint32_t s0 = lane % REC_CAP, s1 = (lane + 64u) % REC_CAP;   // REC_CAP = 128: two slots per lane; 32: lanes 0-31 one slot each
        f32x4 q[8];
        uint64_t hd0, hd1;
        uint32_t c0, c1;
        asm volatile("ds_read_b32 %10, %12\n\tds_read_b32 %11, %13\n\t"
                     "ds_read_b128 %0, %14\n\tds_read_b128 %1, %14 offset:16\n\tds_read_b128 %2, %14 offset:32\n\tds_read_b128 %3, %14 offset:48\n\t"
                     "ds_read_b64 %8, %14 offset:64\n\t"
                     "ds_read_b128 %4, %15\n\tds_read_b128 %5, %15 offset:16\n\tds_read_b128 %6, %15 offset:32\n\tds_read_b128 %7, %15 offset:48\n\t"
                     "ds_read_b64 %9, %15 offset:64\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]), "=&v"(hd0), "=&v"(hd1),
                       "=&v"(c0), "=&v"(c1)
                     : "v"(ctl + 32u + 4u * (s0 / REC_PER_WAVE)), "v"(ctl + 32u + 4u * (s1 / REC_PER_WAVE)), "v"(reca + s0 * REC_BYTES), "v"(reca + s1 * REC_BYTES)
                     : "memory");
        if (lane < (uint32_t)WAVES) asm volatile("ds_write_b32 %0, %1" :: "v"(ctl + 32u + 4u * lane), "v"(0u) : "memory");   // this wave has them all: the counts start over
        uint32_t mask[2] = {0u, 0u};
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int e = 0; e < 16; e++) mask[r] |= q[r * 4 + e / 4][e % 4] >= a.thr_lo ? 1u << e : 0u;
        }
        if (!((s0 % REC_PER_WAVE) < c0) || (uint32_t)(hd0 >> 32) >= a.n_rows || lane >= REC_CAP) mask[0] = 0u;
        if (!((s1 % REC_PER_WAVE) < c1) || (uint32_t)(hd1 >> 32) >= a.n_rows || REC_CAP <= 64u) mask[1] = 0u;
        if (a.n_scan + scan_shift - cur_i0 < (uint32_t)BM || (scan_shift && cur_i0 == 0u)) {   // the first / last panel of scanned rows: rows outside [0, n_scan) are not pairs
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const uint32_t ib = (uint32_t)(r ? hd1 : hd0);
#pragma unroll
                for (uint32_t e = 0; e < 16; e++)
                    if (ib + 8u * (e >> 2) + (e & 3u) >= a.n_scan) mask[r] &= ~(1u << e);
            }
        }
        const uint32_t n_mine = (uint32_t)__builtin_popcount(mask[0]) + (uint32_t)__builtin_popcount(mask[1]);
        if (n_mine) {
            uint32_t pos = lds_add_rtn_u32(ctl, n_mine);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const uint64_t hd = r ? hd1 : hd0;
                uint32_t mk = mask[r];
                while (mk) {
                    const uint32_t e = (uint32_t)__builtin_ctz(mk);
                    mk &= mk - 1u;
                    const uint64_t v = (uint64_t)((uint32_t)hd + 8u * (e >> 2) + (e & 3u)) | (hd & 0xFFFFFFFF00000000ull);
                    if (pos < HL_CAP) {
                        lds_write_u64(hla + pos * 8u, v);
                    } else {
                        const uint32_t p = atomicAdd(a.pair_ctl, 1u);
                        if (p < a.pair_cap) a.pairs[p] = v;
                        else a.pair_ctl[1] = 1u;
                    }
                    pos++;
                }
            }
        }
    };

    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    // ---- the block's life
    // block 0 stamps the shader clock against the 100 MHz reference around its whole life (two scalar reads per launch):
    // what the chip's power management lets this kernel run at is half of every MFMA utilisation figure (tuning.md)
    unsigned long long t_clk = 0, t_ref = 0;
    if (blockIdx.x == 0) {
        t_clk = __builtin_amdgcn_s_memtime();
        t_ref = __builtin_amdgcn_s_memrealtime();
    }
    const uint32_t stride = per_xcd;
    Desc cur = make_desc(first + my), nxt = cur;
#pragma unroll
    for (uint32_t st = 0; st < PF; st++)
#pragma unroll
        for (int w = 0; w < ND; w++) dma(st, st, cur, w);
    wait_ring();
    __builtin_amdgcn_s_barrier();
    {
#pragma unroll
        for (int r = 0; r < NT * 2; r++) fb0[r] = rdB(0u, r >> 1, r & 1);
#pragma unroll
        for (int r = 0; r < MT * 2; r++) fa0[r] = rdA(0u, r >> 1, r & 1);
    }
    uint32_t g = 0, turn = 0;
    for (;;) {
        uint32_t nmy;
        bool more;
        if constexpr (!DYN) {
            nmy = my + stride;
            more = nmy < count;
            nxt = make_desc(first + (more ? nmy : my));
        }
        uint32_t ticket = 1;
        unsigned long long d0 = 0, d1 = 0, d2 = 0;
        if constexpr (DIAG) d0 = __builtin_readcyclecounter();
        strips = 0;
        step(K1{}, g, 0u, cur, nxt, fa0, fb0, fa1, fb1);
        if constexpr (DYN) {   // claim the next tile: issued here, looked at five K-steps later
            if (wave == 0) asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(ticket) : "s"(a.pair_ctl + 8u + xcd) : "memory");
        }
        step(K0{}, g + 1u, 1u, cur, nxt, fa1, fb1, fa0, fb0);
        pick_up_records(turn++);
        // The first eight steps are written out: with the hooks inside a loop over pairs of steps (`if (kt == 2)` ...) the
        // kernel spills fewer scalar registers (12 against 41) but runs 5 % more cycles — the waits hipcc places where
        // control flow joins inside the loop are not the counted ones.
        step(K0{}, g + 2u, 2u, cur, nxt, fa0, fb0, fa1, fb1);
        step(K0{}, g + 3u, 3u, cur, nxt, fa1, fb1, fa0, fb0);
        {   // every wave has passed two barriers since the last append: the count is the same for all of them
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u32(ctl));
            if (cnt > HL_FLUSH) flush(cnt < HL_CAP ? cnt : HL_CAP);
        }
        step(K0{}, g + 4u, 4u, cur, nxt, fa0, fb0, fa1, fb1);
        step(K0{}, g + 5u, 5u, cur, nxt, fa1, fb1, fa0, fb0);
        if constexpr (DYN) {
            if (wave == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ticket) :: "memory");
                if (lane == 0) lds_write_u32(ctl + 4u, ticket);
            }
        }
        step(K0{}, g + 6u, 6u, cur, nxt, fa0, fb0, fa1, fb1);
        step(K0{}, g + 7u, 7u, cur, nxt, fa1, fb1, fa0, fb0);
        if constexpr (DYN) {
            nmy = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u32(ctl + 4u));
            more = nmy < count;
            nxt = make_desc(first + (more ? nmy : my));
        }
        uint32_t kt = 8;
        for (; kt + 2u < KT; kt += 2) {
            step(K0{}, g + kt, kt, cur, nxt, fa0, fb0, fa1, fb1);
            step(K0{}, g + kt + 1u, kt + 1u, cur, nxt, fa1, fb1, fa0, fb0);
        }
        step(K0{}, g + kt, kt, cur, nxt, fa0, fb0, fa1, fb1);
        step(K2{}, g + kt + 1u, kt + 1u, cur, nxt, fa1, fb1, fa0, fb0);
        g += KT;
        if constexpr (DIAG) { d1 = __builtin_readcyclecounter(); d2 = d1; c_main += d1 - d0; c_epi += d2 - d1; c_hook += (unsigned long long)__builtin_popcount(strips); }
        cur_i0 = cur.i0;
        if (!more) break;
        cur = nxt;
        my = nmy;
    }
    // the dummy DMAs of the tile that never came; and this wave's last records (inline assembly: hipcc's own wait before the
    // barrier does not know them) must be in LDS before another wave reads them behind the barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    pick_up_records(0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (DIAG) {
        if (lane == 0) {
            unsigned long long *o = a.diag + ((size_t)blockIdx.x * WAVES + wave) * 8;
            o[0] = c_main; o[1] = c_epi; o[2] = c_vm; o[3] = c_bar; o[4] = g / KT; o[5] = c_hook;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.pair_ctl + 16);
        o[0] = __builtin_amdgcn_s_memtime() - t_clk;
        o[1] = __builtin_amdgcn_s_memrealtime() - t_ref;
        a.pair_ctl[20] = g / KT;   // tiles this block walked
    }
    {
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u32(ctl));
        if (cnt) flush(cnt < HL_CAP ? cnt : HL_CAP);
    }
}

// pairs -> the per-row candidate lists the exact rescore reads.  Symmetric pass, blocks of 2^shift rows: a pair inside a
// diagonal block was found in both orders by the tile that computed the block; a pair above the diagonal enters both rows'
// lists; a pair below it (the 128 x 256 tiles' second column block on odd panels) is the mirror of one found elsewhere
__global__ __launch_bounds__(256) void pair_scatter_kernel(const uint64_t *pairs, const uint32_t *pair_ctl, uint32_t pair_cap,
                                                           uint32_t *cand_cnt, uint32_t *cand, uint32_t cap, uint32_t symmetric, uint32_t shift) {
    const uint32_t total = pair_ctl[0];
    const uint32_t n = total < pair_cap ? total : pair_cap;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const uint64_t v = pairs[t];
        const uint32_t i = (uint32_t)v, j = (uint32_t)(v >> 32);
        if (symmetric && (j >> shift) < (i >> shift)) continue;
        const uint32_t s = atomicAdd(cand_cnt + i, 1u);
        if (s < cap) cand[(size_t)i * cap + s] = j;
        if (symmetric && (i >> shift) != (j >> shift)) {
            const uint32_t s2 = atomicAdd(cand_cnt + j, 1u);
            if (s2 < cap) cand[(size_t)j * cap + s2] = i;
        }
    }
}

bool pair_filter_p_supported(const PairFilterArgs &a) {
    // the scanned vectors as tiled shadow rows in order — a run of the shard's own rows, or a staged panel (shadow_i) —; >= 12 K-steps
    // (the hooks of a tile sit behind its steps 1, 3, 5 and 7)
    const bool own = !a.shadow_i && (!a.scan_rows || a.scan_contig) && (uint64_t)a.scan_lo + a.n_scan <= a.n_rows;
    const bool staged = a.shadow_i && a.scan_lo == 0u && a.scan_contig && !a.symmetric;
    return a.shadow_t && !a.shadow_q && (own || staged) && a.dim % 64u == 0 && a.dim >= 384u && a.pairs && a.pair_ctl && a.pair_cap;
}

// A list of scanned rows as a staged I panel: one 16-byte piece per thread, copied from the row's place in the tiled shadow to
// position i's; the pieces of a K-step are XOR-permuted with bits 3-4 of the row they sit at, so a piece changes its slot when
// (row >> 3) & 3 differs between source and destination — tiled_shadow_off on both sides.  Rows n_scan .. n_pad are zero.
__global__ __launch_bounds__(256) void stage_scan_rows_kernel(const uint16_t *shadow_t, const uint32_t *scan_rows, uint32_t n_scan, uint32_t n_pad, uint32_t dim,
                                                              uint16_t *out) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t pieces = dim / 8u, kt32 = dim / 32u;
    const uint64_t total = (uint64_t)n_pad * pieces;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(t / pieces), p = (uint32_t)(t % pieces);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (i < n_scan) v = *reinterpret_cast<const u32x4 *>(shadow_t + tiled_shadow_off(scan_rows[i], p, kt32));
        *reinterpret_cast<u32x4 *>(out + tiled_shadow_off(i, p, kt32)) = v;
    }
}
int launch_stage_scan_rows(const uint16_t *shadow_t, const uint32_t *d_scan_rows, uint32_t n_scan, uint32_t dim, uint16_t *out, hipStream_t stream) {
    if (!n_scan) return CX_OK;
    if (dim % 32u) return set_err(CX_ERR_VALIDATION, "staged scan rows need dim %% 32 == 0 (got %u)", dim);
    const uint32_t n_pad = (n_scan + 255u) / 256u * 256u;
    const uint64_t total = (uint64_t)n_pad * (dim / 8u);
    const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255u) / 256u, 16384u);
    hipLaunchKernelGGL(stage_scan_rows_kernel, dim3(grid), dim3(256), 0, stream, shadow_t, d_scan_rows, n_scan, n_pad, dim, out);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

template <int BM, bool DYN, bool DIAG = false, int ARM = 0>
static int launch_p(const PairFilterArgs &a, uint32_t grid, hipStream_t stream) {
    using C = pp::Cfg<BM>;
    static std::atomic<uint64_t> attr_devices{0};
    if (first_use_on_device(attr_devices))
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_filter_p_kernel<BM, DYN, DIAG, ARM>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    if (a.ev_begin) CX_HIP(hipEventRecord((hipEvent_t)a.ev_begin, stream));
    hipLaunchKernelGGL((pair_filter_p_kernel<BM, DYN, DIAG, ARM>), dim3(grid), dim3(C::WAVES * 64), C::LDS_BYTES, stream, a);
    if (a.ev_end) CX_HIP(hipEventRecord((hipEvent_t)a.ev_end, stream));
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// CX_PAIR_DIAG: per-phase cycles (s_memtime; the stamps themselves cost ~6 %), results still valid
template <int BM>
static int launch_p_diag(const PairFilterArgs &a, uint32_t grid, bool dyn, hipStream_t stream) {
    PairFilterArgs d = a;
    const size_t n = (size_t)grid * 8 * 8;
    CX_HIP(hipMalloc((void **)&d.diag, n * 8));
    CX_HIP(hipMemsetAsync(d.diag, 0, n * 8, stream));
    if (int rc = dyn ? launch_p<BM, true, true>(d, grid, stream) : launch_p<BM, false, true>(d, grid, stream)) return rc;
    CX_HIP(hipStreamSynchronize(stream));
    std::vector<unsigned long long> h(n);
    CX_HIP(hipMemcpy(h.data(), d.diag, n * 8, hipMemcpyDeviceToHost));
    CX_HIP(hipFree(d.diag));
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (size_t w = 0; w < (size_t)grid * 8; w++) for (int p = 0; p < 6; p++) s[p] += (double)h[w * 8 + p];
    const double nt = s[4] > 0 ? s[4] : 1.0;   // wave-tiles
    fprintf(stderr, "[pair_p diag] %.0f wave-tiles, %.0f 32x32 tiles with a hit; cycles per tile and wave: K loop %.0f (counted waits %.0f, barriers %.0f)\n",
            nt, s[5], s[0] / nt, s[2] / nt, s[3] / nt);
    return CX_OK;
}

uint32_t pair_filter_p_block_rows() { return 256u; }   // scanned rows per tile

// live tiles of the symmetric pass for tiles of bm scanned rows x 256 rows: (ti << 16) | tj with the tile's last column
// at or beyond its first row, GS I-panels (1,024 scanned rows) per J-panel
void pair_filter_p_tile_list(uint32_t n_rows, uint32_t bm, std::vector<uint32_t> &out) {
    const uint32_t tiles_i = (n_rows + bm - 1) / bm, tiles_j = (n_rows + 255u) / 256u, GS = 1024u / bm;
    out.clear();
    for (uint32_t g0 = 0; g0 < tiles_i; g0 += GS) {
        const uint32_t g1 = std::min(g0 + GS, tiles_i);
        for (uint32_t tj = (uint32_t)((uint64_t)g0 * bm / 256u); tj < tiles_j; tj++)
            for (uint32_t ti = g0; ti < g1; ti++)
                if ((uint64_t)tj * 256u + 255u >= (uint64_t)ti * bm) out.push_back((ti << 16) | tj);
    }
}

template <int BM>
static int launch_cfg(const PairFilterArgs &a, hipStream_t stream) {
    using C = pp::Cfg<BM>;
    const int cus = (int)device_cus();
    const int grid_env = getenv("CX_PAIR_P_GRID") ? atoi(getenv("CX_PAIR_P_GRID")) : 0;
    uint32_t grid = (uint32_t)(grid_env > 0 ? grid_env : cus * (256 / BM));
    grid = std::max<uint32_t>(8u, grid / 8u * 8u);   // 256 / BM blocks per CU, a whole number per XCD
    const int dyn = getenv("CX_PAIR_P_DYN") ? atoi(getenv("CX_PAIR_P_DYN")) : 1;   // tile claims: 1 s_atomic_add tickets, 0 static interleave
    const int arm = getenv("CX_PAIR_P_ARM") ? atoi(getenv("CX_PAIR_P_ARM")) : 0;
    (void)sizeof(C);
    if (getenv("CX_PAIR_DIAG")) return launch_p_diag<BM>(a, grid, dyn != 0, stream);
    switch (arm) {
        case 1: return launch_p<BM, true, false, 1>(a, grid, stream);
        case 2: return launch_p<BM, true, false, 2>(a, grid, stream);
        case 4: return launch_p<BM, true, false, 4>(a, grid, stream);
        default: return dyn ? launch_p<BM, true>(a, grid, stream) : launch_p<BM, false>(a, grid, stream);
    }
}

int launch_pair_filter_p(const PairFilterArgs &a, hipStream_t stream) {
    if (!pair_filter_p_supported(a)) return set_err(CX_ERR_VALIDATION, "persistent pair filter: unsupported arguments");
    if (!a.n_scan || !a.n_rows) return CX_OK;
    const uint32_t bm = 256u;
    if (a.symmetric && (!a.tile_list || (a.n_rows + bm - 1) / bm > 0xFFFFu))
        return set_err(CX_ERR_VALIDATION, "persistent pair filter: symmetric pass needs a tile list");
    const uint64_t tiles = a.symmetric ? a.n_tiles : (uint64_t)((a.n_scan + a.scan_lo % 32u + bm - 1) / bm) * ((a.n_rows + 255u) / 256u);
    if (tiles > 0x7FFFFFFFull) return set_err(CX_ERR_VALIDATION, "persistent pair filter: too many tiles");
    CX_HIP(hipMemsetAsync(a.pair_ctl, 0, 128, stream));
    if (int rc = launch_cfg<256>(a, stream)) return rc;
    hipLaunchKernelGGL(pair_scatter_kernel, dim3(1024), dim3(256), 0, stream, a.pairs, a.pair_ctl, a.pair_cap, a.cand_cnt, a.cand, a.cap,
                       a.symmetric, 8u);
    CX_HIP(hipGetLastError());
    if (getenv("CX_PAIR_P_CLOCK")) {   // measurement: the clock block 0 saw, pairs handed over
        uint32_t h[32];
        CX_HIP(hipStreamSynchronize(stream));
        CX_HIP(hipMemcpy(h, a.pair_ctl, sizeof h, hipMemcpyDeviceToHost));
        unsigned long long clk, ref;
        memcpy(&clk, h + 16, 8);
        memcpy(&ref, h + 18, 8);
        fprintf(stderr, "[pair_p] block 0: %llu shader cycles in %.3f ms = %.3f GHz, %u tiles; %u pairs%s\n", clk, (double)ref / 1.0e5,
                ref ? (double)clk / (double)ref * 0.1 : 0.0, h[20], h[0], h[1] ? " (pair buffer overflow)" : "");
    }
    return CX_OK;
}

}  // namespace cx
