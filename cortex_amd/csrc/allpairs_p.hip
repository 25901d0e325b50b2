// allpairs_p.hip — the filter GEMM of the all-pairs pass as PERSISTENT blocks (round 3).
//
// Same contract as pair_filter256_kernel (allpairs256.hip): bf16 shadow rows in, candidate columns out, no score
// matrix; replaces the per-node search loop of AutoLinker::run_cycle (linker/auto_linker.rs:215-264) and of
// DedupScanner::scan (linker/dedup.rs:65-127).  Same 256x256x32 K-step on the same 4-slot LDS ring.  What changes is
// everything AROUND the K loop, which at dim 768 (24 K-steps per tile) was 15 % of every tile (profiles/r02/tuning.md §2):
//
//  - one block per CU for the whole launch; a block walks its tiles and the LDS ring runs THROUGH the tile
//    boundaries: the last three K-steps of a tile issue the LDS-DMAs of the next tile's first three, the last one
//    reads the next tile's first fragments — no prologue, no relaunch, no cold ring (prologue: 1.85k of 38k cycles);
//  - the first K-step of a tile starts its accumulators from the MFMA's inline 0 (no 128-register clear);
//  - hits do not leave the CU inside the tile: (i, j) pairs go to a list in LDS (24 KiB behind the ring) and the list is
//    written out — ONE returning atomic per block for the space, coalesced 8-byte stores — only when it holds more than
//    1,024 pairs, every ~30-60 tiles.  A tile's epilogue is then VALU + LDS work only; it never waits for an
//    acknowledgement from L2 (2.5-5k cycles per tile under this load, the largest part of the old epilogue).
//    pair_scatter_kernel turns the pairs into the per-row candidate lists the exact rescore reads (both directions
//    for the mirrored tiles of the symmetric pass);
//  - tiles are dealt to blocks in list order inside each XCD's contiguous share of the tile list, so that the blocks
//    running at any moment on an XCD work on neighbouring tiles and share panels in its L2: statically interleaved
//    (block b of the XCD takes entries b, b + 32, ...) or claimed with s_atomic_add (returns through lgkmcnt, not
//    through the in-order vmcnt queue the ring's counted waits live on; profiles/r02/tuning.md §7).
//
// vmcnt is one in-order queue per wave (MI355X_MICROARCH.md): a global store or returning atomic issued by a wave that
// also issues the ring's LDS-DMAs delays its next counted wait by the store's acknowledgement.  That is why nothing in
// the steady state writes global memory, and why the rare flush simply drains.
#include "kernels.hpp"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

namespace cx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

namespace pp {
constexpr int BM = 256, BN = 256, BK = 32, NS = 4, PF = 3;
constexpr int OP_BYTES = BM * BK * 2;          // 16 KiB per operand per slot
constexpr int SLOT_BYTES = 2 * OP_BYTES;       // 32 KiB
constexpr int RING_BYTES = NS * SLOT_BYTES;    // 128 KiB
constexpr uint32_t HL_CAP = 3072;              // pairs the block's list holds (24 KiB)
constexpr uint32_t HL_FLUSH = 1024;            // written out at the next check once it holds more than this
constexpr int HL_OFF = RING_BYTES;
constexpr int CTL_OFF = HL_OFF + (int)HL_CAP * 8;   // [0] pairs in the list, [1] next tile (dynamic claims), [2] flush base
constexpr int LDS_BYTES = CTL_OFF + 64;
// same LDS image as allpairs256.hip: 16-byte piece p of a 64-byte row at p ^ (row >> 3 & 3)
__device__ inline uint32_t off(uint32_t row, uint32_t piece) { return row * 64u + ((piece ^ ((row >> 3) & 3u)) << 4); }
// LDS control words and the hit list are touched through inline assembly: hipcc tracks every in-flight LDS-DMA as a
// pending LDS write and puts `s_waitcnt vmcnt(0)` in front of any LDS access it cannot tell apart from the ring —
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

// WAVES = 8: 2 (M) x 4 (N) waves of 128 x 64, two per SIMD (256 registers each).
// WAVES = 4: 2 x 2 waves of 128 x 128, one per SIMD (512 registers, accumulators in the upper half).
template <int WAVES, bool DYN>
__global__ __launch_bounds__(WAVES * 64) void pair_filter_p_kernel(const PairFilterArgs a) {
    using namespace pp;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WN = WAVES == 8 ? 4 : 2;            // waves along N
    constexpr int MT = 4, NT = (BN / WN) / 32;        // 32x32 tiles per wave: 4 x 2 or 4 x 4
    constexpr int NPW = 16 / WAVES;                   // 16-row pieces of each operand a wave loads per K-step
    constexpr int ND = 2 * NPW;                       // LDS-DMA instructions per wave and K-step
    constexpr int NM = MT * NT * 2;                   // MFMAs per wave and K-step
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t wm = wave / WN, wn = wave % WN;
    const uint32_t KT = a.dim / BK;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem;   // LDS byte address of smem
    const uint32_t ctl = lds0 + CTL_OFF, hla = lds0 + HL_OFF;
    const uint64_t *hl = reinterpret_cast<const uint64_t *>(smem + HL_OFF);

    // this block's share of the tile order: XCD x (blocks x, x + 8, ... share one) owns a contiguous eighth
    const uint32_t tiles_i = (a.n_scan + BM - 1) / BM, tiles_j = (a.n_rows + BN - 1) / BN;
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
        } else {   // 4 I-panels per J-panel, as pair_filter256_kernel
            const uint32_t GS = 4u, per_group = GS * tiles_j;
            const uint32_t group = idx / per_group, first_i = group * GS;
            const uint32_t gsz = (tiles_i - first_i) < GS ? (tiles_i - first_i) : GS;
            ti = first_i + (idx % per_group) % gsz;
            tj = (idx % per_group) / gsz;
        }
        Desc d;
        d.i0 = ti * BM;
        d.j0 = tj * BN;
        // the tiled shadow is padded to whole 256-row tiles (ensure_shadow): no clamping of the last panel
        d.A = reinterpret_cast<const char *>(a.shadow_t) + (size_t)(d.i0 / 16u + wave * NPW) * KT * 1024u;
        d.B = reinterpret_cast<const char *>(a.shadow_t) + (size_t)(d.j0 / 16u + wave * NPW) * KT * 1024u;
        return d;
    };

    // tile claims.  static: entries local, local + per_xcd, ... of the share.  dynamic: tickets of the XCD's counter.
    uint32_t my = local;                 // position inside the share
    if constexpr (DYN) {
        uint32_t v = 1;
        if (wave == 0) {
            asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(a.pair_ctl + 8u + xcd) : "memory");
            if (lane == 0) lds_write_u32(ctl + 4u, v);
        }
        if (tid == 0) lds_write_u32(ctl, 0u);
        __syncthreads();
        my = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u32(ctl + 4u));
    } else {
        if (tid == 0) lds_write_u32(ctl, 0u);
        __syncthreads();
    }
    if (my >= count) return;   // block-uniform

    const uint32_t voff = lane * 16u;
    auto dma = [&](uint32_t slot, uint32_t kk, const Desc &d, int which) {   // which: 2 q + (0 A | 1 B)
        const int q = which >> 1;
        char *dst = smem + slot * SLOT_BYTES + ((which & 1) ? OP_BYTES : 0) + (wave * NPW + (uint32_t)q) * 1024u;
        const char *src = ((which & 1) ? d.B : d.A) + ((size_t)q * KT + kk) * 1024u + voff;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    auto wait_ring = [&]() {   // all but this wave's youngest K-step of DMAs have landed
        if constexpr (ND == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };

    f32x16 acc[MT][NT];
    const uint32_t fr = lane & 31u, fq = lane >> 5;
    uint32_t fo[2];
#pragma unroll
    for (uint32_t h = 0; h < 2; h++) fo[h] = off(fr, 2u * h + fq);
    const uint32_t baseA = wm * 128u * 64u, baseB = OP_BYTES + wn * (NT * 32u) * 64u;
    bf16x8 fa0[MT * 2], fb0[NT * 2], fa1[MT * 2], fb1[NT * 2];
    auto rdA = [&](uint32_t slot, int m, int h) { return *reinterpret_cast<const bf16x8 *>(smem + (slot * SLOT_BYTES + baseA + fo[h]) + m * 2048); };
    auto rdB = [&](uint32_t slot, int n, int h) { return *reinterpret_cast<const bf16x8 *>(smem + (slot * SLOT_BYTES + baseB + fo[h]) + n * 2048); };

    // One K-step (g = steps since the block started: ring slot g & 3; kt = step inside the tile): the MFMAs of step kt
    // on the fragments read during the step before, and after each MFMA one other instruction of the step — the
    // LDS-DMAs of step kt + 3 (the NEXT tile's when kt + 3 >= KT) and the fragment reads of step kt + 1 (the next tile's
    // step 0 when kt is the last) — then the counted wait and the raw barrier (allpairs256.hip: RAW / WAR argument).
    auto step = [&](auto first_tag, uint32_t g, uint32_t kt, const Desc &cur, const Desc &nxt, const bf16x8 *fa, const bf16x8 *fb,
                    bf16x8 *na, bf16x8 *nb) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint32_t dslot = (g + PF) & 3u, rslot = (g + 1u) & 3u;
        const bool in_cur = kt + PF < KT;
        Desc dd;
        dd.A = in_cur ? cur.A : nxt.A;
        dd.B = in_cur ? cur.B : nxt.B;
        const uint32_t kk = in_cur ? kt + PF : kt + PF - KT;
        __builtin_amdgcn_sched_barrier(0);
        int di = 0, ri = 0;
#pragma unroll
        for (int idx = 0; idx < NM; idx++) {
            const int h = idx / (MT * NT), m = (idx / NT) % MT, n = idx % NT;
            if (FIRST && h == 0) {
                f32x16 z;
#pragma unroll
                for (int e = 0; e < 16; e++) z[e] = 0.0f;
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m * 2 + h], fb[n * 2 + h], z, 0, 0, 0);
            } else {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m * 2 + h], fb[n * 2 + h], acc[m][n], 0, 0, 0);
            }
            // WAVES = 8: 16 MFMAs, 4 DMAs + 12 reads: one per MFMA.  WAVES = 4: 32 MFMAs, 8 DMAs + 16 reads: three per four.
            const bool want_dma = (idx % 4) == 0 && di < ND;
            const bool want_read = !want_dma && (WAVES == 8 || (idx % 4) != 3) && ri < (MT + NT) * 2;
            if (want_dma) {
                dma(dslot, kk, dd, di);
                di++;
            } else if (want_read) {
                if (ri < NT * 2) nb[ri] = rdB(rslot, ri >> 1, ri & 1);
                else na[ri - NT * 2] = rdA(rslot, (ri - NT * 2) >> 1, (ri - NT * 2) & 1);
                ri++;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_ring();
        __builtin_amdgcn_s_barrier();
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

    // epilogue of a tile: screen the accumulators (running maximum + one ballot per 32x32 tile), walk only the tiles
    // with a hit somewhere in the wave, append (i, j) to the block's list.  No global memory, no barrier.
    auto epilogue = [&](const Desc &cur) {
        uint32_t strips = 0;
#pragma unroll
        for (uint32_t m = 0; m < MT; m++)
#pragma unroll
            for (uint32_t n = 0; n < NT; n++) {
                float mx = acc[m][n][0];
#pragma unroll
                for (uint32_t e = 1; e < 16; e++) mx = fmaxf(mx, acc[m][n][e]);
                strips |= __ballot(mx >= a.thr_lo) != 0ull ? 1u << (m * NT + n) : 0u;
            }
        if (!strips) return;
#pragma unroll
        for (uint32_t m = 0; m < MT; m++)
#pragma unroll
            for (uint32_t n = 0; n < NT; n++) {
                if (!((strips >> (m * NT + n)) & 1u)) continue;
                // C layout of the 32x32 tile: register e of lane l holds row 8 (e / 4) + 4 (l >> 5) + e % 4, column l & 31
                const uint32_t j = cur.j0 + wn * (NT * 32u) + n * 32u + fr;
                const uint32_t ibase = cur.i0 + wm * 128u + m * 32u + 4u * fq;
                uint32_t mask = 0;
#pragma unroll
                for (uint32_t e = 0; e < 16; e++) {
                    const uint32_t i = ibase + 8u * (e >> 2) + (e & 3u);
                    mask |= (acc[m][n][e] >= a.thr_lo && i < a.n_scan && j < a.n_rows) ? (1u << e) : 0u;
                }
                // lanes with a hit take list positions round by round: one LDS atomic per round for the whole wave
                for (;;) {
                    const unsigned long long act = __ballot(mask != 0u);
                    if (!act) break;
                    const uint32_t n_act = (uint32_t)__builtin_popcountll(act);
                    const uint32_t leader = (uint32_t)__builtin_ctzll(act);
                    uint32_t base = 0;
                    if (lane == leader) base = lds_add_rtn_u32(ctl, n_act);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
                    if (!mask) continue;
                    const uint32_t e = (uint32_t)__builtin_ctz(mask);
                    mask &= mask - 1u;
                    const uint32_t i = ibase + 8u * (e >> 2) + (e & 3u);
                    const uint64_t v = (uint64_t)i | ((uint64_t)j << 32);
                    const uint32_t pos = base + (uint32_t)__builtin_popcountll(act & ((1ull << lane) - 1ull));
                    if (pos < HL_CAP) {
                        lds_write_u64(hla + pos * 8u, v);
                    } else {   // list full inside one tile (a block of near-duplicates): this hit pays its own round trip
                        const uint32_t p = atomicAdd(a.pair_ctl, 1u);
                        if (p < a.pair_cap) a.pairs[p] = v;
                        else a.pair_ctl[1] = 1u;
                    }
                }
            }
    };

    // ---- the block's life
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
    uint32_t g = 0;
    for (;;) {
        uint32_t nmy;
        bool more;
        if constexpr (!DYN) {
            nmy = my + stride;
            more = nmy < count;
            nxt = make_desc(first + (more ? nmy : my));
        }
        uint32_t ticket = 1;
        step(std::true_type{}, g, 0u, cur, nxt, fa0, fb0, fa1, fb1);
        if constexpr (DYN) {   // claim the next tile: issued here, looked at four K-steps later
            if (wave == 0) asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(ticket) : "s"(a.pair_ctl + 8u + xcd) : "memory");
        }
        step(std::false_type{}, g + 1u, 1u, cur, nxt, fa1, fb1, fa0, fb0);
        {   // every wave has passed a barrier since the last append: the count is the same for all of them
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u32(ctl));
            if (cnt > HL_FLUSH) flush(cnt < HL_CAP ? cnt : HL_CAP);
        }
        uint32_t kt = 2;
        if constexpr (DYN) {
            for (; kt < 6; kt += 2) {
                step(std::false_type{}, g + kt, kt, cur, nxt, fa0, fb0, fa1, fb1);
                step(std::false_type{}, g + kt + 1u, kt + 1u, cur, nxt, fa1, fb1, fa0, fb0);
            }
            if (wave == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ticket) :: "memory");
                if (lane == 0) lds_write_u32(ctl + 4u, ticket);
            }
            for (; kt < 8; kt += 2) {
                step(std::false_type{}, g + kt, kt, cur, nxt, fa0, fb0, fa1, fb1);
                step(std::false_type{}, g + kt + 1u, kt + 1u, cur, nxt, fa1, fb1, fa0, fb0);
            }
            nmy = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u32(ctl + 4u));
            more = nmy < count;
            nxt = make_desc(first + (more ? nmy : my));
        }
        for (; kt < KT; kt += 2) {
            step(std::false_type{}, g + kt, kt, cur, nxt, fa0, fb0, fa1, fb1);
            step(std::false_type{}, g + kt + 1u, kt + 1u, cur, nxt, fa1, fb1, fa0, fb0);
        }
        g += KT;
        epilogue(cur);
        if (!more) break;
        cur = nxt;
        my = nmy;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy DMAs of the tile that never came
    __syncthreads();
    {
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u32(ctl));
        if (cnt) flush(cnt < HL_CAP ? cnt : HL_CAP);
    }
}

// pairs -> the per-row candidate lists the exact rescore reads; a pair of a mirrored tile (symmetric pass, ti != tj)
// enters both rows' lists
__global__ __launch_bounds__(256) void pair_scatter_kernel(const uint64_t *pairs, const uint32_t *pair_ctl, uint32_t pair_cap,
                                                           uint32_t *cand_cnt, uint32_t *cand, uint32_t cap, uint32_t symmetric) {
    const uint32_t total = pair_ctl[0];
    const uint32_t n = total < pair_cap ? total : pair_cap;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const uint64_t v = pairs[t];
        const uint32_t i = (uint32_t)v, j = (uint32_t)(v >> 32);
        const uint32_t s = atomicAdd(cand_cnt + i, 1u);
        if (s < cap) cand[(size_t)i * cap + s] = j;
        if (symmetric && (i >> 8) != (j >> 8)) {
            const uint32_t s2 = atomicAdd(cand_cnt + j, 1u);
            if (s2 < cap) cand[(size_t)j * cap + s2] = i;
        }
    }
}

bool pair_filter_p_supported(const PairFilterArgs &a) {
    // rows scanned in order from the tiled shadow; >= 12 K-steps (the claim hooks sit at steps 1, 6 and 8)
    return a.shadow_t && !a.shadow_q && !a.scan_rows && a.dim % 64u == 0 && a.dim >= 384u && a.pairs && a.pair_ctl && a.pair_cap;
}

template <int WAVES, bool DYN>
static int launch_p(const PairFilterArgs &a, uint32_t grid, hipStream_t stream) {
    static std::atomic<uint64_t> attr_devices{0};
    if (first_use_on_device(attr_devices))
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_filter_p_kernel<WAVES, DYN>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, pp::LDS_BYTES));
    hipLaunchKernelGGL((pair_filter_p_kernel<WAVES, DYN>), dim3(grid), dim3(WAVES * 64), pp::LDS_BYTES, stream, a);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

int launch_pair_filter_p(const PairFilterArgs &a, hipStream_t stream) {
    using namespace pp;
    if (!pair_filter_p_supported(a)) return set_err(CX_ERR_VALIDATION, "persistent pair filter: unsupported arguments");
    if (!a.n_scan || !a.n_rows) return CX_OK;
    if (a.symmetric && (!a.tile_list || (a.n_rows + BM - 1) / BM > 0xFFFFu))
        return set_err(CX_ERR_VALIDATION, "persistent pair filter: symmetric pass needs a tile list");
    const uint64_t tiles = a.symmetric ? a.n_tiles : (uint64_t)((a.n_scan + BM - 1) / BM) * ((a.n_rows + BN - 1) / BN);
    if (tiles > 0x7FFFFFFFull) return set_err(CX_ERR_VALIDATION, "persistent pair filter: too many tiles");
    const int cus = (int)device_cus();
    const int grid_env = getenv("CX_PAIR_P_GRID") ? atoi(getenv("CX_PAIR_P_GRID")) : 0;
    uint32_t grid = (uint32_t)(grid_env > 0 ? grid_env : cus);
    grid = std::max<uint32_t>(8u, grid / 8u * 8u);   // one block per CU, a whole number per XCD
    CX_HIP(hipMemsetAsync(a.pair_ctl, 0, 64, stream));
    const int waves = getenv("CX_PAIR_P_WAVES") ? atoi(getenv("CX_PAIR_P_WAVES")) : 8;
    const int dyn = getenv("CX_PAIR_P_DYN") ? atoi(getenv("CX_PAIR_P_DYN")) : 0;
    int rc;
    if (waves == 4) rc = dyn ? launch_p<4, true>(a, grid, stream) : launch_p<4, false>(a, grid, stream);
    else rc = dyn ? launch_p<8, true>(a, grid, stream) : launch_p<8, false>(a, grid, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(pair_scatter_kernel, dim3(1024), dim3(256), 0, stream, a.pairs, a.pair_ctl, a.pair_cap, a.cand_cnt, a.cand, a.cap,
                       a.symmetric);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

}  // namespace cx
