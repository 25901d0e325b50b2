// batchq.hip — batched exact search at the reference's default row width (384-d, embedding.rs:43-50; search_batch,
// vector/index.rs:390-410): up to 64 queries per pass over the rows, the QUERIES resident in LDS and the ROWS going
// straight from HBM into MFMA operand registers.
//
// batch2_kernel (batch.hip) keeps the queries in the consumers' registers and moves every row tile HBM -> registers ->
// LDS -> registers through four producer waves, one block barrier per 16-row tile; at 384-d a tile is 24 KiB and the
// per-tile costs (barrier, LDS round trip, candidate bookkeeping) held it at 0.57-0.59 of the HBM peak (tuning.md §4.1).
// At 384-d the split queries of a whole batch are 96 KiB — they fit in LDS — and the index's split store is laid out as
// MFMA A fragments (batch_common.hpp), so here a block is SEVEN WORKER WAVES AND ONE SERVICE WAVE:
//  - a worker does what the single-query scan's waves do, independently of the others: it owns 32-row tiles (two A
//    fragments); per K-step of 32 it loads four 1 KiB fragments (hi, lo of both) with plain coalesced 16-byte buffer
//    loads, reads the eight query fragments of the step from LDS (each feeds six MFMAs) and issues 24
//    v_mfma_f32_16x16x32_bf16 (hi.hi + hi.lo + lo.hi for 4 query groups x 2 row fragments) — the arithmetic of batch2,
//    term for term.  A ring of four K-steps (16 KiB per wave, 112 KiB per CU) stays in flight across tile boundaries.
//    No barrier after the prologue, and NOTHING in its memory queue but row fragments and row norms: under a saturated
//    HBM an agent-scope load, a returning atomic or a fence takes 5-15 us to come back and vmcnt retires in order;
//  - the bound of a query is shared by the whole grid from the first tile on: 2,048 slots per query in HBM, slot
//    (tile mod 2,048) holds the best cosine seen among the rows of those tiles, so the k-th largest slot value is a lower
//    bound of the query's k-th best cosine whatever the timing (k different tiles each hold a row at least that good).
//    Every worker's first tile fills a slot (1,792 workers x 32 rows: a 57k-row sample without a pass of its own), the
//    service waves publish the k-th largest per query, the workers test from then on with
//    `dot > 0 & dot^2 >= bound^2 |q|^2 |r|^2` (no sqrt, no divide) against the block's copy of the bounds in LDS;
//  - a pair that passes is a HIT: (row, query, dot) goes into the worker's ring in LDS.  The service wave drains the
//    rings: exact cosine, row filter, (row, cosine) appended to the query's candidate list in HBM (lists have room for
//    every row: nothing can overflow, no fallback pass exists), the tile's slot raised; it re-reads and re-publishes the
//    bounds at growing intervals, and it deals the tiles: the first one of every worker is static (the sample), the rest
//    are claimed 14 at a time from one grid-wide counter and handed over through a queue in LDS (blocks do not get equal
//    shares of the HBM: at 5M rows the fastest block took 38 % more tiles than the slowest);
//  - batchq_select_kernel takes the k best of each list (a few hundred entries: registers + block_select_kth; any
//    length: chunk by chunk) and clears the control block for the next pass: two stream operations per 64 queries.
// Results do not depend on timing: a bound only ever removes rows that cannot be among the k best.
#include <vector>

#include "batch_common.hpp"
#include "kernels.hpp"
#include "select.hpp"
#include "topk.hpp"

namespace cx {

constexpr uint32_t BQ_HB = 256;        // hit-ring entries per worker wave (a power of two)
constexpr uint32_t BQ_WORK = 7;        // worker waves per block; the eighth wave is the service wave

__device__ inline uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// A published bound is the bits of a cosine > 0; 0 = nothing published yet, BQ_NONE = "this query has no bound" (fewer than
// k of its slots filled during the warm-up: a zero query, a filter that passes next to nothing) — an atomic max with a
// real cosine replaces it the moment enough slots do fill.
constexpr uint32_t BQ_NONE = 1u;
// the bound in the test's terms, t^2 (1 - 1e-4) |q|^2; -1 = no bound (every pair passes); +inf for a query slot beyond nq
__device__ inline float bq_tq(uint32_t bits, float qq, bool live) {
    const float t = __uint_as_float(bits);
    return !live ? __builtin_inff() : (bits <= BQ_NONE ? -1.0f : t * t * (1.0f - 1.0e-4f) * qq);
}

// cross-wave words in LDS: relaxed / acquire loads and release stores at workgroup scope (a plain access in a polling loop
// would be hoisted out of it)
__device__ inline uint32_t lds_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline uint32_t lds_ld_acq(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void lds_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void lds_st_rel(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// LDS control words (s_ctl)
enum : uint32_t { BQL_HEAD = 0, BQL_TAIL = 8, BQL_ARRIVED = 16, BQL_DONE = 17, BQL_READY = 18, BQL_QHEAD = 19, BQL_QTAIL = 20, BQL_WORDS = 24 };
constexpr uint32_t BQ_TQ = 64;         // entries of the block's tile queue (a power of two)
constexpr uint32_t BQ_CLAIM = 14;      // tiles the service wave claims at a time: two per worker
constexpr uint32_t BQ_NO_TILE = 0xFFFFFFFFu;

template <int D, int P>
__global__ __launch_bounds__(512, 2) void batchq_kernel(const BatchQArgs a) {
    constexpr int KS = D / 32;       // K-steps per row
    static_assert(KS % P == 0 && KS % 2 == 0, "dim / 32 must be a multiple of the ring depth");
    constexpr uint32_t T16 = 16u * D * 4u;   // bytes of a 16-row tile of the split store
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [query fragments: KS x 4 groups x (hi | lo) x 1 KiB][|q|^2: 64][control words][hit rings: rows | queries | dots, 7 x HB]
    //      [warm-up maxima 7 x 64][histogram 256][bounds 64]
    char *qimg = smem;
    float *s_qq = reinterpret_cast<float *>(smem + KS * 8192);
    uint32_t *s_ctl = reinterpret_cast<uint32_t *>(s_qq + 64);
    uint32_t *s_hrow = s_ctl + BQL_WORDS;
    uint32_t *s_hq = s_hrow + BQ_WORK * BQ_HB;
    float *s_hdot = reinterpret_cast<float *>(s_hq + BQ_WORK * BQ_HB);
    float *s_qqp = s_hdot;   // prologue only: the two halves of every |q|^2
    uint32_t *s_wm = reinterpret_cast<uint32_t *>(s_hdot + BQ_WORK * BQ_HB);   // [7 workers][64 queries] warm-up maxima
    uint32_t *s_hist = s_wm + BQ_WORK * 64u;                                   // [256] the service wave's digit histogram
    uint32_t *s_bnd = s_hist + 256u;                                           // [64] the block's copy of the published bounds
    uint32_t *s_tq = s_bnd + 64u;                                              // [BQ_TQ] the block's tile queue

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t j = lane & 15u, kq = lane >> 4;
    const uint32_t n_rows = a.n_rows, k = a.k, nq = a.nq;
    const uint32_t n32 = (n_rows + 31u) >> 5;
    const uint32_t nw = gridDim.x * BQ_WORK;                      // worker waves of the grid
    const uint32_t T_first = blockIdx.x * BQ_WORK;                // the block's workers start at tiles T_first .. T_first + 6
    const uint32_t in_block = T_first >= n32 ? 0u : (n32 - T_first < BQ_WORK ? n32 - T_first : BQ_WORK);   // workers with a first tile
    uint32_t *const g_slots = a.ctl, *const g_bound = a.ctl + BQ_CTL_BOUND, *const g_cnt = a.ctl + BQ_CTL_CNT, *const g_next = a.ctl + BQ_CTL_NEXT;
    const bool worker = wave < BQ_WORK;
    // Tiles: the first nw (one per worker wave: the warm-up sample, tile -> slot) are dealt statically; the rest are claimed
    // BQ_CLAIM at a time from one grid-wide counter by the service waves and handed to the block's workers through a queue
    // in LDS — blocks do not get the same share of the HBM (at 5M rows the first block to finish was 25 % ahead of the
    // last), and a worker never has an atomic of its own in flight
    const uint32_t n_static = nw < n32 ? nw : n32;
    uint32_t claim0 = 0u;
    if (!worker && lane == 0u) claim0 = atomicAdd(g_next, BQ_CLAIM);   // (comes back under the query split)

    auto now = [&]() -> uint64_t {   // 100 MHz
        uint64_t t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };

    // ---- workers: the ring.  P K-steps x (2 row fragments x hi, lo) of 16 bytes per lane.  A 32-row tile is 2 x T16
    // contiguous bytes of the split store (the store ends with a spare 16-row tile, so the second half of the last tile
    // exists); one buffer descriptor per tile — SGPR base, the lane's 16 bytes as the only address VGPR
    s16x8 ring[P][2][2];
    const uint32_t voff = lane * 16u;
    auto tile_rsrc = [&](uint32_t T) {
        const uint32_t Tc = (uint32_t)__builtin_amdgcn_readfirstlane((int)(T < n32 ? T : n32 - 1u));
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.split) + (size_t)Tc * (2u * T16), 0, (int)(2u * T16), 0x00020000);
    };
    auto issue = [&](s16x8 (&slot)[2][2], __amdgpu_buffer_rsrc_t rs, int ks) {
#pragma unroll
        for (int f = 0; f < 2; f++)
#pragma unroll
            for (int h = 0; h < 2; h++)
                slot[f][h] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff + (ks & 1) * 2048 + h * 1024, (int)(f * T16) + (ks >> 1) * 4096, 2 /* nt */));
    };
    auto norms_of = [&](uint32_t T, uint32_t f) {   // |row|^2 of rows 32 T + 16 f + 4 kq .. + 3 (the norm array is padded)
        const uint32_t r0 = (T < n32 ? T : n32 - 1u) * 32u + 16u * f + 4u * kq;
        return *reinterpret_cast<const f32x4 *>(a.norms + r0);
    };

    uint32_t T = T_first + wave;
    const bool has_work = worker && T < n32;
    __amdgpu_buffer_rsrc_t crs = tile_rsrc(T);
    f32x4 rr_cur[2] = {norms_of(T, 0), norms_of(T, 1)}, rr_nxt[2] = {rr_cur[0], rr_cur[1]};
    if (has_work) {
#pragma unroll
        for (int p = 0; p < P; p++) issue(ring[p], crs, p);
    }

    // ---- prologue (all eight waves): the queries, split into B fragments, into LDS: wave w takes query group w & 3,
    // K-steps of half w >> 2
    {
        const uint32_t g = wave & 3u, half = wave >> 2, q = g * 16u + j;
        const bool live = q < nq;
        const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.queries + (size_t)(live ? q : 0u) * D) + half * (KS / 2) * 8;
        float qq = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ks++) {
            f32x4 v0 = q4[8 * ks + 2 * kq], v1 = q4[8 * ks + 2 * kq + 1];
            if (!live) { v0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; v1 = v0; }
            qq += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w + v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
            s16x8 H, L;
            split8(v0, v1, H, L);
            char *dst = qimg + (((uint32_t)ks + half * (KS / 2)) * 4u + g) * 2048u + lane * 16u;
            *reinterpret_cast<s16x8 *>(dst) = H;
            *reinterpret_cast<s16x8 *>(dst + 1024) = L;
        }
        qq += __shfl_xor(qq, 16, 64);
        qq += __shfl_xor(qq, 32, 64);
        if (kq == 0u) s_qqp[half * 64u + q] = qq;   // |q|^2 = first half + second half, in that order
    }
    if (tid < BQL_WORDS) s_ctl[tid] = 0u;
    if (tid < 64u) s_bnd[tid] = 0u;
    __syncthreads();
    if (tid < 64u) s_qq[tid] = s_qqp[tid] + s_qqp[64u + tid];
    // the service wave fills the tile queue: tiles n_static + c .. + BQ_CLAIM - 1 of a claim c; BQ_WORK end marks once
    // the counter has passed the last tile
    bool exhausted = false;
    uint32_t q_head = 0u;
    auto push_claim = [&](uint32_t c) {   // service wave, all lanes
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        const uint32_t first = n_static + c;
        const uint32_t have = first >= n32 ? 0u : (n32 - first < BQ_CLAIM ? n32 - first : BQ_CLAIM);
        if (lane < have) lds_st(&s_tq[(q_head + lane) & (BQ_TQ - 1u)], first + lane);
        q_head += have;
        if (have < BQ_CLAIM) {
            if (lane < BQ_WORK) lds_st(&s_tq[(q_head + lane) & (BQ_TQ - 1u)], BQ_NO_TILE);
            q_head += BQ_WORK;
            exhausted = true;
        }
        if (lane == 0u) lds_st_rel(&s_ctl[BQL_QHEAD], q_head);
    };
    if (!worker) push_claim(claim0);
    __syncthreads();

    if (!worker) {
        // =============================================================== the service wave
        // Everything that talks to the rest of the grid lives here — the slots, the published bounds, the candidate lists —
        // so that no worker ever has an agent-scope load, a returning atomic or a fence in its memory queue: under a
        // saturated HBM such an operation takes 5-15 us to come back and vmcnt retires in order (one bound refresh per
        // wave and 8 tiles cost the row stream 8 %; the waves' own warm-up handshakes 50 us).
        // One hit per lane: exact cosine, row filter, candidate list, the tile's slot.  Lanes with the same query take
        // their list positions from ONE atomic add.
        auto process_hits = [&](bool active, uint32_t row, uint32_t q, float dot) {
            active = active && row_passes(a.flt, row);
            row = active ? row : 0u;
            q = active ? q : 0u;
            const float rr = a.norms[row];
            uint64_t same = __ballot(active);
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const uint64_t m = __ballot((q >> b) & 1u);
                same &= ((q >> b) & 1u) ? m : ~m;
            }
            const int leader = __ffsll((unsigned long long)same) - 1;
            uint32_t base = 0;
            if (active && (int)lane == leader) base = atomicAdd(g_cnt + q, (uint32_t)__popcll(same));   // (in flight together with the norm)
            const float cosv = cosine_from_sums(dot, s_qq[q], rr);
            base = (uint32_t)__shfl((int)base, active ? leader : 0, 64);
            const uint32_t pos = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            if (active && pos < a.cap) {
                a.cand_rows[(size_t)q * a.cap + pos] = row;
                a.cand_cos[(size_t)q * a.cap + pos] = cosv;
            }
            if (active && cosv > 0.0f)
                __hip_atomic_fetch_max(g_slots + q * BQ_SL + ((row >> 5) & (BQ_SL - 1u)), __float_as_uint(cosv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // The slots of query q (32 per lane); the k-th largest of them: k different tiles each hold a row with at least
        // this cosine, so it bounds the query's k-th best from below.  A radix walk over four 8-bit digits with a 256-bin
        // histogram in LDS (one wave: its LDS operations stay in order) — a bit-by-bit walk with 32 ballots per bit took 13 us.
        constexpr int NV = (int)(BQ_SL / 64u);
        uint32_t v[NV];
        auto slots_load = [&](uint32_t q) -> uint32_t {   // returns the number of filled slots
            uint32_t nz = 0;
#pragma unroll
            for (int i = 0; i < NV; i++) v[i] = ld_agent(g_slots + q * BQ_SL + lane + 64u * i);
#pragma unroll
            for (int i = 0; i < NV; i++) nz += (uint32_t)__popcll(__ballot(v[i] != 0u));
            return nz;
        };
        auto slots_kth = [&]() -> uint32_t {   // 0 = fewer than k slots are filled
            uint32_t prefix = 0u, mask = 0u, need = k;
#pragma unroll 1
            for (int shift = 24; shift >= 0; shift -= 8) {
                *reinterpret_cast<u32x4 *>(s_hist + 4u * lane) = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                for (int i = 0; i < NV; i++)
                    if (v[i] != 0u && (v[i] & mask) == prefix) atomicAdd(&s_hist[(v[i] >> shift) & 255u], 1u);
                const u32x4 h = *reinterpret_cast<const u32x4 *>(s_hist + 4u * lane);   // bins 4 lane .. 4 lane + 3
                const uint32_t mine_tot = h.x + h.y + h.z + h.w;
                uint32_t suf = mine_tot;   // inclusive suffix sum over the lanes above
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_down((int)suf, off, 64);
                    if (lane + (uint32_t)off < 64u) suf += t;
                }
                uint32_t cum = suf - mine_tot;   // entries in bins above this lane's
                uint32_t bin = 0u, nneed = 0u;
                bool found = false;
                const uint32_t hh[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
                for (int b = 3; b >= 0; b--) {
                    const bool here = !found && cum < need && need <= cum + hh[b];
                    bin = here ? 4u * lane + (uint32_t)b : bin;
                    nneed = here ? need - cum : nneed;
                    found = found || here;
                    cum += hh[b];
                }
                const uint64_t fm = __ballot(found);
                if (!fm) return 0u;
                const int src = __ffsll((unsigned long long)fm) - 1;
                bin = (uint32_t)__builtin_amdgcn_readlane((int)bin, src);
                need = (uint32_t)__builtin_amdgcn_readlane((int)nneed, src);
                prefix |= bin << shift;
                mask |= 0xFFu << shift;
            }
            return prefix;
        };
        // publisher duty: this block looks after the queries b, b + grid, ... (b = blockIdx mod 64: several blocks per
        // query when the grid is larger); `min_filled`: publish only once that many slots are in; final: a query whose
        // slots have not filled by now is declared to have no bound
        auto publish = [&](uint32_t min_filled, bool final) -> bool {
            bool all_done = true;
            for (uint32_t q = blockIdx.x & 63u; q < nq; q += gridDim.x) {
                const uint32_t nz = slots_load(q);
                if (nz < min_filled && !final) { all_done = false; continue; }
                const uint32_t t = slots_kth();
                if (lane == 0u && (t || final)) __hip_atomic_fetch_max(g_bound + q, t ? t : BQ_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!t && !final) all_done = false;
            }
            return all_done;
        };

        // A-C. The warm-up, one polling loop (every step of it is a round trip to the far side of the fabric, so none waits
        // for another): (A) once the block's workers have left their first tiles' maxima in LDS, write them to the tiles'
        // slots — plain write-through stores: at <= 2,048 worker waves every first tile has a slot to itself, and where two
        // share one either value is valid; (B) publish this block's queries as soon as a fraction of the grid's sample is in
        // the slots (the slowest first tile lands ~25 us after the fastest, and a bound from an eighth of the sample
        // already rejects all but a few pairs per tile; the several publishers of a query wait for different fractions, the
        // service loop re-publishes); (C) leave when every live query has a bound or a no-bound mark — after ~150 us the
        // stragglers are given theirs.
        uint32_t bl = 1u;
        {
            const uint32_t sample = (nw < n32 ? nw : n32) < BQ_SL ? (nw < n32 ? nw : n32) : BQ_SL;   // slots the first tiles fill
            const uint32_t frac = ((blockIdx.x >> 6) & 3u) + 1u;                                      // 1/8, 2/8, 3/8, 4/8
            const uint32_t want = sample * frac / 8u > k ? sample * frac / 8u : k;
            bool stored = false, published = false;
            for (int spin = 0; spin < 4096; spin++) {   // bounded: a few ms
                if (!stored && lds_ld_acq(&s_ctl[BQL_ARRIVED]) >= in_block) {
                    stored = true;
                    if (lane < nq)
                        for (uint32_t w = 0; w < in_block; w++)
                            __hip_atomic_store(g_slots + lane * BQ_SL + ((T_first + w) & (BQ_SL - 1u)), s_wm[w * 64u + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!published) published = publish(want, spin >= 96);
                bl = lane < nq ? ld_agent(g_bound + lane) : 1u;
                if (stored && __ballot(bl == 0u) == 0ull) break;
                if (published) __builtin_amdgcn_s_sleep(8);
            }
        }
        lds_st(&s_bnd[lane], bl);
        if (lane == 0u) lds_st_rel(&s_ctl[BQL_READY], 1u);
        // D. service loop: drain the workers' hit rings, keep the block's copy of the bounds fresh, re-publish
        uint32_t gap = 200u;           // x10 ns: the next look at the published bounds
        uint64_t t_next = now() + gap;
        for (;;) {
            // up to 64 pending entries, taken from the rings in worker order, in ONE round (a round is a norm load, a
            // returning atomic and the stores: ~5-8 us under load; a round per ring put 40 us behind the last tile)
            uint32_t hd = 0u, tl = 0u;
            if (lane < BQ_WORK) { hd = lds_ld_acq(&s_ctl[BQL_HEAD + lane]); tl = lds_ld(&s_ctl[BQL_TAIL + lane]); }   // (the tails are this wave's own)
            const uint32_t pend = hd - tl;
            uint32_t incl = pend;        // inclusive prefix sum over the first lanes
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
                if (lane >= (uint32_t)off) incl += t;
            }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)BQ_WORK - 1);
            const bool busy = total != 0u;
            if (busy) {
                uint32_t w_of = 0u, first_of = 0u, tail_of = 0u;   // this lane's entry: ring, the ring's first position in the batch, its tail
#pragma unroll
                for (uint32_t w = 0; w < BQ_WORK; w++) {
                    const uint32_t end_w = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)w);
                    const uint32_t beg_w = end_w - (uint32_t)__builtin_amdgcn_readlane((int)pend, (int)w);
                    const uint32_t tl_w = (uint32_t)__builtin_amdgcn_readlane((int)tl, (int)w);
                    if (lane >= beg_w && lane < end_w) { w_of = w; first_of = beg_w; tail_of = tl_w; }
                }
                const bool on = lane < total;
                const uint32_t e = w_of * BQ_HB + ((tail_of + lane - first_of) & (BQ_HB - 1u));
                if (!(a.arm & 4u)) process_hits(on, on ? s_hrow[e] : 0u, on ? s_hq[e] : 0u, on ? s_hdot[e] : 0.0f);
                // the tails move by what was taken: everything of a ring whose entries all fell inside the first 64
                if (lane < BQ_WORK) {
                    const uint32_t beg = incl - pend;
                    const uint32_t took = beg >= 64u ? 0u : (incl <= 64u ? pend : 64u - beg);
                    if (took) lds_st_rel(&s_ctl[BQL_TAIL + lane], tl + took);
                }
            }
            // The published bounds are re-read, and this block's queries re-published, ever more rarely (2, 4, 8, ... us
            // apart, then every 128 us): 256 service waves polling the same 256 bytes every few microseconds keep one HBM
            // channel busy with themselves — agent-scope loads go past the L2 — and the row stream, which needs every
            // channel, slowed by 8-10 %
            if (!exhausted && (int32_t)(q_head - lds_ld(&s_ctl[BQL_QTAIL])) < (int32_t)BQ_CLAIM) {   // fewer than two tiles per worker queued
                uint32_t c = 0u;
                if (lane == 0u) c = atomicAdd(g_next, BQ_CLAIM);
                push_claim(c);
            }
            const bool workers_done = lds_ld_acq(&s_ctl[BQL_DONE]) >= BQ_WORK;
            const uint64_t t_now = now();
            if (t_now >= t_next && !workers_done) {
                bl = ld_agent(g_bound + lane);
                if (bl > lds_ld(&s_bnd[lane])) lds_st(&s_bnd[lane], bl);
                if (gap >= 800u) publish(0u, true);
                gap = gap < 12800u ? gap * 2u : 12800u;   // x10 ns
                t_next = now() + gap;
            }
            if (!busy) {
                if (workers_done) {   // every worker is through; one more look at the rings, then out
                    bool left = false;
#pragma unroll 1
                    for (uint32_t w = 0; w < BQ_WORK; w++) left = left || lds_ld_acq(&s_ctl[BQL_HEAD + w]) != lds_ld(&s_ctl[BQL_TAIL + w]);
                    if (!left) break;
                } else {
                    __builtin_amdgcn_s_sleep(64);   // ~2 us
                }
            }
        }
        return;
    }

    // =================================================================== worker waves
    if (!has_work) {
        if (lane == 0u) __hip_atomic_fetch_add(&s_ctl[BQL_DONE], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    float qq4[4];
#pragma unroll
    for (int g = 0; g < 4; g++) qq4[g] = s_qq[16 * g + j];
    const char *const my_q = qimg + lane * 16u;
    uint32_t *const hb_row = s_hrow + wave * BQ_HB, *const hb_q = s_hq + wave * BQ_HB;
    float *const hb_dot = s_hdot + wave * BQ_HB;
    uint32_t head = 0;               // entries this wave has put into its hit ring (wave-uniform; s_ctl[BQL_HEAD + wave] mirrors it)
    bool first = true;
    bool liveq[4];
#pragma unroll
    for (int g = 0; g < 4; g++) liveq[g] = 16u * g + j < nq;
    float tq[4] = {-1.0f, -1.0f, -1.0f, -1.0f};   // the bound of each of the lane's four queries in the test's terms (bq_tq)
    // room for n more entries in the ring (the service wave moves the tail); bounded wait
    auto wait_room = [&](uint32_t n) {
        for (int spin = 0; spin < (1 << 20); spin++) {
            if (head + n - lds_ld_acq(&s_ctl[BQL_TAIL + wave]) <= BQ_HB) break;
            __builtin_amdgcn_s_sleep(8);
        }
    };
    // query fragments of the current K-step: [group][hi, lo].  One register set: a pair of groups is re-read for the NEXT
    // K-step right behind its own 12 MFMAs, under the other pair's
    s16x8 B[4][2];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        B[g][0] = *reinterpret_cast<const s16x8 *>(my_q + g * 2048);
        B[g][1] = *reinterpret_cast<const s16x8 *>(my_q + g * 2048 + 1024);
    }

    // the next tile of this wave: an entry of the block's queue (an LDS atomic for the position, then the entry once the
    // service wave has written it — it keeps two tiles per worker ahead)
    auto claim = [&]() -> uint32_t {
        uint32_t idx = 0u;
        if (lane == 0u) idx = __hip_atomic_fetch_add(&s_ctl[BQL_QTAIL], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        while ((int32_t)(lds_ld_acq(&s_ctl[BQL_QHEAD]) - idx) <= 0) __builtin_amdgcn_s_sleep(2);   // the service wave always refills: its claims depend on nobody
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_ld(&s_tq[idx & (BQ_TQ - 1u)]));
    };
    while (T != BQ_NO_TILE) {
        const uint32_t Tn = claim();
        const __amdgpu_buffer_rsrc_t nrs = tile_rsrc(Tn != BQ_NO_TILE ? Tn : T);
        f32x4 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; g++) { acc[g][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[g][1] = acc[g][0]; }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int p = ks % P;
            const int nks = ks + 1 < KS ? ks + 1 : 0;   // the last step reads the next tile's first query fragments
            const s16x8 h0 = ring[p][0][0], l0 = ring[p][0][1], h1 = ring[p][1][0], l1 = ring[p][1][1];
            // x.y = xl.yh + xh.yl + xh.yh, small terms first (batch2's order); four accumulators in rotation
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
#pragma unroll
                for (int g = gp; g < gp + 2; g++) {
                    acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, B[g][0], acc[g][0], 0, 0, 0);
                    acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, B[g][0], acc[g][1], 0, 0, 0);
                }
#pragma unroll
                for (int g = gp; g < gp + 2; g++) {
                    acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, B[g][1], acc[g][0], 0, 0, 0);
                    acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, B[g][1], acc[g][1], 0, 0, 0);
                }
#pragma unroll
                for (int g = gp; g < gp + 2; g++) {
                    acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, B[g][0], acc[g][0], 0, 0, 0);
                    acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, B[g][0], acc[g][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = gp; g < gp + 2; g++) {
                    B[g][0] = *reinterpret_cast<const s16x8 *>(my_q + (nks * 4 + g) * 2048);
                    B[g][1] = *reinterpret_cast<const s16x8 *>(my_q + (nks * 4 + g) * 2048 + 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the slot is free: K-step ks + P of this tile, or of the next one
            if (ks + P < KS) issue(ring[p], crs, ks + P);
            else issue(ring[p], nrs, ks + P - KS);
            if (ks + P == KS) {   // with the next tile's first K-step: its row norms
                rr_nxt[0] = norms_of(Tn != BQ_NO_TILE ? Tn : T, 0);
                rr_nxt[1] = norms_of(Tn != BQ_NO_TILE ? Tn : T, 1);
            }
        }

        const uint32_t row0 = T * 32u + 4u * kq;   // this lane's rows: row0 + 16 f + r
        if (first) {
            // ---- warm-up, once per wave: the tile's best cosine per query (best row by dot / |r|, its exact cosine)
            // into LDS for the service wave, which fills the grid's slots with them and brings the first bounds back
            first = false;
            uint32_t okm = 0;   // bit 4 f + r: row row0 + 16 f + r exists and passes the filter
            if (a.flt.trivial) {
#pragma unroll
                for (uint32_t i = 0; i < 8u; i++) okm |= (row0 + 16u * (i >> 2) + (i & 3u) < n_rows) ? (1u << i) : 0u;
            } else {
                // (the one place a worker reads row metadata: once, before its stream has anything to wait for)
#pragma unroll 1
                for (uint32_t i = 0; i < 8u; i++) {
                    const uint32_t row = row0 + 16u * (i >> 2) + (i & 3u);
                    okm |= (row < n_rows && row_passes(a.flt, row)) ? (1u << i) : 0u;
                }
            }
            uint32_t mine = 0u;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float best = 0.0f, bdot = 0.0f, brr = 1.0f;
#pragma unroll
                for (int f = 0; f < 2; f++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float key = acc[g][f][r] * __builtin_amdgcn_rsqf(rr_cur[f][r]);
                        const bool up = ((okm >> (4 * f + r)) & 1u) && key > best;
                        best = up ? key : best; bdot = up ? acc[g][f][r] : bdot; brr = up ? rr_cur[f][r] : brr;
                    }
                float mx = best > 0.0f ? cosine_from_sums(bdot, qq4[g], brr) : 0.0f;
                mx = mx > 0.0f ? mx : 0.0f;   // NaN -> 0
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                mine = (kq == (uint32_t)g && 16u * g + j < nq) ? __float_as_uint(mx) : mine;   // lane 16 g + j: query 16 g + j
            }
            s_wm[wave * 64u + lane] = mine;
            if (lane == 0u) __hip_atomic_fetch_add(&s_ctl[BQL_ARRIVED], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (int spin = 0; spin < (1 << 16); spin++) {        // bounded: ~30 ms
                if (lds_ld_acq(&s_ctl[BQL_READY]) != 0u) break;
                __builtin_amdgcn_s_sleep(4);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; g++) tq[g] = bq_tq(lds_ld(&s_bnd[16 * g + j]), qq4[g], liveq[g]);

        // ---- the test: nothing but registers; one wave-level branch
        bool any = false;
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int f = 0; f < 2; f++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float dot = acc[g][f][r], rr = rr_cur[f][r];
                    // bitwise, not short-circuit (hipcc turns || and && on float compares into divergent branches)
                    any |= (tq[g] < 0.0f) | ((dot > 0.0f) & (dot * dot >= tq[g] * rr)) | (dot != dot) | (rr != rr);
                }
        if (a.arm & 1u) any = false;
        if (__ballot(any)) {
            uint32_t hm = 0;             // hit mask (bit (g * 2 + f) * 4 + r)
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int f = 0; f < 2; f++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float dot = acc[g][f][r], rr = rr_cur[f][r];
                        const bool hit = ((tq[g] < 0.0f) | ((dot > 0.0f) & (dot * dot >= tq[g] * rr)) | (dot != dot) | (rr != rr)) &
                                         (row0 + 16u * f + r < n_rows) & (16u * g + j < nq);
                        hm |= hit ? (1u << ((g * 2 + f) * 4 + r)) : 0u;
                    }
            const uint32_t mine = (uint32_t)__popc(hm);
            uint32_t incl = mine;        // inclusive prefix sum over the lanes
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
                if (lane >= (uint32_t)off) incl += t;
            }
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (tot <= BQ_HB / 2u) {
                wait_room(tot);
                uint32_t pos = head + incl - mine;
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int f = 0; f < 2; f++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if ((hm >> ((g * 2 + f) * 4 + r)) & 1u) {
                                const uint32_t e = pos & (BQ_HB - 1u);
                                hb_row[e] = row0 + 16u * f + r;
                                hb_q[e] = 16u * g + j;
                                hb_dot[e] = acc[g][f][r];
                                pos++;
                            }
                head += tot;
                if (lane == 0u) lds_st_rel(&s_ctl[BQL_HEAD + wave], head);
            } else {
                // a tile of a query without a bound: one hit per lane and round
#pragma unroll 1
                while (__ballot(hm != 0u)) {
                    const bool on = hm != 0u;
                    const uint32_t idx = on ? (uint32_t)__ffs((int)hm) - 1u : 0u;
                    hm &= hm - 1u;
                    float dot = 0.0f;
#pragma unroll
                    for (int g = 0; g < 4; g++)
#pragma unroll
                        for (int f = 0; f < 2; f++)
#pragma unroll
                            for (int r = 0; r < 4; r++) dot = idx == (uint32_t)((g * 2 + f) * 4 + r) ? acc[g][f][r] : dot;
                    const uint64_t om = __ballot(on);
                    const uint32_t n = (uint32_t)__popcll(om);
                    wait_room(n);
                    if (on) {
                        const uint32_t e = (head + (uint32_t)__popcll(om & ((1ull << lane) - 1ull))) & (BQ_HB - 1u);
                        hb_row[e] = row0 + 16u * ((idx >> 2) & 1u) + (idx & 3u);
                        hb_q[e] = 16u * (idx >> 3) + j;
                        hb_dot[e] = dot;
                    }
                    head += n;
                    if (lane == 0u) lds_st_rel(&s_ctl[BQL_HEAD + wave], head);
                }
            }
        }
        // ---- advance
        T = Tn;
        crs = nrs;
        rr_cur[0] = rr_nxt[0]; rr_cur[1] = rr_nxt[1];
    }
    if (lane == 0u) __hip_atomic_fetch_add(&s_ctl[BQL_DONE], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---------------------------------------------------------------------------------------------------
// The k best of each query's candidate list -> results, one block per query; clears the query's part of the control
// block for the next pass.  A list of <= NV x 1024 entries (the expected few hundred to few thousand) is held in
// registers and selected once; a longer one (weak bounds: massive ties, a selective row filter) is folded chunk by
// chunk, the survivors of the chunks so far riding along — exact whatever the length.
template <int NV>
__global__ __launch_bounds__(1024) void batchq_select_kernel(const BatchQArgs a, uint32_t *out_rows, float *out_scores, float *out_dists,
                                                             uint32_t *out_count) {
    __shared__ uint32_t sh[264];
    __shared__ uint64_t surv_k[256 + 64];
    __shared__ float surv_s[256 + 64];
    __shared__ uint32_t s_n;
    const uint32_t tid = threadIdx.x, q = blockIdx.x, k = a.k;
    uint32_t *const g_slots = a.ctl, *const g_bound = a.ctl + BQ_CTL_BOUND, *const g_cnt = a.ctl + BQ_CTL_CNT;
    if (q == 0u && tid == 0u) a.ctl[BQ_CTL_NEXT] = 0u;
    uint32_t total = g_cnt[q];
    total = total < a.cap ? total : a.cap;
    const uint32_t *rows = a.cand_rows + (size_t)q * a.cap;
    const float *cosv = a.cand_cos + (size_t)q * a.cap;
    constexpr uint32_t CHUNK = (uint32_t)NV * 1024u - 256u;   // room for the survivors so far
    uint32_t n_surv = 0;                                       // entries of surv_* that are live
    for (uint32_t c0 = 0; c0 == 0u || c0 < total; c0 += CHUNK) {
        const uint32_t cn = total - c0 < CHUNK ? total - c0 : CHUNK;
        uint64_t key[NV];
        float sim[NV];
#pragma unroll
        for (int u = 0; u < NV; u++) {
            const uint32_t ci = tid + (uint32_t)u * 1024u;
            key[u] = 0ull;
            sim[u] = 0.0f;
            if (ci < cn) {
                sim[u] = cosv[c0 + ci];
                key[u] = make_key(score_of(distance_of(sim[u])), rows[c0 + ci]);
            } else if (ci - cn < n_surv) {
                key[u] = surv_k[ci - cn];
                sim[u] = surv_s[ci - cn];
            }
        }
        __syncthreads();
        if (tid == 0) s_n = 0u;
        const uint64_t t = block_select_kth<NV>(key, k, 56, 0, sh);   // 0: fewer than k entries — all of them survive
        const uint64_t low = t ? t : 1ull;
#pragma unroll
        for (int u = 0; u < NV; u++)
            if (key[u] >= low && key[u] != 0ull) {   // keys are unique (the row is part of the key): exactly min(k, live) survivors
                const uint32_t pos = atomicAdd(&s_n, 1u);
                if (pos < 256u + 64u) { surv_k[pos] = key[u]; surv_s[pos] = sim[u]; }
            }
        __syncthreads();
        n_surv = s_n < 256u + 64u ? s_n : 256u + 64u;
    }
    const uint32_t S = n_surv;
    uint32_t *o_rows = out_rows + (size_t)q * k;
    float *o_scores = out_scores + (size_t)q * k, *o_dists = out_dists + (size_t)q * k;
    for (uint32_t i = tid; i < S; i += 1024u) {
        const uint64_t ki = surv_k[i];
        uint32_t rank = 0;
        for (uint32_t jj = 0; jj < S; jj++) rank += surv_k[jj] > ki ? 1u : 0u;
        if (rank < k) {
            const float dist = distance_of(surv_s[i]);
            o_rows[rank] = key_row(ki);
            o_dists[rank] = dist;
            o_scores[rank] = score_of(dist);
        }
    }
    if (tid == 0) { out_count[q] = S < k ? S : k; g_cnt[q] = 0u; g_bound[q] = 0u; }
    for (uint32_t s = tid; s < BQ_SL; s += 1024u) g_slots[q * BQ_SL + s] = 0u;
}

// ---------------------------------------------------------------------------------------------------
bool batchq_supported(uint32_t dim, uint32_t k) { return dim == 384u && k >= 1u && k <= 256u; }

uint32_t batchq_min_rows() {
    static const uint32_t v = getenv("CX_BATCHQ_MIN_ROWS") ? (uint32_t)atoi(getenv("CX_BATCHQ_MIN_ROWS")) : 131072u;
    return v;
}

static size_t batchq_lds_bytes(uint32_t dim) { return (size_t)(dim / 32u) * 8192u + 64 * 4 + BQL_WORDS * 4 + (size_t)BQ_WORK * BQ_HB * 12 + BQ_WORK * 64 * 4 + 256 * 4 + 64 * 4 + BQ_TQ * 4; }

int launch_batchq_pass(const BatchQArgs &a_in, hipStream_t stream) {
    BatchQArgs a = a_in;
    static const uint32_t arm_env = getenv("CX_BATCHQ_ARM") ? (uint32_t)atoi(getenv("CX_BATCHQ_ARM")) : 0u;
    a.arm = arm_env;
    if (!batchq_supported(a.dim, a.k) || a.nq == 0 || a.nq > 64u || a.n_rows == 0)
        return set_err(CX_ERR_VALIDATION, "batchq: unsupported shape (dim %u, k %u, %u queries, %u rows)", a.dim, a.k, a.nq, a.n_rows);
    const uint32_t cus = device_cus(), n32 = (a.n_rows + 31u) / 32u;
    // every worker wave needs a first tile of its own (its warm-up fills a slot); blocks of 7 workers + the service wave
    uint32_t grid = n32 / BQ_WORK;
    grid = grid < 1u ? 1u : (grid > cus ? cus : grid);
    const size_t lds = batchq_lds_bytes(a.dim);
    static std::atomic<uint64_t> attr_devices{0};
    static const int ring = getenv("CX_BATCHQ_RING") ? atoi(getenv("CX_BATCHQ_RING")) : 4;
    if (first_use_on_device(attr_devices)) {
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchq_kernel<384, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchq_kernel<384, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (ring == 6) hipLaunchKernelGGL((batchq_kernel<384, 6>), dim3(grid), dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((batchq_kernel<384, 4>), dim3(grid), dim3(512), lds, stream, a);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

int launch_batchq_select(const BatchQArgs &a, uint32_t *out_rows, float *out_scores, float *out_dists, uint32_t *out_count, hipStream_t stream) {
    if (a.k <= 32u) hipLaunchKernelGGL(batchq_select_kernel<4>, dim3(a.nq), dim3(1024), 0, stream, a, out_rows, out_scores, out_dists, out_count);
    else hipLaunchKernelGGL(batchq_select_kernel<8>, dim3(a.nq), dim3(1024), 0, stream, a, out_rows, out_scores, out_dists, out_count);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

}  // namespace cx
