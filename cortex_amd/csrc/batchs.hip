// batchs.hip — batched exact search (search_batch, vector/index.rs:390-410; BASELINE config 4's inner loop) as a SCREENING
// pass over a 2-byte copy of the rows followed by an exact f32 re-score of the survivors — the all-pairs pass's own
// design (allpairs.hip: bf16 filter with a rigorous error bound, then the reference's arithmetic on what is left)
// applied to up to 64 queries per pass (128 at row widths up to 512: the kernel's NB parameter).
//
// The operand is the all-pairs filter's own: the index's tiled shadow (cx_index::d_shadow_t, kernels.hpp: rows L2-normalised,
// rounded to bf16, one KiB per 16 rows x K-step of 32, a row's four 16-byte pieces XOR-permuted with bits 3-4 of the row).
// A lane reads ITS piece of the MFMA A fragment out of that KiB — lane (i, kq): row i's piece kq — so a wave-level load is
// one KiB of contiguous memory in a permuted lane order, which streams exactly as fast as the linear order
// (scripts/probes/ring_probe.hip: 0.862 against 0.864 of the peak): ONE normalised 2-byte copy of the rows serves the
// linker passes and the batched search.  Its dot x^.y^ with a normalised bf16 query is the cosine x.y within
//   eps(y) = d_max (1 + u) + d_y (1 + 1e-6) + 1e-4,   d_max = the largest || x^ - x || over the shadow's rows, d_y = || y^ - y ||
// (|x^.y^ - x.y| <= || x^ - x || || y^ || + || x || || y^ - y ||; u = 2^-8; 1e-4: f32 accumulation and the normalisations) — the
// errors the rounding really made (the shadow's build keeps d_max, the prologue computes d_y): ~1.6e-3 each on embeddings,
// against the worst case u || x || = 3.9e-3 each that the all-pairs filter's fixed bound assumes.  So
//   - a row whose approximate cosine is A has an exact cosine >= A - eps;
//   - if k different rows have approximate cosines >= A_k, the query's exact k-th best is >= A_k - eps, and a row can
//     only be among the exact k best if its approximate cosine is >= A_k - 2 eps (the query's margin).
// The pass therefore streams HALF the bytes of the f32 rows (a quarter of rows + split store), needs ONE
// v_mfma_f32_16x16x32_bf16 per 16 rows x 16 queries x 32 elements instead of three, tests a pair with one compare,
// and hands a few hundred survivors per query to the re-score kernel, which tests each of them once more against the pass's
// LAST bound (a candidate came in under the bound of its moment; the same inequality, the same proof — what fails is struck
// without its row being read), computes the exact cosines of the rest from the stored rows (f32 or bf16 store) with
// sequential-in-lane f32 sums — the scan path's arithmetic class —, and to the select kernel.
// Results are the exact path's: the screening only ever removes rows that cannot be among the k best.
//
// Structure of the pass (every choice below was measured on the way; profiles/r03/tuning.md §8): a block is seven WORKER
// waves and one SERVICE wave.
//  - A worker does what the single-query scan's waves do, independently of the others: it owns 32-row tiles (two A
//    fragments), keeps a ring of K-steps (2 KiB each, 8 deep at 768 / 1024-d) in flight across tile boundaries with plain
//    16-byte buffer loads, reads the four query fragments of a K-step from LDS (64 queries x dim x 2 bytes: 96 KiB at
//    768-d, 128 KiB at 1024-d) and issues 8 MFMAs.  No barrier after the prologue and NOTHING else in its memory queue:
//    under a saturated HBM an agent-scope load, a returning atomic or a fence takes 5-15 us to come back and vmcnt retires
//    in order (one bound refresh per wave and 8 tiles cost the row stream 8 %; per-wave warm-up handshakes 50 us).
//  - The bound of a query is shared by the whole grid from the first tile on: 2,048 slots per query in HBM, slot
//    (tile mod 2,048) holds the best approximate cosine seen among the rows of those tiles, so the k-th largest slot value
//    is A_k whatever the timing.  Every worker's first tile fills a slot (1,792 workers x 32 rows: a 57k-row sample without
//    a pass of its own); the workers test against the block's copy of the bounds in LDS.
//  - A pair that passes is a HIT: (row, query, approximate cosine) goes into the worker's ring in LDS.
//  - The service wave owns everything that WAITS for the rest of the grid (a worker's first-tile maxima go straight to the
//    slots: plain write-through stores the worker does not wait for — an atomic max per wave and query, 131k at agent scope,
//    took 75 us; one arrival counter for 2,048 waves 110 us; a release fence per wave ~75 us): it publishes the k-th largest
//    slot value of its queries (radix walk over four 8-bit digits with an LDS histogram: 2 us; a ballot per bit and value:
//    13; the first publisher of a query polls a 256-slot prefix, everybody else the bounds: one round trip per poll), drains the
//    hit rings — row filter, candidate list (bounded since round 4: a list that runs over is redone exactly, below), the
//    tile's slot raised —, re-reads and re-publishes the bounds at growing intervals (256 service waves polling the same
//    256 bytes every few microseconds keep one HBM channel busy with themselves: -8 %), and deals the tiles: the first
//    one of every worker is static (the sample), the rest are claimed 21 at a time (7 near the end) from one grid-wide counter
//    and handed over through a queue in LDS (blocks do not get equal shares of the HBM: at 5M rows the fastest block took 38 %
//    more tiles than the slowest; statically dealt tiles left a quarter of the chip idle at the end).
//
// Round 4 (profiles/r04/tuning.md):
//  - candidate lists are BOUNDED (BatchSArgs::cap entries per query; round 3: room for every row).  A query whose list runs
//    over, an IRREGULAR query (kernels.hpp: bs_regular — zero, non-finite, |q|^2 outside [1e-30, 1e30]) and every query of a pass
//    in which a worker gave up on its ring are redone by the re-score kernel with the reference's arithmetic over every row
//    (bs_exact_redo); the store's irregular rows (zero shadow rows: never hits) ride behind every query's list;
//  - large passes do not stand still for the grid's first bounds: the k-th largest of the fourteen 16-row-group maxima of a
//    block's own first tiles is a valid bound at once (k <= 14), the hits it lets through wait in the rings until the service wave
//    can drop them under the grid's bounds — which it does without an atomic: survivors are staged in LDS and handed to
//    process_hits 64 at a time;
//  - NB = 2: two banks of 64 queries in one pass at row widths up to 512 (128 queries' fragments fit the LDS): a call of more than
//    64 queries streams the shadow once per 128;
//  - a tile FULL of hits under a row filter (queries without a bound, or with one from the handful of rows that pass) is checked
//    against the rows' metadata by the worker itself, and publishers learn from one arrival count per block that every first tile
//    is in — a query with fewer than k filled slots then has no bound to wait for (a filter passing 125 of 1.25M rows: 0.42 ms per
//    batch, 3.7 before);
//  - the re-score is a chain of round trips, not a gather (nine tenths of a k = 100 list are struck by the second look): a wave
//    looks at a chunk of the list at once, one entry per lane, and takes the survivors four rows at a time with 16-byte loads.
#include <vector>

#include "batch_common.hpp"
#include "kernels.hpp"
#include <algorithm>
#include <vector>
#include "select.hpp"
#include "topk.hpp"

namespace cx {

constexpr uint32_t BS_WORK = 7;        // worker waves per block; the eighth wave is the service wave

__device__ inline uint32_t bs_ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint32_t bs_lds_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline uint32_t bs_lds_ld_acq(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void bs_lds_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void bs_lds_st_rel(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// A published bound is the bits of an approximate cosine > 0 (the k-th largest slot value A_k); 0 = nothing published
// yet, BS_NONE = "this query has no bound" (fewer than k slots filled: a zero query, a filter that passes next to nothing).
constexpr uint32_t BS_NONE = 1u;
// the test's threshold: a pair is a hit unless its approximate cosine is below A_k - margin, margin = 2 eps of the query.  -inf = every pair passes (no
// bound, or one so low that rows with a non-positive cosine — all tied at the clamped score 0 — could be among the k best);
// +inf for a query slot beyond nq
__device__ inline float bs_thr(uint32_t bits, bool live, float margin) {
    const float t = __uint_as_float(bits) - margin;
    return !live ? __builtin_inff() : ((bits <= BS_NONE || !(t > 0.0f)) ? -__builtin_inff() : t);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// LDS control words
enum : uint32_t { BSL_HEAD = 0, BSL_TAIL = 8, BSL_ARRIVED = 16, BSL_DONE = 17, BSL_READY = 18, BSL_QHEAD = 19, BSL_QTAIL = 20, BSL_LOC = 21, BSL_WORDS = 24 };
constexpr uint32_t BS_LOCG = 2u * BS_WORK;   // 16-row groups of a block's first tiles: the values its LOCAL bounds are taken from
constexpr uint32_t BS_TQ = 128;        // entries of the block's tile queue (a power of two)
constexpr uint32_t BS_CLAIM = 21;      // tiles the service wave claims at a time: three per worker (14 / 21 / 28 measured: round 4 tuning notes)
constexpr uint32_t BS_NO_TILE = 0xFFFFFFFFu;
constexpr uint32_t BS_STRUCK = 0xFFFFFFFFu;   // a candidate the re-score struck from its list

// NB: 64-query BANKS of a pass.  One bank is the pass of round 3; two (row widths up to 512: the fragments of 128 queries are 96 KiB
// at 384-d, 128 KiB at 512-d) let a call of more than 64 queries stream the shadow once per 128 of them — index.rs:397-403 is one
// par_iter over any number of queries.  A worker then holds eight query groups' accumulators (64 registers) and issues 16 MFMAs per
// K-step on the same two row fragments; everything per query (slots, bounds, lists) is indexed by query 0..127.
template <int D, int NB = 1>
struct BsCfg {
    static constexpr int KS = D / 32;                                   // K-steps per row
    static constexpr int P = KS % 8 == 0 ? 8 : (KS % 6 == 0 ? 6 : 4);   // K-steps a worker keeps in flight (2 KiB each)
    static constexpr int NG = 4 * NB;                                   // query groups of 16
    static constexpr uint32_t QN = 64u * NB;                            // queries of a pass
    // hit-ring entries per worker (a power of two), as many as the LDS has room for: while a block runs on its LOCAL bounds (see
    // the kernel) its service wave holds the hits back, and the rings are the workers' runway until the grid's bounds arrive
    static constexpr uint32_t HB = NB == 2 ? (D > 384 ? 128u : 512u) : (D > 896 ? 128u : (D > 768 ? 256u : (D > 384 ? 512u : 1024u)));
    static constexpr uint32_t T16 = 16u * D * 2u;                       // bytes of a 16-row block of the tiled shadow
    static constexpr size_t LDS = (size_t)KS * 1024u * NG + BSL_WORDS * 4 + (size_t)BS_WORK * HB * 12 + 256 * 4 + QN * 4 + BS_TQ * 4 + QN * 4 + QN * 4 + BS_LOCG * QN * 4 + 128 * 12;
    static_assert(KS % P == 0 && D % 128 == 0 && LDS <= 160u * 1024u && (NB == 1 || (NB == 2 && D <= 512)), "unsupported row width");
};

// THR: the same pass as the all-pairs filter of a SMALL scan set (<= 64 scanned rows of the shard, or external vectors:
// streaming ingest, BASELINE config 5; allpairs_stream.hip's contract): the "queries" are the scanned rows' own shadow
// pieces, the test is a fixed threshold (thr - eps), there is no bound to establish — no sample, no slots, no wait —, and a
// hit's row goes into the scanned row's candidate list (cand_cnt / cand, `cap` slots: the exact rescore redoes a row whose
// count runs over).
template <int D, bool THR, int NB>
__global__ __launch_bounds__(512, 2) void batchs_kernel(const BatchSArgs a) {
    using C = BsCfg<D, NB>;
    static_assert(!(THR && NB != 1), "the filter pass scans at most 64 rows");
    constexpr int KS = C::KS, P = C::P, NG = C::NG;
    constexpr uint32_t HB = C::HB, T16 = C::T16, QN = C::QN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [query fragments: KS x 4 groups x 1 KiB][control words][hit rings: rows | queries | cosines, 7 x HB]
    //      [histogram 256][bounds 64][tile queue][1 / |q|: 64]
    char *qimg = smem;
    uint32_t *s_ctl = reinterpret_cast<uint32_t *>(smem + KS * 1024 * NG);
    uint32_t *s_hrow = s_ctl + BSL_WORDS;
    uint32_t *s_hq = s_hrow + BS_WORK * HB;
    float *s_hdot = reinterpret_cast<float *>(s_hq + BS_WORK * HB);
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_hdot + BS_WORK * HB); // [256] the service wave's digit histogram
    uint32_t *s_bnd = s_hist + 256u;                                        // [64] the block's copy of the published bounds
    uint32_t *s_tq = s_bnd + QN;                                            // [BS_TQ] the block's tile queue
    float *s_inv = reinterpret_cast<float *>(s_tq + BS_TQ);                 // [64] 1 / |q| (prologue); 0 = an irregular query: no image, no hits, redone exactly
    float *s_mrg = s_inv + QN;                                              // [QN] 2 eps of each query (the service wave's look at a hit)
    uint32_t *s_loc = reinterpret_cast<uint32_t *>(s_mrg + QN);             // [BS_LOCG][QN] best approximate cosine of each 16-row group of the block's first tiles
    uint32_t *s_stg_row = s_loc + BS_LOCG * QN;                            // [128] the service wave's staging area: hits that passed its look under the bounds of the moment
    uint32_t *s_stg_q = s_stg_row + 128u;
    float *s_stg_dot = reinterpret_cast<float *>(s_stg_q + 128u);
    float *s_qqp = s_hdot;                                                  // prologue only: [2][QN] the two halves of every |q|^2, [2][QN] of every || y^ - y ||^2

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t j = lane & 15u, kq = lane >> 4;
    const uint32_t n_rows = a.n_rows, k = a.k, nq = a.nq;
    const uint32_t n32 = (n_rows + 31u) >> 5;
    const uint32_t nw = gridDim.x * BS_WORK;                      // worker waves of the grid
    const uint32_t T_first = blockIdx.x * BS_WORK;                // the block's workers start at tiles T_first .. T_first + 6
    const uint32_t in_block = T_first >= n32 ? 0u : (n32 - T_first < BS_WORK ? n32 - T_first : BS_WORK);   // workers with a first tile
    uint32_t *const g_slots = a.ctl, *const g_bound = a.ctl + BS_CTL_BOUND, *const g_cnt = THR ? a.thr_cand_cnt : a.ctl + BS_CTL_CNT;
    uint32_t *const g_next = THR ? a.thr_next : a.ctl + BS_CTL_NEXT;
    const bool worker = wave < BS_WORK;
    // Tiles: the first nw (one per worker wave: the sample, tile -> slot) are dealt statically; the rest are claimed
    // BS_CLAIM at a time from one grid-wide counter by the service waves and handed over through a queue in LDS
    const uint32_t n_static = nw < n32 ? nw : n32;
    // (a block's first claim is dealt, not taken: 256 blocks' atomics on one word at launch are served one after the other,
    // ~50 ns each, and hipcc waits for the returning atomic where it is issued — the whole block's prologue stood behind it)
    const uint32_t CLAIM = a.claim;   // tiles the service wave claims at a time (BS_CLAIM; CX_BATCHS_CLAIM)
    const uint32_t claim0 = blockIdx.x * CLAIM, claim_base = gridDim.x * CLAIM;

    auto now = [&]() -> uint64_t {   // 100 MHz
        uint64_t t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };
    unsigned long long *const stamps = a.tl ? a.tl + (size_t)blockIdx.x * 48u : nullptr;   // (CX_BATCHS_TL)
    if (stamps && tid == 0u) stamps[0] = now();

    // ---- workers: the ring.  P K-steps x 2 row fragments of 16 bytes per lane.  A 32-row tile is 2 x T16 contiguous bytes
    // of the tiled shadow (padded to whole 256-row tiles, zero beyond the last row); one buffer descriptor per tile — SGPR
    // base; the only address VGPRs are the lane's piece inside a KiB for an even and for an odd 16-row block: row i = lane & 15
    // at 64 i, its piece kq = lane >> 4 at position kq ^ (bits 3-4 of the row) = kq ^ ((i >> 3) | 2 x (block parity))
    s16x8 ring[P][2];
    const uint32_t voff[2] = {j * 64u + (((kq ^ (j >> 3)) & 3u) << 4), j * 64u + (((kq ^ ((j >> 3) | 2u)) & 3u) << 4)};
    auto tile_rsrc = [&](uint32_t T) {
        const uint32_t Tc = (uint32_t)__builtin_amdgcn_readfirstlane((int)(T < n32 ? T : n32 - 1u));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(const_cast<uint16_t *>(a.shadow_t)) + (size_t)Tc * (2u * T16), 0, (int)(2u * T16), 0x00020000);
    };
    auto issue = [&](s16x8 (&slot)[2], __amdgpu_buffer_rsrc_t rs, int ks) {
#pragma unroll
        for (int f = 0; f < 2; f++)
            slot[f] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[f] + (ks & 3) * 1024, (int)(f * T16) + (ks >> 2) * 4096, 2 /* nt */));
    };

    uint32_t T = T_first + wave;
    const bool has_work = worker && T < n32;
    __amdgpu_buffer_rsrc_t crs = tile_rsrc(T);
    auto start_ring = [&]() {   // every wave, unconditionally (the service wave's and an idle worker's loads are a clamped tile's first
                                // K-steps, dropped): a branch here makes hipcc's counted waits behind it assume the shorter queue
#pragma unroll
        for (int p = 0; p < P; p++) issue(ring[p], crs, p);
    };

    if constexpr (THR) {
        start_ring();
        // ---- prologue (all eight waves): the scanned rows' shadow pieces as B fragments in LDS (they are normalised bf16 already)
        const uint32_t g = wave & 3u, half = wave >> 2, q = g * 16u + j;
        const bool live = q < nq;
        const uint32_t gr = live ? (a.scan_rows ? a.scan_rows[q] : q) : 0u;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ks++) {
            const uint32_t kk = (uint32_t)ks + half * (KS / 2);
            s16x8 H = a.shadow_q ? reinterpret_cast<const s16x8 *>(a.shadow_q + (size_t)gr * D)[4u * kk + kq]
                                 : *reinterpret_cast<const s16x8 *>(a.shadow_t + tiled_shadow_off(gr, 4u * kk + kq, KS));
            if (!live) H = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<s16x8 *>(qimg + (kk * NG + g) * 1024u + lane * 16u) = H;
        }
        if (tid < BSL_WORDS) s_ctl[tid] = 0u;
        if (tid < 64u) { s_bnd[tid] = 0u; s_inv[tid] = tid < nq ? 1.0f : 0.0f; s_mrg[tid] = 0.0f; }
        __syncthreads();
    } else {
    // ---- prologue (all eight waves): |q|, then the queries normalised, rounded to bf16, as B fragments in LDS: wave w
    // takes query group w & 3, K-steps of half w >> 2.  Lanes 4 r .. 4 r + 3 read 64 contiguous bytes of query 16 g + r
    // (16 accesses of 64 bytes per load instruction; with the fragment's own lane order — lane 16 kq + j at row j — it was 64
    // accesses of 16 bytes, and the L1's rate of line accesses made the 196 KiB of a 768-d batch a 9 us read); the lane's
    // share stays in registers from the norm to the image: read once, every load in flight at once.
    if constexpr (NB == 1) {
        const uint32_t g = wave & 3u, half = wave >> 2, r = lane >> 2, c = lane & 3u, q = g * 16u + r;
        const bool live = q < nq;
        const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.queries + (size_t)(live ? q : 0u) * D) + half * (KS / 2) * 8 + c;
        f32x4 v[KS / 2][2];   // K-step ks: elements 4 c .. 4 c + 3 and 16 + 4 c .. 16 + 4 c + 3
#pragma unroll
        for (int ks = 0; ks < KS / 2; ks++) { v[ks][0] = q4[8 * ks]; v[ks][1] = q4[8 * ks + 4]; }
        __builtin_amdgcn_sched_barrier(0);
        start_ring();   // behind the query loads in the wave's queue: the prologue does not wait for the first rows
        __builtin_amdgcn_sched_barrier(0);
        float qq = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ks++) {
            const f32x4 v0 = v[ks][0], v1 = v[ks][1];
            qq += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w + v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
        }
        qq += __shfl_xor(qq, 1, 64);
        qq += __shfl_xor(qq, 2, 64);
        if (c == 0u) s_qqp[half * QN + q] = live ? qq : 0.0f;
        if (tid < BSL_WORDS) s_ctl[tid] = 0u;
        if (tid < 64u) s_bnd[tid] = 0u;
        for (uint32_t i = tid; i < BS_LOCG * QN; i += 512u) s_loc[i] = 0u;
        __syncthreads();
        if (stamps && tid == 0u) stamps[30] = now();
        if (tid < 64u) {
            const float ss = s_qqp[tid] + s_qqp[64u + tid];
            // an IRREGULAR query (kernels.hpp: zero, non-finite, or so large / small that the reference's own sums overflow / underflow):
            // a zero image, no hits, no bound to wait for — the re-score kernel redoes it with the reference's arithmetic over every row
            const bool reg = bs_regular(ss);
            s_inv[tid] = reg ? 1.0f / sqrtf(ss) : 0.0f;
            if (blockIdx.x == 0u && tid < nq && !reg) a.ctl[BS_CTL_REDO + tid] = 1u;
        }
        __syncthreads();
        if (stamps && tid == 0u) stamps[31] = now();
        const float inv = s_inv[q];
        // (inv > 0 <=> the query is regular => every element is finite and so is its scaled value)
        const bool ok = live && inv > 0.0f;
        float es = 0.0f;   // this lane's share of || y^ - y ||^2
        // elements 4 p .. 4 p + 3 of a K-step (p = c, 4 + c) are half of fragment lane 16 (p >> 1) + r's 16 bytes
        char *const dst = qimg + (half * (KS / 2) * NG + g) * 1024u + (16u * (c >> 1) + r) * 16u + (c & 1u) * 8u;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ks++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                uint32_t w[2];   // two bf16 pairs (v_cvt_pk_bf16_f32: round to nearest even)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const f32x2 y = {ok ? v[ks][i][2 * h] * inv : 0.0f, ok ? v[ks][i][2 * h + 1] * inv : 0.0f};
                    const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(y, bf16x2_t));
                    const float d0 = y.x - __uint_as_float(pk << 16), d1 = y.y - __uint_as_float(pk & 0xFFFF0000u);
                    es += d0 * d0 + d1 * d1;
                    w[h] = pk;
                }
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2 *>(dst + ks * (NG * 1024) + i * 512) = u32x2{w[0], w[1]};   // (piece 4 + c: fragment lanes 32 further)
            }
        es += __shfl_xor(es, 1, 64);
        es += __shfl_xor(es, 2, 64);
        if (c == 0u) s_qqp[2u * QN + half * QN + q] = es;
    } else {
        // two banks: wave w takes query group w (eight groups), both halves of the K-steps one after the other — the lane's share of
        // a whole row does not fit the registers next to the ring, so the queries are read twice (from the L2): once for |q|, once
        // for the image
        const uint32_t g = wave, r = lane >> 2, c = lane & 3u, q = g * 16u + r;
        const bool live = q < nq;
        const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.queries + (size_t)(live ? q : 0u) * D) + c;
        float qq = 0.0f;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            f32x4 v[KS / 2][2];
#pragma unroll
            for (int ks = 0; ks < KS / 2; ks++) { v[ks][0] = q4[8 * (half * (KS / 2) + ks)]; v[ks][1] = q4[8 * (half * (KS / 2) + ks) + 4]; }
            if (half == 0) {
                __builtin_amdgcn_sched_barrier(0);
                start_ring();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ks = 0; ks < KS / 2; ks++) {
                const f32x4 v0 = v[ks][0], v1 = v[ks][1];
                qq += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w + v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
            }
        }
        qq += __shfl_xor(qq, 1, 64);
        qq += __shfl_xor(qq, 2, 64);
        if (c == 0u) { s_qqp[q] = live ? qq : 0.0f; s_qqp[QN + q] = 0.0f; }
        if (tid < BSL_WORDS) s_ctl[tid] = 0u;
        if (tid < QN) s_bnd[tid] = 0u;
        for (uint32_t i = tid; i < BS_LOCG * QN; i += 512u) s_loc[i] = 0u;
        __syncthreads();
        if (stamps && tid == 0u) stamps[30] = now();
        if (tid < QN) {
            const float ss = s_qqp[tid] + s_qqp[QN + tid];
            const bool reg = bs_regular(ss);
            s_inv[tid] = reg ? 1.0f / sqrtf(ss) : 0.0f;
            if (blockIdx.x == 0u && tid < nq && !reg) a.ctl[BS_CTL_REDO + tid] = 1u;
        }
        __syncthreads();
        if (stamps && tid == 0u) stamps[31] = now();
        const float inv = s_inv[q];
        const bool ok = live && inv > 0.0f;
        float es = 0.0f;
        char *const dst = qimg + g * 1024u + (16u * (c >> 1) + r) * 16u + (c & 1u) * 8u;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            f32x4 v[KS / 2][2];
#pragma unroll
            for (int ks = 0; ks < KS / 2; ks++) { v[ks][0] = q4[8 * (half * (KS / 2) + ks)]; v[ks][1] = q4[8 * (half * (KS / 2) + ks) + 4]; }
#pragma unroll
            for (int ks = 0; ks < KS / 2; ks++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    uint32_t w[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const f32x2 y = {ok ? v[ks][i][2 * h] * inv : 0.0f, ok ? v[ks][i][2 * h + 1] * inv : 0.0f};
                        const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(y, bf16x2_t));
                        const float d0 = y.x - __uint_as_float(pk << 16), d1 = y.y - __uint_as_float(pk & 0xFFFF0000u);
                        es += d0 * d0 + d1 * d1;
                        w[h] = pk;
                    }
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u32x2 *>(dst + (half * (KS / 2) + ks) * (NG * 1024) + i * 512) = u32x2{w[0], w[1]};
                }
        }
        es += __shfl_xor(es, 1, 64);
        es += __shfl_xor(es, 2, 64);
        if (c == 0u) { s_qqp[2u * QN + q] = es; s_qqp[3u * QN + q] = 0.0f; }
    }
    }
    // the service wave fills the tile queue: tiles n_static + c .. + BS_CLAIM - 1 of a claim c; BS_WORK end marks once
    // the counter has passed the last tile
    bool exhausted = false;
    uint32_t q_head = 0u;
    uint32_t seen = claim_base;   // the highest claim start this block has seen (how far the grid is: stale is fine)
    auto push_claim = [&](uint32_t c, uint32_t size) {   // service wave, all lanes
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        seen = c + size > seen ? c + size : seen;
        const uint32_t first = n_static + c;
        const uint32_t have = first >= n32 ? 0u : (n32 - first < size ? n32 - first : size);
        if (lane < have) bs_lds_st(&s_tq[(q_head + lane) & (BS_TQ - 1u)], first + lane);
        q_head += have;
        if (have < size) {
            if (lane < BS_WORK) bs_lds_st(&s_tq[(q_head + lane) & (BS_TQ - 1u)], BS_NO_TILE);
            q_head += BS_WORK;
            exhausted = true;
        }
        if (lane == 0u) bs_lds_st_rel(&s_ctl[BSL_QHEAD], q_head);
    };
    if (!worker) push_claim(claim0, CLAIM);
    __syncthreads();
    if (stamps && tid == 0u) stamps[1] = now();

    if (!worker) {
        // =============================================================== the service wave
        // One hit per lane: row filter, candidate list (the exact cosine is the select kernel's), the tile's slot.  Lanes
        // with the same query take their list positions from ONE atomic add.
        auto process_hits = [&](bool active, uint32_t row, uint32_t q, float approx) {
            if constexpr (!THR) active = active && row_passes(a.flt, row);   // (the filter pass has no row filter: the rescore and the rules look at the rows)
            row = active ? row : 0u;
            q = active ? q : 0u;
            uint64_t same = __ballot(active);
#pragma unroll
            for (int b = 0; b < (NB == 1 ? 6 : 7); b++) {
                const uint64_t m = __ballot((q >> b) & 1u);
                same &= ((q >> b) & 1u) ? m : ~m;
            }
            const int leader = __ffsll((unsigned long long)same) - 1;
            uint32_t base = 0;
            if (active && (int)lane == leader) base = atomicAdd(g_cnt + q, (uint32_t)__popcll(same));
            base = (uint32_t)__shfl((int)base, active ? leader : 0, 64);
            const uint32_t pos = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            if constexpr (THR) {
                if (active && pos < a.thr_cap) a.thr_cand[(size_t)q * a.thr_cap + pos] = row;
            } else {
                if (active && pos < a.cap) {
                    a.cand_rows[(size_t)q * a.cap + pos] = row;
                    a.cand_cos[(size_t)q * a.cap + pos] = approx;   // (the re-score looks at it again under the pass's final bound)
                }
                if (active && approx > 0.0f)
                    __hip_atomic_fetch_max(g_slots + q * BS_SL + ((row >> 5) & (BS_SL - 1u)), __float_as_uint(approx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        // The slots of query q (32 per lane) and the k-th largest of them (a radix walk over four 8-bit digits with a
        // 256-bin histogram in LDS; one wave: its LDS operations stay in order)
        constexpr int NV = (int)(BS_SL / 64u);
        uint32_t v[NV];
        // (`pre`: the first 256 slots only — the first tiles of the grid's first blocks: an eighth of the round trip's bytes for a
        // publisher that wants a bound from the earliest finishers; the k-th largest of any subset of the slots is a valid bound)
        auto slots_load = [&](uint32_t q, bool pre = false) -> uint32_t {   // returns the number of filled slots
            uint32_t nz = 0;
#pragma unroll
            for (int i = 0; i < NV; i++) v[i] = (!pre || i < 4) ? bs_ld_agent(g_slots + q * BS_SL + lane + 64u * i) : 0u;
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
        auto publish = [&](uint32_t min_filled, bool final, bool pre = false) -> bool {
            bool all_done = true;
            for (uint32_t q = blockIdx.x % QN; q < nq; q += gridDim.x) {
                if (!(s_inv[q] > 0.0f)) continue;   // an irregular query has no bound and needs none
                const uint32_t nz = slots_load(q, pre);
                if (nz < min_filled && !final) { all_done = false; continue; }
                const uint32_t t = slots_kth();
                if (lane == 0u && (t || final)) __hip_atomic_fetch_max(g_bound + q, t ? t : BS_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!t && !final) all_done = false;
            }
            return all_done;
        };

        auto refill = [&]() {   // fewer than a claim's worth of tiles queued: claim the next
            if (exhausted) return;
            // Near the end of a pass (fewer than a full claim per block left, as far as this block has seen the counter) the claims
            // are small (a.claim_tail tiles): within the boards' noise at 1.25M rows, 0.6 % at 6.25M.  (Keeping the QUEUE short there as
            // well — tiles in a block's queue are that block's for good, 21 to 41 of them when the counter runs dry — was measured
            // too: the workers then wait for claims, +2-3 % at 384-d.)
            const uint32_t n_dyn = n32 - n_static;
            const bool tail = a.claim_tail && seen + gridDim.x * CLAIM >= n_dyn;
            const uint32_t size = tail ? a.claim_tail : CLAIM, low = CLAIM;
            if ((int32_t)(q_head - bs_lds_ld(&s_ctl[BSL_QTAIL])) < (int32_t)low) {
                uint32_t c = 0u;
                if (lane == 0u) c = claim_base + atomicAdd(g_next, size);
                push_claim(c, size);
            }
        };
        uint32_t bl[NB];   // (lane, bank): the published bound of query 64 bank + lane
#pragma unroll
        for (int bk = 0; bk < NB; bk++) bl[bk] = 1u;
        constexpr uint32_t FRAC_MASK = NB == 1 ? 3u : 1u;   // publishers per query in a grid of 256 blocks: four (one bank) or two
        if constexpr (!THR) {
        // A-C. the warm-up, one polling loop: (A) once the block's workers have left their first tiles' maxima in LDS, write
        // them to the tiles' slots (plain write-through stores); (B) publish this block's queries as soon as a fraction
        // of the grid's sample is in the slots; (C) leave when every live query has a bound or a no-bound mark
        {
            const uint32_t sample = (nw < n32 ? nw : n32) < BS_SL ? (nw < n32 ? nw : n32) : BS_SL;   // slots the first tiles fill
            // the first publisher of a query (blocks 0..63) goes as soon as a.pub_min slots (>= k) are in: a bound from the earliest
            // finishers' rows lets a couple more pairs per tile through for the few microseconds until the others — at 2/8,
            // 3/8, 4/8 of the sample — and the service loop tighten it, and everybody starts testing a round trip earlier
            const uint32_t frac = (blockIdx.x / QN) & FRAC_MASK;
            const uint32_t want0 = frac ? sample * (frac + 1u) / 8u : a.pub_min;
            const uint32_t want = want0 > k ? want0 : k;
            // (all_in: every block of the grid has its first tiles in the slots — a query that has fewer than k filled slots THEN has no
            // bound to wait for: a row filter that passes next to nothing.  Round 3 found that out after 96 polls, ~1 ms into a 0.4 ms pass.)
            bool stored = false, published = false, all_in = false;
            for (int spin = 0; spin < 4096; spin++) {   // bounded: a few ms
                if (bs_lds_ld(&s_ctl[BSL_LOC]) != 0u) refill();   // (the workers are on their way under the block's local bounds)
                if (!stored && bs_lds_ld_acq(&s_ctl[BSL_ARRIVED]) >= in_block) {
                    stored = true;   // (every worker of the block has written its first tile's maxima to the slots)
                    if (lane == 0u) __hip_atomic_fetch_add(a.ctl + BS_CTL_ARR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (stamps && lane == 0u) stamps[27] = now();
                }
                // (the fraction waited for halves every few polls: when the grid's blocks are not all resident — another
                // kernel on the device: a concurrent reader's search — the slots of the missing ones stay empty until the
                // resident ones are through, and a bound from the k-th largest of what IS there is valid; without it the
                // query would run without a bound, every pair a hit, at the service wave's pace)
                // One round trip past the L2 per iteration, not two: a first publisher (blocks 0..63) polls its queries' slots until it
                // has published, then the bounds; the other blocks poll the bounds only — their publications (from 2/8 .. 4/8 of the
                // sample) are refinements, made here only when the warm-up drags on, otherwise by the service loop's refreshes
                const bool try_pub = !published && (frac == 0u || spin >= 16);
                if (try_pub) {
                    if (!all_in && spin >= 8) all_in = bs_ld_agent(a.ctl + BS_CTL_ARR) >= gridDim.x;   // (only a publisher that is still waiting asks)
                    published = publish((want >> (spin >> 2)) > k ? (want >> (spin >> 2)) : k, spin >= 96 || all_in, frac == 0u && want <= 64u && spin < 16 && !all_in);
                    if (stamps && lane == 0u && published) stamps[28] = now();
                }
                if (stamps && lane == 0u) stamps[29] = (unsigned long long)spin + 1ull;
                if (try_pub && !published && spin < 24) continue;
                uint64_t missing[NB];
                bool any_missing = false;
#pragma unroll
                for (int bk = 0; bk < NB; bk++) {
                    const uint32_t ql = 64u * bk + lane;
                    bl[bk] = (ql < nq && s_inv[ql] > 0.0f) ? bs_ld_agent(g_bound + ql) : 1u;
                    missing[bk] = __ballot(bl[bk] == 0u);
                    any_missing = any_missing || missing[bk] != 0ull;
                }
                if (stored && !any_missing) break;
                // queries nobody has published after ~80 us: their publishers are not resident (see above) — any block may
                // publish any query (an atomic max: redundancy is harmless)
                if (spin >= 24 && (spin & 7) == 0) {
#pragma unroll
                    for (int bk = 0; bk < NB; bk++)
                        while (missing[bk]) {
                            const uint32_t q = 64u * bk + (uint32_t)__ffsll((unsigned long long)missing[bk]) - 1u;
                            missing[bk] &= missing[bk] - 1ull;
                            const uint32_t nz = slots_load(q);
                            const uint32_t t = nz >= k ? slots_kth() : 0u;
                            if (lane == 0u && (t || spin >= 96 || all_in)) __hip_atomic_fetch_max(g_bound + q, t ? t : BS_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                }
                if (published) __builtin_amdgcn_s_sleep(8);
            }
        }
#pragma unroll
        for (int bk = 0; bk < NB; bk++)   // (the block's local bound may be there, and may be the higher one)
            __hip_atomic_fetch_max(&s_bnd[64u * bk + lane], bl[bk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0u) bs_lds_st_rel(&s_ctl[BSL_READY], 1u);
        if (stamps && lane == 0u) stamps[2] = now();
        // the first bounds come from the earliest finishers' slots; by now every first tile of the grid is in: one block per query
        // (the second of its publishers) tightens it from all the slots right away — the others' refreshes pick it up 2 / 6 / 14 us on
        if (((blockIdx.x / QN) & FRAC_MASK) == 1u) publish(0u, true);
        }
        // D. service loop: drain the workers' hit rings, keep the tile queue filled and the block's copy of the bounds
        // fresh (re-read and re-published 2, 4, 8, ... us apart, then every 128 us: see the head of the file)
        // A round takes up to 64 pending entries from the rings, in worker order, and looks at each under the block's bounds of the
        // moment (one compare: most hits of a block that ran ahead on its local bounds leave here); what is left goes to a staging
        // area in LDS and is handed to process_hits 64 at a time — ONE returning atomic per 64 survivors: its round trip past the L2
        // takes 2-3 us under the row stream, and paid per round (a survivor or two in nearly every round) it made the service wave
        // the last to leave by 30-60 us.
        uint32_t gap = 200u;           // x10 ns
        uint64_t t_next = now() + gap;
        unsigned long long n_rounds = 0, n_hits = 0, n_kept = 0, max_pend = 0;   // (timeline diagnostics)
        bool drained = false, seen_done = false, workers_done = false;
        uint32_t n_stage = 0u, w_start = 0u;
        // survivors of the staging area, 64 at a time: row filter, list positions, slots
        auto flush_stage = [&]() {
            const uint32_t take = n_stage < 64u ? n_stage : 64u;
            const bool on = lane < take;
            const uint32_t r0 = on ? s_stg_row[lane] : 0u, q0 = on ? s_stg_q[lane] : 0u;
            const float d0 = on ? s_stg_dot[lane] : 0.0f;
            const bool more = lane + 64u < n_stage;
            const uint32_t r1 = more ? s_stg_row[64u + lane] : 0u, q1 = more ? s_stg_q[64u + lane] : 0u;
            const float d1 = more ? s_stg_dot[64u + lane] : 0.0f;
            if (more) { s_stg_row[lane] = r1; s_stg_q[lane] = q1; s_stg_dot[lane] = d1; }   // (one wave: its LDS operations stay in order)
            n_stage -= take;
            process_hits(on, r0, q0, d0);
        };
        for (;;) {
            uint32_t hd = 0u, tl = 0u;
            if (lane < BS_WORK) { hd = bs_lds_ld_acq(&s_ctl[BSL_HEAD + lane]); tl = bs_lds_ld(&s_ctl[BSL_TAIL + lane]); }   // (the tails are this wave's own)
            const uint32_t pend = hd - tl;
            const bool busy = __ballot(pend != 0u) != 0ull;
            if (stamps && !busy && !drained) { drained = true; if (lane == 0u) stamps[32] = now(); }
            if (busy) {
                // ring by ring (starting where the last iteration stopped), chunks of 64 entries, one read of the heads for all of it; the
                // first form packed 64 entries across the rings per iteration — a chain of prefix sums and lane reads of ~2 us per 64
                // entries, and at 384-d the service wave drained a block's ~3,500 held hits until after its workers had left.  An
                // iteration ends after eight chunks or after a flush (a returning atomic: 2-3 us): the bounds' refresh below must
                // stay on time — late refreshes let 40 % more candidates through at k = 100.
                uint32_t budget = 8u, next_start = w_start;
                bool flushed = false;
#pragma unroll 1
                for (uint32_t wi = 0; wi < BS_WORK && budget && !flushed; wi++) {
                    const uint32_t w = (w_start + wi) % BS_WORK;
                    uint32_t pw = (uint32_t)__builtin_amdgcn_readlane((int)pend, (int)w);
                    if (!pw) continue;
                    const uint32_t tlw = (uint32_t)__builtin_amdgcn_readlane((int)tl, (int)w);
                    uint32_t taken = 0u;
                    if (stamps) max_pend = pw > max_pend ? pw : max_pend;
#pragma unroll 1
                    for (; budget && pw && !flushed; budget--) {
                        const uint32_t take = pw < 64u ? pw : 64u;
                        const bool on = lane < take;
                        const uint32_t e = w * HB + ((tlw + taken + lane) & (HB - 1u));
                        const uint32_t h_row = on ? s_hrow[e] : 0u, h_q = on ? s_hq[e] : 0u;
                        const float h_dot = on ? s_hdot[e] : 0.0f;
                        bool keep = on && !(a.arm & 4u);
                        if constexpr (!THR) keep = keep && !(h_dot < bs_thr(bs_lds_ld(&s_bnd[h_q]), true, s_mrg[h_q]));
                        const uint64_t km = __ballot(keep);
                        if (keep) {
                            const uint32_t sp = n_stage + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                            s_stg_row[sp] = h_row; s_stg_q[sp] = h_q; s_stg_dot[sp] = h_dot;
                        }
                        n_stage += (uint32_t)__popcll(km);
                        if (stamps) { n_rounds++; n_hits += take; n_kept += (uint32_t)__popcll(km); }
                        if (n_stage >= 64u) { flush_stage(); flushed = true; }
                        taken += take;
                        pw -= take;
                    }
                    if (lane == 0u) bs_lds_st_rel(&s_ctl[BSL_TAIL + w], tlw + taken);
                    next_start = pw ? w : (w + 1u) % BS_WORK;   // (a ring that still holds entries goes first next time)
                }
                w_start = next_start;
            }
            if (n_stage != 0u && !busy) flush_stage();
            refill();   // (an LDS read unless the queue runs low: the workers of a 384-d pass empty fourteen entries in 11 us)
            // (the clock and the bound refresh in EVERY round: looked at every eighth busy round only, the refreshes of a pass with many
            // survivors — k = 100 — came late, its workers tested against stale bounds and let 40 % more candidates through)
            {
                workers_done = bs_lds_ld_acq(&s_ctl[BSL_DONE]) >= BS_WORK;
                const uint64_t t_now = now();
                if (stamps && workers_done && !seen_done) { seen_done = true; if (lane == 0u) stamps[37] = t_now; }
                if (!THR && t_now >= t_next && !workers_done) {
                    {   // (how far the grid's tile counter is: the same round trip as the bounds')
                        const uint32_t far = claim_base + bs_ld_agent(g_next);
                        seen = far > seen ? far : seen;
                    }
#pragma unroll
                    for (int bk = 0; bk < NB; bk++) {
                        bl[bk] = bs_ld_agent(g_bound + 64u * bk + lane);
                        __hip_atomic_fetch_max(&s_bnd[64u * bk + lane], bl[bk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    if (stamps && lane == 0u) stamps[35] = now();
                    if (gap >= 800u && !(a.arm & 16u)) publish(0u, true);
                    if (stamps && lane == 0u) stamps[36] = now();
                    gap = gap < 12800u ? gap * 2u : 12800u;
                    t_next = now() + gap;
                }
            }
            if (!busy && n_stage == 0u) {
                if (workers_done) {   // every worker is through; one more look at the rings, then out
                    bool left = false;
#pragma unroll 1
                    for (uint32_t w = 0; w < BS_WORK; w++) left = left || bs_lds_ld_acq(&s_ctl[BSL_HEAD + w]) != bs_lds_ld(&s_ctl[BSL_TAIL + w]);
                    if (!left) break;
                } else {
                    __builtin_amdgcn_s_sleep(64);   // ~2 us
                }
            }
        }
        if (stamps && lane == 0u) stamps[34] = n_kept;
        if (stamps && lane == 0u) { stamps[3] = now(); stamps[33] = n_hits; stamps[38] = n_rounds; stamps[39] = max_pend; }
        return;
    }

    // =================================================================== worker waves
    if (!has_work) {
        if (lane == 0u) __hip_atomic_fetch_add(&s_ctl[BSL_DONE], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    const char *const my_q = qimg + lane * 16u;
    uint32_t *const hb_row = s_hrow + wave * HB, *const hb_q = s_hq + wave * HB;
    float *const hb_dot = s_hdot + wave * HB;
    uint32_t head = 0;               // entries this wave has put into its hit ring (wave-uniform; s_ctl[BSL_HEAD + wave] mirrors it)
    bool first = true;
    uint32_t n_tiles_done = 0;       // (timeline diagnostic)
    bool liveq[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) liveq[g] = 16u * g + j < nq && s_inv[16u * g + j] > 0.0f;   // (an irregular query takes no part in the pass)
    float thr[NG];                   // the threshold of each of the lane's four (eight) queries (bs_thr)
    float mrg[NG];                   // 2 eps of each of them (read here, right behind the prologue's last barrier: the ring that
                                     // shares this LDS is first written after every worker of the block is past its first tile)
    {
        const float dmax = a.shadow_err ? __uint_as_float(*a.shadow_err) : 0.00390625f;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const float dq = THR ? 0.0f : sqrtf(s_qqp[2u * QN + 16u * g + j] + s_qqp[3u * QN + 16u * g + j]);
            mrg[g] = 2.0f * (dmax * (1.0f + 0.00390625f) + dq * (1.0f + 1.0e-6f) + 1.0e-4f);
            if (!THR && wave == 0u && kq == 0u) {
                s_mrg[16u * g + j] = mrg[g];   // (read by the service wave once hits exist: after this wave's first tile)
                if (blockIdx.x == 0u) a.ctl[BS_CTL_MRG + 16u * g + j] = __float_as_uint(mrg[g]);
            }
        }
    }
    // room for n more entries in the ring (the service wave moves the tail).  Bounded — half a second —, and a wave that gives up
    // does NOT write into a ring without room (entries the service wave has not read would be lost without a trace): it drops its
    // hits and raises the control block's sticky failure word, and the re-score kernel redoes every query of the pass exactly
    auto wait_room = [&](uint32_t n) -> bool {
        for (int spin = 0; spin < (1 << 21); spin++) {
            if (head + n - bs_lds_ld_acq(&s_ctl[BSL_TAIL + wave]) <= HB) return true;
            __builtin_amdgcn_s_sleep(8);
        }
        // (the filter pass: the word behind its tile counter — the caller reads it with the pass's other flags and sends every scanned
        // row down the exact path; the lists' counts are left alone: a count only ever covers entries that were written)
        if (lane == 0u) __hip_atomic_store(THR ? a.thr_next + 1 : a.ctl + BS_CTL_FAIL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    };
    // query fragments: [buffer][group of four]; one bank: the next K-step's four are read while this one's MFMAs run; two banks: a
    // K-step's eight in two halves — the second half while the first half's MFMAs run, the next step's first half while the second
    // half's do (the same 32 registers; read all eight at the step's head, every step began with the LDS latency in the open)
    s16x8 B[2][4];
#pragma unroll
    for (int g = 0; g < 4; g++) B[0][g] = *reinterpret_cast<const s16x8 *>(my_q + g * 1024);
    auto claim = [&]() -> uint32_t {   // the next tile of this wave: an entry of the block's queue
        uint32_t idx = 0u;
        if (lane == 0u) idx = __hip_atomic_fetch_add(&s_ctl[BSL_QTAIL], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        while ((int32_t)(bs_lds_ld_acq(&s_ctl[BSL_QHEAD]) - idx) <= 0) __builtin_amdgcn_s_sleep(2);   // the service wave always refills: its claims depend on nobody
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)bs_lds_ld(&s_tq[idx & (BS_TQ - 1u)]));
    };

    while (T != BS_NO_TILE) {
        const uint32_t Tn = claim();
        const __amdgpu_buffer_rsrc_t nrs = tile_rsrc(Tn != BS_NO_TILE ? Tn : T);
        f32x4 acc[NG][2];
#pragma unroll
        for (int g = 0; g < NG; g++) { acc[g][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[g][1] = acc[g][0]; }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int p = ks % P;
            const int nks = ks + 1 < KS ? ks + 1 : 0;   // the last step reads the next tile's first query fragments
            const s16x8 h0 = ring[p][0], h1 = ring[p][1];
            if constexpr (NB == 1) {
                const int cur = ks & 1, nxt = cur ^ 1;
#pragma unroll
                for (int g = 0; g < 4; g++) B[nxt][g] = *reinterpret_cast<const s16x8 *>(my_q + (nks * NG + g) * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, B[cur][g], acc[g][0], 0, 0, 0);
                    acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, B[cur][g], acc[g][1], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; g++) B[1][g] = *reinterpret_cast<const s16x8 *>(my_q + (ks * NG + 4 + g) * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, B[0][g], acc[g][0], 0, 0, 0);
                    acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, B[0][g], acc[g][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; g++) B[0][g] = *reinterpret_cast<const s16x8 *>(my_q + (nks * NG + g) * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    acc[4 + g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, B[1][g], acc[4 + g][0], 0, 0, 0);
                    acc[4 + g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, B[1][g], acc[4 + g][1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the slot is free: K-step ks + P of this tile, or of the next one
            if (ks + P < KS) issue(ring[p], crs, ks + P);
            else issue(ring[p], nrs, ks + P - KS);
        }

        const uint32_t row0 = T * 32u + 4u * kq;   // this lane's rows: row0 + 16 f + r
        if (stamps && lane == 0u) { if (first) stamps[4 + wave] = now(); n_tiles_done++; }
        if (!THR && first) {
            // ---- warm-up, once per wave: the tile's best approximate cosine per query into LDS for the service wave,
            // which fills the grid's slots with them and brings the first bounds back
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
            // LOCAL bounds (k <= 14 and a full block): the block's seven first tiles are fourteen 16-row groups; the k-th largest of
            // their best approximate cosines is an A_k as valid as the grid's (k different rows reach it), known the moment the
            // block's last worker is through its first tile — no round trip past the L2, where the grid's first bounds take ~14 us
            // during which the workers of round 3 stood still.  It is weak (the k-th best of 224 rows, not of 57k: some 7 % of the
            // pairs pass), so until the grid's bounds arrive the service wave — busy polling for them anyway — leaves the hits in
            // the rings (sized for it: BsCfg::HB), and then drops nearly all of them under the real bound (the staging step of its
            // loop).  Those hits are work: a pass pays ~3,500 ring entries per block for the 14 us it does not stand still, and the
            // narrower the rows the more tiles go by meanwhile — measured (profiles/r04/tuning.md): a gain from 400k rows at 768-d
            // (0.164 against 0.173 ms) and from ~1M rows at 384-d (1.25M: 0.202 against 0.208), a loss below (100k x 384: 0.115
            // against 0.074) — so only passes of at least a.loc_min_rows rows do it (launch_batchs_pass).
            const bool loc = k <= BS_LOCG && in_block == BS_WORK && n_rows >= a.loc_min_rows && !(a.arm & 8u);
            uint32_t mine[NB];   // (lane, bank): the tile's best approximate cosine of query 64 bank + lane
#pragma unroll
            for (int bk = 0; bk < NB; bk++) mine[bk] = 0u;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                float mx = 0.0f;
#pragma unroll
                for (int f = 0; f < 2; f++) {
                    float m = 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; r++) m = (((okm >> (4 * f + r)) & 1u) && acc[g][f][r] > m) ? acc[g][f][r] : m;
                    m = fmaxf(m, __shfl_xor(m, 16, 64));
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    if (loc && kq == 0u) s_loc[(2u * wave + (uint32_t)f) * QN + 16u * g + j] = __float_as_uint(m);
                    mx = fmaxf(mx, m);
                }
                mine[g >> 2] = (kq == (uint32_t)(g & 3) && 16u * g + j < nq) ? __float_as_uint(mx) : mine[g >> 2];   // lane 16 (g mod 4) + j, bank g / 4: query 16 g + j
            }
            // (straight to the tile's slots, lane q = query q: a store the wave does not wait for — through the service wave it
            // waited for the block's slowest worker and for that wave's next poll, ~8 us on every bound of the grid)
#pragma unroll
            for (int bk = 0; bk < NB; bk++)
                if (64u * bk + lane < nq) __hip_atomic_store(g_slots + (64u * bk + lane) * BS_SL + (T & (BS_SL - 1u)), mine[bk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t before = 0u;
            if (lane == 0u) before = __hip_atomic_fetch_add(&s_ctl[BSL_ARRIVED], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
            if (loc && before + 1u == in_block) {   // the block's last arrival: lane q sorts query q's fourteen values (a network: ~130 instructions), bank by bank
                uint32_t kth[NB];
                bool none = false;
#pragma unroll
                for (int bk = 0; bk < NB; bk++) {
                    const uint32_t ql = 64u * bk + lane;
                    uint32_t v[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) v[i] = i < (int)BS_LOCG ? bs_lds_ld(&s_loc[(uint32_t)i * QN + ql]) : 0u;   // (bits of cosines >= 0 order like integers)
#pragma unroll
                    for (int p2 = 1; p2 < 16; p2 <<= 1)
#pragma unroll
                        for (int k2 = p2; k2 >= 1; k2 >>= 1)
#pragma unroll
                            for (int j2 = k2 % p2; j2 + k2 < 16; j2 += 2 * k2)
#pragma unroll
                                for (int i2 = 0; i2 < k2; i2++)
                                    if ((i2 + j2) / (2 * p2) == (i2 + j2 + k2) / (2 * p2)) {   // Batcher's odd-even merge sort, descending
                                        const uint32_t x = v[i2 + j2], y = v[i2 + j2 + k2];
                                        v[i2 + j2] = x > y ? x : y;
                                        v[i2 + j2 + k2] = x > y ? y : x;
                                    }
                    kth[bk] = 0u;
#pragma unroll
                    for (int i = 0; i < (int)BS_LOCG; i++) kth[bk] = (uint32_t)i + 1u == k ? v[i] : kth[bk];
                    const bool want = ql < nq && s_inv[ql] > 0.0f;
                    none = none || __ballot(want && kth[bk] == 0u) != 0ull;
                    kth[bk] = want ? kth[bk] : 0u;
                }
                if (!none) {   // all or nothing: a query without a local bound would flood the rings
#pragma unroll
                    for (int bk = 0; bk < NB; bk++)
                        if (kth[bk]) __hip_atomic_fetch_max(&s_bnd[64u * bk + lane], kth[bk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (lane == 0u) bs_lds_st_rel(&s_ctl[BSL_LOC], 1u);
                }
            }
            for (int spin = 0; spin < (1 << 16); spin++) {        // bounded: ~30 ms
                if (bs_lds_ld_acq(&s_ctl[BSL_READY]) != 0u || bs_lds_ld_acq(&s_ctl[BSL_LOC]) != 0u) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (stamps && lane == 0u && wave == 0u) stamps[26] = now();
        } else if (stamps && lane == 0u && wave == 0u && n_tiles_done == 2u) stamps[25] = now();
#pragma unroll
        for (int g = 0; g < NG; g++) thr[g] = THR ? (liveq[g] ? a.thr_lo : __builtin_inff()) : bs_thr(bs_lds_ld(&s_bnd[16 * g + j]), liveq[g], mrg[g]);

        // ---- the test: one compare per pair (a NaN passes); one wave-level branch
        bool any = false;
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int f = 0; f < 2; f++)
#pragma unroll
                for (int r = 0; r < 4; r++) any |= !(acc[g][f][r] < thr[g]);
        if (a.arm & 1u) any = false;
        if (__ballot(any)) {
            uint64_t hm = 0;             // hit mask (bit (g * 2 + f) * 4 + r)
#pragma unroll
            for (int g = 0; g < NG; g++)
#pragma unroll
                for (int f = 0; f < 2; f++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const bool hit = !(acc[g][f][r] < thr[g]) & (row0 + 16u * f + r < n_rows) & liveq[g];
                        hm |= hit ? (1ull << ((g * 2 + f) * 4 + r)) : 0ull;
                    }
            // A tile FULL of hits under a row filter: queries without a bound (a filter that passes next to nothing leaves fewer than k
            // slots filled) or with a bound that says little (the k-th best of the handful of rows that pass: below most rows' cosines).
            // The rows are looked at HERE then, before a thousand hits per tile go through the ring for the service wave to drop all but
            // a handful (125 of 1.25M rows passing: 3.7 ms per batch instead of 0.4) — the one case besides its first tile in which a
            // worker reads row metadata.  (Under bounds that do their work a tile has hits in a few lanes: nothing changes for those.)
            if constexpr (!THR) {
                if (!a.flt.trivial && __popcll(__ballot(hm != 0ull)) > 16) {
                    // (lane l < 32 looks at row 32 T + l: one metadata load per lane, the tile's 32 verdicts by ballot)
                    const uint32_t rl = T * 32u + (lane & 31u);
                    const uint32_t m = rl < n_rows ? a.flt.meta[rl] : META_REMOVED;
                    const uint32_t pass32 = (uint32_t)__ballot(lane < 32u && row_passes_meta(a.flt, rl, m));
                    const uint32_t okt = ((pass32 >> (4u * kq)) & 0xFu) | (((pass32 >> (16u + 4u * kq)) & 0xFu) << 4);   // bit 4 f + r: row row0 + 16 f + r passes
                    hm &= (uint64_t)okt * 0x0101010101010101ull;
                }
            }
            const uint32_t mine = (uint32_t)__popcll(hm);
            uint32_t incl = mine;        // inclusive prefix sum over the lanes
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
                if (lane >= (uint32_t)off) incl += t;
            }
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (tot <= HB) {
                if (!wait_room(tot)) hm = 0ull;
                uint32_t pos = head + incl - mine;
#pragma unroll
                for (int g = 0; g < NG; g++)
#pragma unroll
                    for (int f = 0; f < 2; f++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if ((hm >> ((g * 2 + f) * 4 + r)) & 1u) {
                                const uint32_t e = pos & (HB - 1u);
                                hb_row[e] = row0 + 16u * f + r;
                                hb_q[e] = 16u * g + j;
                                hb_dot[e] = acc[g][f][r];
                                pos++;
                            }
                if (__ballot(hm != 0ull)) {
                    head += tot;
                    if (lane == 0u) bs_lds_st_rel(&s_ctl[BSL_HEAD + wave], head);
                }
            } else {
                // a tile of a query without a bound: one hit per lane and round
#pragma unroll 1
                while (__ballot(hm != 0ull)) {
                    const bool on = hm != 0ull;
                    const uint32_t idx = on ? (uint32_t)__ffsll((unsigned long long)hm) - 1u : 0u;
                    hm &= hm - 1ull;
                    float dot = 0.0f;
#pragma unroll
                    for (int g = 0; g < NG; g++)
#pragma unroll
                        for (int f = 0; f < 2; f++)
#pragma unroll
                            for (int r = 0; r < 4; r++) dot = idx == (uint32_t)((g * 2 + f) * 4 + r) ? acc[g][f][r] : dot;
                    const uint64_t om = __ballot(on);
                    const uint32_t n = (uint32_t)__popcll(om);
                    if (!wait_room(n)) break;
                    if (on) {
                        const uint32_t e = (head + (uint32_t)__popcll(om & ((1ull << lane) - 1ull))) & (HB - 1u);
                        hb_row[e] = row0 + 16u * ((idx >> 2) & 1u) + (idx & 3u);
                        hb_q[e] = 16u * (idx >> 3) + j;
                        hb_dot[e] = dot;
                    }
                    head += n;
                    if (lane == 0u) bs_lds_st_rel(&s_ctl[BSL_HEAD + wave], head);
                }
            }
        }
        // ---- advance
        T = Tn;
        crs = nrs;
    }
    if (stamps && lane == 0u) { stamps[11 + wave] = now(); stamps[18 + wave] = n_tiles_done; }
    if (lane == 0u) __hip_atomic_fetch_add(&s_ctl[BSL_DONE], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---------------------------------------------------------------------------------------------------
// Which queries of a pass are REDONE EXACTLY instead of re-scored: the irregular ones (flagged by the pass: no image, no hits), those
// whose candidate list ran over (more hits than `cap` entries — massive ties at the k-th score, a k-th best cosine that is not
// positive, a filter that leaves nothing to bound with), every query when a worker gave up on its hit ring.  The same
// predicate in the re-score and in the select kernel, over words neither of them changes before the last select block is done.
__device__ inline bool bs_redo(const BatchSArgs &a, uint32_t q) {
    const uint32_t cnt = a.ctl[BS_CTL_CNT + q];
    return a.ctl[BS_CTL_REDO + q] != 0u || a.ctl[BS_CTL_FAIL] != 0u || cnt > a.cap || cnt + a.irr_n > a.cap;
}

// The exact redo of one query by the BS_REDO_WAVES waves of its re-score blocks: the reference's arithmetic over EVERY row (wave w
// takes rows 4 w .. 4 w + 3 of every BS_REDO_WAVES x 4: the re-score's own inner loop), each wave's k best by (score desc, row asc)
// in a register list (topk.hpp), written as the query's candidate list: wave w's entries at [w k, (w + 1) k).  Slow — gathered
// rows, 128 waves: a few ms per million rows — and exact whatever the data; only what the screening cannot decide comes here.
template <typename S, int KS>
__device__ inline void bs_exact_redo(const BatchSArgs &a, const S *rows, uint32_t q, uint32_t gw, uint32_t nwq, const float (&ql)[16], float qq,
                                     uint32_t per, uint32_t lane) {
#pragma clang fp contract(off)
    constexpr uint32_t U = 4;
    const uint32_t n = a.n_rows, k = a.k, dim = a.dim;
    WaveTopK<KS> top;
    top.init(k);
    const DevFilter &f = a.flt;
    bool nonzero = false;
#pragma unroll
    for (uint32_t i = 0; i < 16u; i++) nonzero = nonzero || ql[i] != 0.0f;
    // every cosine is NaN whatever the row when an element of the query is NaN (every dot is) or all of them are zero (0 / 0; a row
    // with an infinite element: 0 x inf): the k best are the first k rows that pass the filter — no row is read
    if (qq != qq || !__ballot(nonzero)) {
        if (gw == 0u) {
            uint32_t got = 0;
            for (uint32_t r0 = 0; r0 < n && got < k; r0 += 64u) {
                const uint32_t row = r0 + lane;
                const bool ok = row < n && row_passes(f, row);
                top.offer_lanes(ok ? make_key(__builtin_nanf(""), row) : 0ull, __builtin_nanf(""), [](uint32_t) { return true; });
                got += (uint32_t)__popcll(__ballot(ok));
            }
        }
    } else {
        for (uint32_t c = gw * U; c < n; c += nwq * U) {
            const S *p[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) p[u] = rows + (size_t)(c + u < n ? c + u : c) * dim + lane;
            float dot[U], rr[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) { dot[u] = 0.0f; rr[u] = 0.0f; }
#pragma unroll
            for (uint32_t i = 0; i < 16u; i++)
                if (i < per) {
                    float x[U];
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) x[u] = ldf(p[u] + 64u * i);
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) { dot[u] += x[u] * ql[i]; rr[u] += x[u] * x[u]; }
                }
#pragma unroll
            for (int x = 1; x < 64; x <<= 1)
#pragma unroll
                for (uint32_t u = 0; u < U; u++) { dot[u] += __shfl_xor(dot[u], x, 64); rr[u] += __shfl_xor(rr[u], x, 64); }
            float sim = 0.0f;
#pragma unroll
            for (uint32_t u = 0; u < U; u++) sim = lane == u ? cosine_from_sums(dot[u], qq, rr[u]) : sim;
            const uint64_t kg = (lane < U && c + lane < n) ? make_key(score_of(distance_of(sim)), c + lane) : 0ull;
            top.offer_lanes(kg, sim, [&f](uint32_t rw) { return row_passes(f, rw); });
        }
    }
    uint32_t *o_r = a.cand_rows + (size_t)q * a.cap + (size_t)gw * k;
    float *o_c = a.cand_cos + (size_t)q * a.cap + (size_t)gw * k;
#pragma unroll
    for (int sl = 0; sl < KS; sl++) {
        const uint32_t i = (uint32_t)sl * 64u + lane;
        if (i < k) { o_r[i] = top.key[sl] ? key_row(top.key[sl]) : BS_STRUCK; o_c[i] = top.sim[sl]; }
    }
}

// ---------------------------------------------------------------------------------------------------
// The exact cosine of every candidate from the stored rows: grid = (slices, queries); a wave takes four candidates at a time
// (their row reads in flight together: a gathered 3 KiB row per candidate is a latency, not a bandwidth, problem); each
// lane sums its strided share of dot and |row|^2 in f32 with separately rounded products — the query's share and |q|^2
// are in registers, computed once per wave —, lanes folded by a butterfly, the reference's epilogue (cosine_from_sums).
// The value depends on the row and the query only, not on where the candidate sits in the list.
// Behind the listed candidates come the store's IRREGULAR rows (kernels.hpp; the shadow's build lists them, their shadow rows are
// zero): entry e of that list is candidate cnt + e of every query — kept if the row passes the filter and IS irregular (an upsert
// may have replaced it since), while a LISTED candidate that is irregular (a query without a bound takes every row) is struck: each
// row is in the list once.  Both sides decide by the sum the shadow's build took (same order, same rounding: wave_sum).
template <typename S, int KS>
__device__ inline void bs_rescore_body(const BatchSArgs &a, const S *rows, uint32_t q, uint32_t gw, uint32_t nwq, bool redo) {
#pragma clang fp contract(off)
    constexpr uint32_t U = 4;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = a.ctl[BS_CTL_CNT + q];
    const uint32_t n_all = total + a.irr_n;
    const uint32_t n_it = n_all;
    if (!redo && gw * U >= n_it) return;
    uint32_t *cand = a.cand_rows + (size_t)q * a.cap;
    float *cosv = a.cand_cos + (size_t)q * a.cap;   // in: the approximate cosine the pass saw; out: the exact one
    // A candidate came in under the bound of its moment; the pass's last bound is the tightest.  One that fails the same test
    // against it cannot be among the k best: it is struck from the list (row BS_STRUCK) without its row being read.
    const float thr = bs_thr(a.ctl[BS_CTL_BOUND + q], true, __uint_as_float(a.ctl[BS_CTL_MRG + q]));
    const uint32_t dim = a.dim, per = dim / 64u;   // dim % 128 == 0: at most 16 elements per lane
    const float *qv = a.queries + (size_t)q * dim;
    if (redo) {   // (cold: the exact scan sums in the scan kernels' order, element lane + 64 i)
        float qs[16];
        float qq = 0.0f;
#pragma unroll
        for (uint32_t i = 0; i < 16u; i++) {
            qs[i] = i < per ? qv[lane + 64u * i] : 0.0f;
            qq += qs[i] * qs[i];
        }
#pragma unroll
        for (int x = 1; x < 64; x <<= 1) qq += __shfl_xor(qq, x, 64);
        bs_exact_redo<S, KS>(a, rows, q, gw, nwq, qs, qq, per, lane);
        return;
    }
    // The list is dealt in chunks of up to 64 entries, one per wave when the list is short enough (4 entries per wave for the few
    // hundred candidates of k = 10, 16-32 for the ~1,900 of k = 100).  A wave takes the second look at its whole chunk at once (one
    // entry per lane), strikes what fails, and goes through the REST four rows at a time: with most of a k = 100 list struck (it
    // holds what the early, weak bounds let through), four consecutive entries at a time had a row or two in flight per trip.
    const uint32_t per_wave = (n_it + nwq - 1u) / nwq;
    const uint32_t chunk = per_wave <= 4u ? 4u : (per_wave >= 64u ? 64u : ((per_wave + 3u) & ~3u));
    // (the first chunk's entries are on their way while the query is read: the kernel is a chain of dependent round trips — list
    // length, entries, rows — and the query's is not one of them)
    uint32_t c0 = gw * chunk;
    uint32_t ci = c0 + lane;
    bool valid = lane < chunk && ci < n_it, listed_l = ci < total;
    uint32_t row_l = valid ? (listed_l ? cand[ci] : a.irr_rows[ci - total]) : 0u;
    float cos_l = (valid && listed_l) ? cosv[ci] : 0.0f;
    // a lane's share of a row: the groups of four elements lane, lane + 64, ... (16 bytes per load from an f32 store, 8 from a bf16
    // one: a gathered row in 3 wave-loads at 768-d, not 12)
    const uint32_t per4 = dim / 4u;
    float ql[16];
    float qq = 0.0f;
#pragma unroll
    for (uint32_t i = 0; i < 4u; i++) {
        const uint32_t g = lane + 64u * i;
        const cx_f32x4 v = g < per4 ? reinterpret_cast<const cx_f32x4 *>(qv)[g] : cx_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) { ql[4u * i + j] = v[j]; qq += v[j] * v[j]; }
    }
#pragma unroll
    for (int x = 1; x < 64; x <<= 1) qq += __shfl_xor(qq, x, 64);
    while (c0 < n_it) {
        const bool keep_l = valid && (listed_l ? !(cos_l < thr) : (row_l < a.n_rows && row_passes(a.flt, row_l)));
        if (valid && !keep_l) cand[ci] = BS_STRUCK;
        uint64_t todo = __ballot(keep_l);
        while (todo) {   // (wave-uniform throughout)
            uint32_t row[U], cu[U];
            bool keep[U], listed[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                keep[u] = todo != 0ull;
                const uint32_t src = keep[u] ? (uint32_t)__ffsll((unsigned long long)todo) - 1u : 0u;
                todo &= todo - (keep[u] ? 1ull : 0ull);
                cu[u] = c0 + src;
                listed[u] = cu[u] < total;
                row[u] = keep[u] ? (uint32_t)__builtin_amdgcn_readlane((int)row_l, (int)src) : 0u;
            }
            float dot[U], rr[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) { dot[u] = 0.0f; rr[u] = 0.0f; }
#pragma unroll
            for (uint32_t i = 0; i < 4u; i++)
                if (64u * i < per4) {
                    const uint32_t g = lane + 64u * i;
                    cx_f32x4 x[U];
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) x[u] = (keep[u] && g < per4) ? row4(rows, (size_t)row[u], dim, g) : cx_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (uint32_t j = 0; j < 4u; j++)
#pragma unroll
                        for (uint32_t u = 0; u < U; u++) { dot[u] += x[u][j] * ql[4u * i + j]; rr[u] += x[u][j] * x[u][j]; }
                }
#pragma unroll
            for (int x = 1; x < 64; x <<= 1)
#pragma unroll
                for (uint32_t u = 0; u < U; u++) { dot[u] += __shfl_xor(dot[u], x, 64); rr[u] += __shfl_xor(rr[u], x, 64); }
            // regular or not: as the shadow's build decided.  Its sum (wave_sum's order) and this one (a butterfly) differ in the last
            // bits only, so the butterfly's value decides unless it lies within a factor of two of a limit — only then is the build's own
            // sum taken again from the row (wave-uniform, next to never)
            bool regular[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const float t = rr[u];
                const bool inside = t >= 2.0f * BS_REG_LO && t <= 0.5f * BS_REG_HI, outside = !(t >= 0.5f * BS_REG_LO && t <= 2.0f * BS_REG_HI);
                regular[u] = inside;
                if (!inside && !outside && keep[u]) {
                    float ss = 0.0f;
                    const S *p = rows + (size_t)row[u] * dim + lane;
                    for (uint32_t i = 0; i < per; i++) { const float x = ldf(p + 64u * i); ss += x * x; }
                    regular[u] = bs_regular(wave_sum(ss));
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < U; u++)
                if (lane == u && keep[u]) {
                    const bool ok = regular[u] == listed[u];
                    if (!ok || !listed[u]) cand[cu[u]] = ok ? row[u] : BS_STRUCK;   // (a listed candidate that stays keeps its entry)
                    if (ok) cosv[cu[u]] = cosine_from_sums(dot[u], qq, rr[u]);
                }
        }
        c0 += nwq * chunk;   // (a list longer than 64 entries per wave: the next chunk)
        if (c0 >= n_it) break;
        ci = c0 + lane;
        valid = lane < chunk && ci < n_it;
        listed_l = ci < total;
        row_l = valid ? (listed_l ? cand[ci] : a.irr_rows[ci - total]) : 0u;
        cos_l = (valid && listed_l) ? cosv[ci] : 0.0f;
    }
}

// (four waves per SIMD: the exact redo — a cold path — must not cost the re-score its occupancy; at 129 registers it ran three)
template <typename S, int KS>
__global__ __launch_bounds__(256, 4) void batchs_rescore_kernel(const BatchSArgs a, const S *rows) {
    bs_rescore_body<S, KS>(a, rows, blockIdx.y, blockIdx.x * 4u + (threadIdx.x >> 6), gridDim.x * 4u, bs_redo(a, blockIdx.y));
}

// ---------------------------------------------------------------------------------------------------
// Per query: the k best of the re-scored candidates -> results; clears the query's part of the control block for the
// next pass.  A list of <= NV x 1024
// entries is held in registers and selected once; a longer one (weak bounds: massive ties, a selective row filter, a
// zero query) is folded chunk by chunk, the survivors so far riding along — exact whatever the length.
template <int NV>
__device__ inline void bs_select_body(const BatchSArgs &a, uint32_t q, uint32_t n_query_blocks, uint32_t *out_rows, float *out_scores, float *out_dists,
                                      uint32_t *out_count) {
    __shared__ uint32_t sh[264];
    __shared__ uint64_t surv_k[256 + 64];
    __shared__ float surv_s[256 + 64];
    __shared__ uint32_t s_n;
    const uint32_t tid = threadIdx.x, k = a.k;
    uint32_t *const g_slots = a.ctl, *const g_bound = a.ctl + BS_CTL_BOUND, *const g_cnt = a.ctl + BS_CTL_CNT;
    if (q == 0u && tid == 0u) { a.ctl[BS_CTL_NEXT] = 0u; a.ctl[BS_CTL_ARR] = 0u; }
    // a query that was redone exactly: its list is the redo waves' k best each (already exact); otherwise the listed candidates
    // and, behind them, the store's irregular rows (batchs_rescore_kernel)
    const uint32_t total = bs_redo(a, q) ? BS_REDO_WAVES * k : g_cnt[q] + a.irr_n;
    __syncthreads();   // (every thread has read the control words this block is about to clear)
    const uint32_t *cand = a.cand_rows + (size_t)q * a.cap;
    const float *cosv = a.cand_cos + (size_t)q * a.cap;
    constexpr uint32_t CHUNK = (uint32_t)NV * 1024u - 256u;   // room for the survivors so far
    uint32_t n_surv = 0;
    uint64_t low_so_far = 0ull;   // the k-th best key of the chunks selected so far (1: fewer than k entries yet)
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
                const uint32_t row = cand[c0 + ci];
                if (row != BS_STRUCK) {   // (struck by the re-score: key 0 = no entry)
                    sim[u] = cosv[c0 + ci];
                    key[u] = make_key(score_of(distance_of(sim[u])), row);
                }
            } else if (ci - cn < n_surv) {
                key[u] = surv_k[ci - cn];
                sim[u] = surv_s[ci - cn];
            }
        }
        if (c0 != 0u && low_so_far > 1ull) {   // a later chunk of a long list: does any of its entries beat the k best so far?
            bool beats = false;
#pragma unroll
            for (int u = 0; u < NV; u++) beats = beats || (tid + (uint32_t)u * 1024u < cn && key[u] > low_so_far);
            if (!__syncthreads_or(beats ? 1 : 0)) continue;   // (uniform) no: the survivors stand
        }
        __syncthreads();
        if (tid == 0) s_n = 0u;
        const uint64_t t = block_select_kth<NV>(key, k, 56, 0, sh);   // 0: fewer than k entries — all of them survive
        const uint64_t low = t ? t : 1ull;
        low_so_far = low;
#pragma unroll
        for (int u = 0; u < NV; u++)
            if (key[u] >= low && key[u] != 0ull) {   // keys are unique (the row is part of the key): exactly min(k, live) survivors
                const uint32_t pos = atomicAdd(&s_n, 1u);
                if (pos < 256u + 64u) { surv_k[pos] = key[u]; surv_s[pos] = sim[u]; }
            }
        __syncthreads();
        n_surv = s_n < 256u + 64u ? s_n : 256u + 64u;
    }
    const uint32_t Sn = n_surv;
    uint32_t *o_rows = out_rows + (size_t)q * k;
    float *o_scores = out_scores + (size_t)q * k, *o_dists = out_dists + (size_t)q * k;
    for (uint32_t i = tid; i < Sn; i += 1024u) {
        const uint64_t ki = surv_k[i];
        uint32_t rank = 0;
        for (uint32_t jj = 0; jj < Sn; jj++) rank += surv_k[jj] > ki ? 1u : 0u;
        if (rank < k) {
            const float dist = distance_of(surv_s[i]);
            o_rows[rank] = key_row(ki);
            o_dists[rank] = dist;
            o_scores[rank] = score_of(dist);
        }
    }
    if (tid == 0) {
        out_count[q] = Sn < k ? Sn : k;
        g_cnt[q] = 0u;
        g_bound[q] = 0u;
        a.ctl[BS_CTL_REDO + q] = 0u;
        // the failure word is every block's to read (they have, above): the last block through clears it and the counter that says who is last
        if (atomicAdd(a.ctl + BS_CTL_FAIL + 1, 1u) == n_query_blocks - 1u) { a.ctl[BS_CTL_FAIL] = 0u; a.ctl[BS_CTL_FAIL + 1] = 0u; }
    }
    for (uint32_t s = tid; s < BS_SL; s += 1024u) g_slots[q * BS_SL + s] = 0u;
}

template <int NV>
__global__ __launch_bounds__(1024) void batchs_select_kernel(const BatchSArgs a, uint32_t *out_rows, float *out_scores, float *out_dists,
                                                             uint32_t *out_count) {
    // The launch picks NV for the longest list k can produce; most lists are far shorter (a few hundred entries at k = 10, ~1,900 at
    // k = 100) and a selection over fewer registers per thread is quicker (any NV is correct for any length: longer lists are folded
    // chunk by chunk).  Block-uniform; read before anything of the control block is cleared (bs_select_body's own order).
    const uint32_t q = blockIdx.x;
    const uint32_t total = bs_redo(a, q) ? BS_REDO_WAVES * a.k : a.ctl[BS_CTL_CNT + q] + a.irr_n;
    if (total <= 1024u - 256u) bs_select_body<1>(a, q, gridDim.x, out_rows, out_scores, out_dists, out_count);
    else if (NV > 4 && total <= 4u * 1024u - 256u) bs_select_body<4>(a, q, gridDim.x, out_rows, out_scores, out_dists, out_count);
    else bs_select_body<NV>(a, q, gridDim.x, out_rows, out_scores, out_dists, out_count);
}

// ---------------------------------------------------------------------------------------------------
bool batchs_supported(uint32_t dim, uint32_t k) { return dim >= 128u && dim <= 1024u && dim % 128u == 0u && k >= 1u && k <= 256u; }

// queries one pass serves: 128 (two banks) for calls of more than 64 queries at row widths up to 512 over at least 128 blocks' worth
// of rows (every query needs a publisher block: batchs_kernel), 64 otherwise; CX_BATCHS_QPP=64 switches the second bank off
uint32_t batchs_queries_per_pass(uint32_t dim, uint32_t n_rows, uint64_t nq) {
    static const uint32_t qpp_env = getenv("CX_BATCHS_QPP") ? (uint32_t)atoi(getenv("CX_BATCHS_QPP")) : 128u;
    return (qpp_env >= 128u && dim <= 512u && nq > 64u && n_rows >= 128u * BS_WORK * 32u) ? 128u : 64u;
}

uint32_t batchs_min_rows() {
    static const uint32_t v = getenv("CX_BATCHS_MIN_ROWS") ? (uint32_t)atoi(getenv("CX_BATCHS_MIN_ROWS")) : 131072u;
    return v;
}

template <int D>
static int launch_batchs_d(const BatchSArgs &a, uint32_t grid, hipStream_t stream) {
    static std::atomic<uint64_t> attr_devices{0};
    if (first_use_on_device(attr_devices)) {
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchs_kernel<D, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchs_kernel<D, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if constexpr (D <= 512)
            CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batchs_kernel<D, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (a.thr_cand) {
        hipLaunchKernelGGL((batchs_kernel<D, true, 1>), dim3(grid), dim3(512), BsCfg<D>::LDS, stream, a);
    } else if (a.nq > 64u) {   // two banks of 64 queries (launch_batchs_pass has checked the width)
        if constexpr (D <= 512) hipLaunchKernelGGL((batchs_kernel<D, false, 2>), dim3(grid), dim3(512), (BsCfg<D, 2>::LDS), stream, a);
        else return set_err(CX_ERR_VALIDATION, "batchs: %u queries in one pass need a row width of at most 512 (got %u)", a.nq, a.dim);
    } else {
        hipLaunchKernelGGL((batchs_kernel<D, false, 1>), dim3(grid), dim3(512), BsCfg<D>::LDS, stream, a);
    }
    CX_HIP(hipGetLastError());
    return CX_OK;
}

int launch_batchs_pass(const BatchSArgs &a_in, hipStream_t stream) {
    BatchSArgs a = a_in;
    static const uint32_t arm_env = getenv("CX_BATCHS_ARM") ? (uint32_t)atoi(getenv("CX_BATCHS_ARM")) : 0u;
    static const uint32_t pub_env = getenv("CX_BATCHS_PUB_MIN") ? (uint32_t)atoi(getenv("CX_BATCHS_PUB_MIN")) : 64u;
    static const uint32_t claim_env = getenv("CX_BATCHS_CLAIM") ? (uint32_t)std::min(32, std::max(1, atoi(getenv("CX_BATCHS_CLAIM")))) : BS_CLAIM;
    a.arm = arm_env;
    a.pub_min = pub_env;
    a.claim = claim_env;
    static const uint32_t tail_env = getenv("CX_BATCHS_CLAIM_TAIL") ? (uint32_t)std::min(32, std::max(0, atoi(getenv("CX_BATCHS_CLAIM_TAIL")))) : BS_WORK;
    a.claim_tail = tail_env;
    static const long loc_env = getenv("CX_BATCHS_LOC_MIN") ? atol(getenv("CX_BATCHS_LOC_MIN")) : -1;
    a.loc_min_rows = loc_env >= 0 ? (uint32_t)loc_env : (a.dim >= 640u ? 393216u : 1048576u);   // (block-local first bounds: see the kernel)
    if (!batchs_supported(a.dim, a.k) || a.nq == 0 || a.nq > batchs_queries_per_pass(a.dim, a.n_rows, a.nq) || a.n_rows == 0 || (a.thr_cand && a.nq > 64u))
        return set_err(CX_ERR_VALIDATION, "batchs: unsupported shape (dim %u, k %u, %u queries, %u rows)", a.dim, a.k, a.nq, a.n_rows);
    const uint32_t cus = device_cus(), n32 = (a.n_rows + 31u) / 32u;
    // every worker wave needs a first tile of its own (its warm-up fills a slot); blocks of 7 workers + the service wave
    uint32_t grid = n32 / BS_WORK;
    grid = grid < 1u ? 1u : (grid > cus ? cus : grid);
    auto dispatch = [](const BatchSArgs &b, uint32_t g, hipStream_t st) -> int {
        switch (b.dim) {
            case 128: return launch_batchs_d<128>(b, g, st);
            case 256: return launch_batchs_d<256>(b, g, st);
            case 384: return launch_batchs_d<384>(b, g, st);
            case 512: return launch_batchs_d<512>(b, g, st);
            case 640: return launch_batchs_d<640>(b, g, st);
            case 768: return launch_batchs_d<768>(b, g, st);
            case 896: return launch_batchs_d<896>(b, g, st);
            default: return launch_batchs_d<1024>(b, g, st);   // (batchs_supported: 1024)
        }
    };
    static const bool tl_env = getenv("CX_BATCHS_TL") && atoi(getenv("CX_BATCHS_TL")) != 0;
    if (tl_env) {   // diagnostic: one pass with stamps, its timeline on stderr (one stamp buffer per device)
        static unsigned long long *d_tls[64] = {nullptr};
        int dev = 0;
        CX_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev > 63) return set_err(CX_ERR_DEVICE, "batchs timeline: device %d", dev);
        unsigned long long *&d_tl = d_tls[dev];
        if (!d_tl) CX_HIP(hipMalloc(&d_tl, (size_t)1024 * 48 * 8));
        CX_HIP(hipMemsetAsync(d_tl, 0, (size_t)1024 * 48 * 8, stream));
        a.tl = d_tl;
        int rc = dispatch(a, grid, stream);
        if (rc) return rc;
        CX_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h((size_t)grid * 48);
        CX_HIP(hipMemcpy(h.data(), d_tl, h.size() * 8, hipMemcpyDeviceToHost));
        static int tl_calls = 0;
        if (++tl_calls % 8 == 0) {
            unsigned long long t0 = ~0ull;
            for (uint32_t b = 0; b < grid; b++) t0 = std::min(t0, h[(size_t)b * 48]);
            auto stat = [&](const char *name, int lo, int hi) {
                std::vector<double> v;
                for (uint32_t b = 0; b < grid; b++)
                    for (int i = lo; i < hi; i++) if (h[(size_t)b * 48 + i]) v.push_back((double)(h[(size_t)b * 48 + i] - t0) * 0.01);
                if (v.empty()) return;
                std::sort(v.begin(), v.end());
                fprintf(stderr, "  %-28s min %7.1f  p10 %7.1f  med %7.1f  p90 %7.1f  max %7.1f us\n", name, v.front(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
            };
            fprintf(stderr, "[batchs timeline] %u rows x %u, %u queries, k %u, grid %u\n", a.n_rows, a.dim, a.nq, a.k, grid);
            stat("block entry", 0, 1); stat("prologue: queries read", 30, 31); stat("prologue: norms", 31, 32); stat("prologue done", 1, 2); stat("worker: first tile done", 4, 11); stat("service: bounds in LDS", 2, 3);
            stat("service: maxima stored", 27, 28); stat("service: own queries published", 28, 29);
            stat("service: backlog drained", 32, 33); stat("service: last refresh begins", 35, 36); stat("service: last refresh ends", 36, 37); stat("service: sees workers done", 37, 38);
            {
                unsigned long long hits = 0, rounds = 0, mp = 0;
                for (uint32_t b = 0; b < grid; b++) { hits += h[(size_t)b * 48 + 33]; rounds += h[(size_t)b * 48 + 38]; mp = std::max(mp, h[(size_t)b * 48 + 39]); }
                unsigned long long kept = 0;
                for (uint32_t b = 0; b < grid; b++) kept += h[(size_t)b * 48 + 34];
                fprintf(stderr, "  service waves: %llu hits taken from the rings (%.0f per block), %llu kept under the bounds of the moment, %.0f loop rounds per block, largest backlog %llu\n", hits, (double)hits / grid, kept, (double)rounds / grid, mp);
            }
            stat("worker 0: past the wait", 26, 27); stat("worker 0: second tile done", 25, 26); stat("worker exit", 11, 18); stat("service exit", 3, 4);
            std::vector<unsigned long long> nt;
            for (uint32_t b = 0; b < grid; b++) for (int i = 18; i < 25; i++) nt.push_back(h[(size_t)b * 48 + i]);
            std::sort(nt.begin(), nt.end());
            fprintf(stderr, "  tiles per worker: min %llu med %llu max %llu\n", nt.front(), nt[nt.size() / 2], nt.back());
            for (int w = 0; w < 7; w++) {   // by worker index: tiles done (mean / min / max over the blocks), mean exit time
                double st = 0, se = 0; unsigned long long mn = ~0ull, mx = 0;
                for (uint32_t b = 0; b < grid; b++) { const unsigned long long t = h[(size_t)b * 48 + 18 + w]; st += (double)t; mn = std::min(mn, t); mx = std::max(mx, t); se += (double)(h[(size_t)b * 48 + 11 + w] - t0) * 0.01; }
                fprintf(stderr, "  worker %d: tiles mean %.1f min %llu max %llu, exit mean %.1f us\n", w, st / grid, mn, mx, se / grid);
            }
            for (int x = 0; x < 8; x++) {   // by XCD (block index mod 8): tiles per block, exit of the last worker
                double st = 0, se = 0; int nb = 0;
                for (uint32_t b = x; b < grid; b += 8) { unsigned long long t = 0, e = 0; for (int w = 0; w < 7; w++) { t += h[(size_t)b * 48 + 18 + w]; e = std::max(e, h[(size_t)b * 48 + 11 + w]); } st += (double)t; se += (double)(e - t0) * 0.01; nb++; }
                fprintf(stderr, "  blocks = %d mod 8: tiles per block mean %.1f, last worker's exit mean %.1f us\n", x, st / nb, se / nb);
            }
            std::vector<unsigned long long> sp;
            for (uint32_t b = 0; b < grid; b++) sp.push_back(h[(size_t)b * 48 + 29]);
            std::sort(sp.begin(), sp.end());
            fprintf(stderr, "  warm-up polls per block: min %llu med %llu max %llu\n", sp.front(), sp[sp.size() / 2], sp.back());
            for (uint32_t b : {0u, 1u, 63u, 64u, 128u, 255u}) if (b < grid)
                fprintf(stderr, "  block %3u: stored %.1f published %.1f ready %.1f polls %llu\n", b, (double)(h[(size_t)b * 48 + 27] - t0) * 0.01, h[(size_t)b * 48 + 28] ? (double)(h[(size_t)b * 48 + 28] - t0) * 0.01 : -1.0, (double)(h[(size_t)b * 48 + 2] - t0) * 0.01, h[(size_t)b * 48 + 29]);
        }
        return CX_OK;
    }
    return dispatch(a, grid, stream);
}

// The all-pairs filter of a small scan set through the same pass (THR): candidate columns out, allpairs_stream.hip's contract.
bool batchs_thr_supported(uint32_t n_rows, uint32_t dim, uint32_t n_scan) {
    return n_scan >= 1u && n_scan <= 64u && batchs_supported(dim, 1) && n_rows >= batchs_min_rows();
}
int launch_batchs_thr(const BatchSArgs &a_in, hipStream_t stream) {
    BatchSArgs a = a_in;
    a.k = 1;
    if (!a.thr_cand || !a.thr_cand_cnt || !a.thr_next || !a.shadow_t) return set_err(CX_ERR_VALIDATION, "batchs (threshold mode): incomplete arguments");
    CX_HIP(hipMemsetAsync(a.thr_next, 0, 2 * sizeof(uint32_t), stream));   // the tile counter and the give-up flag behind it
    return launch_batchs_pass(a, stream);
}

template <typename S>
static void launch_rescore_s(const BatchSArgs &a, const S *rows, hipStream_t stream) {
    // 128 waves per query (those past the list's end leave at once): a few hundred to a few thousand candidates; BS_REDO_WAVES
    // when a query is redone exactly
    const dim3 grid(BS_REDO_WAVES / 4u, a.nq), block(256);
    if (a.k <= 64u) hipLaunchKernelGGL((batchs_rescore_kernel<S, 1>), grid, block, 0, stream, a, rows);
    else if (a.k <= 128u) hipLaunchKernelGGL((batchs_rescore_kernel<S, 2>), grid, block, 0, stream, a, rows);
    else hipLaunchKernelGGL((batchs_rescore_kernel<S, 4>), grid, block, 0, stream, a, rows);
}

int launch_batchs_select(const BatchSArgs &a, uint32_t *out_rows, float *out_scores, float *out_dists, uint32_t *out_count, hipStream_t stream) {
    if ((uint64_t)BS_REDO_WAVES * a.k > a.cap) return set_err(CX_ERR_VALIDATION, "batchs: candidate lists of %u entries cannot hold an exact redo (k %u)", a.cap, a.k);
    if (a.rows16) launch_rescore_s(a, a.rows16, stream);
    else launch_rescore_s(a, a.rows, stream);
    if (a.k <= 32u) hipLaunchKernelGGL(batchs_select_kernel<4>, dim3(a.nq), dim3(1024), 0, stream, a, out_rows, out_scores, out_dists, out_count);
    else hipLaunchKernelGGL(batchs_select_kernel<8>, dim3(a.nq), dim3(1024), 0, stream, a, out_rows, out_scores, out_dists, out_count);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

// entries per query of the candidate lists (index.cpp sizes the scratch with it): every row of a small store, CX_BATCHS_CAND_CAP
// (65,536) of a large one — 33.5 MB per context for 64 queries, where round 3 held 512 bytes per ROW —, never fewer than an
// exact redo writes
uint32_t batchs_cand_cap(uint32_t n_rows, uint32_t k) {
    static const uint32_t cap_env = getenv("CX_BATCHS_CAND_CAP") ? (uint32_t)std::max(1, atoi(getenv("CX_BATCHS_CAND_CAP"))) : 65536u;
    return std::max<uint32_t>(std::min<uint32_t>(n_rows, cap_env), BS_REDO_WAVES * k);
}

}  // namespace cx
