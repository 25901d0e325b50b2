// batch.hip — batched exact search: many queries per pass over the corpus (VectorIndex::search_batch,
// vector/index.rs:390-410; BASELINE config 4's inner loop; the auto-linker's top-100 lists).  The reference runs B
// independent scans (rayon, one per query); here the row store is read ONCE per 64 queries (32 for the wide lists of a
// call with no more than 32 queries).
//
// At B = 64 the contraction needs 32 flop per corpus byte — past what f32 VALU FMAs sustain next to an 8 TB/s
// stream — so the dot products run on the matrix cores.  An exact-f32 MFMA (v_mfma_f32_16x16x4_f32) version of this
// kernel was the first attempt: one wave per SIMD serialised every phase and the f32 MFMAs alone needed 6.1k cycles
// per 16-row tile (2.13 ms per 64 queries over 1.25M x 768; profiles/r01/tuning.md); it is gone.  batch2_kernel:
//  - split precision: every f32 value a = hi + lo with hi = bf16(a), lo = bf16(a - hi) (16 mantissa bits), a product
//    is three mfma_f32_16x16x32_bf16 (hi.hi + hi.lo + lo.hi; lo.lo is below what the split already drops): measured
//    |cos error| <= 9e-7, inside the 5e-5 parity tolerance;
//  - the rows arrive ALREADY split: the index keeps a bf16 hi/lo copy of the store in the LDS image layout
//    (build_split_kernel, cx_index::d_split) and their |row|^2 (d_norms);
//  - 512-thread block per CU: 4 consumer waves (16 queries each in registers, MFMA loop, per-tile test, appends)
//    and 4 producer waves (move the next tiles HBM -> registers -> LDS, compact the candidate lists, share a
//    score bound with the other blocks); one block barrier per 16-row tile (two in the single-buffer wide mode, MODE 8);
//  - per-block lists go to HBM, one merge block per query finishes (merge_small_kernel / merge_radix_kernel).
#include <vector>

#include "batch_common.hpp"
#include "kernels.hpp"
#include "topk.hpp"

namespace cx {

constexpr int BT_ROWS = SPLIT_TILE_ROWS;   // rows per tile
constexpr int BT_Q = 64;      // queries per pass (4 waves x 16)
constexpr uint32_t BATCH_K_WIDE = 104;   // largest k of the wide mode (lists of k + 48 for 32 queries next to two 48 KiB tiles)

// key of a stored candidate (row, sim): the same order key the scan path uses
__device__ inline uint64_t cand_key(uint32_t row, float sim) { return make_key(score_of(distance_of(sim)), row); }

// ---------------------------------------------------------------------------------------------------
// tile images: a 16-row tile is D / 32 K-steps of [hi fragment | lo fragment] (batch_common.hpp), 16 rows x D x 4 bytes
template <int D>
struct Batch2Cfg {
    static constexpr int TILE_BYTES = BT_ROWS * D * 4;
    static constexpr int LOADS = BT_ROWS * D * 4 / 1024 / 4; // 1 KiB wave loads per wave per tile (4 waves)
    static_assert(D % 256 == 0 || D == 384, "dim must be 384 or a multiple of 256");
};

// sum over the 64 lanes, result in every lane: four DPP steps inside each 16-lane row (VALU rate), then
// two cross-row exchanges
__device__ inline float wave_sum_dpp(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// E = candidate-list entries a lane holds in a compaction: 1 for k <= 32 (64 queries per pass, lists of 80),
// 4 for k <= 104 ("wide": the auto-linker's top-100 lists — 32 queries per pass with lists of up to 272, or all 64 with
// lists of 216 / 264 beside a single tile buffer when the call has more than 32 queries).
// A list entry is (row, dot) plus, for E = 1, |row|^2 — 12 bytes, so that no compaction and no final ordering goes
// back to HBM for the norm (the wide lists spend that LDS on length and gather it from the index's norm cache,
// a.norms); the producers copy the norms per tile into LDS for the consumers' per-tile test.
// MODE: 1 = lists of 80 (k <= 32), 64 queries; 4 = wide lists, 32 queries; 8 = wide lists, all 64 queries, one tile buffer
template <int D, bool DIAG, int MODE>
__global__ __launch_bounds__(512, 2) void batch2_kernel(const BatchArgs a_in) {
    using C = Batch2Cfg<D>;
    constexpr int E = MODE == 1 ? 1 : 4;
    // tile buffers: two (the producers write tile t + 1 while the consumers read tile t), except for the wide lists of
    // all 64 queries: at 768-d they leave LDS for one 48 KiB buffer only, so the producers write between two barriers
    // while the consumers wait, and compact while the consumers compute; at 384-d two buffers would fit, but one
    // buys lists of 264 instead of 216 (a third fewer compactions, and this mode is compaction-bound): 8 % faster
    constexpr uint32_t NBUF = MODE == 8 ? 1u : 2u;
    // grid = (row chunks, query groups): block (x, y) scans the tiles t = x (mod gridDim.x) for query group y
    // (64 queries, 32 in the wide mode).  Small corpora get few chunks — so that a block still sees enough rows
    // for its own bound to mean something — and many groups per launch; large ones one group on every CU.
    BatchArgs a = a_in;
    {
        constexpr uint32_t QPP = MODE != 4 ? BT_Q : BT_Q / 2;
        const uint32_t grp = blockIdx.y;
        a.queries += (size_t)grp * QPP * D;
        a.nq = a_in.nq - grp * QPP < QPP ? a_in.nq - grp * QPP : QPP;
        a.part_keys += (size_t)grp * QPP * gridDim.x * a_in.k;
        a.part_sims += (size_t)grp * QPP * gridDim.x * a_in.k;
        a.gslots += (size_t)grp * BT_Q * 128u;
        if constexpr (DIAG) a.diag += (size_t)grp * gridDim.x * 64u;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [2 tiles: hi image | lo image][rr: 2 x 16 f32][cand rows | dots | (E = 1) norms: QC x capq][tau][cnt][tsq][pub][shr][qq]
    char *tiles = smem;
    float *c_rr = reinterpret_cast<float *>(smem + NBUF * C::TILE_BYTES);
    const uint32_t capq = a.capq;
    // queries that own a candidate list: 64, or 32 for the wide lists (two 48 KiB tile buffers leave LDS for 32 lists
    // of 248); at 384-d the tiles are half as big and MODE 8 keeps all 64 queries (lists of 216) — 1.4x the queries per
    // second once a call has more than 32 of them, slower below (half the producers then own all the lists)
    constexpr bool FULLQ = MODE != 4;
    constexpr uint32_t QC = FULLQ ? BT_Q : BT_Q / 2;
    uint32_t *c_rows = reinterpret_cast<uint32_t *>(c_rr + 2 * BT_ROWS);
    float *c_dots = reinterpret_cast<float *>(c_rows + QC * capq);
    // E = 1 keeps |row|^2 beside each entry (the consumer has it in hand when it appends), so no compaction and no
    // final ordering goes back to HBM for it; the wide lists spend that LDS on length instead and gather from a.norms
    float *c_nrm = c_dots + QC * capq;
    uint64_t *c_tau = reinterpret_cast<uint64_t *>(c_nrm + (E == 1 ? QC * capq : 0u));
    uint32_t *c_cnt = reinterpret_cast<uint32_t *>(c_tau + BT_Q);
    float *c_tsq = reinterpret_cast<float *>(c_cnt + BT_Q);
    uint32_t *c_pub = reinterpret_cast<uint32_t *>(c_tsq + BT_Q);   // entries of a list that are completely written
    uint32_t *c_shr = c_pub + BT_Q;                                 // != 0: a producer compacted entries [0, c_shr) to [0, k)
    float *c_qq = reinterpret_cast<float *>(c_shr + BT_Q);          // |q|^2 per query, for the producers

    // Wave roles: waves 0-3 are CONSUMERS (16 queries each in registers: MFMA loop, epilogue, candidate
    // buffers), waves 4-7 are PRODUCERS (bring the next tile: global loads, bf16 split, row norms, LDS
    // writes).  A consumer and a producer share each SIMD, so the producer's VALU/LDS work for tile t+1 runs
    // under the consumer's MFMAs for tile t instead of after them (one wave per SIMD serialised every phase:
    // profiles/r01/tuning.md).  One block barrier per tile hands the tile over.
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool consumer = __builtin_amdgcn_readfirstlane(wave) < 4u;
    const uint32_t pw = wave & 3u;                 // consumer: query group; producer: row group of the tile
    const uint32_t k = a.k, n_rows = a.n_rows;
    const uint32_t n_tiles = (n_rows + BT_ROWS - 1) / BT_ROWS;
    if (tid < BT_Q) { c_cnt[tid] = 0; c_tau[tid] = 0ull; c_tsq[tid] = -1.0f; c_pub[tid] = 0; c_shr[tid] = 0; c_qq[tid] = 0.0f; }

    unsigned long long t_stage = 0, t_mfma = 0, t_epi = 0, t_bar = 0, t_prev = 0, t_wait = 0, t_write = 0;
    unsigned long long t_begin = 0;
    if constexpr (DIAG) t_begin = __builtin_readcyclecounter();
    auto stamp = [&](unsigned long long &acc) {
        if constexpr (DIAG) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            acc += t - t_prev;
            t_prev = t;
        }
    };
    auto stamp0 = [&]() {
        if constexpr (DIAG) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); t_prev = t; }
    };

    // The per-tile barrier.  __syncthreads() would also drain every outstanding global load (its fence waits for
    // vmcnt(0)), i.e. the producers' prefetched tiles — measured as ~2k idle cycles per tile on the producers.  What the
    // hand-over needs is only that this wave's LDS writes have completed: lgkmcnt(0), then the raw barrier.
    auto tile_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // Final phase, shared by all eight waves once the last tile's barrier is behind them: wave w orders and writes the
    // lists [w QC/8, (w+1) QC/8) — per-block lists go to part[(q * grid + block) * k + r].  A list holds at most
    // capq - 16 entries at a barrier (the producers' compaction invariant below): 64, or 256 in the wide mode — one,
    // or four, per lane.  All of the wave's lists are read and turned into keys first (their LDS reads, norm
    // gathers and divisions overlap), then each is ranked by counting (entry 64 e + lane sits in slot e of this
    // lane) and the best k go straight from registers to their places in HBM.
    constexpr int NE = E == 1 ? 1 : 4;
    auto finalize_lists = [&]() {
        constexpr uint32_t PER = QC / 8u;
        uint64_t key[PER][NE];
        float sim[PER][NE];
        uint32_t nn[PER];
#pragma unroll
        for (uint32_t l = 0; l < PER; l++) {
            const uint32_t qs = wave * PER + l;
            const bool on = qs < a.nq;
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(on ? c_cnt[qs] : 0u));   // uniform: scalar loop bounds below
            const uint32_t n = cnt < 64u * NE ? cnt : 64u * NE;
            const float qq_l = c_qq[on ? qs : 0u];
            nn[l] = n;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const uint32_t idx = lane + 64u * e;
                key[l][e] = 0ull; sim[l][e] = 0.0f;
                if (idx < n) {
                    const uint32_t r = c_rows[qs * capq + idx];
                    const float nr = E == 1 ? c_nrm[qs * capq + idx] : a.norms[r];
                    sim[l][e] = cosine_from_sums(c_dots[qs * capq + idx], qq_l, nr);
                    key[l][e] = cand_key(r, sim[l][e]);
                }
            }
        }
#pragma unroll
        for (uint32_t l = 0; l < PER; l++) {
            const uint32_t qs = wave * PER + l;
            if (qs >= a.nq) break;
            const uint32_t n = nn[l];
            uint32_t rank[NE];
#pragma unroll
            for (int e = 0; e < NE; e++) rank[e] = 0;
#pragma unroll
            for (int g = 0; g < NE; g++) {
                const uint32_t hi = n < 64u * (g + 1) ? n : 64u * (g + 1);
                for (uint32_t f = 64u * g; f < hi; f++) {
                    const uint64_t kf = readlane_u64(key[l][g], (int)(f - 64u * g));
#pragma unroll
                    for (int e = 0; e < NE; e++) rank[e] += kf > key[l][e] ? 1u : 0u;
                }
            }
            const size_t base = ((size_t)qs * gridDim.x + blockIdx.x) * k;
#pragma unroll
            for (int e = 0; e < NE; e++)
                if (lane + 64u * e < n && rank[e] < k) { a.part_keys[base + rank[e]] = key[l][e]; a.part_sims[base + rank[e]] = sim[l][e]; }
            for (uint32_t idx = lane; idx < k; idx += 64u)   // a block that saw fewer than k rows pads its list
                if (idx >= n) { a.part_keys[base + idx] = 0ull; a.part_sims[base + idx] = 0.0f; }
        }
    };

    if (!consumer) {
        // ------------------------------------------------------------------ producer
        // The rows arrive already split: the index keeps a bf16 hi/lo copy of the store laid out tile by tile exactly
        // like the LDS images (cx_index::d_split, build_split_kernel), 4 bytes per element like the f32 rows, so a
        // producer only moves its 12 KiB share of a tile — 16-byte loads into registers, ds_write_b128 to the same
        // offsets — and the per-batch split (2.5k cycles of VALU per tile, the kernel's critical path) is gone.
        // Two register sets, tiles alternate between them: every load has two tile intervals of flight time.
        f32x4 ldA[C::LOADS], ldB[C::LOADS], nrA, nrB;   // tile bytes and the wave's four cached row norms
        const uint32_t my_off = pw * (uint32_t)C::LOADS * 1024u + lane * 16u;
        auto issue_loads = [&](f32x4 (&ld)[C::LOADS], f32x4 &nr, uint32_t t) {
            const uint32_t tile = t < n_tiles ? t : n_tiles - 1u;
            const char *base = a.split + (size_t)tile * C::TILE_BYTES + my_off;
#pragma unroll
            for (int e = 0; e < C::LOADS; e++) ld[e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + e * 1024));
            nr = *reinterpret_cast<const f32x4 *>(a.norms + (size_t)tile * BT_ROWS + pw * 4u);
        };
        // write the tile held in ld[] and put each register back in flight for tile `reload` as soon as it is out
        auto write_tile = [&](f32x4 (&ld)[C::LOADS], f32x4 &nr, uint32_t buf, uint32_t reload) {
            const uint32_t rl = reload < n_tiles ? reload : n_tiles - 1u;   // past the end: a valid tile, never used
            const char *rbase = a.split + (size_t)rl * C::TILE_BYTES + my_off;
            char *dst = tiles + buf * C::TILE_BYTES + my_off;
            const f32x4 rr_now = nr;
            nr = *reinterpret_cast<const f32x4 *>(a.norms + (size_t)rl * BT_ROWS + pw * 4u);
#pragma unroll
            for (int e = 0; e < C::LOADS; e++) {
                // store first, then reload INTO THE SAME REGISTER: with the load ahead of the store the compiler needs a
                // second register for the new value and later copies it home — a v_mov of an in-flight load result,
                // i.e. an s_waitcnt vmcnt(0) per tile that cut the prefetch distance to a fraction of a tile
                *reinterpret_cast<f32x4 *>(dst + e * 1024) = ld[e];
                ld[e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rbase + e * 1024));
            }
            if (lane == 0) *reinterpret_cast<f32x4 *>(c_rr + buf * BT_ROWS + pw * 4u) = rr_now;
        };
        // Candidate-list compaction runs HERE, on the producer waves (they have the registers; the consumers hold
        // 16 queries in 192 VGPRs).  Producer wave pw owns the lists of consumer wave pw.  A list is compacted when
        // its published length n0 reaches capq - 32: the best k of entries [0, n0) move to [0, k) in any order and
        // tau is refreshed; the consumer keeps appending at [cnt, ...) meanwhile and, behind the next barrier,
        // slides what it appended since down to k (c_shr = n0 tells it).  The k-th best score comes from a radix
        // select over the 32-bit score order: one ballot + popcount per bit, all scalar.  Invariant: cnt <= capq - 16
        // at every barrier, so 16 appends always fit.
        //
        // Global slots: a block only ever sees n_rows / grid rows, so its own k-th best is a loose bound for most
        // of its life.  Blocks therefore share one: every survivor of a compaction is pushed into slot
        // hash(row) % k of its query with an agent-scope atomic max on the bits of its cosine (> 0).  The k slots
        // hold cosines of k DIFFERENT rows, so their minimum is a valid lower bound of the query's global k-th best
        // cosine whichever blocks contributed — rows below it can be skipped by everyone.  Results do not depend
        // on timing: the bound only removes rows that cannot be in the top k.
        unsigned long long n_pcompact = 0, n_tie = 0, cp[6] = {0, 0, 0, 0, 0, 0}, cp_t = 0;
        auto cstamp = [&](int i) {
            if constexpr (DIAG) {
                __builtin_amdgcn_sched_barrier(0);
                unsigned long long t;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
                __builtin_amdgcn_sched_barrier(0);
                cp[i] += t - cp_t; cp_t = t;
            }
        };
        // E = 1: 4 lanes per query, all 16 queries of the group per refresh, slots at stride 32;
        // wide (E = 4): 8 lanes per query, the 8 queries this producer owns per refresh, stride 128
        constexpr uint32_t LPQ = E == 1 ? 4u : 8u, NG = E == 1 ? 4u : 7u, GSTRIDE = E == 1 ? 32u : 128u;
        uint64_t gv[NG];
#pragma unroll
        for (uint32_t i = 0; i < NG; i++) gv[i] = 0ull;
        const uint32_t g_g = lane & (LPQ - 1u);
        // Ownership of the lists: E = 1: producer pw owns the 16 lists of consumer wave pw.  Wide mode has only two
        // live consumer waves, so all four producers share their lists: producer pw owns the queries of group pw & 1
        // whose index has parity pw >> 1 (twice the compaction throughput; the same producer refreshes their bounds).
        constexpr bool SHARED = !FULLQ;   // two live consumer waves: the four producers share their lists by parity
        const uint32_t own_grp = SHARED ? (pw & 1u) : pw, own_par = pw >> 1;
        // wide with all 64 queries: a refresh covers 8 of the producer's 16 queries, alternating halves
        uint32_t g_q = SHARED ? own_grp * 16u + 2u * (lane / LPQ) + own_par : pw * 16u + lane / LPQ;
        const uint32_t n_gld = (k + 2u * LPQ - 1u) / (2u * LPQ);   // lane (query, g) reads slot pairs 2 (g + LPQ i), + 1
        auto refresh_issue = [&](uint32_t half) {
            if constexpr (E != 1 && FULLQ) g_q = pw * 16u + 8u * (half & 1u) + lane / LPQ;
            const uint64_t *G = reinterpret_cast<const uint64_t *>(a.gslots + g_q * GSTRIDE) + g_g;
#pragma unroll
            for (uint32_t i = 0; i < NG; i++)
                if (i < n_gld) gv[i] = __hip_atomic_load(G + LPQ * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto refresh_apply = [&]() {
            uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t i = 0; i < NG; i++) {
                const uint32_t s0 = 2u * (g_g + LPQ * i), lo = (uint32_t)gv[i], hi = (uint32_t)(gv[i] >> 32);
                if (i < n_gld && s0 < k) mn = lo < mn ? lo : mn;
                if (i < n_gld && s0 + 1u < k) mn = hi < mn ? hi : mn;
            }
#pragma unroll
            for (uint32_t x = 1; x < LPQ; x <<= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)mn, (int)x, 64);
                mn = o < mn ? o : mn;
            }
            if (g_g == 0u && g_q < QC && mn != 0u && mn != 0xFFFFFFFFu) {   // every slot filled
                const float sm = __uint_as_float(mn);
                const float t = sm * sm * (1.0f - 1.0e-4f);
                if (t > c_tsq[g_q]) c_tsq[g_q] = t;
            }
        };
        auto producer_compact = [&]() {
            uint32_t pubv = 0, shrv = 1;
            const bool mine_q = lane < 16u && own_grp * 16u < QC && (!SHARED || (lane & 1u) == own_par);
            if (mine_q) { pubv = c_pub[own_grp * 16u + lane]; shrv = c_shr[own_grp * 16u + lane]; }
            uint64_t need = __ballot(mine_q && shrv == 0u && pubv + 32u >= capq && pubv > k);
            while (need) {
                const int l = __ffsll((unsigned long long)need) - 1;
                need &= need - 1;
                if constexpr (DIAG) { n_pcompact++; cstamp(5); }
                const uint32_t qs = own_grp * 16u + (uint32_t)l;
                uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)pubv, l);
                n = n < 64u * E ? n : 64u * E;
                const float qq_of = c_qq[qs];
                uint32_t *rws = c_rows + qs * capq;
                float *dts = c_dots + qs * capq;
                float *nms = c_nrm + qs * capq;
                // entry 64 e + lane of the list sits in slot e of this lane
                bool valid[E];
                uint32_t r0[E], ord[E]; float d0[E], n0[E], sim[E];
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const uint32_t idx = lane + 64u * e;
                    valid[e] = idx < n;
                    r0[e] = 0; ord[e] = 0; d0[e] = 0.0f; n0[e] = 1.0f; sim[e] = 0.0f;
                    if (valid[e]) {
                        r0[e] = rws[idx]; d0[e] = dts[idx]; n0[e] = E == 1 ? nms[idx] : a.norms[r0[e]];
                        sim[e] = cosine_from_sums(d0[e], qq_of, n0[e]);
                        ord[e] = score_ord(score_of(distance_of(sim[e])));
                    }
                }
                if constexpr (DIAG) asm volatile("s_nop 0" :: "v"(ord[0]), "v"(ord[E - 1]));
                cstamp(0);   // entries read, norms gathered, cosines formed
                auto count_ge = [&](uint32_t t) {
                    uint32_t c = 0;
#pragma unroll
                    for (int e = 0; e < E; e++) c += (uint32_t)__popcll(__ballot(valid[e] && ord[e] >= t));
                    return c;
                };
                // radix select of the k-th largest score order, most significant bit first.  Scores are <= 1.0, so
                // bits 31..30 are clear unless something odd is in the list; the walk stops as soon as exactly k
                // entries lie at or above the prefix (then they ARE the top k and no tie can straddle the cut)
                // (every value of the walk is wave-uniform; the readfirstlanes say so to the compiler, which otherwise
                // keeps the bit index and the prefix in VGPRs: 24 instructions per bit instead of 8)
                uint32_t T = 0;
                int b = __builtin_amdgcn_readfirstlane(count_ge(0x40000000u) ? 31 : 29);
                uint32_t c_at_T = n;   // entries at or above the prefix T (T = 0: all of them)
#pragma unroll 1
                while (b >= 0 && c_at_T != k) {
                    const uint32_t candT = T | (1u << b);
                    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)count_ge(candT));
                    T = (uint32_t)__builtin_amdgcn_readfirstlane((int)(c >= k ? candT : T));
                    c_at_T = (uint32_t)__builtin_amdgcn_readfirstlane((int)(c >= k ? c : c_at_T));
                    b = __builtin_amdgcn_readfirstlane(b - 1);
                }
                const bool exact_k = c_at_T == k;
                cstamp(1);   // radix walk
                uint64_t gt[E], eq[E], keep[E];
                uint32_t n_gt = 0, n_eq = 0;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    gt[e] = __ballot(valid[e] && ord[e] > T);
                    eq[e] = __ballot(valid[e] && ord[e] == T);
                    n_gt += (uint32_t)__popcll(gt[e]);
                    n_eq += (uint32_t)__popcll(eq[e]);
                    keep[e] = gt[e] | eq[e];
                }
                if (!exact_k && n_eq != k - n_gt) {   // equal scores straddle the cut: lower rows win
                    if constexpr (DIAG) n_tie++;
                    const uint32_t want_eq = k - n_gt;
                    if (n_eq > 24u) {
                        // many equal scores (a wide list's first compaction: half of the first rows have a non-positive
                        // cosine, i.e. score 0): the want_eq lowest rows among them by a radix walk over the row index —
                        // rows are unique, so exactly want_eq survive — instead of ranking each against all the others
                        uint32_t R = 0;
#pragma unroll 1
                        for (int rb = 31 - __builtin_clz(n_rows | 1u); rb >= 0; rb--) {
                            const uint32_t cand = R | (1u << rb);
                            uint32_t c = 0;
#pragma unroll
                            for (int e = 0; e < E; e++) c += (uint32_t)__popcll(__ballot(valid[e] && ord[e] == T && r0[e] < cand));
                            R = (uint32_t)__builtin_amdgcn_readfirstlane((int)(c < want_eq ? cand : R));
                        }
#pragma unroll
                        for (int e = 0; e < E; e++) keep[e] = gt[e] | __ballot(valid[e] && ord[e] == T && r0[e] <= R);
                    } else {
                        uint32_t rk[E];
#pragma unroll
                        for (int e = 0; e < E; e++) rk[e] = 0;
#pragma unroll
                        for (int f = 0; f < E; f++)
                            for (uint64_t m = eq[f]; m; m &= m - 1) {
                                const uint32_t ro = (uint32_t)__builtin_amdgcn_readlane((int)r0[f], __ffsll((unsigned long long)m) - 1);
#pragma unroll
                                for (int e = 0; e < E; e++) rk[e] += ro < r0[e] ? 1u : 0u;
                            }
#pragma unroll
                        for (int e = 0; e < E; e++) keep[e] = gt[e] | __ballot(valid[e] && ord[e] == T && rk[e] < want_eq);
                    }
                }
                // the k-th best = the smallest kept score order, and among equal ones the largest row
                cstamp(2);   // keep masks and ties
                uint32_t mine = 0xFFFFFFFFu;
#pragma unroll
                for (int e = 0; e < E; e++) mine = (((keep[e] >> lane) & 1ull) && ord[e] < mine) ? ord[e] : mine;
                const uint32_t mn = wave_min_u32(mine);
                uint32_t trow = 0; float tsim = 0.0f;
#pragma unroll
                for (int e = 0; e < E; e++)
                    for (uint64_t m = __ballot(((keep[e] >> lane) & 1ull) && ord[e] == mn); m; m &= m - 1) {
                        const int le = __ffsll((unsigned long long)m) - 1;
                        const uint32_t ro = (uint32_t)__builtin_amdgcn_readlane((int)r0[e], le);
                        if (ro >= trow) { trow = ro; tsim = readlane_f32(sim[e], le); }
                    }
                cstamp(4);   // k-th entry (wave min, its row)
                // survivors to [0, k): rank in (slot, lane) order <= the entry's old index, and every lane read first
                uint32_t base = 0;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    if ((keep[e] >> lane) & 1ull) {
                        const uint32_t slot = base + (uint32_t)__popcll(keep[e] & ((1ull << lane) - 1ull));
                        rws[slot] = r0[e]; dts[slot] = d0[e];
                        if constexpr (E == 1) nms[slot] = n0[e];
                    }
                    base += (uint32_t)__popcll(keep[e]);
                }
#pragma unroll
                for (int e = 0; e < E; e++)
                    if (((keep[e] >> lane) & 1ull) && sim[e] > 0.0f)
                        __hip_atomic_fetch_max(a.gslots + qs * GSTRIDE + ((r0[e] * 2654435761u) >> 16) % k, __float_as_uint(sim[e]),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cstamp(3);   // survivors written, slot atomics issued
                if (lane == 0) {
                    c_tau[qs] = cand_key(trow, tsim);
                    // k-th best score is 0 (a clamped non-positive cosine): every later row with dot <= 0 ties with it
                    // at a higher row and loses, so from here on only positive dots are candidates (t = 0); without
                    // this the list kept refilling with score-0 rows and every compaction ran the tie rule over ~100 of
                    // them.  A NaN k-th score (fewer than k comparable rows yet) leaves the door open (t = -1).
                    const float t = tsim > 0.0f ? tsim * tsim * (1.0f - 1.0e-4f) : (tsim == tsim ? 0.0f : -1.0f);
                    if (t > c_tsq[qs]) c_tsq[qs] = t;
                    c_shr[qs] = n;
                }
            }
        };
        uint32_t tile = blockIdx.x;
        if (tile < n_tiles) {
            issue_loads(ldA, nrA, tile);
            write_tile(ldA, nrA, 0, tile + 2u * gridDim.x);
            issue_loads(ldB, nrB, tile + gridDim.x);   // after tile 0 is out: the first tile does not share the start-up burst with the second
        }
        __syncthreads();
        stamp0();
        uint32_t buf = 0, it = 0;
        // one tile interval: the consumers work on `tile`; this wave writes tile + grid (held in `ld`) into the
        // other buffer and sends `ld` for tile + 3 grid
        auto step = [&](f32x4 (&ld)[C::LOADS], f32x4 &nr) {
            const uint32_t next = tile + gridDim.x;
            // the slots are re-read every 4th tile (agent-scope loads go past the L2) and applied one tile later
            if ((it & 3u) == 1u && !(a.arm & 16u)) refresh_apply();
            // the slot loads go out BEFORE this step's tile reloads: waiting for them next step then leaves the
            // reloads in flight (vmcnt counts in issue order)
            if ((it & 3u) == 0u && !(a.arm & 16u)) refresh_issue(it >> 2);
            if constexpr (NBUF == 2) {
                if (next < n_tiles) {
                    write_tile(ld, nr, buf ^ 1u, next + 2u * gridDim.x);     // buffer last read one tile ago, behind that tile's barrier
                    stamp(t_write);
                }
                it++;
                if (!(a.arm & 8u)) producer_compact();
                stamp(t_stage);
                tile_barrier();
                stamp(t_bar);
                buf ^= 1u;
            } else {
                it++;
                if (!(a.arm & 8u)) producer_compact();          // while the consumers work on `tile`
                stamp(t_stage);
                tile_barrier();              // the consumers are done with the buffer
                stamp(t_bar);
                if (next < n_tiles) {
                    write_tile(ld, nr, 0u, next + 2u * gridDim.x);
                    stamp(t_write);
                }
                tile_barrier();              // the next tile is in place
                stamp(t_bar);
            }
            tile += gridDim.x;
        };
        // tiles are taken two per loop iteration, always both (a step past the last tile only keeps the barrier; the
        // consumers do the same): one back edge, each register set written in one place
        const uint32_t my_tiles = tile < n_tiles ? (n_tiles - 1u - tile) / gridDim.x + 1u : 0u;
        for (uint32_t i = 0; i < my_tiles; i += 2u) {
            step(ldB, nrB);
            step(ldA, nrA);
        }
        if constexpr (DIAG) {
            if (lane == 0) {
                unsigned long long *o = a.diag + ((size_t)blockIdx.x * 8 + wave) * 8;
                o[0] = t_stage; o[3] = t_bar; o[5] = t_wait; o[6] = t_write; o[7] = n_pcompact;
                if (blockIdx.x == 7 && wave == 5 && n_pcompact) printf("[batch2 diag] block 7 producer 1: %llu compactions; cycles each: load+gather+cos %llu, radix %llu, keep masks+ties %llu, k-th entry %llu, scatter+atomics %llu; tie branches %llu\n", n_pcompact, cp[0] / n_pcompact, cp[1] / n_pcompact, cp[2] / n_pcompact, cp[4] / n_pcompact, cp[3] / n_pcompact, n_tie);
            }
        }
        __syncthreads();   // the consumers have slid their last appends into place
        finalize_lists();
        return;
    }

    // ---------------------------------------------------------------------- consumer
    // consumers are the critical path (producers wait >4k cycles per tile at the barrier): they win the
    // VALU-issue arbitration on the SIMD they share with a producer
    __builtin_amdgcn_s_setprio(2);
    const uint32_t j = lane & 15u, kq = lane >> 4;
    const uint32_t qslot = pw * 16u + j;
    // queries -> split registers: step ks covers k = 32 ks .. 32 ks + 31; lane (j, kq) holds q_j[32 ks + 8 kq + e]
    constexpr int KS = D / 32;
    s16x8 qh[KS], ql[KS];
    float qq = 0.0f;
    {
        const bool live = qslot < a.nq;
        const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.queries + (size_t)(live ? qslot : 0) * D);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            f32x4 v0 = q4[8 * ks + 2 * kq], v1 = q4[8 * ks + 2 * kq + 1];
            if (!live) { v0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; v1 = v0; }
            qq += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w + v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
            bf16x4_t h0, l0, h1, l1;
            split4(v0, h0, l0);
            split4(v1, h1, l1);
            const bf16x8_t H = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8_t L = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            qh[ks] = __builtin_bit_cast(s16x8, H);
            ql[ks] = __builtin_bit_cast(s16x8, L);
        }
    }
    qq += __shfl_xor(qq, 16, 64);
    qq += __shfl_xor(qq, 32, 64);

    // A-operand reads: K-step ks of the tile is [hi fragment | lo fragment], a lane's 16 bytes at 16 * lane
    const uint32_t a_off = lane * 16u;

    unsigned long long n_compact = 0, n_append_steps = 0;   // n_compact: kept for the diag record layout
    // behind a barrier: slide what this wave appended since a producer's compaction snapshot down to k
    auto apply_shrink = [&]() {
        const uint32_t shr = c_shr[qslot];
        if (__ballot(shr != 0u)) {
            if (shr != 0u) {
                const uint32_t cnt = c_cnt[qslot], m = cnt - shr;   // m <= 16 and k + 16 <= shr: no overlap
                for (uint32_t e = kq; e < m; e += 4u) {
                    const uint32_t src = qslot * capq + shr + e, dst = qslot * capq + k + e;
                    c_rows[dst] = c_rows[src]; c_dots[dst] = c_dots[src];
                    if constexpr (E == 1) c_nrm[dst] = c_nrm[src];
                }
                if (kq == 0u) { c_cnt[qslot] = k + m; c_pub[qslot] = k + m; c_shr[qslot] = 0u; }
            }
        }
    };
    if (kq == 0u) c_qq[qslot] = qq;
    unsigned long long t_q = 0;
    if constexpr (DIAG) { asm volatile("s_nop 0" :: "v"(qh[KS - 1]), "v"(ql[KS - 1]), "v"(qq)); t_q = __builtin_readcyclecounter() - t_begin; }

    __syncthreads();   // tile 0 is in buffer 0
    unsigned long long t_pro = 0;
    if constexpr (DIAG) t_pro = __builtin_readcyclecounter() - t_begin;
    stamp0();
    uint32_t buf = 0;
    const bool wave_dead = pw * 16u >= QC;   // wide mode: consumer waves 2, 3 own no queries; they only keep the barriers
    const uint32_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - 1u - blockIdx.x) / gridDim.x + 1u : 0u;
    const uint32_t my_steps = (my_tiles + 1u) & ~1u;   // the producers step in pairs
    for (uint32_t st = 0, tile = blockIdx.x; st < my_steps; st++, tile += gridDim.x) {
        if (wave_dead || tile >= n_tiles || (a.arm & 4u)) { tile_barrier(); if constexpr (NBUF == 2) buf ^= 1u; else tile_barrier(); continue; }
        if (!(a.arm & 2u)) apply_shrink();
        stamp(t_stage);
        const char *Thi = tiles + buf * C::TILE_BYTES + a_off, *Tlo = Thi + SPLIT_FRAG_BYTES;
        // epilogue operands are read now, under the MFMA loop, not after it
        const float tsq = c_tsq[qslot];
        const f32x4 rr4 = *reinterpret_cast<const f32x4 *>(c_rr + buf * BT_ROWS + 4u * kq);
        // (three independent accumulator chains instead of one 36- / 72-deep one: 5 % slower at 384-d, again in round 3 —
        // the loop is bound by its LDS reads, 4 bytes read per tile byte, not by the MFMA dependency)
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        constexpr int CH = 2, NCH = KS / CH;
        static_assert(KS % CH == 0, "dim/32 must be even");
        auto rdh = [&](int ks) { return *reinterpret_cast<const s16x8 *>(Thi + (uint32_t)ks * SPLIT_STEP_BYTES); };
        auto rdl = [&](int ks) { return *reinterpret_cast<const s16x8 *>(Tlo + (uint32_t)ks * SPLIT_STEP_BYTES); };
        s16x8 ha[CH], la[CH], hb[CH], lb[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) { ha[u] = rdh(u); la[u] = rdl(u); }
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            s16x8 *ch = (c & 1) ? hb : ha, *cl = (c & 1) ? lb : la, *nh = (c & 1) ? ha : hb, *nl = (c & 1) ? la : lb;
            if (c + 1 < NCH) {
#pragma unroll
                for (int u = 0; u < CH; u++) { nh[u] = rdh((c + 1) * CH + u); nl[u] = rdl((c + 1) * CH + u); }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < CH; u++) {
                const int ks = c * CH + u;
                // x.y = xh.yh + xh.yl + xl.yh (+ xl.yl, left out: <= 2^-18 of a term, below the 2^-17 the two-piece
                // split itself leaves behind; measured |cos error| 9e-7 against 6e-7 with it — tuning.md)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cl[u], qh[ks], acc, 0, 0, 0);   // small terms first
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[u], ql[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[u], qh[ks], acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(t_mfma);
        // single tile buffer: the tile is no longer needed once the MFMA loop has read it, so the hand-back barrier
        // comes BEFORE the epilogue and the producers write the next tile while the consumers test and append (the
        // epilogue touches only registers and the candidate lists)
        if constexpr (NBUF == 1) tile_barrier();

        // epilogue: lane (j, kq) holds rows 4 kq + r of query j.  All four tests first, one wave-level branch:
        // after warm-up no lane has a survivor and the wave falls through
        const uint32_t row0 = tile * BT_ROWS;
        const float tq = tsq * qq;
        const bool live = qslot < a.nq;
        uint32_t mask = 0;
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) {
            const float rr = rr4[r], dot = acc[r];
            // bitwise, not short-circuit: hipcc turns || and && on float compares into a chain of divergent
            // branches (1.2k cycles per tile, measured); these are four v_cmp and a few s_and/s_or
            const bool beats = (dot > 0.0f) & (dot * dot >= tq * rr);
            const bool odd = (dot != dot) | (rr != rr);
            const bool maybe = (tsq < 0.0f) | beats | odd;
            mask |= (maybe & live & (row0 + 4u * kq + r < n_rows)) ? (1u << r) : 0u;
        }
        if (a.arm & 1u) mask = 0u;
        if (__ballot(mask != 0u)) {
            if constexpr (DIAG) n_append_steps++;
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) {
                const uint32_t row = row0 + 4u * kq + r;
                if (((mask >> r) & 1u) && row_passes(a.flt, row)) {
                    const uint32_t slot = atomicAdd(&c_cnt[qslot], 1u);
                    if (slot < capq) {
                        c_rows[qslot * capq + slot] = row; c_dots[qslot * capq + slot] = acc[r];
                        if constexpr (E == 1) c_nrm[qslot * capq + slot] = rr4[r];
                    }
                }
            }
            // entries are written before the length the producers read (LDS operations of one wave stay in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (kq == 0u) c_pub[qslot] = c_cnt[qslot];
        }
        stamp(t_epi);
        tile_barrier();        // two buffers: the tile's only barrier; single buffer: the next tile is in place
        if constexpr (NBUF == 2) buf ^= 1u;
        stamp(t_bar);
    }
    apply_shrink();
    if constexpr (DIAG) {
        if (lane == 0) {
            unsigned long long *o = a.diag + ((size_t)blockIdx.x * 8 + wave) * 8;
            o[0] = t_stage; o[1] = t_mfma; o[2] = t_epi; o[3] = t_bar; o[4] = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
            o[5] = n_compact; o[6] = n_append_steps;
        }
    }
    __syncthreads();   // every wave's lists are final
    unsigned long long t_f0 = 0;
    if constexpr (DIAG) t_f0 = __builtin_readcyclecounter();
    finalize_lists();
    if constexpr (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_f1 = __builtin_readcyclecounter();
        if (lane == 0 && blockIdx.x == 7 && wave == 1)
            printf("[batch2 diag] block 7 wave 1: queries in registers %llu, prologue %llu cycles, tiles %llu, finalize %llu, whole kernel %llu\n", t_q, t_pro,
                   t_f0 - t_begin - t_pro, t_f1 - t_f0, t_f1 - t_begin);
    }
}

// rows [row_lo, row_hi) of the f32 store -> the split store in its fragment-major layout (batch_common.hpp) — byte for
// byte what the consumers read from LDS.  One wave per row.
template <int D>
__global__ __launch_bounds__(256) void build_split_kernel(const float *rows, char *split, uint32_t row_lo, uint32_t row_hi) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t r = row_lo + wave; r < row_hi; r += n_waves) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(rows + (size_t)r * D);
        char *tile = split + (size_t)(r / BT_ROWS) * split_tile_bytes(D);
        const uint32_t i = r % BT_ROWS;
        for (uint32_t c4 = lane; c4 < (uint32_t)D / 4u; c4 += 64u) {
            bf16x4_t h, l;
            split4(src[c4], h, l);
            const uint32_t o = split_piece_off(i, c4 * 4u);
            *reinterpret_cast<bf16x4_t *>(tile + o) = h;
            *reinterpret_cast<bf16x4_t *>(tile + o + SPLIT_FRAG_BYTES) = l;
        }
    }
}

int launch_build_split(const float *rows, char *split, uint32_t row_lo, uint32_t row_hi, uint32_t dim, hipStream_t stream) {
    if (row_hi <= row_lo) return CX_OK;
    const uint32_t n = row_hi - row_lo;
    const uint32_t blocks = n / 4u + 1u < 8192u ? n / 4u + 1u : 8192u;
    if (dim == 384) hipLaunchKernelGGL((build_split_kernel<384>), dim3(blocks), dim3(256), 0, stream, rows, split, row_lo, row_hi);
    else if (dim == 768) hipLaunchKernelGGL((build_split_kernel<768>), dim3(blocks), dim3(256), 0, stream, rows, split, row_lo, row_hi);
    else return set_err(CX_ERR_VALIDATION, "split store: dim %u not supported", dim);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

static uint32_t batch_cus() { return device_cus(); }
uint32_t batch_grid_blocks(uint32_t n_rows) {
    const uint32_t cus = batch_cus();
    const uint32_t tiles = n_rows / BT_ROWS + (n_rows % BT_ROWS ? 1u : 0u);
    return tiles < cus ? (tiles ? tiles : 1u) : cus;
}

bool batch_supported(uint32_t dim, uint32_t k) { return (dim == 384 || dim == 768) && k >= 1 && k <= BATCH_K_WIDE; }
// k above which the wide lists are used
static bool batch_wide_k(uint32_t k) {
    static const int wide_min = getenv("CX_BATCH_WIDE_MIN_K") ? atoi(getenv("CX_BATCH_WIDE_MIN_K")) : 33;
    return (int)k >= wide_min;
}
uint32_t batch_queries_per_pass(uint32_t dim, uint32_t k, uint64_t nq) {
    static const int full = getenv("CX_BATCH_WIDE_FULL") ? atoi(getenv("CX_BATCH_WIDE_FULL")) : 1;   // 0: wide lists always 32 queries per pass
    return (!batch_wide_k(k) || (full && nq > 32) || full == 2) ? 64u : 32u;   // 2: the 64-query wide mode for any query count
}

// launch shape for nq queries over n_rows rows: chunks x groups blocks.  A block should see >= 256 tiles (4096
// rows) when the corpus allows; the CUs that leaves free take further query groups in the same launch.
void batch_launch_shape(uint32_t n_rows, uint32_t dim, uint64_t nq, uint32_t k, uint32_t *chunks, uint32_t *groups) {
    const uint32_t cus = batch_cus(), qpp = batch_queries_per_pass(dim, k, nq);
    const uint32_t tiles = (n_rows + BT_ROWS - 1) / BT_ROWS;
    uint32_t c = tiles / 256u;
    if (c < 1u) c = 1u;
    if (c * 2u > cus) c = cus;                 // large corpus: one group on every CU
    const uint64_t g_all = (nq + qpp - 1) / qpp;
    uint64_t g = cus / c;
    if (g < 1) g = 1;
    if (g > g_all) g = g_all;
    if ((uint64_t)c * g < cus) c = (uint32_t)(cus / g);   // too few groups to fill the chip: more, smaller chunks
    if (c > tiles) c = tiles ? tiles : 1u;
    static const int one = getenv("CX_BATCH_ONE_GROUP") ? atoi(getenv("CX_BATCH_ONE_GROUP")) : 0;
    if (one) { c = batch_grid_blocks(n_rows); g = 1; }
    *chunks = c;
    *groups = (uint32_t)g;
}

template <int D>
static int launch_batch_d(BatchArgs a, uint32_t grid, hipStream_t stream) {
    const bool wide = batch_wide_k(a.k);   // 32 queries per pass, lists of up to 272, four entries per lane in a compaction
    const size_t qc = a.qpp;
    const size_t tail = qc * a.capq * (wide ? 8 : 12) + BT_Q * 8 + BT_Q * 4 + BT_Q * 4 + 3 * BT_Q * 4;
    const size_t nbuf = (wide && a.qpp == 64u) ? 1 : 2;   // batch2_kernel: NBUF
    const size_t lds = nbuf * (size_t)Batch2Cfg<D>::TILE_BYTES + 2 * BT_ROWS * 4 + tail;
    if (lds > 160 * 1024) return set_err(CX_ERR_VALIDATION, "batch scan: %zu bytes of LDS for k = %u", lds, a.k);
    static std::atomic<uint64_t> attr_devices{0};
    if (first_use_on_device(attr_devices)) {
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batch2_kernel<D, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batch2_kernel<D, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batch2_kernel<D, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batch2_kernel<D, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batch2_kernel<D, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batch2_kernel<D, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (getenv("CX_BATCH_DIAG")) {  // diagnostic build: per-phase cycle shares on stderr, results still valid
        const size_t n = (size_t)grid * a.n_groups * 8 * 8;   // up to 8 waves x 8 slots per block
        CX_HIP(hipMalloc((void **)&a.diag, n * 8));
        CX_HIP(hipMemset(a.diag, 0, n * 8));
        if (wide && a.qpp == 64u) hipLaunchKernelGGL((batch2_kernel<D, true, 8>), dim3(grid, a.n_groups), dim3(512), lds, stream, a);
        else if (wide) hipLaunchKernelGGL((batch2_kernel<D, true, 4>), dim3(grid, a.n_groups), dim3(512), lds, stream, a);
        else hipLaunchKernelGGL((batch2_kernel<D, true, 1>), dim3(grid, a.n_groups), dim3(512), lds, stream, a);
        CX_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h(n);
        CX_HIP(hipMemcpy(h.data(), a.diag, n * 8, hipMemcpyDeviceToHost));
        CX_HIP(hipFree(a.diag));
        double s[7] = {0, 0, 0, 0, 0, 0, 0}, tiles = 0;
        double pbar = 0, ncomp = 0, nappend = 0;
        for (size_t w = 0; w < (size_t)grid * a.n_groups * 8; w++) {
            const bool cons = (w % 8) < 4;
            if (cons) { for (int p = 0; p < 4; p++) s[p] += (double)h[w * 8 + p]; tiles += (double)h[w * 8 + 4]; ncomp += (double)h[w * 8 + 5]; nappend += (double)h[w * 8 + 6]; }
            else { pbar += (double)h[w * 8 + 3]; s[5] += (double)h[w * 8 + 5]; s[6] += (double)h[w * 8 + 6]; ncomp += (double)h[w * 8 + 7]; s[4] += (double)h[w * 8 + 0]; }
        }
        fprintf(stderr, "[batch diag] producer barrier wait per tile %.0f; per consumer wave-tile: compactions %.3f, tiles with appends %.3f\n", pbar / (tiles > 0 ? tiles : 1), ncomp / (tiles > 0 ? tiles : 1), nappend / (tiles > 0 ? tiles : 1));
        fprintf(stderr, "[batch diag] cycles per tile per wave: stage+check %.0f  mfma-loop %.0f  epilogue %.0f  barrier %.0f"
                        "  (producer: load wait %.0f, split+write %.0f, compaction %.0f)\n",
                s[0] / tiles, s[1] / tiles, s[2] / tiles, s[3] / tiles, s[5] / tiles, s[6] / tiles, s[4] / tiles);
        return CX_OK;
    }
    if (wide && a.qpp == 64u) hipLaunchKernelGGL((batch2_kernel<D, false, 8>), dim3(grid, a.n_groups), dim3(512), lds, stream, a);
    else if (wide) hipLaunchKernelGGL((batch2_kernel<D, false, 4>), dim3(grid, a.n_groups), dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((batch2_kernel<D, false, 1>), dim3(grid, a.n_groups), dim3(512), lds, stream, a);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

int launch_batch_scan(BatchArgs a, uint32_t grid, hipStream_t stream) {
    if (!batch_supported(a.dim, a.k)) return set_err(CX_ERR_VALIDATION, "batch scan: unsupported dim %u / k %u", a.dim, a.k);
    if (a.n_groups == 0) a.n_groups = 1;
    static const uint32_t arm_env = getenv("CX_BATCH_ARM") ? (uint32_t)atoi(getenv("CX_BATCH_ARM")) : 0u;
    a.arm = arm_env;
    const uint32_t qpp = a.qpp;
    if (qpp != 64u && !(qpp == 32u && batch_wide_k(a.k))) return set_err(CX_ERR_VALIDATION, "batch scan: %u queries per pass for k = %u", qpp, a.k);
    if (a.nq == 0 || a.nq > qpp * a.n_groups || a.nq <= qpp * (a.n_groups - 1))
        return set_err(CX_ERR_VALIDATION, "batch scan: %u queries do not fill %u groups of %u (k = %u)", a.nq, a.n_groups, qpp, a.k);
    // batch2: lists are compacted at capq - 32 entries (<= 64: one per lane) and never exceed capq; k + 16 <= capq - 32
    a.capq = 80u;
    if (batch_wide_k(a.k)) {   // wide: as long as LDS allows (a compaction absorbs capq - 32 - k new entries), at most 4 x 64 + 16
        const size_t tiles = (qpp == 64u ? 1 : 2) * (a.dim == 384 ? (size_t)Batch2Cfg<384>::TILE_BYTES : (size_t)Batch2Cfg<768>::TILE_BYTES);
        const size_t room = 160 * 1024 - tiles - 2 * BT_ROWS * 4 - (BT_Q * 8 + BT_Q * 4 + BT_Q * 4 + 3 * BT_Q * 4);
        uint32_t c = (uint32_t)(room / (qpp * 8)) & ~7u;
        a.capq = c > 272u ? 272u : c;
    }
    if (a.dim == 384) return launch_batch_d<384>(a, grid, stream);
    return launch_batch_d<768>(a, grid, stream);
}

}  // namespace cx
