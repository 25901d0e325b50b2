// autolink.cpp — C ABI of the auto-linker's batched similarity pass and the dedup scan.
// Orchestrates allpairs.hip: shadow -> MFMA filter -> exact rescore -> link rules.
#include "internal.hpp"

namespace {

using namespace cx;

constexpr uint32_t CHUNK_ROWS = 1u << 20;  // scanned rows per filter launch (bounds the candidate scratch: 2 GiB at cap 512)

uint32_t cand_cap() {  // candidate slots per scanned row; rows that overflow are redone on the exact scan path
    const char *e = getenv("CX_PAIR_CAND_CAP");
    int v = e ? atoi(e) : 512;
    return (uint32_t)(v < 16 ? 16 : v);
}

// bf16 error bound of the filter: both operands are rounded to bf16 (unit roundoff u = 2^-8), so
// |x^.y^ - cos| <= (2u + u^2) * sum|x_i y_i| <= 2u + u^2 for unit rows; plus f32 accumulation slack.
constexpr float FILTER_EPS = 2.0f / 256.0f + 1.0f / 65536.0f + 1.0e-4f;

}  // namespace

namespace cx {
int ensure_shadow(const cx_index *ix, hipStream_t s) {
    std::lock_guard<std::mutex> g(ix->shadow_mu);
    const uint64_t n = ix->n_rows;
    const bool tiled = ix->dim % 32 == 0;
    bool work = false;
    // ONE copy of the shadow: the tiled layout every filter kernel reads when dim % 32 == 0 (kernels.hpp:
    // tiled_shadow_off), the row-major one otherwise.  (Rounds 1-2 kept both: 12.8 GB more per 6.25M x 1024 shard.)
    if (ix->shadow_cap < n) {
        if (ix->d_shadow) CX_HIP(hipFree(ix->d_shadow));
        if (ix->d_shadow_t) CX_HIP(hipFree(ix->d_shadow_t));
        ix->d_shadow = nullptr;
        ix->d_shadow_t = nullptr;
        ix->shadow_cap = 0;
        const uint64_t cap = std::max<uint64_t>(n, ix->cap);
        if (tiled) {   // whole 256-row tiles (the filter kernels read their last panel unclamped), zero beyond the last row
            const size_t bytes = (size_t)((cap + 255) / 256) * 256 * ix->dim * sizeof(uint16_t);
            CX_HIP(hipMalloc((void **)&ix->d_shadow_t, bytes));
            CX_HIP(hipMemsetAsync(ix->d_shadow_t, 0, bytes, s));
        } else {
            CX_HIP(hipMalloc((void **)&ix->d_shadow, cap * ix->dim * sizeof(uint16_t) + 64));
        }
        ix->shadow_cap = cap;
        ix->shadow_rows = 0;
        ix->shadow_stale.clear();
        work = true;
    }
    if (!ix->d_shadow_err) CX_HIP(hipMalloc((void **)&ix->d_shadow_err, 64));
    if (!ix->d_irr_rows) CX_HIP(hipMalloc((void **)&ix->d_irr_rows, BS_IRR_CAP * sizeof(uint32_t)));
    if (ix->shadow_rows == 0) {   // built from scratch: the error bound and the irregular list start over
        CX_HIP(hipMemsetAsync(ix->d_shadow_err, 0, 64, s));
        ix->irr_n = 0;
        ix->irr_over = false;
    }
    auto build = [&](uint32_t lo, uint32_t hi) -> int {
        return launch_build_shadow_index(ix->rows32(), ix->rows16(), tiled ? ix->d_shadow_t : ix->d_shadow, tiled, lo, hi, ix->dim, s,
                                         ix->d_shadow_err, ix->d_shadow_err + 1, ix->d_irr_rows, ix->irr_n);
    };
    std::sort(ix->shadow_stale.begin(), ix->shadow_stale.end());   // (a row upserted twice is rebuilt once: the irregular list's de-duplication relies on it)
    ix->shadow_stale.erase(std::unique(ix->shadow_stale.begin(), ix->shadow_stale.end()), ix->shadow_stale.end());
    for (uint32_t r : ix->shadow_stale)
        if (r < ix->shadow_rows) {
            if (int rc = build(r, r + 1)) return rc;
            work = true;
        }
    ix->shadow_stale.clear();
    if (ix->shadow_rows < n) {
        if (int rc = build((uint32_t)ix->shadow_rows, (uint32_t)n)) return rc;
        ix->shadow_rows = n;
        work = true;
    }
    if (work) {
        uint32_t cnt = 0;
        CX_HIP(hipMemcpyAsync(&cnt, ix->d_shadow_err + 1, sizeof cnt, hipMemcpyDeviceToHost, s));
        CX_HIP(hipStreamSynchronize(s));   // (every batched search comes through here: no host wait when nothing changed)
        ix->irr_n = cnt < BS_IRR_CAP ? cnt : BS_IRR_CAP;
        ix->irr_over = cnt > BS_IRR_CAP;
    }
    return CX_OK;
}
}  // namespace cx

namespace {

// Scratch of a pass.  It lives in the pooled Ctx (grow-only), so a steady stream of passes does no
// hipMalloc/hipFree (measured: 1.4 ms of a 11.4 ms pass when allocated per call).
struct RedoScratch {   // exact-path redo of a pass: gathered vectors and their contiguous lists
    float *d_vec = nullptr, *d_scores = nullptr, *d_dists = nullptr;
    uint32_t *d_src = nullptr, *d_pos = nullptr, *d_rows = nullptr, *d_cnt = nullptr, *d_rows2 = nullptr, *d_cnt2 = nullptr, *d_of = nullptr;
    uint64_t *d_src_off = nullptr;                                     // the dedup pass's dense rows: their pairs staged for the splice
    uint32_t *d_from_row = nullptr, *d_to_h = nullptr;
    float *d_w_h = nullptr;
    size_t c_vec = 0, c_scores = 0, c_dists = 0, c_src = 0, c_pos = 0, c_rows = 0, c_cnt = 0, c_rows2 = 0, c_cnt2 = 0, c_of = 0;
    size_t c_src_off = 0, c_from_row = 0, c_to_h = 0, c_w_h = 0;
    ~RedoScratch() {
        (void)hipFree(d_vec); (void)hipFree(d_scores); (void)hipFree(d_dists); (void)hipFree(d_src); (void)hipFree(d_pos);
        (void)hipFree(d_rows); (void)hipFree(d_cnt); (void)hipFree(d_rows2); (void)hipFree(d_cnt2); (void)hipFree(d_of);
        (void)hipFree(d_src_off); (void)hipFree(d_from_row); (void)hipFree(d_to_h); (void)hipFree(d_w_h);
    }
};

struct PassScratch {
    RedoScratch redo;
    uint32_t *d_scan = nullptr, *d_cand_cnt = nullptr, *d_cand = nullptr, *d_overflow = nullptr;
    uint32_t *d_list_rows = nullptr, *d_list_cnt = nullptr, *d_counts = nullptr, *d_ident = nullptr;
    float *d_list_scores = nullptr, *d_list_dists = nullptr, *d_pair_sims = nullptr;
    size_t c_pair_sims = 0;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;   // around the filter GEMM kernel alone (timed passes)
    uint64_t *d_pairs = nullptr;       // persistent filter kernel: hits as (i | j << 32) pairs, before pair_scatter_kernel
    uint32_t *d_pair_ctl = nullptr;    // [16]: pairs written, pairs lost, per-XCD tile tickets
    uint32_t *d_irr_ok = nullptr;      // [BS_IRR_CAP]: which of the index's irregular rows are irregular now (launch_irr_append)
    uint16_t *d_stage_t = nullptr;     // persistent filter kernel: the scanned vectors of a row LIST / an external block as a staged I panel (tiled layout)
    size_t c_pairs = 0, c_pair_ctl = 0, c_irr_ok = 0, c_stage_t = 0;
    uint64_t *d_offsets = nullptr, *d_exist_off = nullptr;
    uint32_t *d_exist_to = nullptr;
    size_t c_exist_off = 0, c_exist_to = 0;
    uint8_t *d_deleted = nullptr;
    char *d_temp = nullptr;
    uint32_t *d_from = nullptr, *d_to = nullptr;
    float *d_w = nullptr;
    size_t c_scan = 0, c_cand_cnt = 0, c_cand = 0, c_overflow = 0, c_list_rows = 0, c_list_cnt = 0, c_counts = 0,
           c_ident = 0, c_list_scores = 0, c_list_dists = 0, c_offsets = 0, c_deleted = 0, c_temp = 0, c_from = 0,
           c_to = 0, c_w = 0;
    ~PassScratch() {
        (void)hipFree(d_scan); (void)hipFree(d_cand_cnt); (void)hipFree(d_cand); (void)hipFree(d_overflow);
        (void)hipFree(d_list_rows); (void)hipFree(d_list_cnt); (void)hipFree(d_counts); (void)hipFree(d_ident);
        (void)hipFree(d_list_scores); (void)hipFree(d_list_dists); (void)hipFree(d_offsets); (void)hipFree(d_pair_sims);
        (void)hipFree(d_deleted); (void)hipFree(d_temp); (void)hipFree(d_from); (void)hipFree(d_to); (void)hipFree(d_w);
        (void)hipFree(d_exist_off); (void)hipFree(d_exist_to); (void)hipFree(d_pairs); (void)hipFree(d_pair_ctl); (void)hipFree(d_irr_ok); (void)hipFree(d_stage_t);
        if (ev_k0) (void)hipEventDestroy(ev_k0);
        if (ev_k1) (void)hipEventDestroy(ev_k1);
    }
};

PassScratch &scratch_of(Ctx *c) {
    if (!c->pass_scratch) {
        c->pass_scratch = new PassScratch();
        c->pass_scratch_free = [](void *p) { delete static_cast<PassScratch *>(p); };
    }
    return *static_cast<PassScratch *>(c->pass_scratch);
}

// Runs the pass and leaves the edges in ps.d_from/d_to/d_w (total of them in *total).
// Exact path for scanned vectors the filter could not serve (candidate overflow, dim % 64 != 0): their top-k
// lists by the batched search (the rows are read once per 32-64 of them; one scan per vector cost 0.44 ms each
// at 1M rows), re-scored with the filter path's exact f32 arithmetic so that a list's scores do not depend on the
// path that produced it, then scattered to their places.  redo[i] = position in the scan set; the vector is row
// scan_rows[redo[i]] (or redo[i]) of the shard, or vector redo[i] of d_ext_queries.
static int redo_lists(const cx_index *ix, Ctx *c, PassScratch &ps, hipStream_t s, const std::vector<uint32_t> &redo,
                      const uint32_t *scan_rows, const float *d_ext_queries, uint32_t topk, uint32_t *dst_rows,
                      float *dst_scores, float *dst_dists, uint32_t *dst_cnt) {
    const uint32_t n_rows = (uint32_t)ix->n_rows;
    DevFilter flt;
    memset(&flt, 0, sizeof flt);
    flt.meta = ix->d_meta;
    flt.agent = ix->d_agent;
    const uint32_t k_eff = std::min<uint32_t>(topk, n_rows);
    const uint32_t blk = (uint32_t)std::min<size_t>(8192, redo.size());
    RedoScratch &rs = ps.redo;
    if (int rc = ensure_dev(rs.d_vec, rs.c_vec, (size_t)blk * ix->dim)) return rc;
    if (int rc = ensure_dev(rs.d_src, rs.c_src, (size_t)blk)) return rc;
    if (int rc = ensure_dev(rs.d_pos, rs.c_pos, (size_t)blk)) return rc;
    if (int rc = ensure_dev(rs.d_rows, rs.c_rows, (size_t)blk * std::max(k_eff, 1u))) return rc;
    if (int rc = ensure_dev(rs.d_scores, rs.c_scores, (size_t)blk * std::max(k_eff, 1u))) return rc;
    if (int rc = ensure_dev(rs.d_dists, rs.c_dists, (size_t)blk * std::max(k_eff, 1u))) return rc;
    if (int rc = ensure_dev(rs.d_cnt, rs.c_cnt, (size_t)blk)) return rc;
    if (int rc = ensure_dev(rs.d_rows2, rs.c_rows2, (size_t)blk * std::max(k_eff, 1u))) return rc;
    if (int rc = ensure_dev(rs.d_cnt2, rs.c_cnt2, (size_t)blk)) return rc;
    if (int rc = ensure_dev(rs.d_of, rs.c_of, (size_t)blk)) return rc;
    std::vector<uint32_t> src(blk);
    for (size_t lo = 0; lo < redo.size(); lo += blk) {
        const uint32_t m = (uint32_t)std::min<size_t>(blk, redo.size() - lo);
        for (uint32_t i = 0; i < m; i++) src[i] = (scan_rows && !d_ext_queries) ? scan_rows[redo[lo + i]] : redo[lo + i];
        CX_HIP(hipMemcpyAsync(rs.d_src, src.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
        CX_HIP(hipMemcpyAsync(rs.d_pos, redo.data() + lo, (size_t)m * 4, hipMemcpyHostToDevice, s));
        if (int rc = ((!d_ext_queries && ix->dtype == 1) ? launch_gather_rows(ix->rows16(), rs.d_vec, rs.d_src, m, ix->dim, s) : launch_gather_rows(d_ext_queries ? d_ext_queries : ix->d_rows, rs.d_vec, rs.d_src, m, ix->dim, s))) return rc;
        const uint32_t *l_rows = rs.d_rows, *l_cnt = rs.d_cnt;
        if (k_eff == 0) {
            CX_HIP(hipMemsetAsync(rs.d_cnt, 0, (size_t)m * 4, s));
        } else {
            if (int rc = search_core(ix, c, rs.d_vec, nullptr, m, k_eff, flt, 0.0f, false, rs.d_rows, rs.d_scores, rs.d_dists,
                                     rs.d_cnt, s))
                return rc;
            RescoreArgs r;
            memset(&r, 0, sizeof r);
            r.rows = ix->rows32();
            r.rows16 = ix->rows16();
            r.q_rows = rs.d_vec;          // the gathered vectors, in redo order
            r.out_dists = rs.d_dists;
            r.meta = ix->d_meta;
            r.cand_cnt = rs.d_cnt;
            r.cand = rs.d_rows;
            r.n_scan = m;
            r.dim = ix->dim;
            r.cap = k_eff;
            r.topk = k_eff;
            r.threshold = -1.0f;          // keep every (non-NaN) entry: the rules apply the threshold
            r.out_rows = rs.d_rows2;
            r.out_scores = rs.d_scores;
            r.out_cnt = rs.d_cnt2;
            r.overflow = rs.d_of;
            if (int rc = launch_rescore(r, s)) return rc;
            l_rows = rs.d_rows2;
            l_cnt = rs.d_cnt2;
        }
        if (int rc = launch_scatter_lists(l_rows, rs.d_scores, dst_dists ? rs.d_dists : nullptr, l_cnt, rs.d_pos, m,
                                          std::max(k_eff, 1u), topk, dst_rows, dst_scores, dst_dists, dst_cnt, s))
            return rc;
        CX_HIP(hipStreamSynchronize(s));   // src / redo staging is reused by the next block
    }
    return CX_OK;
}

// The reference's per-cycle inputs besides the thresholds (auto_linker.rs:226-231, :284-287).
struct CycleInputs {
    const uint64_t *existing_offsets = nullptr;   // host CSR over the scanned nodes, [n_scan + 1]; null = no edges yet
    const uint32_t *existing_to = nullptr;
    uint64_t max_edges_per_cycle = ~0ull;
    // cx_topk_lists_rows' use of the pass: stop once the ordered lists stand in the scratch (no rule walk), and send
    // every scanned row whose thresholded list holds fewer than min_count entries down the exact path as well
    bool lists_only = false;
    uint32_t min_count = 0;
    uint32_t cand_cap = 0;      // candidate slots per scanned row; 0 = cand_cap()
    bool no_persist = false;    // the retry of a pass whose pair buffer ran over: the per-tile kernel (pair_filter_kernel) instead
};

int pass_core(const cx_index *ix, Ctx *c, PassScratch &ps, uint64_t n_scan64, const uint32_t *scan_rows,
              uint32_t topk, float threshold, uint32_t max_edges, const uint8_t *deleted, bool dedup,
              const CycleInputs &cyc, uint64_t *total, double *phase_ms /* optional [4] */) {
    const uint32_t n_rows = (uint32_t)ix->n_rows;
    const uint32_t n_scan = (uint32_t)n_scan64;
    hipStream_t s = c->stream;
    *total = 0;
    if (!n_scan || !n_rows) return CX_OK;
    if (topk == 0 || topk > TOPK_MAX) return set_err(CX_ERR_VALIDATION, "autolink: topk must be in 1..%u", TOPK_MAX);
    for (uint32_t i = 0; scan_rows && i < n_scan; i++)
        if (scan_rows[i] >= n_rows) return set_err(CX_ERR_VALIDATION, "autolink: scan row %u out of range", scan_rows[i]);
    const uint32_t cap = cyc.cand_cap ? cyc.cand_cap : cand_cap();
    bool mfma_path = ix->dim % 64 == 0 && ix->dim > 0;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (phase_ms) for (auto &e : ev) CX_HIP(hipEventCreate(&e));

    uint32_t *d_scan = nullptr;   // null = identity
    if (scan_rows) {
        if (int rc = ensure_dev(ps.d_scan, ps.c_scan, (size_t)n_scan)) return rc;
        CX_HIP(hipMemcpyAsync(ps.d_scan, scan_rows, (size_t)n_scan * 4, hipMemcpyHostToDevice, s));
        d_scan = ps.d_scan;
    }
    uint8_t *d_deleted = nullptr;
    if (deleted) {
        if (int rc = ensure_dev(ps.d_deleted, ps.c_deleted, (size_t)n_rows)) return rc;
        CX_HIP(hipMemcpyAsync(ps.d_deleted, deleted, n_rows, hipMemcpyHostToDevice, s));
        d_deleted = ps.d_deleted;
    }
    // existing edges of the scanned nodes: validated here; sorted per node (the rules kernel bisects a node's segment) and
    // uploaded by stage_existing() below, AFTER the filter and the rescore are queued — the host's share of a rescan
    // (4 ms of copying and sorting at 100k nodes) then runs beside the GPU's instead of in front of it, and the device does
    // not idle (and clock down: tuning.md) between the passes of a cycle loop
    std::vector<uint32_t> exist_sorted;
    bool exist_staged = false;
    if (cyc.existing_offsets) {
        const uint64_t *eo = cyc.existing_offsets;
        if (eo[0] != 0) return set_err(CX_ERR_VALIDATION, "autolink: existing_offsets[0] must be 0");
        for (uint32_t i = 0; i < n_scan; i++)
            if (eo[i + 1] < eo[i]) return set_err(CX_ERR_VALIDATION, "autolink: existing_offsets decrease at node %u", i);
        if (eo[n_scan] && !cyc.existing_to) return set_err(CX_ERR_VALIDATION, "autolink: existing_to is null");
    }
    auto stage_existing = [&]() -> int {
        if (!cyc.existing_offsets || exist_staged || dedup) return CX_OK;
        const uint64_t *eo = cyc.existing_offsets;
        const uint64_t n_exist = eo[n_scan];
        exist_sorted.assign(cyc.existing_to, cyc.existing_to + n_exist);
        for (uint32_t i = 0; i < n_scan; i++)
            if (eo[i + 1] - eo[i] > 1) std::sort(exist_sorted.begin() + eo[i], exist_sorted.begin() + eo[i + 1]);
        if (int rc = ensure_dev(ps.d_exist_off, ps.c_exist_off, (size_t)n_scan + 1)) return rc;
        if (int rc = ensure_dev(ps.d_exist_to, ps.c_exist_to, (size_t)std::max<uint64_t>(n_exist, 1))) return rc;
        CX_HIP(hipMemcpyAsync(ps.d_exist_off, eo, ((size_t)n_scan + 1) * 8, hipMemcpyHostToDevice, s));
        if (n_exist) CX_HIP(hipMemcpyAsync(ps.d_exist_to, exist_sorted.data(), (size_t)n_exist * 4, hipMemcpyHostToDevice, s));
        exist_staged = true;
        return CX_OK;
    };
    if (int rc = ensure_dev(ps.d_list_rows, ps.c_list_rows, (size_t)n_scan * topk)) return rc;
    if (int rc = ensure_dev(ps.d_list_scores, ps.c_list_scores, (size_t)n_scan * topk)) return rc;
    if (int rc = ensure_dev(ps.d_list_cnt, ps.c_list_cnt, (size_t)n_scan)) return rc;
    if (int rc = ensure_dev(ps.d_overflow, ps.c_overflow, (size_t)n_scan)) return rc;
    CX_HIP(hipMemsetAsync(ps.d_overflow, 0, (size_t)n_scan * 4, s));
    if (phase_ms) CX_HIP(hipEventRecord(ev[0], s));

    std::vector<uint32_t> redo;  // scanned positions that must take the exact scan path
    int prof_kind = 2;
    uint64_t prof_tiles = 0;
    uint32_t prof_bm = 256;
    if (mfma_path) {   // (the shadow's build counts the store's irregular rows: more than the passes carry along -> the exact path for every row)
        if (int rc = ensure_shadow(ix, s)) return rc;
        if (ix->irr_over) mfma_path = false;
    }
    if (mfma_path) {
        if (phase_ms) CX_HIP(hipEventRecord(ev[1], s));
        const uint32_t chunk = std::min<uint32_t>(n_scan, CHUNK_ROWS);
        if (int rc = ensure_dev(ps.d_cand_cnt, ps.c_cand_cnt, (size_t)chunk)) return rc;
        if (int rc = ensure_dev(ps.d_cand, ps.c_cand, (size_t)chunk * cap)) return rc;
        // phases are timed over all chunks: filter launches first would need all scratch at once, so
        // filter+rescore alternate per chunk and the two event pairs bracket their sums approximately
        bool used_persist = false, used_stream = false;
        uint32_t pairs_lost = 0;
        int filter_kind = 2;
        uint64_t filter_tiles = 0;
        uint32_t filter_bm = 256;
        for (uint32_t lo = 0; lo < n_scan; lo += chunk) {
            const uint32_t m = std::min<uint32_t>(chunk, n_scan - lo);
            CX_HIP(hipMemsetAsync(ps.d_cand_cnt, 0, (size_t)m * 4, s));
            PairFilterArgs f;
            memset(&f, 0, sizeof f);
            f.shadow = ix->d_shadow;
            f.shadow_t = ix->d_shadow_t;
            f.scan_rows = d_scan ? d_scan + lo : nullptr;
            f.n_scan = m;
            f.n_rows = n_rows;
            f.dim = ix->dim;
            f.thr_lo = threshold - FILTER_EPS;
            f.cand_cnt = ps.d_cand_cnt;
            f.cand = ps.d_cand;
            f.cap = cap;
            // large scan sets: 256x256 tiles on the 4-slot ring (allpairs_p.hip); small ones (streaming ingest) keep
            // the 128x128 kernel, where a mostly empty 256-row tile would waste MFMAs
            static const int big_min = getenv("CX_PAIR_256_MIN") ? atoi(getenv("CX_PAIR_256_MIN")) : 129;
            const bool big = ix->dim % 32 == 0 && (int64_t)m >= big_min;
            static const int sym_ok = getenv("CX_PAIR_SYMMETRIC") ? atoi(getenv("CX_PAIR_SYMMETRIC")) : 1;
            f.symmetric = (sym_ok && !scan_rows && lo == 0 && m == n_rows && (n_rows + 127u) / 128u <= 0xFFFFu) ? 1u : 0u;
            f.tile_list = nullptr;
            f.n_tiles = 0;
            const int persist_ok = getenv("CX_PAIR_PERSIST") ? atoi(getenv("CX_PAIR_PERSIST")) : 1;   // read per call: tests switch it
            bool persist = false;
            if (big && persist_ok && !cyc.no_persist) {   // persistent blocks (allpairs_p.hip): hits leave the GEMM as pairs
                const size_t pair_cap = (size_t)std::min<uint64_t>((uint64_t)m * cap / 2u, 128ull << 20);
                if (int rc = ensure_dev(ps.d_pairs, ps.c_pairs, pair_cap)) return rc;
                if (int rc = ensure_dev(ps.d_pair_ctl, ps.c_pair_ctl, (size_t)32)) return rc;
                f.pairs = ps.d_pairs;
                f.pair_ctl = ps.d_pair_ctl;
                f.pair_cap = (uint32_t)pair_cap;
                f.block_rows = pair_filter_p_block_rows();
                // the scanned rows as a range of the shard: all of them, or a run of consecutive rows (an ingest tick's batch)
                f.scan_lo = 0;
                f.scan_contig = 0;
                if (scan_rows) {
                    bool run = true;
                    for (uint32_t i = 1; i < m && run; i++) run = scan_rows[lo + i] == scan_rows[lo] + i;
                    static const int stage_ok = getenv("CX_PAIR_P_STAGE") ? atoi(getenv("CX_PAIR_P_STAGE")) : 1;
                    // (a run that starts k rows past a 32-row boundary is tiled from that boundary: 500 rows + k > 512 take a THIRD
                    // 256-row panel — config 5's 500-row ticks alternated between 5.5 and 7.5 ms with their start row; staged, a
                    // run always starts at row 0 of its panel)
                    const bool run_spills = run && (m + scan_rows[lo] % 32u + 255u) / 256u > (m + 255u) / 256u;
                    if (run && !(run_spills && stage_ok && ix->d_shadow_t)) {
                        f.scan_lo = scan_rows[lo];
                        f.scan_contig = 1;
                    } else if (stage_ok && ix->d_shadow_t) {
                        // a LIST of rows (a cycle's batch of nodes, a subset dedup): their shadow pieces gathered into a staged I panel
                        // (m x dim x 2 bytes copied once: 0.05 ms at 100k x 768 against a 6 ms GEMM) — the same kernel as a range
                        const size_t words = (size_t)((m + 255u) / 256u) * 256u * ix->dim;
                        if (int rc = ensure_dev(ps.d_stage_t, ps.c_stage_t, words)) return rc;
                        if (int rc = launch_stage_scan_rows(ix->d_shadow_t, f.scan_rows, m, ix->dim, ps.d_stage_t, s)) return rc;
                        f.shadow_i = ps.d_stage_t;
                        f.scan_lo = 0;
                        f.scan_contig = 1;
                    }
                } else {
                    f.scan_lo = lo;
                    f.scan_contig = 1;   // (a later chunk's materialised identity rows are exactly this range)
                }
                persist = pair_filter_p_supported(f);
            }
            const uint32_t tile_kind = persist ? f.block_rows : 0u;   // which tile list: the 128^2 kernel's, or the persistent kernel's bm x 256 tiles
            if (f.symmetric) {
                std::lock_guard<std::mutex> g(ix->shadow_mu);
                if (ix->tile_list_rows != n_rows || !ix->d_tile_list || ix->tile_list_big != tile_kind) {
                    std::vector<uint32_t> tl;
                    if (persist) pair_filter_p_tile_list(n_rows, tile_kind, tl);
                    else pair_filter_tile_list(n_rows, tl);
                    ix->tile_list_big = tile_kind;
                    if (ix->d_tile_list) CX_HIP(hipFree(ix->d_tile_list));
                    ix->d_tile_list = nullptr;
                    CX_HIP(hipMalloc((void **)&ix->d_tile_list, tl.size() * 4));
                    CX_HIP(hipMemcpy(ix->d_tile_list, tl.data(), tl.size() * 4, hipMemcpyHostToDevice));
                    ix->tile_list_rows = n_rows;
                    ix->tile_list_n = (uint32_t)tl.size();
                }
                f.tile_list = ix->d_tile_list;
                f.n_tiles = ix->tile_list_n;
            }
            // identity scan rows of a later chunk still need their global row: materialise them
            if (!d_scan && lo > 0) {
                std::vector<uint32_t> ident(m);
                for (uint32_t i = 0; i < m; i++) ident[i] = lo + i;
                CX_HIP(hipStreamSynchronize(s));  // the previous chunk may still read d_ident
                if (int rc = ensure_dev(ps.d_ident, ps.c_ident, (size_t)m)) return rc;
                CX_HIP(hipMemcpyAsync(ps.d_ident, ident.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
                CX_HIP(hipStreamSynchronize(s));
                f.scan_rows = ps.d_ident;
            }
            static const int stream_ok = getenv("CX_PAIR_STREAM") ? atoi(getenv("CX_PAIR_STREAM")) : 1;
            if (phase_ms && lo == 0) {   // the GEMM kernel alone, for cx_autolink_filter_profile
                if (!ps.ev_k0) { CX_HIP(hipEventCreate(&ps.ev_k0)); CX_HIP(hipEventCreate(&ps.ev_k1)); }
                f.ev_begin = ps.ev_k0;
                f.ev_end = ps.ev_k1;
            }
            if (!big && stream_ok && pair_filter_stream_supported(f)) {
                if (int rc = ensure_dev(ps.d_pair_ctl, ps.c_pair_ctl, (size_t)32)) return rc;   // (two words of it: the tile counter and the give-up flag of the pass on large shards)
                CX_HIP(hipMemsetAsync(ps.d_pair_ctl + 24, 0, 8, s));
                f.pair_ctl = ps.d_pair_ctl;
                if (int rc = launch_pair_filter_stream(f, s)) return rc;
                used_stream = true;
                filter_kind = 2;
            } else if (persist) {
                if (int rc = launch_pair_filter_p(f, s)) return rc;
                used_persist = true;
                filter_kind = 1;
            } else {
                if (int rc = launch_pair_filter(f, s)) return rc;   // (round 4: the 256-tile per-launch kernel is gone — what the persistent kernel
                filter_kind = 2;                                    // does not take — dims below 384, a pair buffer that ran over — runs on 128 x 128 tiles)
            }
            filter_bm = persist ? f.block_rows : 256u;
            if (lo == 0) filter_tiles = f.symmetric ? f.n_tiles : (uint64_t)((m + (persist ? f.scan_lo % 32u : 0u) + filter_bm - 1u) / filter_bm) * ((n_rows + 255u) / 256u);
            if (phase_ms && lo == 0) CX_HIP(hipEventRecord(ev[2], s));
            RescoreArgs r;
            memset(&r, 0, sizeof r);
            r.rows = ix->rows32();
            r.rows16 = ix->rows16();
            r.meta = ix->d_meta;
            r.scan_rows = f.scan_rows;
            r.cand_cnt = ps.d_cand_cnt;
            r.cand = ps.d_cand;
            r.n_scan = m;
            r.dim = ix->dim;
            r.cap = cap;
            r.topk = topk;
            r.threshold = threshold;
            r.out_rows = ps.d_list_rows + (size_t)lo * topk;
            r.out_scores = ps.d_list_scores + (size_t)lo * topk;
            r.out_cnt = ps.d_list_cnt + lo;
            r.overflow = ps.d_overflow + lo;
            static const int sym_rescore = getenv("CX_RESCORE_SYMMETRIC") ? atoi(getenv("CX_RESCORE_SYMMETRIC")) : 1;
            // irregular rows (zero shadow rows: the filter never returns them): as scanned rows they go down the exact path, as
            // neighbours they join every other scanned row's candidates (allpairs.hip: launch_irr_append).  A threshold at or below the
            // filter's eps lets every pair through anyway, zero shadows included.
            const bool irr = ix->irr_n > 0 && f.thr_lo > 0.0f;
            if (irr) {
                if (int rc = ensure_dev(ps.d_irr_ok, ps.c_irr_ok, (size_t)BS_IRR_CAP)) return rc;
                if (int rc = launch_irr_append(ix->rows32(), ix->rows16(), ix->dim, n_rows, ix->d_irr_rows, ix->irr_n, ps.d_irr_ok, m, ps.d_cand_cnt, ps.d_cand, cap, s)) return rc;
            }
            if (f.symmetric && sym_rescore && !irr) {   // every list of the store is present: each pair is scored once (allpairs.hip)
                if (int rc = ensure_dev(ps.d_pair_sims, ps.c_pair_sims, (size_t)m * cap)) return rc;
                r.pair_sims = ps.d_pair_sims;
            }
            if (int rc = launch_rescore(r, s)) return rc;
            if (irr)
                if (int rc = launch_irr_mark(ix->d_irr_rows, ps.d_irr_ok, ix->irr_n, f.scan_rows, nullptr, ix->dim, m, ps.d_overflow + lo, s)) return rc;
            if (used_persist && n_scan > chunk) {   // the control words are reused by the next chunk
                uint32_t lost = 0;
                CX_HIP(hipMemcpyAsync(&lost, ps.d_pair_ctl + 1, 4, hipMemcpyDeviceToHost, s));
                CX_HIP(hipStreamSynchronize(s));
                pairs_lost |= lost;
            }
        }
        prof_kind = filter_kind;
        prof_tiles = filter_tiles;
        prof_bm = filter_bm;
        if (phase_ms && n_scan > chunk) CX_HIP(hipEventRecord(ev[2], s));  // multi-chunk: only the total is meaningful
        if (phase_ms) CX_HIP(hipEventRecord(ev[3], s));
        if (!cyc.lists_only)
            if (int rc = stage_existing()) return rc;   // host work under the queued filter + rescore (before any copy that waits for them)
        std::vector<uint32_t> of(n_scan);
        CX_HIP(hipMemcpyAsync(of.data(), ps.d_overflow, (size_t)n_scan * 4, hipMemcpyDeviceToHost, s));
        if (used_persist && n_scan <= chunk) CX_HIP(hipMemcpyAsync(&pairs_lost, ps.d_pair_ctl + 1, 4, hipMemcpyDeviceToHost, s));
        uint32_t stream_gave_up = 0;   // (batchs.hip's threshold mode: a worker dropped hits — every scanned row goes down the exact path)
        if (used_stream) CX_HIP(hipMemcpyAsync(&stream_gave_up, ps.d_pair_ctl + 25, 4, hipMemcpyDeviceToHost, s));
        CX_HIP(hipStreamSynchronize(s));
        if (stream_gave_up) std::fill(of.begin(), of.end(), 1u);
        if (pairs_lost) {   // more hits than the pair buffer holds (a threshold far below the data's): the per-row path has no such limit
            if (phase_ms) for (auto &e : ev) (void)hipEventDestroy(e);
            CycleInputs again = cyc;
            again.no_persist = true;
            return pass_core(ix, c, ps, n_scan64, scan_rows, topk, threshold, max_edges, deleted, dedup, again, total, phase_ms);
        }
        if (cyc.min_count) {   // short lists: the threshold hid part of this row's top-k
            std::vector<uint32_t> cnt(n_scan);
            CX_HIP(hipMemcpyAsync(cnt.data(), ps.d_list_cnt, (size_t)n_scan * 4, hipMemcpyDeviceToHost, s));
            CX_HIP(hipStreamSynchronize(s));
            for (uint32_t i = 0; i < n_scan; i++)
                if (!of[i] && cnt[i] < cyc.min_count) of[i] = 1u;
        }
        for (uint32_t i = 0; i < n_scan; i++)
            if (of[i]) redo.push_back(i);
    } else {
        if (phase_ms) { CX_HIP(hipEventRecord(ev[1], s)); CX_HIP(hipEventRecord(ev[2], s)); CX_HIP(hipEventRecord(ev[3], s)); }
        redo.resize(n_scan);
        for (uint32_t i = 0; i < n_scan; i++) redo[i] = i;
    }

    // exact scan path for rows the filter could not serve (candidate overflow, dim % 64 != 0):
    // search(emb_i, topk) with the row itself as the query, straight into the list arrays
    if (!redo.empty())
        if (int rc = redo_lists(ix, c, ps, s, redo, scan_rows, nullptr, topk, ps.d_list_rows, ps.d_list_scores, nullptr,
                                ps.d_list_cnt))
            return rc;

    if (cyc.lists_only) {
        CX_HIP(hipStreamSynchronize(s));
        *total = redo.size();   // how many lists came from the exact path (diagnostics)
        return CX_OK;
    }
    // link rules: count, exclusive scan, emit
    if (int rc = stage_existing()) return rc;   // (the exact-path-only passes come here without having staged it)
    if (int rc = ensure_dev(ps.d_counts, ps.c_counts, (size_t)n_scan)) return rc;
    if (int rc = ensure_dev(ps.d_offsets, ps.c_offsets, (size_t)n_scan)) return rc;
    const size_t tb = scan_temp_bytes(n_scan);
    if (int rc = ensure_dev(ps.d_temp, ps.c_temp, tb)) return rc;
    LinkArgs l;
    memset(&l, 0, sizeof l);
    l.scan_rows = d_scan;
    l.list_rows = ps.d_list_rows;
    l.list_scores = ps.d_list_scores;
    l.list_cnt = ps.d_list_cnt;
    l.deleted = d_deleted;
    l.meta = ix->d_meta;
    l.n_scan = n_scan;
    l.topk = topk;
    l.max_edges = dedup ? 0xFFFFFFFFu : max_edges;
    l.max_total = dedup ? ~0ull : cyc.max_edges_per_cycle;
    if (cyc.existing_offsets && !dedup) {
        l.existing_offsets = ps.d_exist_off;
        l.existing_to = ps.d_exist_to;
    }
    l.dedup = dedup ? 1u : 0u;
    l.threshold = threshold;
    l.counts = ps.d_counts;
    // dedup.rs:85-87 is search_threshold: no k.  A scanned row whose list is full at topk with its tail still at or above
    // the threshold ("dense": a cluster of more than topk near-duplicates) gets its COMPLETE list from the exact threshold
    // path (dense keys + radix sort, what cx_search_threshold runs), the reference's walk over it on the host, and its
    // pairs spliced into the device output at the row's offset — never CX_ERR_CAPACITY for data reasons.
    std::vector<uint32_t> dense;                 // scan positions
    std::vector<uint64_t> dense_src_off;         // CSR over the dense rows' pairs
    std::vector<uint32_t> dense_to;
    std::vector<float> dense_w;
    if (dedup) {
        std::vector<uint32_t> cnt(n_scan);
        std::vector<float> tail(n_scan);
        CX_HIP(hipMemcpyAsync(cnt.data(), ps.d_list_cnt, (size_t)n_scan * 4, hipMemcpyDeviceToHost, s));
        CX_HIP(hipMemcpy2DAsync(tail.data(), 4, ps.d_list_scores + (topk - 1), (size_t)topk * 4, 4, n_scan, hipMemcpyDeviceToHost, s));   // ONE copy of the last column
        CX_HIP(hipStreamSynchronize(s));
        for (uint32_t i = 0; i < n_scan; i++)
            if (cnt[i] >= topk && tail[i] >= threshold) dense.push_back(i);   // lists from the exact path are not thresholded: the tail decides
    }
    if (!dense.empty()) {
        DevFilter flt;
        memset(&flt, 0, sizeof flt);
        flt.meta = ix->d_meta;
        flt.agent = ix->d_agent;
        RedoScratch &rs = ps.redo;
        const uint32_t blk = (uint32_t)std::min<size_t>(64, dense.size());
        if (int rc = ensure_dev(rs.d_vec, rs.c_vec, (size_t)blk * ix->dim)) return rc;
        if (int rc = ensure_dev(rs.d_src, rs.c_src, (size_t)blk)) return rc;
        if (int rc = ensure_dev(rs.d_rows, rs.c_rows, (size_t)n_rows)) return rc;
        if (int rc = ensure_dev(rs.d_scores, rs.c_scores, (size_t)n_rows)) return rc;
        if (int rc = ensure_dev(rs.d_dists, rs.c_dists, (size_t)n_rows)) return rc;
        if (int rc = ensure_dev(rs.d_cnt, rs.c_cnt, (size_t)1)) return rc;
        std::vector<uint32_t> src(blk), h_rows(n_rows);
        std::vector<float> h_scores(n_rows);
        dense_src_off.push_back(0);
        for (size_t lo = 0; lo < dense.size(); lo += blk) {
            const uint32_t m = (uint32_t)std::min<size_t>(blk, dense.size() - lo);
            for (uint32_t t = 0; t < m; t++) src[t] = scan_rows ? scan_rows[dense[lo + t]] : dense[lo + t];
            CX_HIP(hipMemcpyAsync(rs.d_src, src.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
            if (int rc = (ix->dtype == 1 ? launch_gather_rows(ix->rows16(), rs.d_vec, rs.d_src, m, ix->dim, s) : launch_gather_rows(ix->d_rows, rs.d_vec, rs.d_src, m, ix->dim, s))) return rc;
            for (uint32_t t = 0; t < m; t++) {
                if (int rc = search_core(ix, c, rs.d_vec + (size_t)t * ix->dim, nullptr, 1, n_rows, flt, threshold, true, rs.d_rows, rs.d_scores,
                                         rs.d_dists, rs.d_cnt, s))
                    return rc;
                uint32_t got = 0;
                CX_HIP(hipMemcpyAsync(&got, rs.d_cnt, 4, hipMemcpyDeviceToHost, s));
                CX_HIP(hipStreamSynchronize(s));
                if (got > n_rows) return set_err(CX_ERR_DEVICE, "dedup: threshold list of %u entries in an index of %u rows", got, n_rows);
                CX_HIP(hipMemcpyAsync(h_rows.data(), rs.d_rows, (size_t)got * 4, hipMemcpyDeviceToHost, s));
                CX_HIP(hipMemcpyAsync(h_scores.data(), rs.d_scores, (size_t)got * 4, hipMemcpyDeviceToHost, s));
                CX_HIP(hipStreamSynchronize(s));
                const uint32_t self = src[t];
                for (uint32_t r = 0; r < got; r++) {   // dedup.rs:89-113, as link_rules_kernel walks a list
                    const uint32_t j = h_rows[r];
                    if (j >= n_rows) return set_err(CX_ERR_DEVICE, "dedup: device list names row %u of %u", j, n_rows);
                    if (j == self) continue;
                    if (j < self && !(deleted && deleted[j])) continue;   // the pair was reported when j was scanned
                    if (!(h_scores[r] >= threshold)) continue;
                    dense_to.push_back(j);
                    dense_w.push_back(h_scores[r]);
                }
                dense_src_off.push_back(dense_to.size());
            }
        }
        // the dense rows' truncated lists emit nothing; their counts are the host's
        if (int rc = ensure_dev(rs.d_pos, rs.c_pos, dense.size())) return rc;
        CX_HIP(hipMemcpyAsync(rs.d_pos, dense.data(), dense.size() * 4, hipMemcpyHostToDevice, s));
        if (int rc = launch_patch_u32(ps.d_list_cnt, rs.d_pos, nullptr, 0u, (uint32_t)dense.size(), s)) return rc;
    }
    if (int rc = launch_link_rules(l, false, s)) return rc;
    std::vector<uint32_t> dense_cnt(dense.size());
    if (!dense.empty()) {
        RedoScratch &rs = ps.redo;
        for (size_t t = 0; t < dense.size(); t++) {
            const uint64_t c64 = dense_src_off[t + 1] - dense_src_off[t];
            if (c64 > 0xFFFFFFFFull) return set_err(CX_ERR_CAPACITY, "dedup: too many pairs for one row");
            dense_cnt[t] = (uint32_t)c64;
        }
        if (int rc = ensure_dev(rs.d_cnt2, rs.c_cnt2, dense.size())) return rc;
        CX_HIP(hipMemcpyAsync(rs.d_cnt2, dense_cnt.data(), dense.size() * 4, hipMemcpyHostToDevice, s));
        if (int rc = launch_patch_u32(ps.d_counts, rs.d_pos, rs.d_cnt2, 0u, (uint32_t)dense.size(), s)) return rc;
    }
    if (int rc = launch_exclusive_scan(ps.d_counts, ps.d_offsets, n_scan, ps.d_temp, tb, s)) return rc;
    uint64_t last_off = 0;
    uint32_t last_cnt = 0;
    CX_HIP(hipMemcpyAsync(&last_off, ps.d_offsets + (n_scan - 1), 8, hipMemcpyDeviceToHost, s));
    CX_HIP(hipMemcpyAsync(&last_cnt, ps.d_counts + (n_scan - 1), 4, hipMemcpyDeviceToHost, s));
    CX_HIP(hipStreamSynchronize(s));
    const uint64_t n_edges = std::min<uint64_t>(last_off + last_cnt, l.max_total);   // :284-287
    if (n_edges) {
        if (int rc = ensure_dev(ps.d_from, ps.c_from, (size_t)n_edges)) return rc;
        if (int rc = ensure_dev(ps.d_to, ps.c_to, (size_t)n_edges)) return rc;
        if (int rc = ensure_dev(ps.d_w, ps.c_w, (size_t)n_edges)) return rc;
        l.offsets = ps.d_offsets;
        l.out_from = ps.d_from;
        l.out_to = ps.d_to;
        l.out_weight = ps.d_w;
        if (int rc = launch_link_rules(l, true, s)) return rc;
        if (!dense.empty() && !dense_to.empty()) {   // splice: segment t of (dense_to, dense_w) -> the output at offsets[dense[t]]
            RedoScratch &rs = ps.redo;
            std::vector<uint32_t> from_row(dense.size());
            for (size_t t = 0; t < dense.size(); t++) from_row[t] = scan_rows ? scan_rows[dense[t]] : dense[t];
            // (grow-only scratch of the pooled context: nothing to leak on an early return, no allocation per dense pass)
            if (int rc = ensure_dev(rs.d_src_off, rs.c_src_off, dense.size() + 1)) return rc;
            if (int rc = ensure_dev(rs.d_from_row, rs.c_from_row, dense.size())) return rc;
            if (int rc = ensure_dev(rs.d_to_h, rs.c_to_h, dense_to.size())) return rc;
            if (int rc = ensure_dev(rs.d_w_h, rs.c_w_h, dense_w.size())) return rc;
            hipError_t e1 = hipMemcpyAsync(rs.d_src_off, dense_src_off.data(), (dense.size() + 1) * 8, hipMemcpyHostToDevice, s);
            hipError_t e2 = hipMemcpyAsync(rs.d_from_row, from_row.data(), dense.size() * 4, hipMemcpyHostToDevice, s);
            hipError_t e3 = hipMemcpyAsync(rs.d_to_h, dense_to.data(), dense_to.size() * 4, hipMemcpyHostToDevice, s);
            hipError_t e4 = hipMemcpyAsync(rs.d_w_h, dense_w.data(), dense_w.size() * 4, hipMemcpyHostToDevice, s);
            int rc = (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess && e4 == hipSuccess)
                         ? launch_copy_edge_segments(ps.d_offsets, rs.d_pos, rs.d_src_off, rs.d_from_row, rs.d_to_h, rs.d_w_h, (uint32_t)dense.size(), ps.d_from,
                                                     ps.d_to, ps.d_w, s)
                         : set_err(CX_ERR_DEVICE, "dedup: staging the dense rows' pairs failed");
            (void)hipStreamSynchronize(s);   // (the host vectors staged above go out of scope)
            if (rc) return rc;
        }
    }
    if (phase_ms) CX_HIP(hipEventRecord(ev[4], s));
    CX_HIP(hipStreamSynchronize(s));
    if (phase_ms) {
        for (int p = 0; p < 4; p++) {
            float ms = 0.0f;
            CX_HIP(hipEventElapsedTime(&ms, ev[p], ev[p + 1]));
            phase_ms[p] = ms;
        }
        for (auto &e : ev) (void)hipEventDestroy(e);
        std::lock_guard<std::mutex> g(ix->shadow_mu);
        for (double &v : ix->filter_prof) v = 0.0;
        if (mfma_path && prof_kind != 2 && ps.ev_k0) {
            float ms = 0.0f;
            CX_HIP(hipEventElapsedTime(&ms, ps.ev_k0, ps.ev_k1));
            ix->filter_prof[0] = ms;
            ix->filter_prof[1] = 2.0 * (double)prof_bm * 256.0 * (double)ix->dim * (double)prof_tiles;
            ix->filter_prof[2] = (double)prof_tiles;
            ix->filter_prof[3] = (double)prof_kind;
            if (prof_kind == 1) {
                uint32_t h[32];
                CX_HIP(hipMemcpy(h, ps.d_pair_ctl, sizeof h, hipMemcpyDeviceToHost));
                unsigned long long clk, ref;
                memcpy(&clk, h + 16, 8);
                memcpy(&ref, h + 18, 8);
                ix->filter_prof[4] = ref ? (double)clk / (double)ref * 0.1 : 0.0;
            }
        }
    }
    *total = n_edges;
    return CX_OK;
}

int copy_edges_rows(const cx_index *ix, Ctx *c, PassScratch &ps, uint64_t total, uint64_t cap, uint32_t *out_from, uint32_t *out_to,
                    float *out_w, uint64_t *n_out, uint64_t *n_needed) {
    if (n_needed) *n_needed = total;
    const uint64_t take = std::min(total, cap);
    if (take) {
        if (!out_from || !out_to || !out_w) return set_err(CX_ERR_VALIDATION, "null output buffer");
        CX_HIP(hipMemcpyAsync(out_from, ps.d_from, take * 4, hipMemcpyDeviceToHost, c->stream));
        CX_HIP(hipMemcpyAsync(out_to, ps.d_to, take * 4, hipMemcpyDeviceToHost, c->stream));
        CX_HIP(hipMemcpyAsync(out_w, ps.d_w, take * 4, hipMemcpyDeviceToHost, c->stream));
        CX_HIP(hipStreamSynchronize(c->stream));
        // the caller maps these rows to ids (cx_row_id) and indexes its own arrays with them: check before handing over
        for (uint64_t i = 0; i < take; i++)
            if (out_from[i] >= ix->n_rows || out_to[i] >= ix->n_rows)
                return set_err(CX_ERR_DEVICE, "device edge list is corrupt: edge %llu is %u -> %u in an index of %llu rows",
                               (unsigned long long)i, out_from[i], out_to[i], (unsigned long long)ix->n_rows);
    }
    *n_out = take;
    if (total > cap) return set_err(CX_ERR_CAPACITY, "%llu edges, buffer holds %llu", (unsigned long long)total, (unsigned long long)cap);
    return CX_OK;
}

}  // namespace

extern "C" {

int cx_autolink_pass_rows(const cx_index *ix, uint64_t n_scan, const uint32_t *scan_rows, uint64_t topk,
                          float threshold, uint64_t max_edges_per_node, uint64_t max_edges_per_cycle,
                          const uint8_t *deleted, const uint64_t *existing_offsets, const uint32_t *existing_to,
                          uint64_t cap, uint32_t *out_from, uint32_t *out_to, float *out_weight, uint64_t *n_out,
                          uint64_t *n_needed) try {
    if (!ix || !n_out) return set_err(CX_ERR_VALIDATION, "null argument");
    *n_out = 0;
    if (n_needed) *n_needed = 0;
    if (!scan_rows) n_scan = ix->n_rows;
    if (n_scan > 0xFFFFFFF0ull) return set_err(CX_ERR_VALIDATION, "too many scanned rows");
    if (int rc = use_device(ix)) return rc;
    CtxLease lease(ix);
    if (!lease.c) return CX_ERR_DEVICE;
    PassScratch &ps = scratch_of(lease.c);
    uint64_t total = 0;
    CycleInputs cyc;
    cyc.existing_offsets = existing_offsets;
    cyc.existing_to = existing_to;
    cyc.max_edges_per_cycle = max_edges_per_cycle;
    if (int rc = pass_core(ix, lease.c, ps, n_scan, scan_rows, (uint32_t)std::min<uint64_t>(topk, 0xFFFFFFFFull), threshold,
                           (uint32_t)std::min<uint64_t>(max_edges_per_node, 0xFFFFFFFFull), deleted, false, cyc, &total, nullptr))
        return rc;
    return copy_edges_rows(ix, lease.c, ps, total, cap, out_from, out_to, out_weight, n_out, n_needed);
} catch (...) { return cx::on_exception(); }

int cx_dedup_scan_rows(const cx_index *ix, float dedup_threshold, const uint8_t *deleted, uint64_t cap,
                       uint32_t *out_a, uint32_t *out_b, float *out_similarity, uint64_t *n_out,
                       uint64_t *n_needed) try {
    if (!ix || !n_out) return set_err(CX_ERR_VALIDATION, "null argument");
    *n_out = 0;
    if (n_needed) *n_needed = 0;
    if (int rc = use_device(ix)) return rc;
    // scanned nodes: every row that is in the index and not storage-deleted (dedup.rs:70-81), in row order
    std::vector<uint32_t> scan;
    scan.reserve(ix->n_alive);
    for (uint64_t r = 0; r < ix->n_rows; r++)
        if (!(ix->h_meta[r] & META_REMOVED) && !(deleted && deleted[r])) scan.push_back((uint32_t)r);
    CtxLease lease(ix);
    if (!lease.c) return CX_ERR_DEVICE;
    PassScratch &ps = scratch_of(lease.c);
    uint64_t total = 0;
    if (int rc = pass_core(ix, lease.c, ps, scan.size(), scan.data(), TOPK_MAX, dedup_threshold, 0, deleted, true, CycleInputs(), &total, nullptr))
        return rc;
    return copy_edges_rows(ix, lease.c, ps, total, cap, out_a, out_b, out_similarity, n_out, n_needed);
} catch (...) { return cx::on_exception(); }

int cx_autolink_pass_timed(const cx_index *ix, uint64_t n_scan, const uint32_t *scan_rows, uint64_t topk,
                           float threshold, uint64_t max_edges_per_node, uint64_t max_edges_per_cycle,
                           const uint64_t *existing_offsets, const uint32_t *existing_to, uint64_t *n_edges,
                           double *phase_ms) try {
    if (!ix || !n_edges || !phase_ms) return set_err(CX_ERR_VALIDATION, "null argument");
    if (!scan_rows) n_scan = ix->n_rows;
    if (n_scan > 0xFFFFFFF0ull) return set_err(CX_ERR_VALIDATION, "too many scanned rows");
    if (int rc = use_device(ix)) return rc;
    CtxLease lease(ix);
    if (!lease.c) return CX_ERR_DEVICE;
    PassScratch &ps = scratch_of(lease.c);
    CycleInputs cyc;
    cyc.existing_offsets = existing_offsets;
    cyc.existing_to = existing_to;
    cyc.max_edges_per_cycle = max_edges_per_cycle;
    return pass_core(ix, lease.c, ps, n_scan, scan_rows, (uint32_t)std::min<uint64_t>(topk, 0xFFFFFFFFull), threshold,
                     (uint32_t)std::min<uint64_t>(max_edges_per_node, 0xFFFFFFFFull), nullptr, false, cyc, n_edges, phase_ms);
} catch (...) { return cx::on_exception(); }

int cx_autolink_filter_profile(const cx_index *ix, double out[5]) try {
    if (!ix || !out) return set_err(CX_ERR_VALIDATION, "null argument");
    std::lock_guard<std::mutex> g(ix->shadow_mu);
    for (int i = 0; i < 5; i++) out[i] = ix->filter_prof[i];
    return CX_OK;
} catch (...) { return cx::on_exception(); }

/* Ordered neighbour lists of nq external vectors against this shard (the multi-GPU building block of the
 * all-pairs pass): bf16 shadow of the queries -> MFMA filter against the shard's shadow -> exact rescore.
 * Lists that overflowed the candidate cap (and dims that are not a multiple of 64) are redone on the exact
 * scan path; those lists are not thresholded (the rule walk applies the threshold anyway). */
int cx_autolink_lists_dev(const cx_index *ix, uint64_t nq64, const float *d_queries, uint64_t topk64, float threshold,
                          uint32_t *d_out_rows, float *d_out_scores, float *d_out_dists, uint32_t *d_out_counts,
                          void *stream) try {
    if (!ix || !d_queries || !d_out_rows || !d_out_scores || !d_out_dists || !d_out_counts)
        return set_err(CX_ERR_VALIDATION, "null argument");
    if (topk64 == 0 || topk64 > TOPK_MAX) return set_err(CX_ERR_VALIDATION, "autolink lists: topk must be in 1..%u", TOPK_MAX);
    if (nq64 > 0xFFFFFFF0ull) return set_err(CX_ERR_VALIDATION, "too many queries");
    if (int rc = use_device(ix)) return rc;
    const uint32_t nq = (uint32_t)nq64, topk = (uint32_t)topk64, n_rows = (uint32_t)ix->n_rows;
    hipStream_t s = (hipStream_t)stream;
    if (!nq) return CX_OK;
    if (!n_rows) { CX_HIP(hipMemsetAsync(d_out_counts, 0, (size_t)nq * 4, s)); return CX_OK; }
    CtxLease lease(ix);
    if (!lease.c) return CX_ERR_DEVICE;
    Ctx *c = lease.c;
    PassScratch &ps = scratch_of(c);
    const uint32_t cap = cand_cap();
    std::vector<uint32_t> redo;
    bool filtered = ix->dim % 64 == 0 && ix->dim > 0;
    if (filtered) {
        if (int rc = ensure_shadow(ix, s)) return rc;
        filtered = !ix->irr_over;   // (more irregular rows than a pass carries along: the exact path for every vector)
    }
    if (filtered) {
        // query shadow lives in d_from scratch (uint32 words): nq*dim bf16 = nq*dim/2 words
        if (int rc = ensure_dev(ps.d_from, ps.c_from, (size_t)nq * ix->dim / 2 + 16)) return rc;
        uint16_t *d_qsh = reinterpret_cast<uint16_t *>(ps.d_from);
        if (int rc = launch_build_shadow(d_queries, d_qsh, 0, nq, ix->dim, s)) return rc;
        if (int rc = ensure_dev(ps.d_cand_cnt, ps.c_cand_cnt, (size_t)nq)) return rc;
        if (int rc = ensure_dev(ps.d_cand, ps.c_cand, (size_t)nq * cap)) return rc;
        if (int rc = ensure_dev(ps.d_overflow, ps.c_overflow, (size_t)nq)) return rc;
        CX_HIP(hipMemsetAsync(ps.d_cand_cnt, 0, (size_t)nq * 4, s));
        PairFilterArgs f;
        memset(&f, 0, sizeof f);
        f.shadow = ix->d_shadow;
        f.shadow_t = ix->d_shadow_t;
        f.shadow_q = d_qsh;
        f.n_scan = nq;
        f.n_rows = n_rows;
        f.dim = ix->dim;
        f.thr_lo = threshold - FILTER_EPS;
        f.cand_cnt = ps.d_cand_cnt;
        f.cand = ps.d_cand;
        f.cap = cap;
        static const int big_min = getenv("CX_PAIR_256_MIN") ? atoi(getenv("CX_PAIR_256_MIN")) : 129;
        static const int stream_ok = getenv("CX_PAIR_STREAM") ? atoi(getenv("CX_PAIR_STREAM")) : 1;
        static const int stage_ok = getenv("CX_PAIR_P_STAGE") ? atoi(getenv("CX_PAIR_P_STAGE")) : 1;
        const int persist_ok = getenv("CX_PAIR_PERSIST") ? atoi(getenv("CX_PAIR_PERSIST")) : 1;
        bool done = false, used_stream = false;
        if ((int64_t)nq >= big_min && stage_ok && persist_ok && ix->d_shadow_t) {
            // a block of external vectors (the sharded pass's Q blocks): their shadow built straight into a staged I panel in the
            // tiled layout, then the persistent kernel — the single-GPU pass's own (round 3: the 256-tile per-launch kernel, 15 % slower, retired in round 4)
            PairFilterArgs g = f;
            const uint32_t n_pad = (nq + 255u) / 256u * 256u;
            if (int rc = ensure_dev(ps.d_stage_t, ps.c_stage_t, (size_t)n_pad * ix->dim)) return rc;
            CX_HIP(hipMemsetAsync(ps.d_stage_t, 0, (size_t)n_pad * ix->dim * sizeof(uint16_t), s));
            if (int rc = launch_build_shadow_index(d_queries, nullptr, ps.d_stage_t, true, 0, nq, ix->dim, s, nullptr, nullptr, nullptr, 0)) return rc;
            const size_t pair_cap = (size_t)std::min<uint64_t>((uint64_t)nq * cap / 2u, 128ull << 20);
            if (int rc = ensure_dev(ps.d_pairs, ps.c_pairs, pair_cap)) return rc;
            if (int rc = ensure_dev(ps.d_pair_ctl, ps.c_pair_ctl, (size_t)32)) return rc;
            g.shadow_q = nullptr;
            g.shadow_i = ps.d_stage_t;
            g.scan_lo = 0;
            g.scan_contig = 1;
            g.pairs = ps.d_pairs;
            g.pair_ctl = ps.d_pair_ctl;
            g.pair_cap = (uint32_t)pair_cap;
            g.block_rows = pair_filter_p_block_rows();
            if (pair_filter_p_supported(g)) {
                if (int rc = launch_pair_filter_p(g, s)) return rc;
                uint32_t lost = 0;
                CX_HIP(hipMemcpyAsync(&lost, ps.d_pair_ctl + 1, 4, hipMemcpyDeviceToHost, s));
                CX_HIP(hipStreamSynchronize(s));
                if (lost) CX_HIP(hipMemsetAsync(ps.d_cand_cnt, 0, (size_t)nq * 4, s));   // more hits than the pair buffer holds: the per-row kernel below has no such limit
                else done = true;
            }
        }
        if (done) {
        } else if ((int64_t)nq < big_min && stream_ok && pair_filter_stream_supported(f)) {
            if (int rc = ensure_dev(ps.d_pair_ctl, ps.c_pair_ctl, (size_t)32)) return rc;
            CX_HIP(hipMemsetAsync(ps.d_pair_ctl + 24, 0, 8, s));
            f.pair_ctl = ps.d_pair_ctl;
            if (int rc = launch_pair_filter_stream(f, s)) return rc;
            used_stream = true;
        } else if (int rc = launch_pair_filter(f, s)) return rc;
        if (f.thr_lo > 0.0f) {   // irregular vectors (zero shadows) go down the exact path, the shard's irregular rows join every other list
            if (int rc = ensure_dev(ps.d_irr_ok, ps.c_irr_ok, (size_t)BS_IRR_CAP)) return rc;
            if (int rc = launch_irr_append(ix->rows32(), ix->rows16(), ix->dim, n_rows, ix->d_irr_rows, ix->irr_n, ps.d_irr_ok, nq, ps.d_cand_cnt, ps.d_cand, cap, s)) return rc;
        }
        RescoreArgs r;
        memset(&r, 0, sizeof r);
        r.rows = ix->rows32();
            r.rows16 = ix->rows16();
        r.q_rows = d_queries;
        r.meta = ix->d_meta;
        r.cand_cnt = ps.d_cand_cnt;
        r.cand = ps.d_cand;
        r.n_scan = nq;
        r.dim = ix->dim;
        r.cap = cap;
        r.topk = topk;
        r.threshold = threshold;
        r.out_rows = d_out_rows;
        r.out_scores = d_out_scores;
        r.out_dists = d_out_dists;
        r.out_cnt = d_out_counts;
        r.overflow = ps.d_overflow;
        if (int rc = launch_rescore(r, s)) return rc;
        if (f.thr_lo > 0.0f)
            if (int rc = launch_irr_mark(ix->d_irr_rows, ps.d_irr_ok, ix->irr_n, nullptr, d_queries, ix->dim, nq, ps.d_overflow, s)) return rc;
        std::vector<uint32_t> of(nq);
        CX_HIP(hipMemcpyAsync(of.data(), ps.d_overflow, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        uint32_t stream_gave_up = 0;
        if (used_stream) CX_HIP(hipMemcpyAsync(&stream_gave_up, ps.d_pair_ctl + 25, 4, hipMemcpyDeviceToHost, s));
        CX_HIP(hipStreamSynchronize(s));
        for (uint32_t i = 0; i < nq; i++)
            if (of[i] || stream_gave_up) redo.push_back(i);
    } else {
        redo.resize(nq);
        for (uint32_t i = 0; i < nq; i++) redo[i] = i;
    }
    if (!redo.empty())
        if (int rc = redo_lists(ix, c, ps, s, redo, nullptr, d_queries, topk, d_out_rows, d_out_scores, d_out_dists, d_out_counts))
            return rc;
    CX_HIP(hipStreamSynchronize(s));   // the pooled scratch goes back with the lease
    return CX_OK;
} catch (...) { return cx::on_exception(); }

}  // extern "C"

namespace {

// Ordered top-k lists of MANY scanned rows through the all-pairs machinery instead of the batched search: the bf16
// filter GEMM + exact rescore produce, per scanned row, every neighbour with score >= tau, best first; a row whose
// list holds k entries has its exact top k there (every row scoring >= its k-th best scores >= tau, and the filter
// loses no pair with exact score >= tau); a row with fewer — tau hid part of its top k — or with an overflowed
// candidate list goes down the exact path (redo_lists).  tau comes from a sample of the scanned rows: the smallest
// k-th best score among 64 of them, searched exactly, minus a margin.  Exactness never depends on tau; only the share
// of rows that take the slow path does (1/65 of them in expectation for scores without ties).
// 100k x 768, k = 100, every row: 0.196 s through the batched search's wide lists -> see profiles/r02/tuning.md.
// Returns false when this path does not apply (the caller's batched-search loop runs); true with *rc otherwise.
bool lists_by_filter(const cx_index *ix, Ctx *c, PassScratch &ps, uint32_t n_scan, const uint32_t *scan_rows, uint32_t topk,
                     uint32_t *out_rows, float *out_scores, uint32_t *out_counts, int *rc) {
    static const int min_scan = getenv("CX_LISTS_FILTER_MIN") ? atoi(getenv("CX_LISTS_FILTER_MIN")) : 256;
    const uint32_t n_rows = (uint32_t)ix->n_rows;
    if ((int64_t)n_scan < min_scan || ix->dim % 64 != 0 || ix->dim == 0 || ix->n_alive < 4ull * topk || topk > TOPK_MAX) return false;
    hipStream_t s = c->stream;
    auto fail = [&](int e) { *rc = e; return true; };
    // 1. tau from a sample
    const uint32_t S = std::min<uint32_t>(64u, n_scan);
    std::vector<uint32_t> samp(S);
    for (uint32_t i = 0; i < S; i++) {
        const uint64_t pos = (uint64_t)i * n_scan / S;
        samp[i] = scan_rows ? scan_rows[pos] : (uint32_t)pos;
    }
    if (int e = ensure_dev(ps.d_w, ps.c_w, (size_t)S * ix->dim)) return fail(e);
    if (int e = ensure_dev(ps.d_scan, ps.c_scan, (size_t)std::max(S, n_scan))) return fail(e);
    if (int e = ensure_dev(ps.d_list_rows, ps.c_list_rows, (size_t)S * topk)) return fail(e);
    if (int e = ensure_dev(ps.d_list_scores, ps.c_list_scores, (size_t)S * topk)) return fail(e);
    if (int e = ensure_dev(ps.d_list_dists, ps.c_list_dists, (size_t)S * topk)) return fail(e);
    if (int e = ensure_dev(ps.d_list_cnt, ps.c_list_cnt, (size_t)S)) return fail(e);
    if (hipMemcpyAsync(ps.d_scan, samp.data(), (size_t)S * 4, hipMemcpyHostToDevice, s) != hipSuccess) return fail(set_err(CX_ERR_DEVICE, "copy failed"));
    if (int e = (ix->dtype == 1 ? launch_gather_rows(ix->rows16(), ps.d_w, ps.d_scan, S, ix->dim, s) : launch_gather_rows(ix->d_rows, ps.d_w, ps.d_scan, S, ix->dim, s))) return fail(e);
    DevFilter flt;
    memset(&flt, 0, sizeof flt);
    flt.meta = ix->d_meta;
    flt.agent = ix->d_agent;
    if (int e = search_core(ix, c, ps.d_w, nullptr, S, topk, flt, 0.0f, false, ps.d_list_rows, ps.d_list_scores, ps.d_list_dists,
                            ps.d_list_cnt, s))
        return fail(e);
    std::vector<float> sc((size_t)S * topk);
    std::vector<uint32_t> cn(S);
    if (hipMemcpyAsync(sc.data(), ps.d_list_scores, sc.size() * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(cn.data(), ps.d_list_cnt, (size_t)S * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return fail(set_err(CX_ERR_DEVICE, "copy failed"));
    float tau = 2.0f;
    for (uint32_t i = 0; i < S; i++) {
        if (ix->h_meta[samp[i]] & META_REMOVED) continue;
        if (cn[i] < topk) return false;                 // fewer than k comparable rows: nothing for a threshold to do
        const float kth = sc[(size_t)i * topk + (topk - 1)];
        if (!(kth == kth)) return false;
        tau = std::min(tau, kth);
    }
    static const float margin = getenv("CX_LISTS_MARGIN") ? (float)atof(getenv("CX_LISTS_MARGIN")) : 0.003f;
    tau -= margin;
    // a k-th best score near 0 is a clamped non-positive cosine: no useful bound, and thr - eps <= 0 would make every pair a candidate
    if (!(tau >= 0.05f) || tau > 1.0f) {
        if (getenv("CX_LISTS_DIAG")) fprintf(stderr, "[lists] sampled threshold %.4f: no use, batched search instead\n", tau);
        return false;
    }
    // 2. the pass, lists only
    CycleInputs cyc;
    cyc.lists_only = true;
    cyc.min_count = topk;
    // at a k-th-best threshold (far below a link threshold) the filter's rigorous bf16 slack (0.008) lets several times k
    // candidates through per row: wider candidate lists than the link passes' 512, unless the environment pins them
    static const int lcap = getenv("CX_LISTS_CAND_CAP") ? atoi(getenv("CX_LISTS_CAND_CAP")) : 1024;
    if (!getenv("CX_PAIR_CAND_CAP")) cyc.cand_cap = (uint32_t)std::max(lcap, 16);
    uint64_t n_exact = 0;
    if (int e = pass_core(ix, c, ps, n_scan, scan_rows, topk, tau, 0, nullptr, false, cyc, &n_exact, nullptr)) return fail(e);
    if (getenv("CX_LISTS_DIAG")) fprintf(stderr, "[lists] tau %.4f, %llu of %u lists from the exact path\n", tau, (unsigned long long)n_exact, n_scan);
    // 3. out
    if (hipMemcpyAsync(out_rows, ps.d_list_rows, (size_t)n_scan * topk * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(out_scores, ps.d_list_scores, (size_t)n_scan * topk * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(out_counts, ps.d_list_cnt, (size_t)n_scan * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return fail(set_err(CX_ERR_DEVICE, "copy failed"));
    if (int e = check_result_block(out_counts, out_rows, n_scan, topk, topk, n_rows)) {
        memset(out_counts, 0, (size_t)n_scan * 4);
        return fail(e);
    }
    for (uint32_t i = 0; i < n_scan; i++) {
        const uint32_t r = scan_rows ? scan_rows[i] : i;
        if (ix->h_meta[r] & META_REMOVED) out_counts[i] = 0;   // no embedding, no list (auto_linker.rs:217-218)
    }
    *rc = CX_OK;
    return true;
}

}  // namespace

extern "C" {

/* Ordered top-k lists of the scanned rows (see cortex_hip.h): for many scanned rows the all-pairs machinery
 * (lists_by_filter above); else gather the scanned vectors on the device, run the batched search over them block by
 * block, copy the lists out. */
int cx_topk_lists_rows(const cx_index *ix, uint64_t n_scan64, const uint32_t *scan_rows, uint64_t topk64,
                       uint32_t *out_rows, float *out_scores, uint32_t *out_counts) try {
    if (!ix || !out_rows || !out_scores || !out_counts) return set_err(CX_ERR_VALIDATION, "null argument");
    if (topk64 == 0 || topk64 > TOPK_MAX) return set_err(CX_ERR_VALIDATION, "topk lists: topk must be in 1..%u", TOPK_MAX);
    if (int rc = use_device(ix)) return rc;
    const uint32_t n_rows = (uint32_t)ix->n_rows, topk = (uint32_t)topk64;
    const uint64_t n_scan = scan_rows ? n_scan64 : n_rows;
    if (n_scan > 0xFFFFFFF0ull) return set_err(CX_ERR_VALIDATION, "too many scanned rows");
    for (uint64_t i = 0; scan_rows && i < n_scan; i++)
        if (scan_rows[i] >= n_rows) return set_err(CX_ERR_VALIDATION, "scan row %u out of range", scan_rows[i]);
    if (!n_scan) return CX_OK;
    CtxLease lease(ix);
    if (!lease.c) return CX_ERR_DEVICE;
    Ctx *c = lease.c;
    hipStream_t s = c->stream;
    PassScratch &ps = scratch_of(c);
    const uint32_t k_eff = std::min<uint32_t>(topk, n_rows);
    {
        int rc = CX_OK;
        if (lists_by_filter(ix, c, ps, (uint32_t)n_scan, scan_rows, topk, out_rows, out_scores, out_counts, &rc)) return rc;
    }
    const uint32_t BLOCK = 16384;
    const uint32_t blk = (uint32_t)std::min<uint64_t>(BLOCK, n_scan);
    // scratch: gathered vectors (d_w), their row indices (d_scan), lists (d_list_*)
    if (int rc = ensure_dev(ps.d_w, ps.c_w, (size_t)blk * ix->dim)) return rc;
    if (int rc = ensure_dev(ps.d_scan, ps.c_scan, (size_t)blk)) return rc;
    if (int rc = ensure_dev(ps.d_list_rows, ps.c_list_rows, (size_t)blk * topk)) return rc;
    if (int rc = ensure_dev(ps.d_list_scores, ps.c_list_scores, (size_t)blk * topk)) return rc;
    if (int rc = ensure_dev(ps.d_list_dists, ps.c_list_dists, (size_t)blk * topk)) return rc;
    if (int rc = ensure_dev(ps.d_list_cnt, ps.c_list_cnt, (size_t)blk)) return rc;
    DevFilter flt;
    memset(&flt, 0, sizeof flt);
    flt.meta = ix->d_meta;
    flt.agent = ix->d_agent;
    std::vector<uint32_t> ident;
    for (uint64_t lo = 0; lo < n_scan; lo += blk) {
        const uint32_t m = (uint32_t)std::min<uint64_t>(blk, n_scan - lo);
        const float *d_q;
        if (scan_rows) {
            CX_HIP(hipMemcpyAsync(ps.d_scan, scan_rows + lo, (size_t)m * 4, hipMemcpyHostToDevice, s));
            if (int rc = (ix->dtype == 1 ? launch_gather_rows(ix->rows16(), ps.d_w, ps.d_scan, m, ix->dim, s) : launch_gather_rows(ix->d_rows, ps.d_w, ps.d_scan, m, ix->dim, s))) return rc;
            d_q = ps.d_w;
        } else if (ix->dtype == 1) {   // bf16 store: the block of rows, expanded
            if (int rc = launch_gather_rows(ix->rows16() + (size_t)lo * ix->dim, ps.d_w, nullptr, m, ix->dim, s)) return rc;
            d_q = ps.d_w;
        } else {
            d_q = ix->d_rows + (size_t)lo * ix->dim;   // every row in order: the store itself is the query block
        }
        if (k_eff == 0) CX_HIP(hipMemsetAsync(ps.d_list_cnt, 0, (size_t)m * 4, s));
        else if (int rc = search_core(ix, c, d_q, nullptr, m, k_eff, flt, 0.0f, false, ps.d_list_rows, ps.d_list_scores,
                                      ps.d_list_dists, ps.d_list_cnt, s))
            return rc;
        // lists are k_eff wide on the device, topk wide for the caller
        CX_HIP(hipMemcpy2DAsync(out_rows + (size_t)lo * topk, (size_t)topk * 4, ps.d_list_rows, (size_t)std::max(k_eff, 1u) * 4,
                                (size_t)k_eff * 4, m, hipMemcpyDeviceToHost, s));
        CX_HIP(hipMemcpy2DAsync(out_scores + (size_t)lo * topk, (size_t)topk * 4, ps.d_list_scores, (size_t)std::max(k_eff, 1u) * 4,
                                (size_t)k_eff * 4, m, hipMemcpyDeviceToHost, s));
        CX_HIP(hipMemcpyAsync(out_counts + lo, ps.d_list_cnt, (size_t)m * 4, hipMemcpyDeviceToHost, s));
        CX_HIP(hipStreamSynchronize(s));
        if (int rc = check_result_block(out_counts + lo, out_rows + (size_t)lo * topk, m, topk, k_eff, n_rows)) {
            memset(out_counts + lo, 0, (size_t)m * 4);
            return rc;
        }
        // a scanned row that was removed from the index has no embedding: like the passes, it gets no list
        // (auto_linker.rs:217-218)
        for (uint32_t i = 0; i < m; i++) {
            const uint32_t r = scan_rows ? scan_rows[lo + i] : (uint32_t)(lo + i);
            if (ix->h_meta[r] & META_REMOVED) out_counts[lo + i] = 0;
        }
    }
    return CX_OK;
} catch (...) { return cx::on_exception(); }

/* rows [row_lo, row_lo + n) of the shard copied to a caller buffer in HBM (e.g. to broadcast them to the other
 * ranks as the scanned block of a sharded pass) */
int cx_copy_rows_dev(const cx_index *ix, uint64_t row_lo, uint64_t n, float *d_dst, void *stream) try {
    if (!ix || !d_dst) return set_err(CX_ERR_VALIDATION, "null argument");
    if (row_lo + n > ix->n_rows) return set_err(CX_ERR_VALIDATION, "rows [%llu, %llu) out of range", (unsigned long long)row_lo, (unsigned long long)(row_lo + n));
    if (int rc = use_device(ix)) return rc;
    if (n && ix->dtype == 1) return launch_gather_rows(ix->rows16() + (size_t)row_lo * ix->dim, d_dst, nullptr, (uint32_t)n, ix->dim, (hipStream_t)stream);
    if (n) CX_HIP(hipMemcpyAsync(d_dst, ix->d_rows + (size_t)row_lo * ix->dim, (size_t)n * ix->dim * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CX_OK;
} catch (...) { return cx::on_exception(); }

}  // extern "C"
