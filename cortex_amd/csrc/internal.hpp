// internal.hpp — host-side structures shared by the C-ABI translation units (not installed).
#pragma once

#include <algorithm>
#include <functional>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.hpp"

namespace cx {

struct IdKey {
    uint64_t a, b;
    bool operator==(const IdKey &o) const { return a == o.a && b == o.b; }
};
struct IdHash {
    size_t operator()(const IdKey &k) const {
        uint64_t h = k.a * 0x9E3779B97F4A7C15ull ^ (k.b + 0xC2B2AE3D27D4EB4Full + (k.a << 6) + (k.a >> 2));
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        return (size_t)(h ^ (h >> 32));
    }
};
inline IdKey id_key(const uint8_t *id) {
    IdKey k;
    memcpy(&k.a, id, 8);
    memcpy(&k.b, id + 8, 8);
    return k;
}

template <typename T>
inline int ensure_dev(T *&p, size_t &cap, size_t need) {
    if (need <= cap) return CX_OK;
    size_t ncap = std::max(need, cap * 2);
    if (p) CX_HIP(hipFree(p));
    p = nullptr;
    cap = 0;
    CX_HIP(hipMalloc((void **)&p, ncap * sizeof(T)));
    cap = ncap;
    return CX_OK;
}
template <typename T>
inline int ensure_pinned(T *&p, size_t &cap, size_t need) {
    if (need <= cap) return CX_OK;
    size_t ncap = std::max(need, cap * 2);
    if (p) CX_HIP(hipHostFree(p));
    p = nullptr;
    cap = 0;
    CX_HIP(hipHostMalloc((void **)&p, ncap * sizeof(T), hipHostMallocDefault));
    cap = ncap;
    return CX_OK;
}

// Per-call scratch.  Host-API calls take one from the pool for the duration
// of the call (re-entrancy under the callers' read lock); *_dev calls get the
// one bound to their stream, so back-to-back calls on a stream reuse it in
// stream order.
struct Ctx {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    float *d_query = nullptr; size_t q_cap = 0;
    uint64_t *d_part_keys = nullptr; size_t pk_cap = 0;
    float *d_part_sims = nullptr; size_t ps_cap = 0;
    uint32_t *d_gslots = nullptr; size_t gs_cap = 0;   // batched search: cross-block bound slots [64][32]
    float *d_dense = nullptr; size_t dn_cap = 0;       // batchg: dense cosines [64][rows]
    char *d_qimg = nullptr; size_t qi_cap = 0;         // batchg: split query images (+ 64 floats of |q|^2 behind them)
    uint64_t *d_cand_keys = nullptr; size_t ck_cap = 0; // batchg filter mode: per-block candidate lists [64][grid][cb]
    float *d_cand_sims = nullptr; size_t cs_cap = 0;
    uint32_t *d_bg_ctl = nullptr; size_t bc_cap = 0;   // [64] bounds + [1] overflow flag
    uint32_t *d_bs_ctl = nullptr; size_t bsc_cap = 0;  // batchs: bound slots, published bounds, list lengths, tile counter (BS_CTL_WORDS; zero between passes)
    uint32_t *d_bs_rows = nullptr; size_t bsr_cap = 0; // batchs: candidate lists [queries of a pass][cap]: row, approximate then exact cosine
    float *d_bs_cos = nullptr; size_t bss_cap = 0;
    // a call of several passes runs them on two side streams in turn, each with its own control block and lists: pass i + 1's
    // prologue and first tiles under pass i's tail, re-score and selection (what a stream of batches gains, inside one call)
    hipStream_t bs_aux[2] = {nullptr, nullptr};
    hipEvent_t bs_ev_in = nullptr, bs_ev_done[2] = {nullptr, nullptr};
    uint32_t *d_bs_ctl2 = nullptr; size_t bsc2_cap = 0;
    uint32_t *d_bs_rows2 = nullptr; size_t bsr2_cap = 0;
    float *d_bs_cos2 = nullptr; size_t bss2_cap = 0;
    uint32_t *d_out_rows = nullptr; size_t or_cap = 0;
    float *d_out_scores = nullptr; size_t os_cap = 0;
    float *d_out_dists = nullptr; size_t od_cap = 0;
    uint32_t *d_out_counts = nullptr; size_t oc_cap = 0;
    uint32_t *d_excl = nullptr; size_t ex_cap = 0;
    uint32_t *d_kinds = nullptr; size_t kd_cap = 0;
    uint64_t *d_keys = nullptr; size_t k1_cap = 0;
    uint64_t *d_keys2 = nullptr; size_t k2_cap = 0;
    float *d_sims = nullptr; size_t s1_cap = 0;
    float *d_sims2 = nullptr; size_t s2_cap = 0;
    char *d_temp = nullptr; size_t tmp_cap = 0;
    float *h_query = nullptr; size_t hq_cap = 0;
    uint32_t *h_rows = nullptr; size_t hr_cap = 0;
    float *h_scores = nullptr; size_t hs_cap = 0;
    float *h_dists = nullptr; size_t hd_cap = 0;
    uint32_t *h_counts = nullptr; size_t hc_cap = 0;
    void *pass_scratch = nullptr;              // autolink.cpp's PassScratch (grow-only), freed through the hook
    void (*pass_scratch_free)(void *) = nullptr;

    ~Ctx() {
        if (pass_scratch && pass_scratch_free) pass_scratch_free(pass_scratch);
        (void)hipFree(d_dense); (void)hipFree(d_qimg); (void)hipFree(d_cand_keys); (void)hipFree(d_cand_sims); (void)hipFree(d_bg_ctl);
        (void)hipFree(d_bs_ctl); (void)hipFree(d_bs_rows); (void)hipFree(d_bs_cos);
        (void)hipFree(d_bs_ctl2); (void)hipFree(d_bs_rows2); (void)hipFree(d_bs_cos2);
        for (int i = 0; i < 2; i++) {
            if (bs_aux[i]) (void)hipStreamDestroy(bs_aux[i]);
            if (bs_ev_done[i]) (void)hipEventDestroy(bs_ev_done[i]);
        }
        if (bs_ev_in) (void)hipEventDestroy(bs_ev_in);
        (void)hipFree(d_query); (void)hipFree(d_gslots); (void)hipFree(d_part_keys); (void)hipFree(d_part_sims); (void)hipFree(d_out_rows);
        (void)hipFree(d_out_scores); (void)hipFree(d_out_dists); (void)hipFree(d_out_counts); (void)hipFree(d_excl);
        (void)hipFree(d_kinds); (void)hipFree(d_keys); (void)hipFree(d_keys2); (void)hipFree(d_sims); (void)hipFree(d_sims2);
        (void)hipFree(d_temp);
        (void)hipHostFree(h_query); (void)hipHostFree(h_rows); (void)hipHostFree(h_scores); (void)hipHostFree(h_dists);
        (void)hipHostFree(h_counts);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

}  // namespace cx

using namespace cx;

// what apply_score_decay reads of a node (vector/scoring.rs:84-114), per row, host side (decay.cpp)
struct NodeStats {
    int64_t last_s = 0;     // last_accessed_at: seconds since the epoch (default = the epoch, types.rs:56)
    uint32_t last_ns = 0;
    uint32_t kind = 0;      // interned NodeKind (cx_intern)
    uint64_t access = 0;    // access_count
};

struct cx_index {
    uint32_t dim = 0;
    int device = 0;
    // Row store.  dtype CX_DTYPE_F32: f32 [cap][dim].  CX_DTYPE_BF16 (cx_create_ex): the SAME pointer holds bf16 [cap][dim]
    // — every inserted row rounded to nearest even once, at insert — and every consumer goes through rows32() / rows16()
    // below (exactly one of them is non-null), sizes through elem.
    int dtype = 0;
    uint32_t elem = 4;
    float *d_rows = nullptr;
    const float *rows32() const { return dtype == 0 ? d_rows : nullptr; }
    const uint16_t *rows16() const { return dtype == 1 ? reinterpret_cast<const uint16_t *>(d_rows) : nullptr; }
    uint16_t *rows16_mut() { return dtype == 1 ? reinterpret_cast<uint16_t *>(d_rows) : nullptr; }
    char *row_bytes_ptr(uint64_t row) { return reinterpret_cast<char *>(d_rows) + (size_t)row * dim * elem; }
    float *d_stage = nullptr; size_t stage_cap = 0;   // bf16 stores: host rows land here as f32 before they are rounded
    uint32_t *d_meta = nullptr;
    uint32_t *d_agent = nullptr;
    uint64_t cap = 0;
    uint64_t n_rows = 0;
    uint64_t n_alive = 0;
    uint64_t n_removed = 0;
    std::vector<uint8_t> ids;
    std::vector<uint32_t> h_meta, h_agent;
    std::vector<NodeStats> h_stats;   // empty until cx_set_node_stats_batch; rows beyond its size read as defaults
    std::unordered_map<IdKey, uint32_t, IdHash> map;
    // string -> code table of kinds / agents.  cx_intern (&mut self paths) adds, cx_lookup (&self: filters) only reads;
    // intern_mu makes the two safe against each other whatever lock the caller holds.
    std::unordered_map<std::string, uint32_t> interned;
    mutable std::mutex intern_mu;
    // set_metadata for an id that has no vector yet (the reference keeps metadata in a map of its own,
    // vector/index.rs:219-222; its integration test sets it before insert, vector/tests.rs:65-66): (kind, agent)
    // codes waiting for the id's row; applied by the upsert that creates the row, dropped by cx_remove (:318)
    std::unordered_map<IdKey, std::pair<uint32_t, uint32_t>, IdHash> pending_meta;
    hipStream_t up_stream = nullptr;
    mutable std::mutex mu;
    mutable std::vector<Ctx *> pool;
    mutable std::unordered_map<void *, Ctx *> by_stream;
    // bf16 L2-normalised shadow of the rows for the all-pairs pass (allpairs.hip); built lazily
    // under shadow_mu, rows [0, shadow_rows) valid, in-place upserts listed in shadow_stale
    mutable std::mutex shadow_mu;
    mutable uint16_t *d_shadow = nullptr;
    // the same shadow cut into the 256-tile filter kernel's LDS-DMA pieces: [16-row block][K-step of 32][16 rows x 64 B,
    // pieces pre-swizzled] — one DMA instruction = 1 KiB of contiguous HBM/L2 (allpairs_p.hip); maintained with d_shadow
    mutable uint16_t *d_shadow_t = nullptr;
    mutable uint32_t *d_shadow_err = nullptr;   // [0] the largest || bf16(x) - x || over the shadow's rows (f32 bits; an upper bound: never lowered by removals); [1] irregular rows counted so far
    // IRREGULAR rows (kernels.hpp: bs_regular): zero shadow rows, listed here by the shadow's build; every screening path adds them
    // to its candidates and computes them with the reference's arithmetic.  irr_n = the count as of the last build (read back
    // at the sync the build ends with); more than BS_IRR_CAP of them switch the screening paths off (irr_over) until a rebuild.
    mutable uint32_t *d_irr_rows = nullptr;     // [BS_IRR_CAP]
    mutable uint32_t irr_n = 0;
    mutable bool irr_over = false;
    mutable uint64_t shadow_cap = 0;
    mutable uint64_t shadow_rows = 0;
    mutable std::vector<uint32_t> shadow_stale;
    // |row|^2 per row for the batched search (batch.hip reads it instead of re-summing every row per batch);
    // same lazy scheme as the shadow: rows [0, norms_rows) valid, in-place upserts listed in norms_stale
    mutable std::mutex norms_mu;
    mutable float *d_norms = nullptr;
    mutable uint64_t norms_cap = 0;
    mutable uint64_t norms_rows = 0;
    // LOSSY rows: a row whose squares all underflow to 0 while it is not the zero row (elements below ~1e-23, down to f32 denormals).
    // The reference divides its tiny dot by 0: +-inf, score 1.0 or 0.0 (vector/index.rs:172-177).  batch.hip / batchg.hip split rows
    // into bf16 terms, where an f32 denormal is 0 and the dot comes out 0 -> NaN: a store that holds such a row (found while the
    // norms are taken; sticky until they are all retaken) takes the per-query scans — plain f32, denormals kept — for its batches.
    mutable uint32_t *d_norms_lossy = nullptr;
    mutable uint32_t norms_lossy = 0;
    mutable std::vector<uint32_t> norms_stale;
    // bf16 hi/lo split copy of the rows for the batched search (same bytes as the f32 rows, tile-image layout);
    // maintained together with the norms (same validity prefix and stale list)
    mutable char *d_split = nullptr;
    mutable uint32_t *d_tile_list = nullptr;   // live tiles of the symmetric all-pairs pass, cached per row count
    mutable uint32_t tile_list_rows = 0, tile_list_n = 0, tile_list_big = 0;
    // cx_autolink_filter_profile: the filter GEMM of the last timed all-pairs pass (under shadow_mu)
    mutable double filter_prof[5] = {0, 0, 0, 0, 0};
    // measurement (cx_profile_*): event pairs around the scan kernel
    bool profiling = false;
    mutable std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    double prof_ms = 0.0;
    uint64_t prof_n = 0;
};


namespace cx {
int use_device(const cx_index *ix);
Ctx *acquire_ctx(const cx_index *ix);
void release_ctx(const cx_index *ix, Ctx *c);
struct CtxLease {
    const cx_index *ix;
    Ctx *c;
    explicit CtxLease(const cx_index *i) : ix(i), c(acquire_ctx(i)) {}
    ~CtxLease() { if (c) release_ctx(ix, c); }
    CtxLease(const CtxLease &) = delete;
    CtxLease &operator=(const CtxLease &) = delete;
};
// counts[i] <= k_max and every listed row < n_rows, or CX_ERR_DEVICE: guards every host-side use of a result block
// read back from the device (rows of list i start at rows + i * stride)
int check_result_block(const uint32_t *counts, const uint32_t *rows, uint64_t nq, uint64_t stride, uint64_t k_max,
                       uint64_t n_rows);
// where the start-up bulk load (nodes.cpp) puts what it decoded: one index, or the shards of a cx_sharded
struct BulkSink {
    uint32_t dim = 0;
    int device = 0;     // for the pinned staging buffer
    std::function<int(uint64_t, const uint8_t *, const float *)> upsert;
    std::function<uint32_t(const char *, uint64_t)> intern;
    std::function<int(uint64_t, const uint8_t *, const uint32_t *, const int64_t *, const uint32_t *, const uint64_t *)> set_stats;
    std::function<int(uint64_t, const uint8_t *, const uint32_t *, const uint32_t *)> set_meta;
};
int bulk_load_impl(const BulkSink &sink, uint64_t n, const uint8_t *blob, const uint64_t *offsets, uint32_t flags, ::cx_bulk_stats *stats);

// ---- the reference's index file (vector/index.rs:437-473), shared by the single index and the sharded handle (persist.cpp)
struct IndexFileWriter {
    FILE *f;
    bool ok = true;
    void bytes(const void *p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
    void u64(uint64_t v) { bytes(&v, 8); }  // host is little endian (x86-64)
    void str(const std::string &s) { u64(s.size()); bytes(s.data(), s.size()); }
    void uuid(const uint8_t *id) { u64(16); bytes(id, 16); }
};
struct IndexFileMeta { uint8_t id[16]; std::string kind, agent; };
// writes the tuple: n_alive vectors through write_vectors (uuid, u64 dim, dim x f32 each), the metadata map, the dimension
int save_index_file(const char *path, uint32_t dim, uint64_t n_alive, const std::function<int(IndexFileWriter &)> &write_vectors,
                    const std::vector<IndexFileMeta> &metas);
// where a loaded file goes: one index or a sharded handle.  create() is called once the row width is known.
struct IndexFileSink {
    std::function<int(uint64_t dim, uint64_t n_vec)> create;
    std::function<int(uint64_t n, const uint8_t *ids, const float *rows, uint64_t dim)> upsert;
    std::function<uint32_t(const char *, uint64_t)> intern;
    std::function<int(const uint8_t *id, uint32_t kind, uint32_t agent)> set_meta;
};
int load_index_file(const char *path, const IndexFileSink &sink);   // on failure the caller destroys what create() made
int read_rows_host(const cx_index *ix, uint64_t r0, uint64_t m, float *dst, std::vector<uint16_t> &tmp16);
void collect_metas(const cx_index *ix, std::vector<IndexFileMeta> &out);

struct FilterUpload {
    DevFilter f;
    bool needs_sync = false;
};
// VectorFilter -> its device view in c's scratch (index.cpp); exclude ids that are not in the index are dropped
int build_filter(const cx_index *ix, Ctx *c, const cx_filter *filter, hipStream_t s, FilterUpload &out);
// host queries -> c->d_query (dim floats each, zero-padded / truncated; tails[i] = sum of squares beyond dim), on c->stream
int stage_queries(const cx_index *ix, Ctx *c, uint64_t nq, const float *queries, uint64_t len, std::vector<float> &tails);
bool use_nontemporal(const cx_index *ix);
// nq single-query scans (query i = d_queries + i*dim) enqueued on s; results at [i*k_eff, ...)
int ensure_norms(const cx_index *ix, hipStream_t s);   // index.cpp
int ensure_shadow(const cx_index *ix, hipStream_t s);  // autolink.cpp: the normalised bf16 shadow (linker passes, batched search)
int search_core(const cx_index *ix, Ctx *c, const float *d_queries, const float *tails, uint64_t nq, uint32_t k_eff,
                const DevFilter &flt, float thr, bool has_thr, uint32_t *d_rows, float *d_scores, float *d_dists,
                uint32_t *d_counts, hipStream_t s);
}  // namespace cx
