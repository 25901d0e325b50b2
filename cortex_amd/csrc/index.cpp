// index.cpp — host side of libcortex_hip: the embedding store (id <-> row map,
// tombstones, metadata codes, HBM row store) and the C ABI of
// include/cortex_hip.h.  The arithmetic lives in the .hip files; this file
// only moves bytes, keeps the maps and launches kernels.
//
// HBM layout per index (one index = one shard on one device):
//   d_rows  f32 [cap][dim]  row-major, rows in insertion order (append-only;
//                            an upsert of a known id rewrites its row in place)
//   d_meta  u32 [cap]       bit0 removed, bit1 has-metadata, bits 8.. kind code
//   d_agent u32 [cap]       interned source_agent code
// Host: ids (16 B per row), id -> row hash map, mirrors of meta/agent.
#include "internal.hpp"

namespace cx {

static thread_local char g_err[512];
char *err_buf() { return g_err; }
int set_err(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int on_exception() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return set_err(CX_ERR_DEVICE, "out of host memory");
    } catch (const std::exception &e) {
        return set_err(CX_ERR_DEVICE, "internal error: %s", e.what());
    } catch (...) {
        return set_err(CX_ERR_DEVICE, "internal error");
    }
}

}  // namespace cx

using namespace cx;

namespace cx {

int use_device(const cx_index *ix) {
    CX_HIP(hipSetDevice(ix->device));
    return CX_OK;
}

int check_result_block(const uint32_t *counts, const uint32_t *rows, uint64_t nq, uint64_t stride, uint64_t k_max,
                       uint64_t n_rows) {
    for (uint64_t i = 0; i < nq; i++) {
        const uint32_t cnt = counts[i];
        if (cnt > k_max)
            return set_err(CX_ERR_DEVICE, "device result block is corrupt: list %llu holds %u entries, at most %llu possible",
                           (unsigned long long)i, cnt, (unsigned long long)k_max);
        for (uint32_t j = 0; j < cnt; j++)
            if (rows[i * stride + j] >= n_rows)
                return set_err(CX_ERR_DEVICE, "device result block is corrupt: list %llu entry %u names row %u of %llu",
                               (unsigned long long)i, j, rows[i * stride + j], (unsigned long long)n_rows);
    }
    return CX_OK;
}

Ctx *acquire_ctx(const cx_index *ix) {
    {
        std::lock_guard<std::mutex> g(ix->mu);
        if (!ix->pool.empty()) {
            Ctx *c = ix->pool.back();
            ix->pool.pop_back();
            return c;
        }
    }
    Ctx *c = new Ctx();
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        set_err(CX_ERR_DEVICE, "hipStreamCreate failed");
        return nullptr;
    }
    c->own_stream = true;
    return c;
}
void release_ctx(const cx_index *ix, Ctx *c) {
    std::lock_guard<std::mutex> g(ix->mu);
    ix->pool.push_back(c);
}
}  // namespace cx

namespace {

Ctx *ctx_for_stream(const cx_index *ix, hipStream_t s) {
    std::lock_guard<std::mutex> g(ix->mu);
    auto it = ix->by_stream.find((void *)s);
    if (it != ix->by_stream.end()) return it->second;
    Ctx *c = new Ctx();
    c->stream = s;
    c->own_stream = false;
    ix->by_stream[(void *)s] = c;
    return c;
}

int grow_rows(cx_index *ix, uint64_t need) {
    if (need <= ix->cap) return CX_OK;
    if (need >= 0xFFFFFFF0ull) return set_err(CX_ERR_VALIDATION, "row store limited to 2^32-16 rows per shard");
    uint64_t ncap = std::max<uint64_t>(need, std::max<uint64_t>(ix->cap * 2, 1024));
    float *nr = nullptr;
    uint32_t *nm = nullptr, *na = nullptr;
    CX_HIP(hipMalloc((void **)&nr, (ncap + 256) * ix->dim * ix->elem + 64));  // + one row tile of readable padding (batch.hip: 16 rows, batchg.hip: 256)
    CX_HIP(hipMalloc((void **)&nm, ncap * sizeof(uint32_t)));
    CX_HIP(hipMalloc((void **)&na, ncap * sizeof(uint32_t)));
    CX_HIP(hipMemsetAsync(nm, 0, ncap * sizeof(uint32_t), ix->up_stream));
    CX_HIP(hipMemsetAsync(na, 0, ncap * sizeof(uint32_t), ix->up_stream));
    if (ix->n_rows) {
        CX_HIP(hipMemcpyAsync(nr, ix->d_rows, ix->n_rows * ix->dim * ix->elem, hipMemcpyDeviceToDevice, ix->up_stream));
        CX_HIP(hipMemcpyAsync(nm, ix->d_meta, ix->n_rows * sizeof(uint32_t), hipMemcpyDeviceToDevice, ix->up_stream));
        CX_HIP(hipMemcpyAsync(na, ix->d_agent, ix->n_rows * sizeof(uint32_t), hipMemcpyDeviceToDevice, ix->up_stream));
    }
    CX_HIP(hipStreamSynchronize(ix->up_stream));
    if (ix->d_rows) CX_HIP(hipFree(ix->d_rows));
    if (ix->d_meta) CX_HIP(hipFree(ix->d_meta));
    if (ix->d_agent) CX_HIP(hipFree(ix->d_agent));
    ix->d_rows = nr;
    ix->d_meta = nm;
    ix->d_agent = na;
    ix->cap = ncap;
    return CX_OK;
}

int push_meta(cx_index *ix, uint32_t row) {
    CX_HIP(hipMemcpyAsync(ix->d_meta + row, &ix->h_meta[row], 4, hipMemcpyHostToDevice, ix->up_stream));
    CX_HIP(hipMemcpyAsync(ix->d_agent + row, &ix->h_agent[row], 4, hipMemcpyHostToDevice, ix->up_stream));
    CX_HIP(hipStreamSynchronize(ix->up_stream));
    return CX_OK;
}

// vector/index.rs:298-314 for n rows.  Runs of fresh ids land in consecutive
// rows and move with one copy each.
int upsert_impl(cx_index *ix, uint64_t n, const uint8_t *ids, const float *embs, uint64_t len, bool on_device) {
    if (!ix) return set_err(CX_ERR_VALIDATION, "null index");
    if (len != ix->dim)
        return set_err(CX_ERR_VALIDATION, "Embedding dimension mismatch: expected %u, got %llu", ix->dim,
                       (unsigned long long)len);
    if (!n) return CX_OK;
    if (!ids || !embs) return set_err(CX_ERR_VALIDATION, "null ids/embeddings");
    if (int rc = use_device(ix)) return rc;
    if (int rc = grow_rows(ix, ix->n_rows + n)) return rc;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const size_t row_bytes = (size_t)ix->dim * sizeof(float);
    // bf16 store: the f32 rows (staged on the device if they came from the host) are rounded into place, run by run
    const float *d_src = embs;
    if (ix->dtype == 1 && row_bytes) {
        if (!on_device) {
            const size_t need = (size_t)n * ix->dim;
            if (ix->stage_cap < need) {
                if (ix->d_stage) CX_HIP(hipFree(ix->d_stage));
                ix->d_stage = nullptr;
                ix->stage_cap = 0;
                CX_HIP(hipMalloc((void **)&ix->d_stage, need * sizeof(float)));
                ix->stage_cap = need;
            }
            CX_HIP(hipMemcpyAsync(ix->d_stage, embs, need * sizeof(float), hipMemcpyHostToDevice, ix->up_stream));
            d_src = ix->d_stage;
        }
    }
    auto put_rows = [&](uint64_t first_row, uint64_t src_row, uint64_t count) -> int {   // count rows of the batch -> rows of the store
        if (!row_bytes || !count) return CX_OK;
        if (ix->dtype == 1)
            return launch_gather_rows(d_src + src_row * len, ix->rows16_mut() + (size_t)first_row * ix->dim, nullptr, (uint32_t)count, ix->dim, ix->up_stream);
        CX_HIP(hipMemcpyAsync(ix->d_rows + (size_t)first_row * ix->dim, embs + src_row * len, count * row_bytes, kind, ix->up_stream));
        return CX_OK;
    };
    uint64_t i = 0;
    uint64_t meta_lo = UINT64_MAX, meta_hi = 0;   // new rows that took pending metadata: [meta_lo, meta_hi)
    while (i < n) {
        const IdKey key = id_key(ids + 16 * i);
        auto it = ix->map.find(key);
        if (it != ix->map.end()) {  // replace in place, row position kept
            if (it->second < ix->shadow_rows) ix->shadow_stale.push_back(it->second);
            if (it->second < ix->norms_rows) ix->norms_stale.push_back(it->second);
            if (int rc = put_rows(it->second, i, 1)) return rc;
            i++;
            continue;
        }
        // extend a run of fresh ids (stop at a known id or a repeat inside the run)
        const uint64_t run_start = i;
        const uint64_t first_row = ix->n_rows;
        while (i < n) {
            const IdKey kk = id_key(ids + 16 * i);
            if (ix->map.find(kk) != ix->map.end()) break;
            ix->map.emplace(kk, (uint32_t)ix->n_rows);
            ix->ids.insert(ix->ids.end(), ids + 16 * i, ids + 16 * i + 16);
            uint32_t meta0 = 0, agent0 = 0;
            if (!ix->pending_meta.empty()) {   // metadata that arrived before the vector (vector/tests.rs:65-66)
                auto pm = ix->pending_meta.find(kk);
                if (pm != ix->pending_meta.end()) {
                    meta0 = META_HAS | (pm->second.first << 8);
                    agent0 = pm->second.second;
                    ix->pending_meta.erase(pm);
                    meta_lo = std::min<uint64_t>(meta_lo, ix->n_rows);
                    meta_hi = std::max<uint64_t>(meta_hi, ix->n_rows + 1);
                }
            }
            ix->h_meta.push_back(meta0);
            ix->h_agent.push_back(agent0);
            ix->n_rows++;
            ix->n_alive++;
            i++;
        }
        if (int rc = put_rows(first_row, run_start, i - run_start)) return rc;
    }
    if (meta_lo < meta_hi) {
        CX_HIP(hipMemcpyAsync(ix->d_meta + meta_lo, &ix->h_meta[meta_lo], (size_t)(meta_hi - meta_lo) * 4, hipMemcpyHostToDevice, ix->up_stream));
        CX_HIP(hipMemcpyAsync(ix->d_agent + meta_lo, &ix->h_agent[meta_lo], (size_t)(meta_hi - meta_lo) * 4, hipMemcpyHostToDevice, ix->up_stream));
    }
    CX_HIP(hipStreamSynchronize(ix->up_stream));
    if (ix->stage_cap * sizeof(float) > ((size_t)256 << 20)) {   // a bulk load's staging area is not kept
        (void)hipFree(ix->d_stage);
        ix->d_stage = nullptr;
        ix->stage_cap = 0;
    }
    return CX_OK;
}

}  // namespace
namespace cx {
// VectorFilter -> device view.  Exclude ids that are not in the index cannot
// match any row and are dropped.
int build_filter(const cx_index *ix, Ctx *c, const cx_filter *filter, hipStream_t s, FilterUpload &out) {
    DevFilter &f = out.f;
    memset(&f, 0, sizeof f);
    f.meta = ix->d_meta;
    f.agent = ix->d_agent;
    f.trivial = (!filter && ix->n_removed == 0) ? 1u : 0u;
    if (!filter) return CX_OK;
    if (filter->has_exclude && filter->n_exclude) {
        if (!filter->exclude_ids) return set_err(CX_ERR_VALIDATION, "filter: exclude_ids is null");
        std::vector<uint32_t> rows;
        rows.reserve(filter->n_exclude);
        for (uint64_t i = 0; i < filter->n_exclude; i++) {
            auto it = ix->map.find(id_key(filter->exclude_ids + 16 * i));
            if (it != ix->map.end()) rows.push_back(it->second);
        }
        std::sort(rows.begin(), rows.end());
        if (!rows.empty()) {
            if (int rc = ensure_dev(c->d_excl, c->ex_cap, rows.size())) return rc;
            CX_HIP(hipMemcpyAsync(c->d_excl, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, s));
            CX_HIP(hipStreamSynchronize(s));  // rows is a local; the copy must finish before it dies
            f.exclude_rows = c->d_excl;
            f.n_exclude = (uint32_t)rows.size();
        }
    }
    if (filter->has_kinds) {
        f.has_kinds = 1;
        f.n_kinds = (uint32_t)filter->n_kinds;
        if (filter->n_kinds) {
            if (!filter->kind_codes) return set_err(CX_ERR_VALIDATION, "filter: kind_codes is null");
            if (int rc = ensure_dev(c->d_kinds, c->kd_cap, (size_t)filter->n_kinds)) return rc;
            CX_HIP(hipMemcpyAsync(c->d_kinds, filter->kind_codes, filter->n_kinds * 4, hipMemcpyHostToDevice, s));
            CX_HIP(hipStreamSynchronize(s));
            f.kind_codes = c->d_kinds;
        }
    }
    if (filter->has_agent) {
        f.has_agent = 1;
        f.agent_code = filter->agent_code;
    }
    return CX_OK;
}

bool use_nontemporal(const cx_index *ix) {
    static int forced = -1;
    if (forced == -1) {
        const char *e = getenv("CX_SCAN_NT");
        forced = e ? (atoi(e) ? 1 : 0) : 2;
    }
    if (forced != 2) return forced == 1;
    // rows larger than the 256 MiB Infinity Cache cannot stay resident between queries
    return (uint64_t)ix->n_rows * ix->dim * sizeof(float) > (256ull << 20);
}
}  // namespace cx
namespace {

}  // namespace
namespace cx {
// nq single-query scans enqueued on s; query i's results at [i*k_out, ...).
// threshold searches (has_thr) and k > TOPK_MAX take the dense+sort path.
// What the batched search reads besides the rows — |row|^2 and, for the dims its kernel serves, the bf16 hi/lo split
// copy of the store — kept lazily (see internal.hpp).  Work is enqueued on s and waited for under the mutex: another
// reader on another stream must not see norms_rows advance before the values exist.
int ensure_norms(const cx_index *ix, hipStream_t s) {
    std::lock_guard<std::mutex> g(ix->norms_mu);
    const uint64_t n = ix->n_rows;
    const bool want_split = ix->dtype == 0 && batch_supported(ix->dim, 1);   // a bf16 store is its own MFMA operand
    bool work = false;
    if (ix->norms_cap < n) {
        if (ix->d_norms) CX_HIP(hipFree(ix->d_norms));
        if (ix->d_split) CX_HIP(hipFree(ix->d_split));
        ix->d_norms = nullptr;
        ix->d_split = nullptr;
        ix->norms_cap = 0;
        const uint64_t cap = std::max<uint64_t>(n, ix->cap);
        CX_HIP(hipMalloc((void **)&ix->d_norms, (cap + 256) * sizeof(float)));   // + a row tile of readable padding (batchg.hip reads the norms of a whole 128- / 256-row tile)
        CX_HIP(hipMemsetAsync(ix->d_norms, 0, (cap + 256) * sizeof(float), s));
        if (want_split) {
            const size_t bytes = (size_t)((cap + 31) / 16) * 16 * ix->dim * sizeof(float);   // whole tiles
            CX_HIP(hipMalloc((void **)&ix->d_split, bytes));
            CX_HIP(hipMemsetAsync(ix->d_split, 0, bytes, s));
        }
        ix->norms_cap = cap;
        ix->norms_rows = 0;
        ix->norms_stale.clear();
        work = true;
    }
    if (!ix->d_norms_lossy) CX_HIP(hipMalloc((void **)&ix->d_norms_lossy, sizeof(uint32_t)));
    if (ix->norms_rows == 0) {   // every norm is (re)taken below: so is the verdict on lossy rows (internal.hpp)
        CX_HIP(hipMemsetAsync(ix->d_norms_lossy, 0, sizeof(uint32_t), s));
        ix->norms_lossy = 0;
    }
    auto refresh = [&](uint32_t lo, uint32_t hi) -> int {
        if (ix->dtype == 1) return launch_row_norms(ix->rows16(), ix->d_norms, lo, hi, ix->dim, ix->d_norms_lossy, s);
        if (int rc = launch_row_norms(ix->d_rows, ix->d_norms, lo, hi, ix->dim, ix->d_norms_lossy, s)) return rc;
        if (want_split)
            if (int rc = launch_build_split(ix->d_rows, ix->d_split, lo, hi, ix->dim, s)) return rc;
        return CX_OK;
    };
    for (uint32_t r : ix->norms_stale)
        if (r < ix->norms_rows) {
            if (int rc = refresh(r, r + 1)) return rc;
            work = true;
        }
    ix->norms_stale.clear();
    if (ix->norms_rows < n) {
        if (int rc = refresh((uint32_t)ix->norms_rows, (uint32_t)n)) return rc;
        ix->norms_rows = n;
        work = true;
    }
    if (work) {
        uint32_t lossy = 0;
        CX_HIP(hipMemcpyAsync(&lossy, ix->d_norms_lossy, sizeof lossy, hipMemcpyDeviceToHost, s));
        CX_HIP(hipStreamSynchronize(s));
        if (lossy) ix->norms_lossy = 1;
    }
    return CX_OK;
}

int search_core(const cx_index *ix, Ctx *c, const float *d_queries, const float *tails, uint64_t nq, uint32_t k_eff,
                const DevFilter &flt, float thr, bool has_thr, uint32_t *d_rows, float *d_scores, float *d_dists,
                uint32_t *d_counts, hipStream_t s) {
    const uint32_t n = (uint32_t)ix->n_rows;
    const bool nt = use_nontemporal(ix);
    const bool topk_path = !has_thr && k_eff <= TOPK_MAX;
    uint32_t grid = 0;
    if (topk_path) {
        grid = scan_grid_blocks(n, ix->dim, ix->dtype == 1);
        if (int rc = ensure_dev(c->d_part_keys, c->pk_cap, (size_t)grid * std::max(k_eff, 1u))) return rc;
        if (int rc = ensure_dev(c->d_part_sims, c->ps_cap, (size_t)grid * std::max(k_eff, 1u))) return rc;
    } else {
        if (int rc = ensure_dev(c->d_keys, c->k1_cap, n)) return rc;
        if (int rc = ensure_dev(c->d_keys2, c->k2_cap, n)) return rc;
        if (int rc = ensure_dev(c->d_sims, c->s1_cap, n)) return rc;
        if (int rc = ensure_dev(c->d_sims2, c->s2_cap, n)) return rc;
        if (int rc = ensure_dev(c->d_temp, c->tmp_cap, sort_temp_bytes(n))) return rc;
    }
    // batched pass: the row store is read once per <= 64 queries (batch.hip)
    bool no_tails = true;
    for (uint64_t i = 0; tails && i < nq; i++) no_tails = no_tails && tails[i] == 0.0f;
    static const int batch_min = getenv("CX_BATCH_MIN") ? atoi(getenv("CX_BATCH_MIN")) : 3;
    static const int b2_ok = getenv("CX_BATCH2") ? atoi(getenv("CX_BATCH2")) : 1;   // 0: 768-d through batchg.hip (tests)
    static const int bg_ok = getenv("CX_BATCHG") ? atoi(getenv("CX_BATCHG")) : 1;
    static const int filter_ok = getenv("CX_BATCHG_FILTER") ? atoi(getenv("CX_BATCHG_FILTER")) : 1;
    static const uint32_t filter_min = getenv("CX_BATCHG_FILTER_MIN") ? (uint32_t)atoi(getenv("CX_BATCHG_FILTER_MIN")) : 262144u;
    // wide lists (k > 32) on a large unfiltered store: batchg.hip's bound + candidates pass runs at 0.71-0.75 of the HBM
    // peak whatever k is, batch2_kernel's in-kernel wide lists at 0.40-0.49 (1.25M x 384 / 768, k = 100)
    // (1.25M rows, per step: k = 100 at 768 / 384-d 0.82 / 0.51 ms against 1.02 / 0.64; k = 32: 0.78 / 0.46 against 0.81 / 0.58; k = 20
    // at 384-d 0.44 against 0.48; k = 10: 0.77 / 0.45 against 0.65 / 0.43 — batch2's fused lists win while they are short)
    // Large stores, every width up to 1024 that is a multiple of 128, both store types: batchs.hip — a screening pass over the
    // normalised bf16 shadow the linker passes keep (half the bytes of the f32 rows, one MFMA per pair instead of three, a
    // rigorous error bound), the survivors re-scored exactly from the stored rows
    static const int bs_ok = getenv("CX_BATCHS") ? atoi(getenv("CX_BATCHS")) : 1;
    // (a pass costs ~50 us whatever the store's size and serves 64 queries: from 131,072 rows it wins for any number of queries; from
    // 32,768 rows for calls of up to 256 queries — 40k x 384, k = 10 / 100: 0.062 / 0.072 ms against 0.082 / 0.157; 100k x 768, k = 100:
    // 0.133 against 0.387 —, while many queries over a small store are better served by batch.hip's chunk x group launches)
    static const uint32_t bs_small_rows = getenv("CX_BATCHS_SMALL_ROWS") ? (uint32_t)atoi(getenv("CX_BATCHS_SMALL_ROWS")) : 32768u;
    static const uint64_t bs_small_nq = getenv("CX_BATCHS_SMALL_NQ") ? (uint64_t)atoll(getenv("CX_BATCHS_SMALL_NQ")) : 256u;
    const bool bs_rows_ok = n >= batchs_min_rows() || (n >= bs_small_rows && nq <= bs_small_nq);
    // The screening pass needs the shadow (0.5 x an f32 store, built on the first batched search — inside a `&self` call) and its
    // candidate scratch: when either cannot be had (out of device memory), or the store holds more irregular rows than the pass
    // carries along (kernels.hpp: BS_IRR_CAP), the search goes on below with the kernels that read the stored rows themselves.
    // CX_SINGLE_SCREENED=1 (read per call: a routed option, off by default — the single-query scan stays the contract's path):
    // one or two queries take the screening pass as well — half the bytes of an f32 store per query (1.536 GB instead of 3.072 at
    // 1M x 768), the same exact results; for callers that search node by node (the linker's per-node loop, the HTTP handler)
    const char *ss_env = getenv("CX_SINGLE_SCREENED");
    const bool single_screened = ss_env && atoi(ss_env) != 0 && ix->dtype == 0;
    bool bs_go = bs_ok && topk_path && no_tails && (nq >= (uint64_t)batch_min || (single_screened && nq >= 1)) && k_eff >= 1 && batchs_supported(ix->dim, k_eff) && bs_rows_ok;
    const uint32_t bs_cap = bs_go ? batchs_cand_cap(n, k_eff) : 0u;
    if (bs_go) {
        int rc = ensure_shadow(ix, s);
        if (!rc && ix->irr_over) bs_go = false;
        if (!rc && bs_go && c->bsc_cap < BS_CTL_WORDS) {
            rc = ensure_dev(c->d_bs_ctl, c->bsc_cap, (size_t)BS_CTL_WORDS);
            if (!rc && hipMemsetAsync(c->d_bs_ctl, 0, (size_t)BS_CTL_WORDS * sizeof(uint32_t), s) != hipSuccess) rc = CX_ERR_DEVICE;
        }
        // (calls of more than 64 queries at row widths up to 512 run 128 queries per pass: batchs.hip's two banks)
        const size_t bs_q = batchs_queries_per_pass(ix->dim, n, nq);
        if (!rc && bs_go) rc = ensure_dev(c->d_bs_rows, c->bsr_cap, bs_q * bs_cap);
        if (!rc && bs_go) rc = ensure_dev(c->d_bs_cos, c->bss_cap, bs_q * bs_cap);
        if (rc) {
            (void)hipGetLastError();   // (a failed hipMalloc is sticky until read)
            bs_go = false;
        }
    }
    if (bs_go) {
        // Several passes of one call (more queries than a pass serves): on two side streams in turn, each with its own control block
        // and candidate lists, joined back into s at the end — consecutive passes overlap like the batches of a stream do
        // (profiles/r04/tuning.md: +12-14 %).  One pass, or the diagnostics that stop the stream: everything on s.
        static const int pipe_env = getenv("CX_BATCHS_PIPE") ? atoi(getenv("CX_BATCHS_PIPE")) : 1;
        static const int bs_diag = getenv("CX_BATCHS_DIAG") ? atoi(getenv("CX_BATCHS_DIAG")) : 0;
        static const bool tl_env = getenv("CX_BATCHS_TL") && atoi(getenv("CX_BATCHS_TL")) != 0;
        const uint32_t qpp0 = batchs_queries_per_pass(ix->dim, n, nq);
        // (one-bank passes only: two two-bank passes in each other's way lose — 256 queries k = 100 x 1.25M x 384: 0.694 ms piped
        // against 0.625; 768-d, 1024-d and small stores gain 4-7 %)
        bool piped = pipe_env && !bs_diag && !tl_env && nq > qpp0 && qpp0 == 64u;
        if (piped) {
            int rc = CX_OK;
            for (int i = 0; i < 2 && !rc; i++) {
                if (!c->bs_aux[i] && hipStreamCreateWithFlags(&c->bs_aux[i], hipStreamNonBlocking) != hipSuccess) rc = CX_ERR_DEVICE;
                if (!rc && !c->bs_ev_done[i] && hipEventCreateWithFlags(&c->bs_ev_done[i], hipEventDisableTiming) != hipSuccess) rc = CX_ERR_DEVICE;
            }
            if (!rc && !c->bs_ev_in && hipEventCreateWithFlags(&c->bs_ev_in, hipEventDisableTiming) != hipSuccess) rc = CX_ERR_DEVICE;
            if (!rc && c->bsc2_cap < BS_CTL_WORDS) {
                rc = ensure_dev(c->d_bs_ctl2, c->bsc2_cap, (size_t)BS_CTL_WORDS);
                if (!rc && hipMemsetAsync(c->d_bs_ctl2, 0, (size_t)BS_CTL_WORDS * sizeof(uint32_t), s) != hipSuccess) rc = CX_ERR_DEVICE;
            }
            if (!rc) rc = ensure_dev(c->d_bs_rows2, c->bsr2_cap, (size_t)qpp0 * bs_cap);
            if (!rc) rc = ensure_dev(c->d_bs_cos2, c->bss2_cap, (size_t)qpp0 * bs_cap);
            if (!rc && hipEventRecord(c->bs_ev_in, s) != hipSuccess) rc = CX_ERR_DEVICE;   // (queries, filter, everything queued on s so far)
            for (int i = 0; i < 2 && !rc; i++)
                if (hipStreamWaitEvent(c->bs_aux[i], c->bs_ev_in, 0) != hipSuccess) rc = CX_ERR_DEVICE;
            if (rc) {   // (no second scratch set, no streams: the passes run one after the other on s)
                (void)hipGetLastError();
                piped = false;
            }
        }
        int rc_all = CX_OK;
        uint32_t pass_i = 0;
        for (uint64_t q0 = 0; q0 < nq && !rc_all; pass_i++) {
            const uint32_t m = (uint32_t)std::min<uint64_t>(batchs_queries_per_pass(ix->dim, n, nq - q0), nq - q0);
            const bool second = piped && (pass_i & 1u);
            hipStream_t st = piped ? c->bs_aux[pass_i & 1u] : s;
            BatchSArgs b;
            memset(&b, 0, sizeof b);
            b.shadow_t = ix->d_shadow_t;
            b.shadow_err = ix->d_shadow_err;
            b.queries = d_queries + q0 * ix->dim;
            b.rows = ix->rows32();
            b.rows16 = ix->rows16();
            b.n_rows = n;
            b.nq = m;
            b.dim = ix->dim;
            b.k = k_eff;
            b.flt = flt;
            b.ctl = second ? c->d_bs_ctl2 : c->d_bs_ctl;
            b.cand_rows = second ? c->d_bs_rows2 : c->d_bs_rows;
            b.cand_cos = second ? c->d_bs_cos2 : c->d_bs_cos;
            b.cap = bs_cap;
            b.irr_rows = ix->d_irr_rows;
            b.irr_n = ix->irr_n;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            int rc = CX_OK;
            if (ix->profiling) {
                if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = set_err(CX_ERR_DEVICE, "hipEventCreate failed");
                if (!rc) {
                    std::lock_guard<std::mutex> g(ix->mu);
                    ix->prof_events.emplace_back(e0, e1);
                }
                if (!rc && hipEventRecord(e0, st) != hipSuccess) rc = set_err(CX_ERR_DEVICE, "hipEventRecord failed");
            }
            if (!rc) rc = launch_batchs_pass(b, st);
            if (e1 && !rc && hipEventRecord(e1, st) != hipSuccess) rc = set_err(CX_ERR_DEVICE, "hipEventRecord failed");   // (no early return between a pass and its select: see below)
            if (bs_diag && !rc) {   // candidates per query and published bounds of this pass, on stderr
                std::vector<uint32_t> dg(2 * BS_MAXQ);   // bounds | list lengths
                (void)hipStreamSynchronize(st);
                (void)hipMemcpy(dg.data(), b.ctl + BS_CTL_BOUND, 2 * BS_MAXQ * sizeof(uint32_t), hipMemcpyDeviceToHost);
                uint64_t tot = 0; uint32_t mx = 0, nob = 0;
                for (uint32_t q = 0; q < m; q++) { tot += dg[BS_MAXQ + q]; mx = std::max(mx, dg[BS_MAXQ + q]); nob += dg[q] <= 1u; }
                float b0; memcpy(&b0, &dg[0], 4);
                fprintf(stderr, "[batchs diag] %u queries: %llu candidates (max %u per query), %u without a bound, bound[0] = %.4f\n", m, (unsigned long long)tot, mx, nob, b0);
            }
            if (!rc) rc = launch_batchs_select(b, d_rows + q0 * k_eff, d_scores + q0 * k_eff, d_dists + q0 * k_eff, d_counts + q0, st);
            if (rc) {   // a pass that did not run to its select leaves the control block in an unknown state
                (void)hipMemsetAsync(b.ctl, 0, (size_t)BS_CTL_WORDS * sizeof(uint32_t), st);
                rc_all = rc;
            }
            q0 += m;
        }
        if (piped) {   // s goes on behind both side streams (also after an error: nothing of this call may still be running behind the caller's back)
            for (int i = 0; i < 2; i++) {
                if (hipEventRecord(c->bs_ev_done[i], c->bs_aux[i]) != hipSuccess || hipStreamWaitEvent(s, c->bs_ev_done[i], 0) != hipSuccess) {
                    (void)hipStreamSynchronize(c->bs_aux[i]);
                    if (!rc_all) rc_all = set_err(CX_ERR_DEVICE, "joining the passes' side streams failed");
                }
            }
        }
        return rc_all;
    }
    const bool wide_to_bg = (k_eff > 32 || (ix->dim <= 384 && k_eff >= 20)) && bg_ok && filter_ok && n >= filter_min && batchg_supported(ix->dim, k_eff);
    const bool b2_go = b2_ok && ix->dtype == 0 && !wide_to_bg && topk_path && no_tails && nq >= (uint64_t)batch_min && k_eff >= 1 && batch_supported(ix->dim, k_eff);
    const bool bg_go = bg_ok && topk_path && no_tails && nq >= (uint64_t)batch_min && k_eff >= 1 && batchg_supported(ix->dim, k_eff);
    // both kernels split the rows into bf16 terms: a store with a LOSSY row (internal.hpp: squares all 0 under a non-zero row) takes
    // the per-query scans at the end of this function instead — the reference's +-inf from a denormal dot over a zero norm
    bool lossy_rows = false;
    if (b2_go || bg_go) {
        if (int rc = ensure_norms(ix, s)) return rc;
        std::lock_guard<std::mutex> g(ix->norms_mu);
        lossy_rows = ix->norms_lossy != 0;
    }
    if (b2_go && !lossy_rows) {
        const uint32_t qpp = batch_queries_per_pass(ix->dim, k_eff, nq);
        uint32_t bgrid = 1, groups = 1;
        batch_launch_shape(n, ix->dim, nq, k_eff, &bgrid, &groups);
        const uint64_t per_launch = (uint64_t)qpp * groups;
        if (int rc = ensure_dev(c->d_part_keys, c->pk_cap, (size_t)per_launch * bgrid * k_eff)) return rc;
        if (int rc = ensure_dev(c->d_part_sims, c->ps_cap, (size_t)per_launch * bgrid * k_eff)) return rc;
        if (int rc = ensure_dev(c->d_gslots, c->gs_cap, (size_t)groups * 64 * 128)) return rc;
        for (uint64_t q0 = 0; q0 < nq; q0 += per_launch) {
            const uint32_t m = (uint32_t)std::min<uint64_t>(per_launch, nq - q0);
            BatchArgs b;
            memset(&b, 0, sizeof b);
            b.rows = ix->d_rows;
            b.queries = d_queries + q0 * ix->dim;
            b.norms = ix->d_norms;
            b.split = ix->d_split;
            b.n_rows = n;
            b.nq = m;
            b.n_groups = (m + qpp - 1) / qpp;
            b.qpp = qpp;
            b.dim = ix->dim;
            b.k = k_eff;
            b.flt = flt;
            b.part_keys = c->d_part_keys;
            b.part_sims = c->d_part_sims;
            b.gslots = c->d_gslots;
            CX_HIP(hipMemsetAsync(c->d_gslots, 0, (size_t)b.n_groups * 64 * 128 * sizeof(uint32_t), s));
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (ix->profiling) {
                CX_HIP(hipEventCreate(&e0));
                CX_HIP(hipEventCreate(&e1));
                std::lock_guard<std::mutex> g(ix->mu);
                ix->prof_events.emplace_back(e0, e1);
                CX_HIP(hipEventRecord(e0, s));
            }
            if (int rc = launch_batch_scan(b, bgrid, s)) return rc;
            if (e1) CX_HIP(hipEventRecord(e1, s));
            MergeArgs mg;
            mg.part_keys = c->d_part_keys;
            mg.part_sims = c->d_part_sims;
            mg.n_lists = bgrid;
            mg.k = k_eff;
            mg.out_rows = d_rows + q0 * k_eff;
            mg.out_scores = d_scores + q0 * k_eff;
            mg.out_dists = d_dists + q0 * k_eff;
            mg.out_count = d_counts + q0;
            if (int rc = launch_merge_batch(mg, m, s)) return rc;
        }
        return CX_OK;
    }
    // the other row widths (1024-d: BGE-large): batchg.hip — dense cosines of <= 64 queries per pass over the f32 rows,
    // then the top k of each
    if (bg_go && !lossy_rows) {
        const uint32_t stride = (n + 3u) & ~3u, chunks = dense_topk_chunks(n);
        const size_t qimg = batchg_qimg_bytes(ix->dim);
        if (int rc = ensure_dev(c->d_dense, c->dn_cap, (size_t)64 * stride)) return rc;
        if (int rc = ensure_dev(c->d_qimg, c->qi_cap, qimg + 64 * sizeof(float))) return rc;
        if (int rc = ensure_dev(c->d_part_keys, c->pk_cap, (size_t)64 * chunks * k_eff)) return rc;
        if (int rc = ensure_dev(c->d_part_sims, c->ps_cap, (size_t)64 * chunks * k_eff)) return rc;
        // Large stores: bound each query from a 1-in-64 (k <= 32) or 1-in-32 sample of the row tiles, then write only the
        // rows that reach the bound (kernels.hpp: BatchGFilter) — the 4 bytes per row and query of the dense pass cost
        // the row stream a fifth of its rate, and launch_dense_topk reads them all back.
        // expected candidates per query = k * step (the sample's k-th best against step times as many rows)
        static const uint32_t step_env = getenv("CX_BATCHG_SAMPLE_STEP") ? (uint32_t)std::max(1, atoi(getenv("CX_BATCHG_SAMPLE_STEP"))) : 0u;
        // (A coarser sample for big shards — 1 tile in 191 at 6.25M rows instead of 1 in 64 — was tried in round 3 to shorten the
        // bound: k * step candidates per query then spread widely around their mean of 1,910 and one query of a batch overflowed
        // cand_select_kernel's 4,096 now and then: the exact dense fallback ran, 4.5 ms instead of 2.4.  The bound got cheaper
        // another way: launch_bound_topk below.)
        // The sample is as coarse as the candidate stage allows: a bound from a 1-in-s sample lets k s candidates per query
        // through on average, spread with a standard deviation of ~sqrt(k) s; cand_select_kernel holds 4,096 (k <= 32) and
        // has to stay ~6 sigma above the mean — s = 128 at k = 10, 64 at k = 32 (the 1-in-191 sample that failed above sat
        // 3.6 sigma under the cap).  A coarser sample is a shorter sample pass (latency-bound: 31 us for 77 tiles, 49 for
        // 306) and a shorter bound selection (<= 64k sampled scores: in registers, bound_select_reg_kernel).
        uint32_t tile_step = k_eff <= 32u ? 64u : 32u;
        if (k_eff <= 32u) {
            const float room = 4096.0f / ((float)k_eff + 6.0f * sqrtf((float)k_eff));
            tile_step = room >= 128.0f ? 128u : (room >= 96.0f ? 96u : 64u);
        }
        if (step_env) tile_step = step_env;
        uint32_t s_tiles = 0;
        const uint32_t s_rows = batchg_sample_rows(n, tile_step, &s_tiles, ix->dtype == 1), s_stride = (s_rows + 3u) & ~3u;
        const bool filtered = filter_ok && n >= filter_min && s_rows >= k_eff;
        const uint32_t bgrid = batchg_grid(n, ix->dtype == 1);
        uint32_t cb = std::max<uint32_t>(k_eff, 32u);
        if (getenv("CX_BATCHG_CAND_CAP")) cb = std::max<uint32_t>(k_eff, (uint32_t)atoi(getenv("CX_BATCHG_CAND_CAP")));   // tests: force the fallback
        cb = (cb + k_eff - 1u) / k_eff * k_eff;
        if (filtered) {
            if (int rc = ensure_dev(c->d_cand_keys, c->ck_cap, (size_t)64 * bgrid * cb)) return rc;
            if (int rc = ensure_dev(c->d_cand_sims, c->cs_cap, (size_t)64 * bgrid * cb)) return rc;
            if (int rc = ensure_dev(c->d_bg_ctl, c->bc_cap, (size_t)80 + (size_t)64 * bgrid)) return rc;   // bounds, flag, list counts
        }
        for (uint64_t q0 = 0; q0 < nq; q0 += 64) {
            const uint32_t m = (uint32_t)std::min<uint64_t>(64, nq - q0);
            float *d_qq = reinterpret_cast<float *>(c->d_qimg + qimg);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            auto prof_begin = [&]() -> int {
                if (!ix->profiling) return CX_OK;
                CX_HIP(hipEventCreate(&e0));
                CX_HIP(hipEventCreate(&e1));
                std::lock_guard<std::mutex> g(ix->mu);
                ix->prof_events.emplace_back(e0, e1);
                CX_HIP(hipEventRecord(e0, s));
                return CX_OK;
            };
            MergeArgs mg;
            mg.part_keys = c->d_part_keys;
            mg.part_sims = c->d_part_sims;
            mg.k = k_eff;
            mg.out_rows = d_rows + q0 * k_eff;
            mg.out_scores = d_scores + q0 * k_eff;
            mg.out_dists = d_dists + q0 * k_eff;
            mg.out_count = d_counts + q0;
            if (int rc = launch_batchg_split(d_queries + q0 * ix->dim, m, ix->dim, c->d_qimg, d_qq, s, ix->dtype == 1)) return rc;
            const uint32_t *run_if = nullptr;
            if (filtered) {
                uint32_t *tau = c->d_bg_ctl, *overflow = c->d_bg_ctl + 64;
                // 1. the bound: the k-th best score of the sampled tiles
                uint32_t *counts = c->d_bg_ctl + 80;
                if (int rc = launch_batchg_pass(ix->rows32(), ix->d_norms, n, ix->dim, m, c->d_qimg, d_qq, c->d_dense, s_stride, tile_step, nullptr, nullptr, s, ix->rows16())) return rc;
                // the k-th best score of the sample.  No row filter: the chunked top-k kernels the dense fallback uses (256
                // blocks per query, then one merge) instead of ONE block per query walking all its sampled scores — 141 us
                // -> ~25 us at the 97k sampled rows of a 6.25M-row shard, 6 % of that step.  With a filter the sample's
                // columns have to be mapped back to rows: bound_select_kernel does that.
                if (flt.trivial && k_eff <= 32u && s_rows > 65536u) {   // more sampled scores than bound_select_reg_kernel holds (wide lists: the chunked kernels' k = 100 lists cost more than the one-block select, 0.94 against 0.83 ms per step)
                    // (2k sampled scores per block, not 16k: 64 blocks walking 15.6k scores each took 40 us of a 0.85 ms step at 1M rows)
                    // at 6.25M rows (97k sampled) 47 chunks cost 12 us more than 6: a block's fixed cost is the merge of its four waves' lists)
                    const uint32_t s_chunks = std::min<uint32_t>(std::min<uint32_t>(chunks, 12u), std::max<uint32_t>(1u, s_rows / 2048u));
                    if (int rc = launch_dense_topk(c->d_dense, s_stride, s_rows, m, k_eff, flt, c->d_part_keys, c->d_part_sims, s_chunks, s)) return rc;
                    MergeArgs mb = mg;
                    mb.n_lists = s_chunks;
                    const bool fused = merge_batch_writes_bound(k_eff, s_chunks);   // the merge writes the bounds and clears the flag itself
                    if (fused) { mb.bound_out = tau; mb.clear_word = overflow; }
                    if (int rc = launch_merge_batch(mb, m, s)) return rc;
                    if (!fused) {
                        if (int rc = launch_tau_from_lists(mg.out_scores, mg.out_count, m, k_eff, tau, s)) return rc;
                        CX_HIP(hipMemsetAsync(overflow, 0, sizeof(uint32_t), s));
                    }
                } else {
                    if (int rc = launch_bound_select(c->d_dense, s_stride, s_rows, m, k_eff, tau, flt, batchg_tile_rows(ix->dtype == 1), tile_step, s)) return rc;
                    CX_HIP(hipMemsetAsync(overflow, 0, sizeof(uint32_t), s));
                }
                // 2. every row, candidates only
                BatchGFilter f{tau, c->d_cand_keys, c->d_cand_sims, counts, overflow, cb, flt};
                if (int rc = prof_begin()) return rc;
                if (int rc = launch_batchg_pass(ix->rows32(), ix->d_norms, n, ix->dim, m, c->d_qimg, d_qq, nullptr, 0, 1, &f, nullptr, s, ix->rows16())) return rc;
                if (e1) CX_HIP(hipEventRecord(e1, s));
                MergeArgs mc = mg;
                mc.part_keys = c->d_cand_keys;
                mc.part_sims = c->d_cand_sims;
                mc.n_lists = bgrid * (cb / k_eff);
                mc.seg_counts = counts;
                mc.seg_len = cb;
                if (int rc = launch_cand_select(mc, m, overflow, s)) return rc;
                // 3. the exact fallback below runs only if some list overflowed
                run_if = overflow;
                e0 = e1 = nullptr;
            } else if (int rc = prof_begin()) return rc;
            if (int rc = launch_batchg_pass(ix->rows32(), ix->d_norms, n, ix->dim, m, c->d_qimg, d_qq, c->d_dense, stride, 1, nullptr, run_if, s, ix->rows16())) return rc;
            if (e1) CX_HIP(hipEventRecord(e1, s));
            if (int rc = launch_dense_topk(c->d_dense, stride, n, m, k_eff, flt, c->d_part_keys, c->d_part_sims, chunks, s, run_if)) return rc;
            mg.n_lists = chunks;
            mg.run_if = run_if;
            if (int rc = launch_merge_batch(mg, m, s)) return rc;
        }
        return CX_OK;
    }
    for (uint64_t i = 0; i < nq; i++) {
        ScanArgs a;
        memset(&a, 0, sizeof a);
        a.rows = ix->rows32();
        a.rows16 = ix->rows16();
        a.query = d_queries + i * ix->dim;
        a.q_tail_sumsq = tails ? tails[i] : 0.0f;
        a.n_rows = n;
        a.dim = ix->dim;
        a.k = k_eff;
        a.flt = flt;
        if (topk_path) {
            if (k_eff == 0) {
                CX_HIP(hipMemsetAsync(d_counts + i, 0, 4, s));
                continue;
            }
            a.part_keys = c->d_part_keys;
            a.part_sims = c->d_part_sims;
            MergeArgs m;
            m.part_keys = c->d_part_keys;
            m.part_sims = c->d_part_sims;
            m.n_lists = grid;
            m.k = k_eff;
            m.out_rows = d_rows + i * k_eff;
            m.out_scores = d_scores + i * k_eff;
            m.out_dists = d_dists + i * k_eff;
            m.out_count = d_counts + i;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (ix->profiling) {
                CX_HIP(hipEventCreate(&e0));
                CX_HIP(hipEventCreate(&e1));
                std::lock_guard<std::mutex> g(ix->mu);
                ix->prof_events.emplace_back(e0, e1);
            }
            if (int rc = launch_scan_topk(a, m, nt, s, e0, e1)) return rc;
        } else {
            a.dense_keys = c->d_keys;
            a.dense_sims = c->d_sims;
            a.threshold = thr;
            a.has_threshold = has_thr ? 1u : 0u;
            if (int rc = launch_scan_dense(a, nt, s)) return rc;
            if (int rc = launch_sort_select(c->d_keys, c->d_sims, c->d_keys2, c->d_sims2, n, k_eff, c->d_temp,
                                            c->tmp_cap, d_rows + i * k_eff, d_scores + i * k_eff,
                                            d_dists + i * k_eff, d_counts + i, s))
                return rc;
        }
    }
    return CX_OK;
}

// Host queries -> pinned staging, zero-padded / truncated to dim; the sum of
// squares of any elements beyond dim still belongs to |q| (the reference
// zips for the dot but takes the norm over the whole query, index.rs:172-173).
int stage_queries(const cx_index *ix, Ctx *c, uint64_t nq, const float *queries, uint64_t len, std::vector<float> &tails) {
    const uint32_t dim = ix->dim;
    if (int rc = ensure_pinned(c->h_query, c->hq_cap, (size_t)nq * std::max(dim, 1u))) return rc;
    if (int rc = ensure_dev(c->d_query, c->q_cap, (size_t)nq * std::max(dim, 1u) + 4)) return rc;
    tails.assign(nq, 0.0f);
    for (uint64_t i = 0; i < nq; i++) {
        const float *q = queries + i * len;
        float *dst = c->h_query + i * dim;
        const uint64_t m = std::min<uint64_t>(len, dim);
        memcpy(dst, q, m * sizeof(float));
        for (uint64_t j = m; j < dim; j++) dst[j] = 0.0f;
        float t = 0.0f;
        for (uint64_t j = dim; j < len; j++) t += q[j] * q[j];
        tails[i] = t;
    }
    if (dim) CX_HIP(hipMemcpyAsync(c->d_query, c->h_query, (size_t)nq * dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    return CX_OK;
}

}  // namespace cx
namespace {

// Results of a host-API call live in ONE device block [counts nq | pad][rows][scores][dists] mirrored by one
// pinned host block, so they come back with a single D2H copy (four separate copies cost ~10 us of a ~40 us
// small-corpus query).
struct OutView {
    uint32_t *counts, *rows;
    float *scores, *dists;
    size_t words;
};
OutView out_view(uint32_t *base, size_t entries, size_t nq) {
    const size_t cpad = (nq + 3) / 4 * 4;
    OutView v;
    v.counts = base;
    v.rows = base + cpad;
    v.scores = reinterpret_cast<float *>(base + cpad + entries);
    v.dists = reinterpret_cast<float *>(base + cpad + 2 * entries);
    v.words = cpad + 3 * entries;
    return v;
}
int ensure_out(Ctx *c, size_t entries, size_t nq, OutView &dev, OutView &host) {
    entries = std::max<size_t>(entries, 1);
    nq = std::max<size_t>(nq, 1);
    const size_t words = (nq + 3) / 4 * 4 + 3 * entries;
    if (int rc = ensure_dev(c->d_out_rows, c->or_cap, words)) return rc;
    if (int rc = ensure_pinned(c->h_rows, c->hr_cap, words)) return rc;
    dev = out_view(c->d_out_rows, entries, nq);
    host = out_view(c->h_rows, entries, nq);
    return CX_OK;
}
// counts + the first `entries_used` entries of each array would need gathers; the block is small, copy it whole
int fetch_block(Ctx *c, const OutView &dev, const OutView &host) {
    CX_HIP(hipMemcpyAsync(host.counts, dev.counts, dev.words * 4, hipMemcpyDeviceToHost, c->stream));
    CX_HIP(hipStreamSynchronize(c->stream));
    return CX_OK;
}

}  // namespace

extern "C" {

const char *cx_last_error(void) { return err_buf(); }

int cx_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} catch (...) { return cx::on_exception(); }

cx_index *cx_create(uint32_t dimension, int device) { return cx_create_ex(dimension, device, CX_DTYPE_F32); }

int cx_dtype(const cx_index *ix) { return ix ? ix->dtype : -1; }

cx_index *cx_create_ex(uint32_t dimension, int device, int dtype) try {
    if (dtype != CX_DTYPE_F32 && dtype != CX_DTYPE_BF16) {
        set_err(CX_ERR_VALIDATION, "unknown storage dtype %d", dtype);
        return nullptr;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_err(CX_ERR_DEVICE, "no HIP device available (%s); libcortex_hip has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= n) {
        set_err(CX_ERR_VALIDATION, "device %d out of range (0..%d)", device, n - 1);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        set_err(CX_ERR_DEVICE, "hipSetDevice(%d) failed", device);
        return nullptr;
    }
    cx_index *ix = new cx_index();
    ix->dim = dimension;
    ix->device = device;
    ix->dtype = dtype;
    ix->elem = dtype == CX_DTYPE_BF16 ? 2u : 4u;
    if (hipStreamCreateWithFlags(&ix->up_stream, hipStreamNonBlocking) != hipSuccess) {
        set_err(CX_ERR_DEVICE, "hipStreamCreate failed");
        delete ix;
        return nullptr;
    }
    return ix;
} catch (...) { cx::on_exception(); return nullptr; }

void cx_destroy(cx_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    for (Ctx *c : ix->pool) delete c;
    for (auto &kv : ix->by_stream) delete kv.second;
    (void)hipFree(ix->d_rows);
    (void)hipFree(ix->d_stage);
    (void)hipFree(ix->d_meta);
    (void)hipFree(ix->d_agent);
    (void)hipFree(ix->d_shadow);
    (void)hipFree(ix->d_shadow_t);
    (void)hipFree(ix->d_shadow_err);
    (void)hipFree(ix->d_irr_rows);
    (void)hipFree(ix->d_norms);
    (void)hipFree(ix->d_norms_lossy);
    (void)hipFree(ix->d_split);
    (void)hipFree(ix->d_tile_list);
    if (ix->up_stream) (void)hipStreamDestroy(ix->up_stream);
    delete ix;
}

int cx_reserve(cx_index *ix, uint64_t rows) try {
    if (!ix) return set_err(CX_ERR_VALIDATION, "null index");
    if (int rc = use_device(ix)) return rc;
    return grow_rows(ix, rows);
} catch (...) { return cx::on_exception(); }

int cx_upsert(cx_index *ix, const uint8_t id[16], const float *embedding, uint64_t len) try {
    return upsert_impl(ix, 1, id, embedding, len, false);
} catch (...) { return cx::on_exception(); }
int cx_upsert_batch(cx_index *ix, uint64_t n, const uint8_t *ids, const float *embeddings, uint64_t len) try {
    return upsert_impl(ix, n, ids, embeddings, len, false);
} catch (...) { return cx::on_exception(); }
int cx_upsert_batch_dev(cx_index *ix, uint64_t n, const uint8_t *ids, const float *d_embeddings, uint64_t len) try {
    return upsert_impl(ix, n, ids, d_embeddings, len, true);
} catch (...) { return cx::on_exception(); }

int cx_remove(cx_index *ix, const uint8_t id[16]) try {
    if (!ix || !id) return set_err(CX_ERR_VALIDATION, "null argument");
    ix->pending_meta.erase(id_key(id));     // vector/index.rs:318: metadata goes whether or not a vector exists
    auto it = ix->map.find(id_key(id));
    if (it == ix->map.end()) return CX_OK;  // vector/index.rs:317 — HashMap::remove of a missing key
    if (int rc = use_device(ix)) return rc;
    const uint32_t row = it->second;
    ix->map.erase(it);
    ix->h_meta[row] = META_REMOVED;  // metadata goes with the vector (:318)
    ix->h_agent[row] = 0;
    ix->n_alive--;
    ix->n_removed++;
    return push_meta(ix, row);
} catch (...) { return cx::on_exception(); }

int cx_set_metadata(cx_index *ix, const uint8_t id[16], uint32_t kind_code, uint32_t agent_code) try {
    if (!ix || !id) return set_err(CX_ERR_VALIDATION, "null argument");
    if (kind_code >= (1u << 24)) return set_err(CX_ERR_VALIDATION, "kind code out of range");
    auto it = ix->map.find(id_key(id));
    // The reference keeps metadata in its own map (:219-222) and consults it for ids that have a
    // vector (:234): metadata for an id without a vector waits for the insert (vector/tests.rs:65-66).
    if (it == ix->map.end()) {
        ix->pending_meta[id_key(id)] = {kind_code, agent_code};
        return CX_OK;
    }
    if (int rc = use_device(ix)) return rc;
    const uint32_t row = it->second;
    ix->h_meta[row] = META_HAS | (kind_code << 8);
    ix->h_agent[row] = agent_code;
    return push_meta(ix, row);
} catch (...) { return cx::on_exception(); }

int cx_set_metadata_batch(cx_index *ix, uint64_t n, const uint8_t *ids, const uint32_t *kind_codes, const uint32_t *agent_codes) try {
    if (!ix) return set_err(CX_ERR_VALIDATION, "null index");
    if (!n) return CX_OK;
    if (!ids || !kind_codes || !agent_codes) return set_err(CX_ERR_VALIDATION, "null argument");
    for (uint64_t i = 0; i < n; i++)
        if (kind_codes[i] >= (1u << 24)) return set_err(CX_ERR_VALIDATION, "kind code out of range");
    uint32_t lo = UINT32_MAX, hi = 0;
    for (uint64_t i = 0; i < n; i++) {
        auto it = ix->map.find(id_key(ids + 16 * i));
        if (it == ix->map.end()) {           // as cx_set_metadata: kept until the id gets its vector
            ix->pending_meta[id_key(ids + 16 * i)] = {kind_codes[i], agent_codes[i]};
            continue;
        }
        const uint32_t row = it->second;
        ix->h_meta[row] = META_HAS | (kind_codes[i] << 8);
        ix->h_agent[row] = agent_codes[i];
        lo = std::min(lo, row);
        hi = std::max(hi, row);
    }
    if (lo > hi) return CX_OK;
    if (int rc = use_device(ix)) return rc;
    // one copy of the touched row range (bulk loads touch consecutive rows)
    CX_HIP(hipMemcpyAsync(ix->d_meta + lo, &ix->h_meta[lo], (size_t)(hi - lo + 1) * 4, hipMemcpyHostToDevice, ix->up_stream));
    CX_HIP(hipMemcpyAsync(ix->d_agent + lo, &ix->h_agent[lo], (size_t)(hi - lo + 1) * 4, hipMemcpyHostToDevice, ix->up_stream));
    CX_HIP(hipStreamSynchronize(ix->up_stream));
    return CX_OK;
} catch (...) { return cx::on_exception(); }

uint32_t cx_intern(cx_index *ix, const char *utf8, uint64_t len) try {
    if (!ix || (!utf8 && len)) return 0;
    std::string s(utf8 ? utf8 : "", (size_t)len);
    std::lock_guard<std::mutex> g(ix->intern_mu);
    auto it = ix->interned.find(s);
    if (it != ix->interned.end()) return it->second;
    const uint32_t code = (uint32_t)ix->interned.size() + 1;
    ix->interned.emplace(std::move(s), code);
    return code;
} catch (...) { cx::on_exception(); return 0; }

uint32_t cx_lookup(const cx_index *ix, const char *utf8, uint64_t len) try {
    if (!ix || (!utf8 && len)) return 0;
    const std::string s(utf8 ? utf8 : "", (size_t)len);
    std::lock_guard<std::mutex> g(ix->intern_mu);
    auto it = ix->interned.find(s);
    return it == ix->interned.end() ? 0u : it->second;   // 0: a string no row was ever tagged with matches no row
} catch (...) { cx::on_exception(); return 0; }

int cx_rebuild(cx_index *ix) try {
    if (!ix) return set_err(CX_ERR_VALIDATION, "null index");
    if (!ix->n_removed) return CX_OK;
    if (int rc = use_device(ix)) return rc;
    std::vector<uint32_t> keep;
    keep.reserve(ix->n_alive);
    for (uint64_t r = 0; r < ix->n_rows; r++)
        if (!(ix->h_meta[r] & META_REMOVED)) keep.push_back((uint32_t)r);
    const uint64_t n_new = keep.size();
    const uint64_t ncap = std::max<uint64_t>(n_new, 1024);
    // Everything new is built beside the live state and swapped in only after the last device operation succeeded:
    // a failure on the way (allocation, gather, copy) leaves the index exactly as it was.
    struct NewStore {
        float *rows = nullptr;
        uint32_t *meta = nullptr, *agent = nullptr, *keep = nullptr;
        ~NewStore() { (void)hipFree(rows); (void)hipFree(meta); (void)hipFree(agent); (void)hipFree(keep); }
    } ns;
    CX_HIP(hipMalloc((void **)&ns.rows, (ncap + 256) * ix->dim * ix->elem + 64));  // + one row tile of readable padding (batch.hip: 16 rows, batchg.hip: 256)
    CX_HIP(hipMalloc((void **)&ns.meta, ncap * sizeof(uint32_t)));
    CX_HIP(hipMalloc((void **)&ns.agent, ncap * sizeof(uint32_t)));
    CX_HIP(hipMemsetAsync(ns.meta, 0, ncap * sizeof(uint32_t), ix->up_stream));
    CX_HIP(hipMemsetAsync(ns.agent, 0, ncap * sizeof(uint32_t), ix->up_stream));
    std::vector<uint8_t> nids(n_new * 16);
    std::vector<uint32_t> nmeta(n_new), nagent(n_new);
    std::vector<NodeStats> nstats(ix->h_stats.empty() ? 0 : n_new);
    std::unordered_map<IdKey, uint32_t, IdHash> nmap;
    nmap.reserve(n_new);
    for (uint64_t i = 0; i < n_new; i++) {
        const uint32_t r = keep[i];
        memcpy(&nids[16 * i], &ix->ids[16 * (size_t)r], 16);
        nmeta[i] = ix->h_meta[r];
        nagent[i] = ix->h_agent[r];
        if (!nstats.empty() && r < ix->h_stats.size()) nstats[i] = ix->h_stats[r];
        nmap.emplace(id_key(&nids[16 * i]), (uint32_t)i);
    }
    if (n_new) {
        CX_HIP(hipMalloc((void **)&ns.keep, n_new * 4));
        CX_HIP(hipMemcpyAsync(ns.keep, keep.data(), n_new * 4, hipMemcpyHostToDevice, ix->up_stream));
        if (ix->dim) {
            const int rc = ix->dtype == 1 ? launch_gather_rows(ix->rows16(), reinterpret_cast<uint16_t *>(ns.rows), ns.keep, (uint32_t)n_new, ix->dim, ix->up_stream)
                                          : launch_gather_rows(ix->d_rows, ns.rows, ns.keep, (uint32_t)n_new, ix->dim, ix->up_stream);
            if (rc) return rc;
        }
        CX_HIP(hipMemcpyAsync(ns.meta, nmeta.data(), n_new * 4, hipMemcpyHostToDevice, ix->up_stream));
        CX_HIP(hipMemcpyAsync(ns.agent, nagent.data(), n_new * 4, hipMemcpyHostToDevice, ix->up_stream));
    }
    CX_HIP(hipStreamSynchronize(ix->up_stream));
    // commit: nothing below can fail
    std::swap(ix->d_rows, ns.rows);      // the old buffers leave with ns
    std::swap(ix->d_meta, ns.meta);
    std::swap(ix->d_agent, ns.agent);
    ix->cap = ncap;
    ix->map.swap(nmap);
    ix->ids.swap(nids);
    ix->h_meta.swap(nmeta);
    ix->h_agent.swap(nagent);
    ix->h_stats.swap(nstats);
    ix->n_rows = n_new;
    ix->n_alive = n_new;
    ix->n_removed = 0;
    ix->shadow_rows = 0;  // rows moved: the bf16 shadow is rebuilt on next use
    ix->shadow_stale.clear();
    ix->norms_rows = 0;   // and so are the row norms
    ix->norms_stale.clear();
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_debug_check_result_block(const uint32_t *counts, const uint32_t *rows, uint64_t nq, uint64_t stride, uint64_t k_max,
                                uint64_t n_rows) try {
    if (nq && (!counts || !rows)) return set_err(CX_ERR_VALIDATION, "null argument");
    return check_result_block(counts, rows, nq, stride, k_max, n_rows);
} catch (...) { return cx::on_exception(); }

int cx_profile_enable(cx_index *ix, int on) try {
    if (!ix) return set_err(CX_ERR_VALIDATION, "null index");
    ix->profiling = on != 0;
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_profile_read(cx_index *ix, double *kernel_ms_sum, uint64_t *launches, int reset) try {
    if (!ix || !kernel_ms_sum || !launches) return set_err(CX_ERR_VALIDATION, "null argument");
    if (int rc = use_device(ix)) return rc;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
    {
        std::lock_guard<std::mutex> g(ix->mu);
        evs.swap(ix->prof_events);
    }
    for (auto &p : evs) {
        float ms = 0.0f;
        CX_HIP(hipEventSynchronize(p.second));
        CX_HIP(hipEventElapsedTime(&ms, p.first, p.second));
        ix->prof_ms += ms;
        ix->prof_n++;
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    *kernel_ms_sum = ix->prof_ms;
    *launches = ix->prof_n;
    if (reset) {
        ix->prof_ms = 0.0;
        ix->prof_n = 0;
    }
    return CX_OK;
} catch (...) { return cx::on_exception(); }

uint64_t cx_len(const cx_index *ix) { return ix ? ix->n_alive : 0; }
uint32_t cx_dimension(const cx_index *ix) { return ix ? ix->dim : 0; }
// (cortex_hip.h: how many HIP streams a stream of batched searches of nq queries is worth rotating over — search_core's routing,
// read only: two-bank screening passes want 2, everything else batched 4)
uint32_t cx_search_batch_streams_hint(const cx_index *ix, uint64_t nq) {
    if (!ix || nq < 3) return 1;
    const uint32_t n = (uint32_t)ix->n_rows;
    const bool bs_on = !getenv("CX_BATCHS") || atoi(getenv("CX_BATCHS")) != 0;
    const bool screened = bs_on && cx::batchs_supported(ix->dim, 1) && !ix->irr_over &&
                          (n >= cx::batchs_min_rows() || (n >= 32768u && nq <= 256u));
    return (screened && cx::batchs_queries_per_pass(ix->dim, n, nq) > 64u) ? 2u : 4u;
}
uint64_t cx_row_count(const cx_index *ix) { return ix ? ix->n_rows : 0; }
const float *cx_device_rows(const cx_index *ix) { return ix ? ix->rows32() : nullptr; }   // null for a bf16 store

int cx_row_id(const cx_index *ix, uint64_t row, uint8_t out_id[16]) try {
    if (!ix || !out_id) return set_err(CX_ERR_VALIDATION, "null argument");
    if (row >= ix->n_rows) return set_err(CX_ERR_VALIDATION, "row %llu out of range", (unsigned long long)row);
    memcpy(out_id, &ix->ids[16 * (size_t)row], 16);
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_rows_alive(const cx_index *ix, uint64_t row_lo, uint64_t n, uint8_t *out_alive) try {
    if (!ix || (n && !out_alive)) return set_err(CX_ERR_VALIDATION, "null argument");
    if (row_lo + n > ix->n_rows) return set_err(CX_ERR_VALIDATION, "rows %llu..%llu beyond the %llu rows of the index",
                                                (unsigned long long)row_lo, (unsigned long long)(row_lo + n), (unsigned long long)ix->n_rows);
    for (uint64_t i = 0; i < n; i++) out_alive[i] = (ix->h_meta[(size_t)(row_lo + i)] & META_REMOVED) ? 0 : 1;
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_rows_of(const cx_index *ix, uint64_t n, const uint8_t *ids, uint32_t *out_rows) try {
    if (!ix || (n && (!ids || !out_rows))) return set_err(CX_ERR_VALIDATION, "null argument");
    for (uint64_t i = 0; i < n; i++) {
        auto it = ix->map.find(id_key(ids + 16 * i));
        out_rows[i] = it == ix->map.end() ? 0xFFFFFFFFu : it->second;
    }
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_search_batch(const cx_index *ix, uint64_t nq, const float *queries, uint64_t len, uint64_t k,
                    const cx_filter *filter, uint8_t *out_ids, float *out_scores, float *out_distances,
                    uint64_t *out_counts) try {
    if (!ix) return set_err(CX_ERR_VALIDATION, "null index");
    if (nq && (!queries || !out_counts)) return set_err(CX_ERR_VALIDATION, "null queries/out_counts");
    for (uint64_t i = 0; i < nq; i++) out_counts[i] = 0;
    if (!nq || ix->n_alive == 0 || k == 0) return CX_OK;  // vector/index.rs:331-333
    if (!out_ids || !out_scores || !out_distances) return set_err(CX_ERR_VALIDATION, "null output buffer");
    if (int rc = use_device(ix)) return rc;
    CtxLease lease(ix);
    Ctx *c = lease.c;
    if (!c) return CX_ERR_DEVICE;
    const uint32_t k_eff = (uint32_t)std::min<uint64_t>(k, ix->n_rows);
    std::vector<float> tails;
    if (int rc = stage_queries(ix, c, nq, queries, len, tails)) return rc;
    FilterUpload fu;
    if (int rc = build_filter(ix, c, filter, c->stream, fu)) return rc;
    const size_t entries = (size_t)nq * k_eff;
    OutView dv, hv;
    if (int rc = ensure_out(c, entries, nq, dv, hv)) return rc;
    // A small result block is written by the merge kernel straight into the pinned host buffer (device-visible
    // memory): a search of a small corpus is launch-latency-bound and the separate device-to-host copy — a blit kernel
    // of its own — was 5 of its 35 us.  Large blocks keep the HBM buffer and one bulk copy.
    static const int direct_max = getenv("CX_DIRECT_RESULTS_MAX") ? atoi(getenv("CX_DIRECT_RESULTS_MAX")) : 4096;
    const bool direct = k_eff <= TOPK_MAX && entries <= (size_t)direct_max;
    const OutView &ov = direct ? hv : dv;
    if (int rc = search_core(ix, c, c->d_query, tails.data(), nq, k_eff, fu.f, 0.0f, false, ov.rows, ov.scores, ov.dists,
                             ov.counts, c->stream))
        return rc;
    if (direct) CX_HIP(hipStreamSynchronize(c->stream));
    else if (int rc = fetch_block(c, dv, hv)) return rc;
    // nothing read back from the device is used as an index or a length before it was checked: a kernel bug must
    // surface as CX_ERR_DEVICE, never as a fault on the caller's side of the FFI
    if (int rc = check_result_block(hv.counts, hv.rows, nq, k_eff, k_eff, ix->n_rows)) {
        for (uint64_t i = 0; i < nq; i++) out_counts[i] = 0;
        return rc;
    }
    for (uint64_t i = 0; i < nq; i++) {
        const uint32_t cnt = hv.counts[i];
        out_counts[i] = cnt;
        for (uint32_t j = 0; j < cnt; j++) {
            const size_t src = (size_t)i * k_eff + j, dst = (size_t)i * k + j;
            memcpy(out_ids + 16 * dst, &ix->ids[16 * (size_t)hv.rows[src]], 16);
            out_scores[dst] = hv.scores[src];
            out_distances[dst] = hv.dists[src];
        }
    }
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_search(const cx_index *ix, const float *query, uint64_t len, uint64_t k, const cx_filter *filter,
              uint8_t *out_ids, float *out_scores, float *out_distances, uint64_t *n_out) try {
    if (!n_out) return set_err(CX_ERR_VALIDATION, "null n_out");
    return cx_search_batch(ix, 1, query, len, k, filter, out_ids, out_scores, out_distances, n_out);
} catch (...) { return cx::on_exception(); }

int cx_search_threshold(const cx_index *ix, const float *query, uint64_t len, float threshold,
                        const cx_filter *filter, uint64_t cap, uint8_t *out_ids, float *out_scores,
                        float *out_distances, uint64_t *n_out, uint64_t *n_needed) try {
    if (!ix || !query || !n_out) return set_err(CX_ERR_VALIDATION, "null argument");
    *n_out = 0;
    if (n_needed) *n_needed = 0;
    if (ix->n_alive == 0) return CX_OK;
    if (int rc = use_device(ix)) return rc;
    CtxLease lease(ix);
    Ctx *c = lease.c;
    if (!c) return CX_ERR_DEVICE;
    const uint32_t n = (uint32_t)ix->n_rows;
    std::vector<float> tails;
    if (int rc = stage_queries(ix, c, 1, query, len, tails)) return rc;
    FilterUpload fu;
    if (int rc = build_filter(ix, c, filter, c->stream, fu)) return rc;
    OutView dv, hv;
    if (int rc = ensure_out(c, n, 1, dv, hv)) return rc;
    if (int rc = search_core(ix, c, c->d_query, tails.data(), 1, n, fu.f, threshold, true, dv.rows, dv.scores, dv.dists,
                             dv.counts, c->stream))
        return rc;
    CX_HIP(hipMemcpyAsync(hv.counts, dv.counts, 4, hipMemcpyDeviceToHost, c->stream));
    CX_HIP(hipStreamSynchronize(c->stream));
    const uint64_t total = hv.counts[0];
    if (total > n) return set_err(CX_ERR_DEVICE, "search_threshold: device reported %llu results for %u rows", (unsigned long long)total, n);
    if (n_needed) *n_needed = total;
    const uint64_t take = std::min<uint64_t>(total, cap);
    if (take) {
        if (!out_ids || !out_scores || !out_distances) return set_err(CX_ERR_VALIDATION, "null output buffer");
        CX_HIP(hipMemcpyAsync(hv.rows, dv.rows, take * 4, hipMemcpyDeviceToHost, c->stream));
        CX_HIP(hipMemcpyAsync(hv.scores, dv.scores, take * 4, hipMemcpyDeviceToHost, c->stream));
        CX_HIP(hipMemcpyAsync(hv.dists, dv.dists, take * 4, hipMemcpyDeviceToHost, c->stream));
        CX_HIP(hipStreamSynchronize(c->stream));
        const uint32_t one = (uint32_t)take;
        if (int rc = check_result_block(&one, hv.rows, 1, take, take, ix->n_rows)) return rc;
        for (uint64_t j = 0; j < take; j++) {
            memcpy(out_ids + 16 * j, &ix->ids[16 * (size_t)hv.rows[j]], 16);
            out_scores[j] = hv.scores[j];
            out_distances[j] = hv.dists[j];
        }
    }
    *n_out = take;
    if (total > cap)
        return set_err(CX_ERR_CAPACITY, "search_threshold: %llu results, buffer holds %llu",
                       (unsigned long long)total, (unsigned long long)cap);
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_search_batch_dev(const cx_index *ix, uint64_t nq, const float *d_queries, uint64_t k,
                        const cx_filter *filter, uint32_t *d_rows, float *d_scores, float *d_distances,
                        uint32_t *d_counts, void *stream) try {
    if (!ix) return set_err(CX_ERR_VALIDATION, "null index");
    if (!nq) return CX_OK;
    if (!d_queries || !d_counts) return set_err(CX_ERR_VALIDATION, "null device pointer");
    if (int rc = use_device(ix)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (ix->n_rows == 0 || k == 0) {
        CX_HIP(hipMemsetAsync(d_counts, 0, nq * 4, s));
        return CX_OK;
    }
    if (!d_rows || !d_scores || !d_distances) return set_err(CX_ERR_VALIDATION, "null device output");
    if (k > ix->n_rows) return set_err(CX_ERR_VALIDATION, "k=%llu exceeds the %llu rows of this shard; clamp it",
                                        (unsigned long long)k, (unsigned long long)ix->n_rows);
    Ctx *c = ctx_for_stream(ix, s);
    FilterUpload fu;
    if (int rc = build_filter(ix, c, filter, s, fu)) return rc;
    return search_core(ix, c, d_queries, nullptr, nq, (uint32_t)k, fu.f, 0.0f, false, d_rows, d_scores,
                       d_distances, d_counts, s);
} catch (...) { return cx::on_exception(); }

int cx_search_dev(const cx_index *ix, const float *d_query, uint64_t k, const cx_filter *filter, uint32_t *d_rows,
                  float *d_scores, float *d_distances, uint32_t *d_count, void *stream) try {
    return cx_search_batch_dev(ix, 1, d_query, k, filter, d_rows, d_scores, d_distances, d_count, stream);
} catch (...) { return cx::on_exception(); }

int cx_merge_topk_dev(int device, uint64_t n_parts, uint64_t nq, uint64_t k, uint64_t part_stride, const uint64_t *part_base,
                      const uint32_t *d_rows, const float *d_scores, const float *d_distances,
                      const uint32_t *d_counts, uint64_t *d_out_rows, float *d_out_scores,
                      float *d_out_distances, uint32_t *d_out_counts, void *stream) try {
    if (!part_base || !d_rows || !d_scores || !d_distances || !d_counts || !d_out_rows || !d_out_scores ||
        !d_out_distances || !d_out_counts)
        return set_err(CX_ERR_VALIDATION, "null argument");
    if (n_parts > MAX_PARTS) return set_err(CX_ERR_VALIDATION, "merge: at most %u parts", MAX_PARTS);
    CX_HIP(hipSetDevice(device));
    PartBase pb;
    memset(&pb, 0, sizeof pb);
    for (uint64_t p = 0; p < n_parts; p++) pb.base[p] = part_base[p];
    return launch_merge_parts((uint32_t)n_parts, (uint32_t)nq, (uint32_t)k, part_stride, pb, d_rows, d_scores, d_distances,
                              d_counts, d_out_rows, d_out_scores, d_out_distances, d_out_counts,
                              (hipStream_t)stream);
} catch (...) { return cx::on_exception(); }

}  // extern "C"
