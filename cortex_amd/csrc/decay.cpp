// decay.cpp — query-time score decay and re-rank: what the reference's HTTP search handler does
// with the index's results (cortex-server/src/http/routes.rs:889-947) — fetch max(3 limit, 30) candidates, turn
// each raw score into `apply_score_decay(node, raw, cfg, recency_bias)` (cortex-core/src/vector/scoring.rs:84-114),
// stable-sort by the decayed score, keep `limit` — done behind the ABI, so the caller hydrates `limit` nodes from
// storage instead of 3 x limit.  The per-node inputs of the formula (kind, last_accessed_at, access_count) are
// kept per row on the host: they are what the bulk loader reads from the stored nodes anyway (nodes.cpp).
//
//   temporal = max(min_factor, exp(-kind_rate * min(days_idle, max_age_days)))      f64, then `as f32`
//   echo     = min(echo_cap, 1 + access_count * echo_weight)                        f64, then `as f32`
//   final    = raw * (1 - rb) + raw * temporal * echo * rb                          f32, left to right, no FMA
//
// The candidate scan is the index's exact search; the decay is a few flops on <= candidate_limit rows and runs on
// the host (it is not worth a kernel: the cost it removes is the storage round trips).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "internal.hpp"

#pragma clang fp contract(off)   // Rust never contracts a * b + c; clang would, within one expression

using namespace cx;

namespace {

// chrono: now.signed_duration_since(t).num_seconds() — whole seconds of the difference, truncated toward zero
int64_t num_seconds_between(int64_t now_s, uint32_t now_ns, int64_t t_s, uint32_t t_ns) {
    int64_t s = now_s - t_s;
    int64_t ns = (int64_t)now_ns - (int64_t)t_ns;
    if (s > 0 && ns < 0) s -= 1;        // +s seconds minus a fraction: one whole second less
    else if (s < 0 && ns > 0) s += 1;   // toward zero from below
    return s;
}

float decayed(const cx_decay_config &cfg, float raw, float rb, int64_t now_s, uint32_t now_ns, const NodeStats &st) {
    if (!cfg.enabled || rb == 0.0f) return raw;   // scoring.rs:90-92
    const int64_t idle_s = std::max<int64_t>(num_seconds_between(now_s, now_ns, st.last_s, st.last_ns), 0);
    const double days_idle = (double)idle_s / 86400.0;
    double kind_rate = cfg.daily_rate;
    for (uint32_t i = 0; i < cfg.n_by_kind; i++)
        if (cfg.kind_codes[i] != 0 && cfg.kind_codes[i] == st.kind) { kind_rate = cfg.kind_rates[i]; break; }   // 0 = cx_lookup's "never interned"
    const double effective_days = std::min(days_idle, cfg.max_age_days);
    const float temporal = (float)std::max(std::exp(-kind_rate * effective_days), cfg.min_factor);
    const float echo = (float)std::min(1.0 + (double)st.access * cfg.echo_weight, cfg.echo_cap);
    const float a = raw * (1.0f - rb);
    float b = raw * temporal;
    b = b * echo;
    b = b * rb;
    return a + b;
}

}  // namespace

extern "C" {

int cx_set_node_stats_batch(cx_index *ix, uint64_t n, const uint8_t *ids, const uint32_t *kind_codes,
                            const int64_t *last_accessed_s, const uint32_t *last_accessed_ns, const uint64_t *access_counts) try {
    if (!ix) return set_err(CX_ERR_VALIDATION, "null index");
    if (!n) return CX_OK;
    if (!ids || !kind_codes || !last_accessed_s || !access_counts) return set_err(CX_ERR_VALIDATION, "null argument");
    if (ix->h_stats.size() < ix->n_rows) ix->h_stats.resize((size_t)ix->n_rows);
    for (uint64_t i = 0; i < n; i++) {
        auto it = ix->map.find(id_key(ids + 16 * i));
        if (it == ix->map.end()) continue;   // a node without a vector never shows up in a search
        NodeStats &s = ix->h_stats[it->second];
        s.last_s = last_accessed_s[i];
        s.last_ns = last_accessed_ns ? last_accessed_ns[i] : 0u;
        s.kind = kind_codes[i];
        s.access = access_counts[i];
    }
    return CX_OK;
} catch (...) { return cx::on_exception(); }

float cx_apply_score_decay(const cx_decay_config *cfg, float raw_score, float recency_bias, int64_t now_s, uint32_t now_ns,
                           uint32_t kind_code, int64_t last_accessed_s, uint32_t last_accessed_ns, uint64_t access_count) {
    if (!cfg) return raw_score;
    NodeStats st;
    st.last_s = last_accessed_s; st.last_ns = last_accessed_ns; st.kind = kind_code; st.access = access_count;
    return decayed(*cfg, raw_score, recency_bias, now_s, now_ns, st);
}

int cx_search_decayed(const cx_index *ix, const float *query, uint64_t len, uint64_t limit, uint64_t candidate_limit,
                      const cx_filter *filter, const cx_decay_config *cfg, float recency_bias, int64_t now_s, uint32_t now_ns,
                      uint8_t *out_ids, float *out_scores, float *out_raw_scores, uint64_t *n_out) try {
    if (!ix || !query || !cfg || !n_out) return set_err(CX_ERR_VALIDATION, "null argument");
    *n_out = 0;
    if (cfg->n_by_kind && (!cfg->kind_codes || !cfg->kind_rates)) return set_err(CX_ERR_VALIDATION, "null by_kind table");
    if (candidate_limit < limit) candidate_limit = limit;
    const uint64_t cap = std::max<uint64_t>(1, std::min<uint64_t>(candidate_limit, cx_row_count(ix)));
    std::vector<uint8_t> ids(16 * (size_t)cap);
    std::vector<float> raw((size_t)cap), dist((size_t)cap);
    uint64_t n = 0;
    if (int rc = cx_search(ix, query, len, candidate_limit, filter, ids.data(), raw.data(), dist.data(), &n)) return rc;
    if (n && limit && (!out_ids || !out_scores || !out_raw_scores)) return set_err(CX_ERR_VALIDATION, "null output buffer");
    std::vector<uint32_t> rows((size_t)n);
    if (n) if (int rc = cx_rows_of(ix, n, ids.data(), rows.data())) return rc;
    std::vector<float> fin((size_t)n);
    std::vector<uint32_t> order((size_t)n);
    const NodeStats fresh{};   // nodes nobody described: last_accessed_at = the epoch (types.rs:56), never accessed
    for (uint64_t i = 0; i < n; i++) {
        const NodeStats &st = rows[i] < ix->h_stats.size() ? ix->h_stats[rows[i]] : fresh;
        fin[i] = decayed(*cfg, raw[i], recency_bias, now_s, now_ns, st);
        order[i] = (uint32_t)i;
    }
    // scored.sort_by(|a, b| b.1.partial_cmp(&a.1).unwrap_or(Equal)) — stable, descending, NaN compares equal
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return fin[a] > fin[b]; });
    const uint64_t take = std::min<uint64_t>(n, limit);
    for (uint64_t j = 0; j < take; j++) {
        const uint32_t i = order[j];
        memcpy(out_ids + 16 * j, &ids[16 * (size_t)i], 16);
        out_scores[j] = fin[i];
        out_raw_scores[j] = raw[i];
    }
    *n_out = take;
    return CX_OK;
} catch (...) { return cx::on_exception(); }

}  // extern "C"
