// cortex_hip.hpp — C++ host-side mirror of the reference's vector-layer interface over the C ABI
// (header only).  Same names, argument meaning and error behaviour as
// crates/cortex-core/src/vector/index.rs: trait VectorIndex (:50-99), HnswIndex (:182-473),
// VectorFilter (:18-47), SimilarityResult (:11-15); SimilarityConfig from vector/config.rs:3-87.
// Result<T> becomes "returns T or throws CortexError" (the vector layer only produces
// CortexError::Validation(String)).  Everything numeric happens in libcortex_hip.so.
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <exception>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "cortex_hip.h"

namespace cortex {

using NodeId = std::array<uint8_t, 16>;  // types.rs:9 — Uuid
using Embedding = std::vector<float>;    // types.rs:22

struct CortexError : std::runtime_error {  // error.rs:7-50
    int code;
    CortexError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

struct SimilarityResult {  // index.rs:11-15
    NodeId node_id;
    float score;
    float distance;
};

struct VectorFilter {  // index.rs:18-47
    std::optional<std::vector<std::string>> kinds;
    std::optional<std::vector<NodeId>> exclude;
    std::optional<std::string> source_agent;
    static VectorFilter new_() { return {}; }
    VectorFilter with_kinds(std::vector<std::string> k) && { kinds = std::move(k); return std::move(*this); }
    VectorFilter excluding(std::vector<NodeId> ids) && { exclude = std::move(ids); return std::move(*this); }
    VectorFilter with_source_agent(std::string a) && { source_agent = std::move(a); return std::move(*this); }
};

struct VectorIndex {  // index.rs:50-99
    virtual ~VectorIndex() = default;
    virtual void insert(const NodeId &id, const Embedding &embedding) = 0;
    virtual void remove(const NodeId &id) = 0;
    virtual std::vector<SimilarityResult> search(const Embedding &query, size_t k, const VectorFilter *filter = nullptr) const = 0;
    virtual std::vector<SimilarityResult> search_threshold(const Embedding &query, float threshold, const VectorFilter *filter = nullptr) const = 0;
    virtual std::map<NodeId, std::vector<SimilarityResult>> search_batch(const std::vector<std::pair<NodeId, Embedding>> &queries,
                                                                        size_t k, const VectorFilter *filter = nullptr) const = 0;
    virtual size_t len() const = 0;
    bool is_empty() const { return len() == 0; }
    virtual void rebuild() = 0;
    virtual void save(const std::string &path) const = 0;
};

// ScoreDecayConfig (vector/scoring.rs:22-78) with the reference's defaults
struct ScoreDecayConfig {
    bool enabled = true;
    double daily_rate = 0.02, max_age_days = 365.0, min_factor = 0.1, echo_weight = 0.05, echo_cap = 2.0;
    float recency_weight = 0.15f;
    std::vector<std::pair<std::string, double>> by_kind{{"event", 0.05}, {"observation", 0.04}, {"decision", 0.005},
                                                         {"pattern", 0.005}, {"fact", 0.01}, {"preference", 0.005}};
};
struct DecayedResult {  // "node", "score", "raw_score" of the HTTP search response (routes.rs:928-931)
    NodeId node_id;
    float score, raw_score;
};

class HipIndex final : public VectorIndex {
    cx_index *h_ = nullptr;
    explicit HipIndex(cx_index *h) : h_(h) {}
    static void check(int rc) {
        if (rc != CX_OK) throw CortexError(rc, cx_last_error());
    }
    struct FilterBuf {
        std::vector<uint8_t> ex;
        std::vector<uint32_t> kinds;
        cx_filter c{};
    };
    uint32_t intern(const std::string &s) { return cx_intern(h_, s.data(), s.size()); }            // &mut self: adds
    uint32_t lookup(const std::string &s) const { return cx_lookup(h_, s.data(), s.size()); }      // &self: filters
    bool marshal(const VectorFilter *f, FilterBuf &b) const {
        if (!f) return false;
        if (f->exclude) {
            for (auto &id : *f->exclude) b.ex.insert(b.ex.end(), id.begin(), id.end());
            b.c.has_exclude = 1; b.c.n_exclude = f->exclude->size(); b.c.exclude_ids = b.ex.data();
        }
        if (f->kinds) {
            for (auto &k : *f->kinds) b.kinds.push_back(lookup(k));
            b.c.has_kinds = 1; b.c.n_kinds = f->kinds->size(); b.c.kind_codes = b.kinds.data();
        }
        if (f->source_agent) { b.c.has_agent = 1; b.c.agent_code = lookup(*f->source_agent); }
        return true;
    }
    static std::vector<SimilarityResult> collect(const uint8_t *ids, const float *s, const float *d, size_t n) {
        std::vector<SimilarityResult> out(n);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(out[i].node_id.data(), ids + 16 * i, 16);
            out[i].score = s[i];
            out[i].distance = d[i];
        }
        return out;
    }

public:
    // dtype CX_DTYPE_BF16: vectors rounded to bf16 once at insert, 2 bytes per element in HBM (cortex_hip.h: cx_create_ex)
    explicit HipIndex(size_t dimension, int device = 0, int dtype = CX_DTYPE_F32) : h_(cx_create_ex((uint32_t)dimension, device, dtype)) {  // index.rs:204
        if (!h_) throw CortexError(CX_ERR_DEVICE, cx_last_error());
    }
    static HipIndex with_metadata(size_t dimension, int device = 0) { return HipIndex(dimension, device); }  // :214
    HipIndex(HipIndex &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    HipIndex(const HipIndex &) = delete;
    ~HipIndex() override { if (h_) cx_destroy(h_); }

    void set_metadata(const NodeId &id, const std::string &kind, const std::string &source_agent) {  // :219
        check(cx_set_metadata(h_, id.data(), intern(kind), intern(source_agent)));
    }
    void insert(const NodeId &id, const Embedding &e) override { check(cx_upsert(h_, id.data(), e.data(), e.size())); }
    void remove(const NodeId &id) override { check(cx_remove(h_, id.data())); }

    std::vector<SimilarityResult> search(const Embedding &q, size_t k, const VectorFilter *filter = nullptr) const override {
        const size_t cap = std::max<size_t>(1, std::min<size_t>(k, (size_t)cx_row_count(h_)));
        std::vector<uint8_t> ids(16 * cap);
        std::vector<float> s(cap), d(cap);
        uint64_t n = 0;
        FilterBuf b;
        const bool has = marshal(filter, b);
        check(cx_search(h_, q.data(), q.size(), k, has ? &b.c : nullptr, ids.data(), s.data(), d.data(), &n));
        return collect(ids.data(), s.data(), d.data(), (size_t)n);
    }
    std::vector<SimilarityResult> search_threshold(const Embedding &q, float threshold, const VectorFilter *filter = nullptr) const override {
        FilterBuf b;
        const bool has = marshal(filter, b);
        size_t cap = 256;
        for (;;) {
            std::vector<uint8_t> ids(16 * cap);
            std::vector<float> s(cap), d(cap);
            uint64_t n = 0, need = 0;
            const int rc = cx_search_threshold(h_, q.data(), q.size(), threshold, has ? &b.c : nullptr, cap, ids.data(),
                                               s.data(), d.data(), &n, &need);
            if (rc == CX_ERR_CAPACITY) { cap = (size_t)need; continue; }
            check(rc);
            return collect(ids.data(), s.data(), d.data(), (size_t)n);
        }
    }
    std::map<NodeId, std::vector<SimilarityResult>> search_batch(const std::vector<std::pair<NodeId, Embedding>> &queries, size_t k,
                                                                const VectorFilter *filter = nullptr) const override {
        std::map<NodeId, std::vector<SimilarityResult>> out;
        if (queries.empty()) return out;
        const size_t nq = queries.size(), len = queries[0].second.size(), kk = std::max<size_t>(k, 1);
        std::vector<float> flat;
        for (auto &q : queries) flat.insert(flat.end(), q.second.begin(), q.second.end());
        std::vector<uint8_t> ids(16 * nq * kk);
        std::vector<float> s(nq * kk), d(nq * kk);
        std::vector<uint64_t> counts(nq);
        FilterBuf b;
        const bool has = marshal(filter, b);
        check(cx_search_batch(h_, nq, flat.data(), len, k, has ? &b.c : nullptr, ids.data(), s.data(), d.data(), counts.data()));
        for (size_t i = 0; i < nq; i++)
            out[queries[i].first] = collect(ids.data() + 16 * i * kk, s.data() + i * kk, d.data() + i * kk, (size_t)counts[i]);
        return out;
    }
    size_t len() const override { return (size_t)cx_len(h_); }
    void rebuild() override { check(cx_rebuild(h_)); }
    void save(const std::string &path) const override { check(cx_save(h_, path.c_str())); }
    static HipIndex load(const std::string &path, int device = 0, int dtype = CX_DTYPE_F32) {  // :447-473
        cx_index *h = cx_load_ex(path.c_str(), device, dtype);
        if (!h) throw CortexError(CX_ERR_VALIDATION, cx_last_error());
        return HipIndex(h);
    }
    // The start-up loop over stored nodes (serve.rs:105-123 / api.rs:56-70) in one call: `records` are the raw
    // bincode values of the reference's nodes table.  strict = Cortex::open (a wrong-length embedding throws).
    cx_bulk_stats bulk_load_nodes(const std::vector<std::vector<uint8_t>> &records, bool strict = false, uint32_t extra_flags = 0) {
        std::vector<uint64_t> offs(records.size() + 1, 0);
        for (size_t i = 0; i < records.size(); i++) offs[i + 1] = offs[i] + records[i].size();
        std::vector<uint8_t> blob((size_t)offs.back() + 1);
        for (size_t i = 0; i < records.size(); i++) std::copy(records[i].begin(), records[i].end(), blob.begin() + (size_t)offs[i]);
        cx_bulk_stats st{};
        check(cx_bulk_load_nodes(h_, records.size(), blob.data(), offs.data(), (strict ? CX_BULK_STRICT : 0u) | extra_flags, &st));
        return st;
    }
    // kind / last_accessed_at / access_count of nodes: the inputs of apply_score_decay (scoring.rs:84-114)
    void set_node_stats(const std::vector<NodeId> &ids, const std::vector<std::string> &kinds,
                        const std::vector<int64_t> &last_accessed_s, const std::vector<uint64_t> &access_counts) {
        std::vector<uint8_t> flat(16 * ids.size());
        std::vector<uint32_t> kc(ids.size());
        for (size_t i = 0; i < ids.size(); i++) { std::copy(ids[i].begin(), ids[i].end(), flat.begin() + 16 * i); kc[i] = intern(kinds[i]); }
        check(cx_set_node_stats_batch(h_, ids.size(), flat.data(), kc.data(), last_accessed_s.data(), nullptr, access_counts.data()));
    }
    // the HTTP search handler's candidates -> decay -> stable re-rank -> truncate (routes.rs:889-947)
    std::vector<DecayedResult> search_decayed(const Embedding &q, size_t limit, const ScoreDecayConfig &cfg, float recency_bias,
                                              int64_t now_s, uint32_t now_ns = 0, const VectorFilter *filter = nullptr) const {
        std::vector<uint32_t> codes;
        std::vector<double> rates;
        for (auto &kv : cfg.by_kind) { codes.push_back(lookup(kv.first)); rates.push_back(kv.second); }
        cx_decay_config c{cfg.enabled ? 1 : 0, cfg.daily_rate, cfg.max_age_days, cfg.min_factor, cfg.echo_weight, cfg.echo_cap,
                          cfg.recency_weight, (uint32_t)codes.size(), codes.data(), rates.data()};
        const size_t cand = cfg.enabled && recency_bias > 0.0f ? std::max<size_t>(3 * limit, 30) : limit;   // :899-903
        const size_t cap = std::max<size_t>(1, limit);
        std::vector<uint8_t> ids(16 * cap);
        std::vector<float> sc(cap), raw(cap);
        uint64_t n = 0;
        FilterBuf fb;
        const cx_filter *cf = marshal(filter, fb) ? &fb.c : nullptr;
        check(cx_search_decayed(h_, q.data(), q.size(), limit, cand, cf, &c, recency_bias, now_s, now_ns, ids.data(), sc.data(), raw.data(), &n));
        std::vector<DecayedResult> out(n);
        for (uint64_t i = 0; i < n; i++) { std::copy(ids.begin() + 16 * i, ids.begin() + 16 * i + 16, out[i].node_id.begin()); out[i].score = sc[i]; out[i].raw_score = raw[i]; }
        return out;
    }
    cx_index *raw() const { return h_; }
};

// One process, several GPUs (SURVEY §8b/§8e for a server that is a single process, like the reference's): the same
// trait over `cx_sharded` (cortex_hip.h) — one shard per listed device, shard scans concurrent, partial top-k lists
// published peer-to-peer into the root device's gather buffer and merged there, ties by global insertion order exactly
// as in a single index.  Everything happens behind the C ABI; this class only marshals.  (The multi-PROCESS variant with
// an RCCL all-gather is cortex_amd/sharded.py.)
class ShardedHipIndex final : public VectorIndex {
    cx_sharded *h_;
    static void check(int rc) {
        if (rc != CX_OK) throw CortexError(rc, cx_last_error());
    }
    struct FilterBuf {
        std::vector<uint8_t> ex;
        std::vector<uint32_t> kinds;
        cx_filter c{};
    };
    bool marshal(const VectorFilter *f, FilterBuf &b) const {
        if (!f) return false;
        if (f->exclude) {
            for (auto &id : *f->exclude) b.ex.insert(b.ex.end(), id.begin(), id.end());
            b.c.has_exclude = 1; b.c.n_exclude = f->exclude->size(); b.c.exclude_ids = b.ex.data();
        }
        if (f->kinds) {
            for (auto &k : *f->kinds) b.kinds.push_back(cx_sharded_lookup(h_, k.data(), k.size()));
            b.c.has_kinds = 1; b.c.n_kinds = f->kinds->size(); b.c.kind_codes = b.kinds.data();
        }
        if (f->source_agent) { b.c.has_agent = 1; b.c.agent_code = cx_sharded_lookup(h_, f->source_agent->data(), f->source_agent->size()); }
        return true;
    }
    static std::vector<SimilarityResult> collect(const uint8_t *ids, const float *s, const float *d, size_t n) {
        std::vector<SimilarityResult> out(n);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(out[i].node_id.data(), ids + 16 * i, 16);
            out[i].score = s[i];
            out[i].distance = d[i];
        }
        return out;
    }

public:
    ShardedHipIndex(size_t dimension, const std::vector<int> &devices, int dtype = CX_DTYPE_F32)
        : h_(cx_sharded_create_ex((uint32_t)dimension, (uint32_t)devices.size(), devices.data(), dtype)) {
        if (!h_) throw CortexError(CX_ERR_DEVICE, cx_last_error());
    }
    explicit ShardedHipIndex(cx_sharded *adopt) : h_(adopt) {}
    ShardedHipIndex(ShardedHipIndex &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ShardedHipIndex(const ShardedHipIndex &) = delete;
    ~ShardedHipIndex() override { if (h_) cx_sharded_destroy(h_); }
    size_t n_shards() const { return cx_sharded_n_shards(h_); }
    cx_sharded *raw() const { return h_; }
    void insert(const NodeId &id, const Embedding &e) override { check(cx_sharded_upsert(h_, id.data(), e.data(), e.size())); }
    void remove(const NodeId &id) override { check(cx_sharded_remove(h_, id.data())); }
    void set_metadata(const NodeId &id, const std::string &kind, const std::string &agent) {
        check(cx_sharded_set_metadata(h_, id.data(), cx_sharded_intern(h_, kind.data(), kind.size()), cx_sharded_intern(h_, agent.data(), agent.size())));
    }
    std::vector<SimilarityResult> search(const Embedding &q, size_t k, const VectorFilter *f = nullptr) const override {
        const size_t cap = std::max<size_t>(1, std::min<size_t>(k, cx_sharded_row_count(h_)));
        std::vector<uint8_t> ids(16 * cap);
        std::vector<float> sc(cap), di(cap);
        uint64_t n = 0;
        FilterBuf fb;
        const cx_filter *cf = marshal(f, fb) ? &fb.c : nullptr;
        check(cx_sharded_search(h_, q.data(), q.size(), k, cf, ids.data(), sc.data(), di.data(), &n));
        return collect(ids.data(), sc.data(), di.data(), n);
    }
    std::vector<SimilarityResult> search_threshold(const Embedding &q, float thr, const VectorFilter *f = nullptr) const override {
        FilterBuf fb;
        const cx_filter *cf = marshal(f, fb) ? &fb.c : nullptr;
        uint64_t cap = 256;
        for (;;) {
            std::vector<uint8_t> ids(16 * cap);
            std::vector<float> sc(cap), di(cap);
            uint64_t n = 0, need = 0;
            const int rc = cx_sharded_search_threshold(h_, q.data(), q.size(), thr, cf, cap, ids.data(), sc.data(), di.data(), &n, &need);
            if (rc == CX_ERR_CAPACITY && need > cap) { cap = need; continue; }
            check(rc);
            return collect(ids.data(), sc.data(), di.data(), n);
        }
    }
    std::map<NodeId, std::vector<SimilarityResult>> search_batch(const std::vector<std::pair<NodeId, Embedding>> &queries, size_t k,
                                                                const VectorFilter *f = nullptr) const override {
        std::map<NodeId, std::vector<SimilarityResult>> out;
        if (queries.empty()) return out;
        const size_t nq = queries.size(), len = queries[0].second.size(), kk = std::max<size_t>(1, k);
        std::vector<float> flat;
        flat.reserve(nq * len);
        for (auto &q : queries) {
            if (q.second.size() != len) throw CortexError(CX_ERR_VALIDATION, "search_batch: queries of different lengths");
            flat.insert(flat.end(), q.second.begin(), q.second.end());
        }
        std::vector<uint8_t> ids(16 * nq * kk);
        std::vector<float> sc(nq * kk), di(nq * kk);
        std::vector<uint64_t> counts(nq);
        FilterBuf fb;
        const cx_filter *cf = marshal(f, fb) ? &fb.c : nullptr;
        check(cx_sharded_search_batch(h_, nq, flat.data(), len, k, cf, ids.data(), sc.data(), di.data(), counts.data()));
        for (size_t i = 0; i < nq; i++) out[queries[i].first] = collect(&ids[16 * i * kk], &sc[i * kk], &di[i * kk], counts[i]);
        return out;
    }
    size_t len() const override { return cx_sharded_len(h_); }
    void rebuild() override { check(cx_sharded_rebuild(h_)); }
    void save(const std::string &path) const override { check(cx_sharded_save(h_, path.c_str())); }   // :437-445: one file, the single index's layout
    static ShardedHipIndex load(const std::string &path, const std::vector<int> &devices, int dtype = CX_DTYPE_F32) {   // :447-473
        cx_sharded *h = cx_sharded_load_ex(path.c_str(), (uint32_t)devices.size(), devices.data(), dtype);
        if (!h) throw CortexError(CX_ERR_VALIDATION, cx_last_error());
        return ShardedHipIndex(h);
    }
};

struct SimilarityConfig {  // vector/config.rs:3-87
    float auto_link_threshold = 0.75f, dedup_threshold = 0.92f, contradiction_threshold = 0.80f;
    size_t auto_link_k = 20;
    static float clamp01(float t) { return t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t); }
    SimilarityConfig &with_auto_link_threshold(float t) { auto_link_threshold = clamp01(t); return *this; }
    SimilarityConfig &with_dedup_threshold(float t) { dedup_threshold = clamp01(t); return *this; }
    SimilarityConfig &with_contradiction_threshold(float t) { contradiction_threshold = clamp01(t); return *this; }
    SimilarityConfig &with_auto_link_k(size_t k) { auto_link_k = k; return *this; }
    void validate() const {
        if (auto_link_threshold >= dedup_threshold) throw CortexError(CX_ERR_VALIDATION, "auto_link_threshold must be less than dedup_threshold");
        if (contradiction_threshold >= dedup_threshold) throw CortexError(CX_ERR_VALIDATION, "contradiction_threshold must be less than dedup_threshold");
        if (auto_link_k == 0) throw CortexError(CX_ERR_VALIDATION, "auto_link_k must be greater than 0");
    }
};


// ---- auto-linker / dedup scanner over the index (linker/auto_linker.rs:215-264, rules.rs:42-62, dedup.rs:65-127) ----

struct ProposedEdge {   // linker/rules.rs:7-14 as SimilarityLinkRule fills it: relation "related_to", weight = score
    NodeId from, to;
    float weight;
};
struct DuplicatePair {  // linker/dedup.rs: DuplicatePair { node_a, node_b, similarity }
    NodeId node_a, node_b;
    float similarity;
};
struct NeighbourLists {  // `search(&embedding, 100, None)` of every scanned node, in scan order
    size_t topk = 0;
    std::vector<NodeId> scanned;
    std::vector<uint32_t> rows, counts;   // rows [n][topk] (insertion rows), counts [n]
    std::vector<float> scores;            // [n][topk]
};

class Linker {
    const HipIndex &ix_;
    static void check(int rc) {
        if (rc != CX_OK) throw CortexError(rc, cx_last_error());
    }
    std::vector<uint32_t> rows_of(const std::vector<NodeId> &ids) const {
        std::vector<uint8_t> flat;
        flat.reserve(ids.size() * 16);
        for (auto &id : ids) flat.insert(flat.end(), id.begin(), id.end());
        std::vector<uint32_t> rows(ids.size());
        check(cx_rows_of(ix_.raw(), ids.size(), flat.data(), rows.data()));
        for (uint32_t r : rows)
            if (r == 0xFFFFFFFFu) throw CortexError(CX_ERR_VALIDATION, "node has no embedding in the index");
        return rows;
    }
    std::vector<uint8_t> deleted_flags(const std::vector<NodeId> *deleted) const {
        std::vector<uint8_t> f;
        if (!deleted || deleted->empty()) return f;
        f.assign((size_t)cx_row_count(ix_.raw()), 0);
        for (uint32_t r : rows_of(*deleted)) f[r] = 1;
        return f;
    }
    NodeId id_of(uint32_t row) const {
        NodeId id;
        check(cx_row_id(ix_.raw(), row, id.data()));
        return id;
    }

public:
    explicit Linker(const HipIndex &ix) : ix_(ix) {}

    /// The edges `run_cycle` proposes from SimilarityLinkRule for the scanned nodes (scan order, then score order).
    /// scan = nullptr scans every node.  existing = per scanned node (scan order; with scan = nullptr: row order) the
    /// targets of its outgoing related_to edges — `storage.edges_from(node.id)` restricted to the relation this rule
    /// proposes (auto_linker.rs:226-231); such neighbours are dropped without counting towards max_edges_per_node
    /// (:249-258).  max_edges_per_cycle: :284-287 (SIZE_MAX = no truncation).
    std::vector<ProposedEdge> similarity_edges(const std::vector<NodeId> *scan, const SimilarityConfig &cfg,
                                               size_t max_edges_per_node = 50, const std::vector<NodeId> *deleted = nullptr,
                                               size_t topk = 100, const std::vector<std::vector<NodeId>> *existing = nullptr,
                                               size_t max_edges_per_cycle = SIZE_MAX) const {
        cfg.validate();
        std::vector<uint32_t> scan_rows;
        if (scan) scan_rows = rows_of(*scan);
        const std::vector<uint8_t> del = deleted_flags(deleted);
        const uint64_t n_scan = scan ? scan_rows.size() : cx_row_count(ix_.raw());
        std::vector<uint64_t> ex_off;
        std::vector<uint32_t> ex_to;
        if (existing) {
            if (existing->size() != n_scan) throw CortexError(CX_ERR_VALIDATION, "existing: one entry per scanned node");
            ex_off.assign(1, 0);
            for (auto &targets : *existing) {
                if (!targets.empty()) {
                    std::vector<uint8_t> flat;
                    for (auto &id : targets) flat.insert(flat.end(), id.begin(), id.end());
                    std::vector<uint32_t> rows(targets.size());
                    check(cx_rows_of(ix_.raw(), targets.size(), flat.data(), rows.data()));
                    for (uint32_t r : rows)
                        if (r != 0xFFFFFFFFu) ex_to.push_back(r);   // a target without an embedding never shows up in a list
                }
                ex_off.push_back(ex_to.size());
            }
        }
        uint64_t cap = std::max<uint64_t>(1024, n_scan * 4), n = 0, need = 0;
        std::vector<uint32_t> from, to;
        std::vector<float> w;
        for (;;) {
            from.resize(cap); to.resize(cap); w.resize(cap);
            const int rc = cx_autolink_pass_rows(ix_.raw(), n_scan, scan ? scan_rows.data() : nullptr, topk, cfg.auto_link_threshold,
                                                 max_edges_per_node, max_edges_per_cycle == SIZE_MAX ? UINT64_MAX : max_edges_per_cycle,
                                                 del.empty() ? nullptr : del.data(), existing ? ex_off.data() : nullptr,
                                                 existing ? ex_to.data() : nullptr, cap, from.data(), to.data(), w.data(), &n, &need);
            if (rc == CX_ERR_CAPACITY && need > cap) { cap = need; continue; }
            check(rc);
            break;
        }
        std::vector<ProposedEdge> out(n);
        for (uint64_t i = 0; i < n; i++) out[i] = ProposedEdge{id_of(from[i]), id_of(to[i]), w[i]};
        return out;
    }

    /// DedupScanner::scan's pairs: every indexed node in row order, neighbours with score >= dedup_threshold.
    std::vector<DuplicatePair> dedup_scan(const SimilarityConfig &cfg, const std::vector<NodeId> *deleted = nullptr) const {
        const std::vector<uint8_t> del = deleted_flags(deleted);
        uint64_t cap = std::max<uint64_t>(1024, cx_row_count(ix_.raw())), n = 0, need = 0;
        std::vector<uint32_t> a, b;
        std::vector<float> s;
        for (;;) {
            a.resize(cap); b.resize(cap); s.resize(cap);
            const int rc = cx_dedup_scan_rows(ix_.raw(), cfg.dedup_threshold, del.empty() ? nullptr : del.data(), cap, a.data(),
                                              b.data(), s.data(), &n, &need);
            if (rc == CX_ERR_CAPACITY && need > cap) { cap = need; continue; }
            check(rc);
            break;
        }
        std::vector<DuplicatePair> out(n);
        for (uint64_t i = 0; i < n; i++) out[i] = DuplicatePair{id_of(a[i]), id_of(b[i]), s[i]};
        return out;
    }

    /// `search(&embedding, 100, None)` (auto_linker.rs:221) for a whole batch of nodes: what every other rule walks.
    NeighbourLists neighbour_lists(const std::vector<NodeId> &scan, size_t topk = 100) const {
        NeighbourLists L;
        L.topk = topk;
        L.scanned = scan;
        const std::vector<uint32_t> scan_rows = rows_of(scan);
        L.rows.assign(scan.size() * topk, 0);
        L.scores.assign(scan.size() * topk, 0.0f);
        L.counts.assign(scan.size(), 0);
        check(cx_topk_lists_rows(ix_.raw(), scan.size(), scan_rows.data(), topk, L.rows.data(), L.scores.data(), L.counts.data()));
        return L;
    }
    NodeId node_of_row(uint32_t row) const { return id_of(row); }
};

}  // namespace cortex
