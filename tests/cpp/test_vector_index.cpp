// The reference's own vector-layer tests (crates/cortex-core/src/vector/index.rs:484-728 and
// vector/config.rs:93-135), written against the C++ host mirror (include/cortex_hip.hpp).
// Needs a GPU; run by tests/test_hip_cpp_mirror.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "cortex_hip.hpp"

using namespace cortex;

static int g_failed = 0, g_run = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); g_failed++; } \
    } while (0)
#define TEST(name) static void name(); static void run_##name() { g_run++; std::fprintf(stderr, "[ RUN ] %s\n", #name); name(); } static void name()

static NodeId now_v7() {
    static std::mt19937_64 rng(42);
    NodeId id;
    for (auto &b : id) b = (uint8_t)rng();
    return id;
}

TEST(test_index_insert_and_search) {  // :484-510
    HipIndex index(3);
    auto id1 = now_v7(), id2 = now_v7(), id3 = now_v7();
    index.insert(id1, {1.0f, 0.0f, 0.0f});
    index.insert(id2, {0.9f, 0.1f, 0.0f});
    index.insert(id3, {0.0f, 1.0f, 0.0f});
    index.rebuild();
    auto results = index.search({1.0f, 0.0f, 0.0f}, 2);
    CHECK(results.size() == 2);
    CHECK(results[0].node_id == id1);
}
TEST(test_threshold_search) {  // :513-535
    HipIndex index(3);
    auto id1 = now_v7(), id2 = now_v7();
    index.insert(id1, {1.0f, 0.0f, 0.0f});
    index.insert(id2, {0.0f, 1.0f, 0.0f});
    index.rebuild();
    auto results = index.search_threshold({1.0f, 0.0f, 0.0f}, 0.95f);
    CHECK(results.size() == 1);
    CHECK(results[0].node_id == id1);
}
TEST(test_index_persistence) {  // :538-566
    const std::string path = std::string(std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp") + "/cx_cpp_test.hnsw";
    HipIndex index(3);
    auto id1 = now_v7();
    index.insert(id1, {1.0f, 0.0f, 0.0f});
    index.rebuild();
    index.save(path);
    HipIndex loaded = HipIndex::load(path);
    CHECK(loaded.len() == 1);
    auto results = loaded.search({1.0f, 0.0f, 0.0f}, 1);
    CHECK(results.size() == 1 && results[0].node_id == id1);
    std::remove(path.c_str());
}
TEST(test_dimension_mismatch_rejected) {  // :579-583
    HipIndex index(3);
    bool threw = false;
    try { index.insert(now_v7(), {1.0f, 2.0f}); }
    catch (const CortexError &e) { threw = std::string(e.what()) == "Embedding dimension mismatch: expected 3, got 2"; }
    CHECK(threw);
}
TEST(test_empty_index_search) {  // :586-590
    HipIndex index(3);
    CHECK(index.search({1.0f, 0.0f, 0.0f}, 5).empty());
}
TEST(test_brute_force_fallback) {  // :593-606
    HipIndex index(3);
    auto id1 = now_v7(), id2 = now_v7();
    index.insert(id1, {1.0f, 0.0f, 0.0f});
    index.insert(id2, {0.0f, 1.0f, 0.0f});
    auto results = index.search({1.0f, 0.0f, 0.0f}, 2);  // no rebuild
    CHECK(results.size() == 2 && results[0].node_id == id1);
}
TEST(test_filter_by_kind) {  // :609-627
    HipIndex index(3);
    auto id1 = now_v7(), id2 = now_v7();
    index.insert(id1, {1.0f, 0.0f, 0.0f});
    index.set_metadata(id1, "fact", "test");
    index.insert(id2, {0.9f, 0.1f, 0.0f});
    index.set_metadata(id2, "decision", "test");
    index.rebuild();
    auto filter = VectorFilter::new_().with_kinds({"decision"});
    auto results = index.search({1.0f, 0.0f, 0.0f}, 5, &filter);
    CHECK(results.size() == 1 && results[0].node_id == id2);
}
TEST(test_filter_exclude) {  // :630-646
    HipIndex index(3);
    auto id1 = now_v7(), id2 = now_v7();
    index.insert(id1, {1.0f, 0.0f, 0.0f});
    index.insert(id2, {0.9f, 0.1f, 0.0f});
    index.rebuild();
    auto filter = VectorFilter::new_().excluding({id1});
    auto results = index.search({1.0f, 0.0f, 0.0f}, 5, &filter);
    CHECK(results.size() == 1 && results[0].node_id == id2);
}
TEST(test_remove_doesnt_crash_search) {  // :649-664
    HipIndex index(3);
    auto id1 = now_v7(), id2 = now_v7();
    index.insert(id1, {1.0f, 0.0f, 0.0f});
    index.insert(id2, {0.0f, 1.0f, 0.0f});
    index.rebuild();
    index.remove(id1);
    CHECK(index.len() == 1);
    CHECK(!index.search({1.0f, 0.0f, 0.0f}, 5).empty());
}
TEST(test_search_batch) {  // :667-684
    HipIndex index(3);
    auto id1 = now_v7(), id2 = now_v7(), id3 = now_v7();
    index.insert(id1, {1.0f, 0.0f, 0.0f});
    index.insert(id2, {0.0f, 1.0f, 0.0f});
    index.insert(id3, {0.0f, 0.0f, 1.0f});
    index.rebuild();
    auto results = index.search_batch({{id1, {1.0f, 0.0f, 0.0f}}, {id2, {0.0f, 1.0f, 0.0f}}}, 1);
    CHECK(results.size() == 2);
    CHECK(results[id1][0].node_id == id1);
    CHECK(results[id2][0].node_id == id2);
}
TEST(test_similarity_score_range) {  // :687-708
    HipIndex index(3);
    index.insert(now_v7(), {1.0f, 0.0f, 0.0f});
    index.insert(now_v7(), {-1.0f, 0.0f, 0.0f});
    index.rebuild();
    auto results = index.search({1.0f, 0.0f, 0.0f}, 2);
    for (auto &r : results) CHECK(r.score >= 0.0f && r.score <= 1.0f);
    CHECK(results[0].score > 0.99f);
}
TEST(test_threshold_returns_only_above) {  // :711-728
    HipIndex index(3);
    auto id_close = now_v7(), id_far = now_v7();
    index.insert(id_close, {1.0f, 0.0f, 0.0f});
    index.insert(id_far, {0.0f, 0.0f, 1.0f});
    index.rebuild();
    auto results = index.search_threshold({1.0f, 0.0f, 0.0f}, 0.5f);
    bool all_above = true, any_close = false;
    for (auto &r : results) { all_above &= r.score >= 0.5f; any_close |= r.node_id == id_close; }
    CHECK(all_above && any_close);
}
TEST(test_config) {  // vector/config.rs:93-135
    SimilarityConfig c;
    CHECK(c.auto_link_threshold == 0.75f && c.dedup_threshold == 0.92f && c.contradiction_threshold == 0.80f && c.auto_link_k == 20);
    c.validate();
    SimilarityConfig bad;
    bad.with_auto_link_threshold(0.95f).with_dedup_threshold(0.90f);
    bool threw = false;
    try { bad.validate(); } catch (const CortexError &) { threw = true; }
    CHECK(threw);
    SimilarityConfig cl;
    cl.with_auto_link_threshold(1.5f).with_dedup_threshold(-0.5f);
    CHECK(cl.auto_link_threshold == 1.0f && cl.dedup_threshold == 0.0f);
}

// ---- linker passes over the index (no counterpart test in the reference: its auto-linker tests need storage and an
// embedding model; these pin the rule semantics of auto_linker.rs:215-264 / rules.rs:42-62 / dedup.rs:65-127 on
// hand-made vectors)
static Embedding unit3(float a, float b, float c, size_t dim) {
    Embedding v(dim, 0.0f);
    v[0] = a; v[1] = b; v[2] = c;
    return v;
}
TEST(test_linker_similarity_edges_and_dedup) {
    const size_t dim = 64;   // a multiple of 64: the MFMA filter path; exact rescoring decides
    HipIndex index(dim);
    NodeId a = now_v7(), b = now_v7(), c = now_v7(), d = now_v7();
    index.insert(a, unit3(1.0f, 0.0f, 0.0f, dim));
    index.insert(b, unit3(0.99f, 0.1f, 0.0f, dim));    // cos(a,b) = 0.9949
    index.insert(c, unit3(0.8f, 0.6f, 0.0f, dim));     // cos(a,c) = 0.8, cos(b,c) = 0.852/0.9950*... ~0.8563
    index.insert(d, unit3(0.0f, 0.0f, 1.0f, dim));     // orthogonal to all
    Linker linker(index);
    SimilarityConfig cfg;
    cfg.with_auto_link_threshold(0.75f).with_dedup_threshold(0.99f);
    auto edges = linker.similarity_edges(nullptr, cfg);
    // a: b (0.995), c (0.8);  b: a, c (0.856);  c: b, a;  d: none  -> 6 directed edges, scan order then score order
    CHECK(edges.size() == 6);
    if (edges.size() == 6) {
        CHECK(edges[0].from == a && edges[0].to == b && edges[1].from == a && edges[1].to == c);
        CHECK(edges[2].from == b && edges[2].to == a && edges[3].from == b && edges[3].to == c);
        CHECK(edges[4].from == c && edges[4].to == b && edges[5].from == c && edges[5].to == a);
        CHECK(std::fabs(edges[0].weight - edges[2].weight) < 1e-6f);   // the same pair scores the same both ways
        CHECK(edges[1].weight > 0.79f && edges[1].weight < 0.81f);
    }
    // per-node cap of 1: only the best neighbour of each node
    auto capped = linker.similarity_edges(nullptr, cfg, 1);
    CHECK(capped.size() == 3);
    // a deleted (storage-tombstoned) neighbour is skipped; its own scan still runs (the reference scans new nodes only,
    // the caller decides what to scan)
    std::vector<NodeId> del{b};
    std::vector<NodeId> scan{a};
    auto e2 = linker.similarity_edges(&scan, cfg, 50, &del);
    CHECK(e2.size() == 1 && e2[0].to == c);
    // dedup: only (a, b) is >= 0.99, reported once by the node scanned first
    auto dups = linker.dedup_scan(cfg);
    CHECK(dups.size() == 1);
    if (dups.size() == 1) CHECK(dups[0].node_a == a && dups[0].node_b == b && dups[0].similarity > 0.99f);
    // the ordered neighbour lists every other rule walks: self first (score 1), then by score
    auto L = linker.neighbour_lists(scan, 3);
    CHECK(L.counts.size() == 1 && L.counts[0] == 3);
    if (L.counts.size() == 1 && L.counts[0] == 3) {
        CHECK(linker.node_of_row(L.rows[0]) == a && linker.node_of_row(L.rows[1]) == b && linker.node_of_row(L.rows[2]) == c);
        CHECK(L.scores[0] > 0.9999f && L.scores[1] > L.scores[2]);
    }
}

// Start-up bulk load (serve.rs:105-123) over stored-node records; the first record is the reference's golden
// Node (redb_storage.rs:1827-1857: no embedding), the others are the same layout with an embedding appended.
static void put_u64(std::vector<uint8_t> &o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }
static void put_str(std::vector<uint8_t> &o, const std::string &s) { put_u64(o, s.size()); o.insert(o.end(), s.begin(), s.end()); }
static std::vector<uint8_t> node_record(const NodeId &id, const Embedding *e, const std::string &created, bool deleted) {
    std::vector<uint8_t> o;
    put_u64(o, 16); o.insert(o.end(), id.begin(), id.end());
    put_str(o, "fact"); put_str(o, "title"); put_str(o, "body");
    put_u64(o, 0); put_u64(o, 0);                      // metadata {}, tags []
    if (e) { o.push_back(1); put_u64(o, e->size()); const uint8_t *p = (const uint8_t *)e->data(); o.insert(o.end(), p, p + 4 * e->size()); }
    else o.push_back(0);
    put_str(o, "kai"); o.push_back(0); o.push_back(0);  // source {agent, None, None}
    const float imp = 0.5f; const uint8_t *ip = (const uint8_t *)&imp; o.insert(o.end(), ip, ip + 4);
    put_u64(o, 0);
    put_str(o, "1970-01-01T00:00:00Z"); put_str(o, created); put_str(o, created);
    o.push_back(deleted ? 1 : 0);
    return o;
}
TEST(test_bulk_load_nodes) {
    HipIndex index(3);
    NodeId a = now_v7(), b = now_v7(), c = now_v7(), d = now_v7(), e = now_v7();
    Embedding ea{1.0f, 0.0f, 0.0f}, eb{0.9f, 0.1f, 0.0f}, ec{0.0f, 1.0f, 0.0f}, ed{1.0f, 0.0f}, ee{0.0f, 0.0f, 1.0f};
    std::vector<std::vector<uint8_t>> recs{
        node_record(a, &ea, "2024-01-01T00:00:00Z", false),
        node_record(b, &eb, "2024-01-02T00:00:00Z", false),      // newest: row 0
        node_record(c, &ec, "2024-01-01T00:00:00Z", true),       // deleted: not listed
        node_record(d, &ed, "2024-01-01T12:00:00Z", false),      // wrong length: insert fails, skipped by the server
        node_record(e, nullptr, "2024-01-03T00:00:00Z", false),  // no embedding
        node_record(e, &ee, "2024-01-01T00:00:00Z", false),
        std::vector<uint8_t>{1, 2, 3}};                          // corrupt
    recs.back().resize(3);
    auto st = index.bulk_load_nodes(recs);
    CHECK(st.records == 7 && st.undecodable == 1 && st.deleted == 1 && st.no_embedding == 1 && st.dim_mismatch == 1 && st.indexed == 3);
    CHECK(index.len() == 3);
    uint8_t rid[16];
    CHECK(cx_row_id(index.raw(), 0, rid) == 0 && std::equal(rid, rid + 16, b.begin()));   // newest first
    CHECK(cx_row_id(index.raw(), 1, rid) == 0 && std::equal(rid, rid + 16, a.begin()));   // ties keep table order
    CHECK(cx_row_id(index.raw(), 2, rid) == 0 && std::equal(rid, rid + 16, e.begin()));
    auto res = index.search({1.0f, 0.0f, 0.0f}, 2);
    CHECK(res.size() == 2 && res[0].node_id == a && res[1].node_id == b);
    HipIndex strict(3);
    bool threw = false;
    try { strict.bulk_load_nodes(recs, true); } catch (const CortexError &err) {
        threw = std::string(err.what()).find("Embedding dimension mismatch: expected 3, got 2") != std::string::npos;
    }
    CHECK(threw);
}

// Query-time decay + re-rank (scoring.rs:84-114, routes.rs:889-947): a stale event falls behind a fresh fact that
// the raw cosine ranks lower.
TEST(test_search_decayed_reranks) {
    HipIndex index(3);
    NodeId stale = now_v7(), fresh = now_v7(), far = now_v7();
    index.insert(stale, {1.0f, 0.0f, 0.0f});      // raw 1.0
    index.insert(fresh, {0.98f, 0.199f, 0.0f});   // raw ~0.98
    index.insert(far, {0.0f, 1.0f, 0.0f});
    const int64_t now = 1760000000;
    index.set_node_stats({stale, fresh, far}, {"event", "fact", "fact"}, {now - 400 * 86400, now, now}, {0, 10, 0});
    ScoreDecayConfig cfg;
    auto plain = index.search_decayed({1.0f, 0.0f, 0.0f}, 2, cfg, 0.0f, now);
    CHECK(plain.size() == 2 && plain[0].node_id == stale && plain[0].score == plain[0].raw_score);
    auto r = index.search_decayed({1.0f, 0.0f, 0.0f}, 2, cfg, 0.5f, now);
    // stale: 1.0 * 0.5 + 1.0 * 0.1 * 1.0 * 0.5 = 0.55;  fresh: 0.98 * 0.5 + 0.98 * 1.0 * 1.5 * 0.5 = 1.225
    CHECK(r.size() == 2 && r[0].node_id == fresh && r[1].node_id == stale);
    if (r.size() == 2) {
        CHECK(std::fabs(r[1].score - 0.55f) < 1e-3f && std::fabs(r[1].raw_score - 1.0f) < 1e-6f);
        CHECK(std::fabs(r[0].score - 1.225f) < 5e-3f);
    }
    // scoring.rs:136-151: disabled config / zero bias return the raw score
    cx_decay_config off{};
    CHECK(cx_apply_score_decay(&off, 0.8f, 0.15f, now, 0, 0, now, 0, 0) == 0.8f);
}

// One process, several shards (here: three shards on the one device of the test box): results must be those of a
// single index over the same insertions — ids, order (ties by global insertion order) and scores.
TEST(test_sharded_single_process) {
    const size_t dim = 384, n = 3000;
    setenv("CX_SHARD_PLACEMENT_BLOCK", "128", 1);   // 24 placement blocks over the three shards instead of one
    std::mt19937 rng(7);
    std::normal_distribution<float> g(0.0f, 1.0f);
    std::vector<NodeId> ids(n);
    std::vector<Embedding> rows(n, Embedding(dim));
    for (size_t i = 0; i < n; i++) { ids[i] = now_v7(); for (auto &x : rows[i]) x = g(rng); }
    for (size_t i = 10; i < n; i += 97) rows[i] = rows[i - 7];       // exact duplicates: the tie order is checked too
    HipIndex one(dim);
    ShardedHipIndex many(dim, {0, 0, 0});
    for (size_t i = 0; i < n; i++) { one.insert(ids[i], rows[i]); many.insert(ids[i], rows[i]); }
    one.insert(ids[5], rows[6]); many.insert(ids[5], rows[6]);        // upsert of a known id keeps its place
    for (size_t i = 0; i < n; i += 11) { one.remove(ids[i]); many.remove(ids[i]); }
    CHECK(many.len() == one.len() && many.n_shards() == 3);
    auto same = [&](const std::vector<SimilarityResult> &a, const std::vector<SimilarityResult> &b) {
        if (a.size() != b.size()) return false;
        for (size_t i = 0; i < a.size(); i++) if (!(a[i].node_id == b[i].node_id) || a[i].score != b[i].score) return false;
        return true;
    };
    for (size_t qi : {1u, 17u, 107u, 2999u}) {
        CHECK(same(many.search(rows[qi], 10), one.search(rows[qi], 10)));
        CHECK(same(many.search(rows[qi], 200), one.search(rows[qi], 200)));
        CHECK(same(many.search_threshold(rows[qi], 0.2f), one.search_threshold(rows[qi], 0.2f)));
    }
    VectorFilter f = VectorFilter::new_().excluding({ids[17], ids[18]});
    CHECK(same(many.search(rows[17], 5, &f), one.search(rows[17], 5, &f)));
    one.rebuild(); many.rebuild();
    CHECK(many.len() == one.len());
    CHECK(same(many.search(rows[107], 50), one.search(rows[107], 50)));
    one.insert(now_v7(), rows[3]); 
    bool threw = false;
    try { many.insert(now_v7(), Embedding(3)); } catch (const CortexError &) { threw = true; }
    CHECK(threw);
}

TEST(test_bf16_store_runs_the_reference_tests) {   // the reference's own cases (:484-535, :538-566, :667-684) on a bf16 row store
    HipIndex index(3, 0, CX_DTYPE_BF16);
    auto id1 = now_v7(), id2 = now_v7(), id3 = now_v7();
    index.insert(id1, {1.0f, 0.0f, 0.0f});
    index.insert(id2, {0.9f, 0.1f, 0.0f});
    index.insert(id3, {0.0f, 1.0f, 0.0f});
    index.rebuild();
    auto results = index.search({1.0f, 0.0f, 0.0f}, 2);
    CHECK(results.size() == 2 && results[0].node_id == id1 && results[1].node_id == id2);
    CHECK(std::fabs(results[0].score - 1.0f) < 1e-6f);
    auto thr = index.search_threshold({1.0f, 0.0f, 0.0f}, 0.95f);
    CHECK(thr.size() == 2);   // 0.9/0.1 rounds to bf16 and still scores 0.9939 >= 0.95
    const std::string path = std::string(std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp") + "/cx_cpp_test_bf16.hnsw";
    index.save(path);
    HipIndex loaded = HipIndex::load(path, 0, CX_DTYPE_BF16);
    CHECK(loaded.len() == 3);
    auto again = loaded.search({1.0f, 0.0f, 0.0f}, 2);
    CHECK(again.size() == 2 && again[0].node_id == id1 && again[0].score == results[0].score && again[1].score == results[1].score);
    std::remove(path.c_str());
    // a wider store through the batched path: bf16 index against an f32 index fed the rounded vectors gives the same ids
    const size_t dim = 256, n = 600;
    std::mt19937 rng(11);
    std::normal_distribution<float> g(0.0f, 1.0f);
    auto round_bf16 = [](float x) { uint32_t u; std::memcpy(&u, &x, 4); u += 0x7FFFu + ((u >> 16) & 1u); u &= 0xFFFF0000u; float y; std::memcpy(&y, &u, 4); return y; };
    HipIndex b(dim, 0, CX_DTYPE_BF16), f(dim);
    std::vector<NodeId> ids(n);
    std::vector<Embedding> rows(n, Embedding(dim));
    for (size_t i = 0; i < n; i++) {
        ids[i] = now_v7();
        for (auto &x : rows[i]) x = g(rng);
        b.insert(ids[i], rows[i]);
        Embedding r = rows[i];
        for (auto &x : r) x = round_bf16(x);
        f.insert(ids[i], r);
    }
    std::vector<std::pair<NodeId, Embedding>> qs;
    for (size_t i = 0; i < 40; i++) qs.emplace_back(ids[i], rows[i]);
    auto rb = b.search_batch(qs, 5), rf = f.search_batch(qs, 5);
    CHECK(rb.size() == rf.size());
    for (auto &kv : rb) {
        auto &x = kv.second; auto &y = rf[kv.first];
        CHECK(x.size() == y.size());
        for (size_t i = 0; i < x.size() && i < y.size(); i++) CHECK(x[i].node_id == y[i].node_id && std::fabs(x[i].score - y[i].score) < 5e-5f);
    }
}

int main() {
    if (cx_device_count() <= 0) { std::fprintf(stderr, "no HIP device: %s\n", "this test needs a GPU"); return 2; }
    run_test_index_insert_and_search(); run_test_threshold_search(); run_test_index_persistence();
    run_test_dimension_mismatch_rejected(); run_test_empty_index_search(); run_test_brute_force_fallback();
    run_test_filter_by_kind(); run_test_filter_exclude(); run_test_remove_doesnt_crash_search(); run_test_search_batch();
    run_test_similarity_score_range(); run_test_threshold_returns_only_above(); run_test_config();
    run_test_linker_similarity_edges_and_dedup(); run_test_bulk_load_nodes(); run_test_search_decayed_reranks(); run_test_sharded_single_process();
    run_test_bf16_store_runs_the_reference_tests();
    std::printf("%d tests run, %d checks failed\n", g_run, g_failed);
    return g_failed ? 1 : 0;
}
