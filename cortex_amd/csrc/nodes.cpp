// nodes.cpp — bulk load of the index from stored `Node` records: what the reference does at
// start-up with `for node in storage.list_nodes(NodeFilter::new()) { if let Some(emb) = &node.embedding
// { index.insert(node.id, emb) } }` (cortex-server/src/serve.rs:105-123, cortex-core/src/api.rs:56-70), here
// one call over the raw table values, so the embeddings go from the records to HBM without a `Node` per row.
//
// Record = bincode 1.3 (`bincode::deserialize`: little endian, fixed-width integers, u64 lengths, trailing
// bytes allowed) of `Node` (types.rs:26-68); the layout is pinned by the reference's golden bytes
// (storage/redb_storage.rs:1827-1857, tests/golden/node_schema_golden.json):
//
//   id                Uuid            u64 16, 16 raw bytes
//   kind              NodeKind(String) u64 len, UTF-8          (derive(Deserialize): not re-validated)
//   data.title, .body String
//   data.metadata     HashMap<String, serde_json::Value>  u64 count, entries.  bincode cannot DEserialize a
//                     `Value` (deserialize_any), so a record with count > 0 is unreadable for the reference
//                     too: list_nodes skips it (redb_storage.rs:709-712) — and so does this decoder.
//   data.tags         Vec<String>     u64 count, strings
//   embedding         Option<Vec<f32>> tag 0 | 1, u64 len, len x f32 LE
//   source            agent String, session Option<String>, channel Option<String>
//   importance f32, access_count u64
//   last_accessed_at, created_at, updated_at   DateTime<Utc> = RFC 3339 string (chrono serde)
//   deleted           bool (0 | 1; anything else is an error)
//
// What list_nodes does around the decode is kept: undecodable records are skipped (:709-712), deleted nodes
// are filtered out (node_matches_filter :345-349 with the default filter), and the result is ordered newest
// first by `created_at` with a stable sort (:727-728) — that order is the insertion order, i.e. the tie order
// of every later search.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "internal.hpp"

namespace {

using namespace cx;

// str::from_utf8: well-formed UTF-8 only (no overlongs, no surrogates, <= U+10FFFF)
bool utf8_ok(const uint8_t *p, uint64_t n) {
    uint64_t i = 0;
    while (i < n) {
        const uint8_t c = p[i];
        if (c < 0x80) { i++; continue; }
        uint32_t need;
        uint8_t lo = 0x80, hi = 0xBF;
        if (c >= 0xC2 && c <= 0xDF) need = 1;
        else if (c == 0xE0) { need = 2; lo = 0xA0; }
        else if ((c >= 0xE1 && c <= 0xEC) || c == 0xEE || c == 0xEF) need = 2;
        else if (c == 0xED) { need = 2; hi = 0x9F; }
        else if (c == 0xF0) { need = 3; lo = 0x90; }
        else if (c >= 0xF1 && c <= 0xF3) need = 3;
        else if (c == 0xF4) { need = 3; hi = 0x8F; }
        else return false;
        if (i + need >= n) return false;  // truncated sequence
        if (p[i + 1] < lo || p[i + 1] > hi) return false;
        for (uint32_t t = 2; t <= need; t++)
            if ((p[i + t] & 0xC0) != 0x80) return false;
        i += need + 1;
    }
    return true;
}

struct Cur {
    const uint8_t *p;
    uint64_t n, o = 0;
    const char *why = nullptr;
    bool fail(const char *w) { if (!why) why = w; return false; }
    bool take(uint64_t k, const uint8_t **out) {
        if (why) return false;
        if (k > n - o) return fail("unexpected end of record");
        *out = p + o;
        o += k;
        return true;
    }
    bool u8(uint8_t *v) { const uint8_t *q; if (!take(1, &q)) return false; *v = *q; return true; }
    bool u64(uint64_t *v) { const uint8_t *q; if (!take(8, &q)) return false; memcpy(v, q, 8); return true; }
    bool f32(float *v) { const uint8_t *q; if (!take(4, &q)) return false; memcpy(v, q, 4); return true; }
    bool str(const char **s, uint64_t *len) {
        uint64_t k;
        const uint8_t *q;
        if (!u64(&k) || !take(k, &q)) return false;
        if (!utf8_ok(q, k)) return fail("invalid utf-8 in string");
        *s = (const char *)q;
        *len = k;
        return true;
    }
    bool tag(uint8_t *t) {
        if (!u8(t)) return false;
        if (*t > 1) return fail("invalid Option tag");
        return true;
    }
    bool opt_str() {
        uint8_t t;
        const char *s;
        uint64_t l;
        if (!tag(&t)) return false;
        return t ? str(&s, &l) : true;
    }
};

bool digits(const char *s, uint32_t n, uint32_t *v) {
    uint32_t x = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        x = x * 10 + (uint32_t)(s[i] - '0');
    }
    *v = x;
    return true;
}

// days since 1970-01-01 of a proleptic Gregorian date
int64_t days_from_civil(int64_t y, uint32_t m, uint32_t d) {
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const uint32_t yoe = (uint32_t)(y - era * 400);
    const uint32_t doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const uint32_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + (int64_t)doe - 719468;
}

// chrono writes DateTime<Utc> as RFC 3339 ("2023-11-14T22:13:20Z", fraction of 3/6/9 digits when non-zero) and
// reads any RFC 3339 offset back.  Four-digit years only (chrono signs and widens the others; never produced
// by Utc::now()).
bool parse_rfc3339(const char *s, uint64_t n, int64_t *secs, uint32_t *nanos) {
    if (n < 20) return false;
    uint32_t Y, M, D, h, m, sec;
    if (!digits(s, 4, &Y) || s[4] != '-' || !digits(s + 5, 2, &M) || s[7] != '-' || !digits(s + 8, 2, &D)) return false;
    if (s[10] != 'T' && s[10] != 't' && s[10] != ' ') return false;
    if (!digits(s + 11, 2, &h) || s[13] != ':' || !digits(s + 14, 2, &m) || s[16] != ':' || !digits(s + 17, 2, &sec)) return false;
    static const uint32_t mdays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    if (M < 1 || M > 12 || D < 1) return false;
    const bool leap = (Y % 4 == 0 && Y % 100 != 0) || Y % 400 == 0;
    if (D > mdays[M - 1] + (M == 2 && leap ? 1u : 0u)) return false;
    if (h > 23 || m > 59 || sec > 60) return false;
    uint64_t o = 19;
    uint64_t frac = 0;
    if (s[o] == '.') {
        o++;
        uint32_t nd = 0;
        while (o < n && s[o] >= '0' && s[o] <= '9') {
            if (nd < 9) { frac = frac * 10 + (uint64_t)(s[o] - '0'); nd++; }
            o++;
        }
        if (!nd) return false;
        for (; nd < 9; nd++) frac *= 10;
    }
    if (o >= n) return false;
    int64_t off = 0;
    if (s[o] == 'Z' || s[o] == 'z') {
        o++;
    } else if (s[o] == '+' || s[o] == '-') {
        uint32_t oh, om;
        if (o + 6 > n || !digits(s + o + 1, 2, &oh) || s[o + 3] != ':' || !digits(s + o + 4, 2, &om) || oh > 23 || om > 59) return false;
        off = ((int64_t)oh * 3600 + (int64_t)om * 60) * (s[o] == '-' ? -1 : 1);
        o += 6;
    } else {
        return false;
    }
    if (o != n) return false;
    uint32_t ns = (uint32_t)frac;
    if (sec == 60) { sec = 59; ns += 1000000000u; }  // chrono's leap-second representation
    *secs = days_from_civil((int64_t)Y, M, D) * 86400 + (int64_t)h * 3600 + (int64_t)m * 60 + (int64_t)sec - off;
    *nanos = ns;
    return true;
}

bool decode(const uint8_t *rec, uint64_t len, cx_node_view *v, const char **why) {
    Cur c{rec, len};
    memset(v, 0, sizeof(*v));
    uint64_t k = 0;
    const uint8_t *q = nullptr;
    if (c.u64(&k) && k != 16) c.fail("invalid uuid length");
    if (c.take(16, &q)) memcpy(v->id, q, 16);
    c.str(&v->kind, &v->kind_len);
    c.str(&v->title, &v->title_len);
    c.str(&v->body, &v->body_len);
    if (c.u64(&k) && k != 0) c.fail("node metadata is not decodable (bincode cannot deserialize serde_json::Value)");
    if (c.u64(&v->n_tags)) {
        if (v->n_tags > len / 8) c.fail("unexpected end of record");
        for (uint64_t t = 0; t < v->n_tags && !c.why; t++) {
            const char *s;
            uint64_t l;
            c.str(&s, &l);
        }
    }
    uint8_t tag = 0;
    if (c.tag(&tag) && tag) {
        if (c.u64(&v->embedding_len)) {
            if (v->embedding_len > (len - c.o) / 4) c.fail("unexpected end of record");
            else if (c.take(v->embedding_len * 4, &q)) { v->embedding = q; v->has_embedding = 1; }
        }
    }
    c.str(&v->agent, &v->agent_len);
    c.opt_str();
    c.opt_str();
    c.f32(&v->importance);
    c.u64(&v->access_count);
    const char *ts[3] = {nullptr, nullptr, nullptr};
    uint64_t tl[3] = {0, 0, 0};
    for (int t = 0; t < 3; t++) c.str(&ts[t], &tl[t]);
    uint8_t del = 0;
    if (c.u8(&del) && del > 1) c.fail("invalid bool encoding");
    if (!c.why) {
        if (!parse_rfc3339(ts[0], tl[0], &v->last_accessed_at_s, &v->last_accessed_at_ns) ||
            !parse_rfc3339(ts[1], tl[1], &v->created_at_s, &v->created_at_ns) ||
            !parse_rfc3339(ts[2], tl[2], &v->updated_at_s, &v->updated_at_ns))
            c.fail("invalid RFC 3339 timestamp");
    }
    if (c.why) { *why = c.why; return false; }
    v->deleted = del;
    v->bytes_used = c.o;
    return true;
}

}  // namespace

extern "C" {

int cx_node_decode(const uint8_t *record, uint64_t len, cx_node_view *out) try {
    if (!record || !out) return set_err(CX_ERR_VALIDATION, "null argument");
    const char *why = "";
    if (!decode(record, len, out, &why)) return set_err(CX_ERR_VALIDATION, "Failed to deserialize node: %s", why);
    return CX_OK;
} catch (...) { return cx::on_exception(); }

}  // extern "C"

namespace cx {
// The loader behind cx_bulk_load_nodes and cx_sharded_bulk_load_nodes: decode, filter and order like list_nodes, then hand
// the embeddings (pinned staging, 256 MB chunks) and the per-node side data to the sink.
int bulk_load_impl(const BulkSink &ix, uint64_t n, const uint8_t *blob, const uint64_t *offsets, uint32_t flags,
                   cx_bulk_stats *stats) {
    cx_bulk_stats st{};
    if (stats) *stats = st;
    if (n && (!blob || !offsets)) return set_err(CX_ERR_VALIDATION, "null records");
    for (uint64_t i = 0; i < n; i++)
        if (offsets[i + 1] < offsets[i]) return set_err(CX_ERR_VALIDATION, "record offsets must not decrease (record %llu)", (unsigned long long)i);
    st.records = n;
    const bool diag = getenv("CX_BULK_DIAG") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_0 = now();

    // 1. decode every record (host threads; the embedding bytes stay where they are)
    struct Slot { int64_t s; uint32_t ns; uint8_t state; };  // state: 0 undecodable, 1 usable, 2 deleted, 3 no embedding, 4 wrong dimension
    std::vector<cx_node_view> views((size_t)n);
    std::vector<Slot> slots((size_t)n);
    const unsigned hw = std::thread::hardware_concurrency();
    const uint64_t n_thr = std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)(hw ? hw : 1), 16, (n + 4095) / 4096}));
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            const char *why;
            cx_node_view &v = views[(size_t)i];
            Slot &s = slots[(size_t)i];
            if (!decode(blob + offsets[i], offsets[i + 1] - offsets[i], &v, &why)) { s = {0, 0, 0}; continue; }
            uint8_t state = 1;
            if (v.deleted && !(flags & CX_BULK_INCLUDE_DELETED)) state = 2;
            else if (!v.has_embedding) state = 3;
            else if (v.embedding_len != ix.dim) state = 4;
            s = {v.created_at_s, v.created_at_ns, state};
        }
    };
    if (n_thr <= 1) {
        work(0, n);
    } else {
        std::vector<std::thread> pool;
        const uint64_t per = (n + n_thr - 1) / n_thr;
        for (uint64_t t = 0; t < n_thr; t++) pool.emplace_back(work, std::min(n, t * per), std::min(n, (t + 1) * per));
        for (auto &t : pool) t.join();
    }

    const auto t_1 = now();
    // 2. list_nodes order: created_at descending, stable (redb_storage.rs:727-728)
    std::vector<uint64_t> order;
    order.reserve((size_t)n);
    for (uint64_t i = 0; i < n; i++) {
        switch (slots[(size_t)i].state) {
            case 0: st.undecodable++; break;
            case 2: st.deleted++; break;
            default: order.push_back(i);
        }
    }
    if (!(flags & CX_BULK_KEEP_ORDER))
        std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
            const Slot &x = slots[(size_t)a], &y = slots[(size_t)b];
            return x.s != y.s ? x.s > y.s : x.ns > y.ns;
        });

    const auto t_2 = now();
    // 3. the insert loop (serve.rs:111-117 skips a failed insert; api.rs:61-62 returns the error)
    std::vector<uint64_t> take;
    take.reserve(order.size());
    for (uint64_t i : order) {
        const uint8_t state = slots[(size_t)i].state;
        if (state == 3) { st.no_embedding++; continue; }
        if (state == 4) {
            st.dim_mismatch++;
            if (flags & CX_BULK_STRICT) {
                if (stats) *stats = st;
                return set_err(CX_ERR_VALIDATION, "Embedding dimension mismatch: expected %u, got %llu", ix.dim,
                               (unsigned long long)views[(size_t)i].embedding_len);
            }
            continue;
        }
        take.push_back(i);
    }
    const uint64_t dim = ix.dim;
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(take.size(), (256ull << 20) / std::max<uint64_t>(1, dim * 4)));
    // staging in pinned host memory (the H2D copy runs at link rate), filled by the same host threads
    if (take.empty()) { if (stats) *stats = st; return CX_OK; }
    CX_HIP(hipSetDevice(ix.device));
    float *stage = nullptr;
    CX_HIP(hipHostMalloc((void **)&stage, (size_t)(chunk * dim * 4 + 16), hipHostMallocDefault));
    struct Unpin { float *p; ~Unpin() { (void)hipHostFree(p); } } unpin{stage};
    std::vector<uint8_t> ids((size_t)chunk * 16);
    std::vector<uint32_t> kinds, agents;
    for (uint64_t lo = 0; lo < take.size(); lo += chunk) {
        const uint64_t m = std::min<uint64_t>(chunk, take.size() - lo);
        auto fill = [&](uint64_t a, uint64_t b) {
            for (uint64_t j = a; j < b; j++) {
                const cx_node_view &v = views[(size_t)take[(size_t)(lo + j)]];
                memcpy(ids.data() + 16 * j, v.id, 16);
                memcpy(stage + j * dim, v.embedding, (size_t)dim * 4);  // records are byte streams: unaligned source
            }
        };
        const uint64_t f_thr = std::max<uint64_t>(1, std::min<uint64_t>(n_thr, (m + 1023) / 1024));
        if (f_thr <= 1) {
            fill(0, m);
        } else {
            std::vector<std::thread> pool;
            const uint64_t per = (m + f_thr - 1) / f_thr;
            for (uint64_t t = 0; t < f_thr; t++) pool.emplace_back(fill, std::min(m, t * per), std::min(m, (t + 1) * per));
            for (auto &t : pool) t.join();
        }
        if (int rc = ix.upsert(m, ids.data(), stage)) { if (stats) *stats = st; return rc; }
        st.indexed += m;
        if (flags & CX_BULK_SET_STATS) {
            std::vector<uint32_t> kc((size_t)m), lns((size_t)m);
            std::vector<int64_t> ls((size_t)m);
            std::vector<uint64_t> ac((size_t)m);
            for (uint64_t j = 0; j < m; j++) {
                const cx_node_view &v = views[(size_t)take[(size_t)(lo + j)]];
                kc[(size_t)j] = ix.intern(v.kind, v.kind_len);
                ls[(size_t)j] = v.last_accessed_at_s; lns[(size_t)j] = v.last_accessed_at_ns; ac[(size_t)j] = v.access_count;
            }
            if (int rc = ix.set_stats(m, ids.data(), kc.data(), ls.data(), lns.data(), ac.data())) { if (stats) *stats = st; return rc; }
        }
        if (flags & CX_BULK_SET_METADATA) {
            kinds.resize((size_t)m);
            agents.resize((size_t)m);
            for (uint64_t j = 0; j < m; j++) {
                const cx_node_view &v = views[(size_t)take[(size_t)(lo + j)]];
                kinds[(size_t)j] = ix.intern(v.kind, v.kind_len);
                agents[(size_t)j] = ix.intern(v.agent, v.agent_len);
            }
            if (int rc = ix.set_meta(m, ids.data(), kinds.data(), agents.data())) { if (stats) *stats = st; return rc; }
        }
    }
    if (diag) fprintf(stderr, "[bulk] decode %.1f ms, order %.1f ms, stage+insert %.1f ms\n", ms(t_0, t_1), ms(t_1, t_2), ms(t_2, now()));
    if (stats) *stats = st;
    return CX_OK;
}
}  // namespace cx

extern "C" {

int cx_bulk_load_nodes(cx_index *ix, uint64_t n, const uint8_t *blob, const uint64_t *offsets, uint32_t flags,
                       cx_bulk_stats *stats) try {
    if (stats) *stats = cx_bulk_stats{};
    if (!ix) return set_err(CX_ERR_VALIDATION, "null index");
    cx::BulkSink sink;
    sink.dim = ix->dim;
    sink.device = ix->device;
    sink.upsert = [ix](uint64_t m, const uint8_t *ids, const float *embs) { return cx_upsert_batch(ix, m, ids, embs, ix->dim); };
    sink.intern = [ix](const char *p, uint64_t len) { return cx_intern(ix, p, len); };
    sink.set_stats = [ix](uint64_t m, const uint8_t *ids, const uint32_t *kc, const int64_t *ls, const uint32_t *lns, const uint64_t *ac) {
        return cx_set_node_stats_batch(ix, m, ids, kc, ls, lns, ac);
    };
    sink.set_meta = [ix](uint64_t m, const uint8_t *ids, const uint32_t *k, const uint32_t *a) { return cx_set_metadata_batch(ix, m, ids, k, a); };
    return cx::bulk_load_impl(sink, n, blob, offsets, flags, stats);
} catch (...) { return cx::on_exception(); }

}  // extern "C"
