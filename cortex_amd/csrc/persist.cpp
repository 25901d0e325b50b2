// persist.cpp — VectorIndex::save / load (vector/index.rs:437-473) in the reference's own file
// format, so an index written by either side loads on the other:
//
//   bincode 1.3 (little endian, fixed-width integers, u64 lengths) of the tuple
//     ( HashMap<Uuid, Vec<f32>>, HashMap<Uuid, NodeMetadata>, usize )
//   HashMap  = u64 count, then count x (key, value), any order
//   Uuid     = serialize_bytes: u64 16, 16 raw bytes        (pinned by the reference's golden Node
//                                                            bytes, storage/redb_storage.rs:1827-1857)
//   Vec<f32> = u64 len, len x f32 LE
//   NodeMetadata { kind: NodeKind(String), source_agent: String } = two strings (u64 len + UTF-8)
//   usize    = u64
//
// Entries are written in row order; load keeps file order as insertion order (the reference's order
// is a HashMap's and therefore arbitrary).  load() then calls rebuild() in the reference (:469) — a
// no-op for the exact engine.
#include <cerrno>
#include <cstdio>

#include "internal.hpp"

namespace {

using namespace cx;

struct Reader {
    FILE *f;
    bool ok = true;
    const char *why = "";
    void bytes(void *p, size_t n) { if (ok && n && fread(p, 1, n, f) != n) { ok = false; why = "unexpected end of file"; } }
    uint64_t u64() { uint64_t v = 0; bytes(&v, 8); return v; }
    bool uuid(uint8_t *id) {
        if (u64() != 16 && ok) { ok = false; why = "invalid uuid length"; }
        bytes(id, 16);
        return ok;
    }
    bool str(std::string &s, uint64_t limit = 1u << 20) {
        const uint64_t n = u64();
        if (ok && n > limit) { ok = false; why = "string too long"; }
        if (!ok) return false;
        s.resize((size_t)n);
        bytes(s.data(), (size_t)n);
        return ok;
    }
};

}  // namespace

namespace cx {

int save_index_file(const char *path, uint32_t dim, uint64_t n_alive, const std::function<int(IndexFileWriter &)> &write_vectors,
                    const std::vector<IndexFileMeta> &metas) {
    FILE *f = fopen(path, "wb");
    if (!f) return set_err(CX_ERR_IO, "Failed to write index file: %s", strerror(errno));
    IndexFileWriter w{f};
    w.u64(n_alive);
    if (int rc = write_vectors(w)) { fclose(f); return rc; }
    w.u64(metas.size());
    for (const IndexFileMeta &m : metas) { w.uuid(m.id); w.str(m.kind); w.str(m.agent); }
    w.u64(dim);
    const bool ok = w.ok;
    if (fclose(f) != 0 || !ok) return set_err(CX_ERR_IO, "Failed to write index file: %s", strerror(errno));
    return CX_OK;
}

int load_index_file(const char *path, const IndexFileSink &sink) {
    FILE *f = fopen(path, "rb");
    if (!f) return set_err(CX_ERR_IO, "Failed to read index file: %s", strerror(errno));
    Reader r{f};
    auto fail = [&](const char *msg) -> int {
        fclose(f);
        return set_err(CX_ERR_VALIDATION, "Failed to deserialize index: %s", msg);
    };
    // the dimension is the LAST field of the tuple; every vector carries its own length, so the first
    // one tells us the row width and the trailer is checked against it
    const uint64_t n_vec = r.u64();
    if (!r.ok) return fail(r.why);
    bool created = false;
    uint64_t dim = 0;
    const uint64_t batch_rows = 4096;
    std::vector<uint8_t> ids;
    std::vector<float> rows;
    auto flush = [&]() -> int {
        if (ids.empty()) return CX_OK;
        const int rc = sink.upsert(ids.size() / 16, ids.data(), rows.data(), dim);
        ids.clear();
        rows.clear();
        return rc;
    };
    for (uint64_t i = 0; i < n_vec; i++) {
        uint8_t id[16];
        if (!r.uuid(id)) return fail(r.why);
        const uint64_t len = r.u64();
        if (!r.ok) return fail(r.why);
        if (!created) {
            if (len > 0xFFFFFFFFull) return fail("vector too long");
            dim = len;
            if (int rc = sink.create(dim, n_vec)) { fclose(f); return rc; }
            created = true;
        } else if (len != dim) {
            return fail("vectors of different lengths");
        }
        const size_t at = rows.size();
        rows.resize(at + (size_t)dim);
        r.bytes(rows.data() + at, (size_t)dim * 4);
        if (!r.ok) return fail(r.why);
        ids.insert(ids.end(), id, id + 16);
        if (ids.size() / 16 >= batch_rows)
            if (int rc = flush()) { fclose(f); return rc; }
    }
    if (created)
        if (int rc = flush()) { fclose(f); return rc; }
    const uint64_t n_meta = r.u64();
    if (!r.ok) return fail(r.why);
    std::vector<IndexFileMeta> metas;
    for (uint64_t i = 0; i < n_meta; i++) {
        IndexFileMeta m;
        if (!r.uuid(m.id) || !r.str(m.kind) || !r.str(m.agent)) return fail(r.why);
        metas.push_back(std::move(m));
    }
    const uint64_t dimension = r.u64();
    if (!r.ok) return fail(r.why);
    if (fgetc(f) != EOF) return fail("trailing bytes");
    fclose(f);
    if (!created) {  // no vectors: the trailer is the only source of the dimension
        if (dimension > 0xFFFFFFFFull) return set_err(CX_ERR_VALIDATION, "Failed to deserialize index: dimension out of range");
        if (int rc = sink.create(dimension, 0)) return rc;
    } else if (dimension != dim) {
        return set_err(CX_ERR_VALIDATION, "Failed to deserialize index: dimension %llu does not match the stored vectors (%llu)",
                       (unsigned long long)dimension, (unsigned long long)dim);
    }
    for (auto &m : metas) {
        const uint32_t kc = sink.intern(m.kind.data(), m.kind.size());
        const uint32_t ac = sink.intern(m.agent.data(), m.agent.size());
        if (int rc = sink.set_meta(m.id, kc, ac)) return rc;
    }
    return CX_OK;
}

// rows [r0, r0 + m) of a shard's store as f32 on the host (a bf16 store is written as the f32 values it holds)
int read_rows_host(const cx_index *ix, uint64_t r0, uint64_t m, float *dst, std::vector<uint16_t> &tmp16) {
    const uint32_t dim = ix->dim;
    if (!dim || !m) return CX_OK;
    if (ix->dtype == 1) {
        tmp16.resize((size_t)m * dim);
        CX_HIP(hipMemcpy(tmp16.data(), ix->rows16() + (size_t)r0 * dim, (size_t)m * dim * 2, hipMemcpyDeviceToHost));
        for (size_t t = 0; t < (size_t)m * dim; t++) dst[t] = bf16_bits_to_f32(tmp16[t]);
    } else {
        CX_HIP(hipMemcpy(dst, ix->d_rows + (size_t)r0 * dim, (size_t)m * dim * 4, hipMemcpyDeviceToHost));
    }
    return CX_OK;
}

// (id, kind, agent) of every live row of one index that has metadata, plus its ids that only have metadata so far
void collect_metas(const cx_index *ix, std::vector<IndexFileMeta> &out) {
    std::lock_guard<std::mutex> intern_guard(ix->intern_mu);
    std::vector<const std::string *> names(ix->interned.size() + 1, nullptr);
    for (auto &kv : ix->interned) names[kv.second] = &kv.first;
    static const std::string empty;
    auto name = [&](uint32_t c) -> const std::string & { return c < names.size() && names[c] ? *names[c] : empty; };
    for (uint64_t r = 0; r < ix->n_rows; r++) {
        const uint32_t m = ix->h_meta[r];
        if ((m & META_REMOVED) || !(m & META_HAS)) continue;
        IndexFileMeta e;
        memcpy(e.id, &ix->ids[16 * (size_t)r], 16);
        e.kind = name(m >> 8);
        e.agent = name(ix->h_agent[r]);
        out.push_back(std::move(e));
    }
    for (auto &kv : ix->pending_meta) {   // metadata of ids without a vector is part of the map too (:438)
        IndexFileMeta e;
        memcpy(e.id, &kv.first.a, 8);
        memcpy(e.id + 8, &kv.first.b, 8);
        e.kind = name(kv.second.first);
        e.agent = name(kv.second.second);
        out.push_back(std::move(e));
    }
}

}  // namespace cx

extern "C" {

int cx_save(const cx_index *ix, const char *path) try {
    if (!ix || !path) return set_err(CX_ERR_VALIDATION, "null argument");
    if (int rc = use_device(ix)) return rc;
    const uint32_t dim = ix->dim;
    std::vector<IndexFileMeta> metas;
    collect_metas(ix, metas);
    return save_index_file(path, dim, ix->n_alive, [&](IndexFileWriter &w) -> int {
        // vectors: stream the row store back in 32 MiB slabs, entries in row order
        const uint64_t slab_rows = std::max<uint64_t>(1, (32ull << 20) / std::max<uint64_t>(1, (uint64_t)dim * 4));
        std::vector<float> host((size_t)slab_rows * std::max(dim, 1u));
        std::vector<uint16_t> tmp16;
        for (uint64_t r0 = 0; r0 < ix->n_rows; r0 += slab_rows) {
            const uint64_t m = std::min(slab_rows, ix->n_rows - r0);
            if (read_rows_host(ix, r0, m, host.data(), tmp16) != CX_OK)
                return set_err(CX_ERR_DEVICE, "Failed to write index file: device read failed");
            for (uint64_t i = 0; i < m; i++) {
                const uint64_t r = r0 + i;
                if (ix->h_meta[r] & META_REMOVED) continue;
                w.uuid(&ix->ids[16 * (size_t)r]);
                w.u64(dim);
                w.bytes(host.data() + (size_t)i * dim, (size_t)dim * 4);
            }
        }
        return CX_OK;
    }, metas);
} catch (...) { return cx::on_exception(); }

cx_index *cx_load(const char *path, int device) { return cx_load_ex(path, device, CX_DTYPE_F32); }

cx_index *cx_load_ex(const char *path, int device, int dtype) try {
    if (!path) {
        set_err(CX_ERR_VALIDATION, "null path");
        return nullptr;
    }
    cx_index *ix = nullptr;
    IndexFileSink sink;
    sink.create = [&](uint64_t dim, uint64_t n_vec) -> int {
        ix = cx_create_ex((uint32_t)dim, device, dtype);
        if (!ix) return CX_ERR_DEVICE;
        return n_vec ? cx_reserve(ix, n_vec) : CX_OK;
    };
    sink.upsert = [&](uint64_t n, const uint8_t *ids, const float *rows, uint64_t dim) { return cx_upsert_batch(ix, n, ids, rows, dim); };
    sink.intern = [&](const char *s, uint64_t n) { return cx_intern(ix, s, n); };
    sink.set_meta = [&](const uint8_t *id, uint32_t kc, uint32_t ac) { return cx_set_metadata(ix, id, kc, ac); };
    if (const int rc = load_index_file(path, sink)) {
        if (ix) {
            const std::string msg = err_buf();   // cx_destroy may touch the message
            cx_destroy(ix);
            set_err(rc, "%s", msg.c_str());
        }
        return nullptr;
    }
    return ix;
} catch (...) { cx::on_exception(); return nullptr; }

}  // extern "C"
