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

struct Writer {
    FILE *f;
    bool ok = true;
    void bytes(const void *p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
    void u64(uint64_t v) { bytes(&v, 8); }  // host is little endian (x86-64)
    void str(const std::string &s) { u64(s.size()); bytes(s.data(), s.size()); }
    void uuid(const uint8_t *id) { u64(16); bytes(id, 16); }
};

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

extern "C" {

int cx_save(const cx_index *ix, const char *path) try {
    if (!ix || !path) return set_err(CX_ERR_VALIDATION, "null argument");
    if (int rc = use_device(ix)) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) return set_err(CX_ERR_IO, "Failed to write index file: %s", strerror(errno));
    Writer w{f};
    const uint32_t dim = ix->dim;
    w.u64(ix->n_alive);
    // vectors: stream the row store back in 32 MiB slabs
    const uint64_t slab_rows = std::max<uint64_t>(1, (32ull << 20) / std::max<uint64_t>(1, (uint64_t)dim * 4));
    std::vector<float> host((size_t)slab_rows * std::max(dim, 1u));
    std::vector<uint16_t> host16(ix->dtype == 1 ? host.size() : 0);   // a bf16 store is written as the f32 values it holds
    for (uint64_t r0 = 0; r0 < ix->n_rows; r0 += slab_rows) {
        const uint64_t m = std::min(slab_rows, ix->n_rows - r0);
        if (dim) {
            hipError_t e;
            if (ix->dtype == 1) {
                e = hipMemcpy(host16.data(), ix->rows16() + (size_t)r0 * dim, (size_t)m * dim * 2, hipMemcpyDeviceToHost);
                for (size_t t = 0; t < (size_t)m * dim; t++) host[t] = bf16_bits_to_f32(host16[t]);
            } else
                e = hipMemcpy(host.data(), ix->d_rows + (size_t)r0 * dim, (size_t)m * dim * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                fclose(f);
                return set_err(CX_ERR_DEVICE, "Failed to write index file: device read failed: %s", hipGetErrorString(e));
            }
        }
        for (uint64_t i = 0; i < m; i++) {
            const uint64_t r = r0 + i;
            if (ix->h_meta[r] & META_REMOVED) continue;
            w.uuid(&ix->ids[16 * (size_t)r]);
            w.u64(dim);
            w.bytes(host.data() + (size_t)i * dim, (size_t)dim * 4);
        }
    }
    // metadata
    std::lock_guard<std::mutex> intern_guard(ix->intern_mu);
    std::vector<const std::string *> names(ix->interned.size() + 1, nullptr);
    for (auto &kv : ix->interned) names[kv.second] = &kv.first;
    uint64_t n_meta = ix->pending_meta.size();   // metadata of ids without a vector is part of the map too (:438)
    for (uint64_t r = 0; r < ix->n_rows; r++)
        if (!(ix->h_meta[r] & META_REMOVED) && (ix->h_meta[r] & META_HAS)) n_meta++;
    w.u64(n_meta);
    static const std::string empty;
    for (uint64_t r = 0; r < ix->n_rows; r++) {
        const uint32_t m = ix->h_meta[r];
        if ((m & META_REMOVED) || !(m & META_HAS)) continue;
        const uint32_t kc = m >> 8, ac = ix->h_agent[r];
        w.uuid(&ix->ids[16 * (size_t)r]);
        w.str(kc < names.size() && names[kc] ? *names[kc] : empty);
        w.str(ac < names.size() && names[ac] ? *names[ac] : empty);
    }
    for (auto &kv : ix->pending_meta) {
        uint8_t id[16];
        memcpy(id, &kv.first.a, 8);
        memcpy(id + 8, &kv.first.b, 8);
        const uint32_t kc = kv.second.first, ac = kv.second.second;
        w.uuid(id);
        w.str(kc < names.size() && names[kc] ? *names[kc] : empty);
        w.str(ac < names.size() && names[ac] ? *names[ac] : empty);
    }
    w.u64(dim);
    const bool ok = w.ok;
    if (fclose(f) != 0 || !ok) return set_err(CX_ERR_IO, "Failed to write index file: %s", strerror(errno));
    return CX_OK;
} catch (...) { return cx::on_exception(); }

cx_index *cx_load(const char *path, int device) { return cx_load_ex(path, device, CX_DTYPE_F32); }

cx_index *cx_load_ex(const char *path, int device, int dtype) try {
    if (!path) {
        set_err(CX_ERR_VALIDATION, "null path");
        return nullptr;
    }
    FILE *f = fopen(path, "rb");
    if (!f) {
        set_err(CX_ERR_IO, "Failed to read index file: %s", strerror(errno));
        return nullptr;
    }
    Reader r{f};
    auto fail = [&](cx_index *ix, const char *msg) -> cx_index * {
        set_err(CX_ERR_VALIDATION, "Failed to deserialize index: %s", msg);
        fclose(f);
        if (ix) cx_destroy(ix);
        return nullptr;
    };
    // the dimension is the LAST field of the tuple; every vector carries its own length, so the first
    // one tells us the row width and the trailer is checked against it
    const uint64_t n_vec = r.u64();
    if (!r.ok) return fail(nullptr, r.why);
    cx_index *ix = nullptr;
    uint64_t dim = 0;
    const uint64_t batch_rows = 4096;
    std::vector<uint8_t> ids;
    std::vector<float> rows;
    auto flush = [&]() -> bool {
        if (ids.empty()) return true;
        const int rc = cx_upsert_batch(ix, ids.size() / 16, ids.data(), rows.data(), dim);
        ids.clear();
        rows.clear();
        return rc == CX_OK;
    };
    for (uint64_t i = 0; i < n_vec; i++) {
        uint8_t id[16];
        if (!r.uuid(id)) return fail(ix, r.why);
        const uint64_t len = r.u64();
        if (!r.ok) return fail(ix, r.why);
        if (!ix) {
            if (len > 0xFFFFFFFFull) return fail(nullptr, "vector too long");
            dim = len;
            ix = cx_create_ex((uint32_t)dim, device, dtype);
            if (!ix) { fclose(f); return nullptr; }
            if (cx_reserve(ix, n_vec) != CX_OK) { fclose(f); cx_destroy(ix); return nullptr; }
        } else if (len != dim) {
            return fail(ix, "vectors of different lengths");
        }
        const size_t at = rows.size();
        rows.resize(at + (size_t)dim);
        r.bytes(rows.data() + at, (size_t)dim * 4);
        if (!r.ok) return fail(ix, r.why);
        ids.insert(ids.end(), id, id + 16);
        if (ids.size() / 16 >= batch_rows && !flush()) { fclose(f); cx_destroy(ix); return nullptr; }
    }
    if (ix && !flush()) { fclose(f); cx_destroy(ix); return nullptr; }
    const uint64_t n_meta = r.u64();
    if (!r.ok) return fail(ix, r.why);
    struct Meta { uint8_t id[16]; std::string kind, agent; };
    std::vector<Meta> metas;
    for (uint64_t i = 0; i < n_meta; i++) {
        Meta m;
        if (!r.uuid(m.id) || !r.str(m.kind) || !r.str(m.agent)) return fail(ix, r.why);
        metas.push_back(std::move(m));
    }
    const uint64_t dimension = r.u64();
    if (!r.ok) return fail(ix, r.why);
    if (fgetc(f) != EOF) return fail(ix, "trailing bytes");
    fclose(f);
    if (!ix) {  // no vectors: the trailer is the only source of the dimension
        if (dimension > 0xFFFFFFFFull) { set_err(CX_ERR_VALIDATION, "Failed to deserialize index: dimension out of range"); return nullptr; }
        ix = cx_create_ex((uint32_t)dimension, device, dtype);
        if (!ix) return nullptr;
    } else if (dimension != dim) {
        cx_destroy(ix);
        set_err(CX_ERR_VALIDATION, "Failed to deserialize index: dimension %llu does not match the stored vectors (%llu)",
                (unsigned long long)dimension, (unsigned long long)dim);
        return nullptr;
    }
    for (auto &m : metas) {
        const uint32_t kc = cx_intern(ix, m.kind.data(), m.kind.size());
        const uint32_t ac = cx_intern(ix, m.agent.data(), m.agent.size());
        if (cx_set_metadata(ix, m.id, kc, ac) != CX_OK) { cx_destroy(ix); return nullptr; }
    }
    return ix;
} catch (...) { cx::on_exception(); return nullptr; }

}  // extern "C"
