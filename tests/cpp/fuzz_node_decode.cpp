// Mutation fuzzer for the stored-node decoder (cortex_amd/csrc/nodes.cpp: cx_node_decode), built with
// -fsanitize=address,undefined on the CPU (tests/test_node_records.py::test_decoder_survives_mutations).
// The decoder reads database bytes: whatever they hold it must return a status, never read out of bounds.
// The rest of the library is stubbed out; only the decoder is exercised.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../cortex_amd/csrc/nodes.cpp"

namespace cx {
static thread_local char g_err[512];
int set_err(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return code; }
int on_exception() noexcept { return CX_ERR_DEVICE; }
int use_device(const cx_index *) { return CX_ERR_DEVICE; }
}  // namespace cx
extern "C" {
int cx_upsert_batch(cx_index *, uint64_t, const uint8_t *, const float *, uint64_t) { return CX_ERR_DEVICE; }
uint32_t cx_intern(cx_index *, const char *, uint64_t) { return 0; }
int cx_set_metadata_batch(cx_index *, uint64_t, const uint8_t *, const uint32_t *, const uint32_t *) { return CX_ERR_DEVICE; }
int cx_set_node_stats_batch(cx_index *, uint64_t, const uint8_t *, const uint32_t *, const int64_t *, const uint32_t *, const uint64_t *) { return CX_ERR_DEVICE; }
}

static void put_u64(std::vector<uint8_t> &o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }
static void put_str(std::vector<uint8_t> &o, const char *s) { put_u64(o, strlen(s)); o.insert(o.end(), s, s + strlen(s)); }

int main(int argc, char **argv) {
    const long iters = argc > 1 ? atol(argv[1]) : 200000;
    std::vector<uint8_t> base;
    put_u64(base, 16); for (int i = 0; i < 16; i++) base.push_back((uint8_t)i);
    put_str(base, "fact"); put_str(base, "t\xc3\xa9 \xe2\x98\x83"); put_str(base, "body");
    put_u64(base, 0); put_u64(base, 2); put_str(base, "a"); put_str(base, "bb");
    base.push_back(1); put_u64(base, 5); for (int i = 0; i < 20; i++) base.push_back((uint8_t)(i * 7));
    put_str(base, "kai"); base.push_back(1); put_str(base, "sess"); base.push_back(0);
    for (int i = 0; i < 4; i++) base.push_back(0x3f); put_u64(base, 3);
    put_str(base, "1970-01-01T00:00:00Z"); put_str(base, "2024-02-29T23:59:59.123456789Z"); put_str(base, "2024-03-01T00:00:00.5+01:30");
    base.push_back(0);
    cx_node_view v;
    if (cx_node_decode(base.data(), base.size(), &v) != 0) { fprintf(stderr, "base record does not decode: %s\n", cx::g_err); return 2; }
    std::mt19937_64 rng(12345);
    long ok = 0, bad = 0;
    for (long it = 0; it < iters; it++) {
        // exact-size heap copy so that any read past the end is an ASan report
        size_t len = base.size();
        const int kind = (int)(rng() % 6);
        if (kind == 0) len = rng() % (base.size() + 1);                       // truncation
        uint8_t *rec = (uint8_t *)malloc(len ? len : 1);
        memcpy(rec, base.data(), len);
        const int flips = kind == 0 ? 0 : 1 + (int)(rng() % 4);
        for (int f = 0; f < flips && len; f++) {
            const size_t at = rng() % len;
            switch (rng() % 4) {
                case 0: rec[at] ^= (uint8_t)(1u << (rng() % 8)); break;       // bit flip
                case 1: rec[at] = (uint8_t)rng(); break;                      // random byte
                case 2: rec[at] = 0xff; break;                                // big length bytes / invalid utf-8
                default: if (at + 8 <= len) { uint64_t big = rng() >> (rng() % 64); memcpy(rec + at, &big, 8); } break;   // random u64
            }
        }
        const int rc = cx_node_decode(rec, len, &v);
        if (rc == 0) {
            ok++;
            // every pointer the view hands out lies inside the record
            const uint8_t *lo = rec, *hi = rec + len;
            auto in = [&](const void *p, uint64_t n) { return n == 0 || ((const uint8_t *)p >= lo && (const uint8_t *)p + n <= hi); };
            if (!in(v.kind, v.kind_len) || !in(v.title, v.title_len) || !in(v.body, v.body_len) || !in(v.agent, v.agent_len) ||
                (v.has_embedding && !in(v.embedding, v.embedding_len * 4)) || v.bytes_used > len) { fprintf(stderr, "view outside the record at iteration %ld\n", it); return 1; }
            volatile uint8_t sink = 0;
            for (uint64_t i = 0; v.has_embedding && i < v.embedding_len * 4; i++) sink ^= v.embedding[i];
        } else bad++;
        free(rec);
    }
    printf("%ld mutated records: %ld decoded, %ld rejected\n", iters, ok, bad);
    return 0;
}
