// sharded.cpp — one index over several GPUs of the node behind the C ABI (include/cortex_hip.h, "cx_sharded").
//
// The reference's host is ONE process holding ONE index (serve.rs:101, api.rs:41); a drop-in that wants the 8 GPUs of
// a node therefore has to shard under the boundary, not above it.  A cx_sharded owns one cx_index per shard (one per
// listed device), places new ids block-round-robin, and remembers for every row its GLOBAL insertion sequence number
// ("global row").  Everything the caller sees — result order, tie order, row-indexed linker interfaces — is expressed
// in global rows, so a sharded index is indistinguishable from a single index that saw the same calls.
//
// Search data path (k <= 256):   shard scans run concurrently, one stream per shard
//   shard p:  search_core -> local lists -> publish_part_kernel: local rows -> global rows, written straight into
//             part p of the gather buffer ON THE ROOT DEVICE (peer-to-peer stores over xGMI; KBs, single hop)
//   root:     waits for the P events, merge_parts_kernel<SEQ> (ties by global row), result block written into
//             pinned host memory, one host wait.
// No ring, no collective library: the exchange is P small posted writes (the payload is latency-bound).
// The multi-PROCESS variant (one rank per GPU, RCCL all-gather) is cortex_amd/sharded.py, used by bench.py.
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <thread>

#include "internal.hpp"

namespace {

using namespace cx;

constexpr uint64_t PLACEMENT_BLOCK = 4096;   // consecutive new ids per shard
constexpr uint32_t LINK_BLOCK = 2048;        // scanned nodes per block of the all-pairs pass

inline size_t part_words(uint64_t nq, uint64_t k) { return (size_t)((3 * nq * k + nq + 3) / 4 * 4); }

// Pooled per-call scratch on the root device.
struct RootCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t *d_gather = nullptr; size_t c_gather = 0;   // [P][part_words]  (root HBM: peers write into it)
    uint32_t *h_gather = nullptr; size_t c_hgather = 0;  // the same in pinned host memory when a shard has no peer access
    uint32_t *h_out = nullptr; size_t c_hout = 0;        // merged block, pinned: counts | rows | scores | dists
    std::vector<hipEvent_t> ev;                           // one per shard, created on the shard's device
    // all-pairs pass (root side)
    uint32_t *d_lists = nullptr; size_t c_lists = 0;     // merged lists: rows[nq*k] | scores | dists | counts
    uint32_t *d_scan = nullptr; size_t c_scan = 0;
    uint8_t *d_deleted = nullptr; size_t c_deleted = 0;
    uint64_t *d_exist_off = nullptr; size_t c_exist_off = 0;
    uint32_t *d_exist_to = nullptr; size_t c_exist_to = 0;
    uint32_t *d_counts = nullptr; size_t c_counts = 0;
    uint64_t *d_offsets = nullptr; size_t c_offsets = 0;
    char *d_temp = nullptr; size_t c_temp = 0;
    uint32_t *d_from = nullptr, *d_to = nullptr; float *d_w = nullptr; size_t c_from = 0, c_to = 0, c_w = 0;
    ~RootCtx() {
        (void)hipSetDevice(device);
        (void)hipFree(d_gather); (void)hipHostFree(h_gather); (void)hipHostFree(h_out); (void)hipFree(d_lists); (void)hipFree(d_scan);
        (void)hipFree(d_deleted); (void)hipFree(d_exist_off); (void)hipFree(d_exist_to); (void)hipFree(d_counts);
        (void)hipFree(d_offsets); (void)hipFree(d_temp); (void)hipFree(d_from); (void)hipFree(d_to); (void)hipFree(d_w);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// Per-shard scratch of one all-pairs call (lives for the call).
struct LinkShard {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t scattered = nullptr, done = nullptr;
    float *d_q = nullptr;                                  // [LINK_BLOCK][dim] the block's vectors on this device
    uint32_t *d_rows = nullptr, *d_cnt = nullptr, *d_src = nullptr, *d_pos = nullptr;
    float *d_scores = nullptr, *d_dists = nullptr;
    size_t c_q = 0, c_rows = 0, c_cnt = 0, c_src = 0, c_pos = 0, c_scores = 0, c_dists = 0;
    ~LinkShard() {
        (void)hipSetDevice(device);
        (void)hipFree(d_q); (void)hipFree(d_rows); (void)hipFree(d_cnt); (void)hipFree(d_src); (void)hipFree(d_pos);
        (void)hipFree(d_scores); (void)hipFree(d_dists);
        if (scattered) (void)hipEventDestroy(scattered);
        if (done) (void)hipEventDestroy(done);
        if (stream) (void)hipStreamDestroy(stream);
    }
};
// The per-shard streams and scratch of one linker pass.  They live in the handle's pool, not in the call: every shard
// index keeps one Ctx (with its device scratch) per stream handle it has been called on until cx_destroy, so a fresh
// stream per pass — once per linker cycle in a long-running server — leaked HBM without bound (round-2 ADVICE).
struct LinkSet {
    std::vector<std::unique_ptr<LinkShard>> ls;
};

// One enqueue thread per shard, alive as long as the handle: a search's per-shard work (query upload, filter, kernel
// launches, publish — ~10 us of host time each) is queued on all shards at once instead of one shard after the other
// from the caller's thread (8 shards: 80 us in front of a 0.5 ms scan).  Callers (`&self`, concurrent) push one job per
// shard and wait for their own jobs; a worker runs the jobs of its shard in arrival order.
class ShardWorkers {
public:
    explicit ShardWorkers(size_t n) : q_(n) {
        for (size_t s = 0; s < n; s++) th_.emplace_back([this, s] { loop(s); });
    }
    ~ShardWorkers() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    // runs job(s) for every s in [0, n) on shard s's thread; returns when all are done
    void run_all(size_t n, const std::function<void(size_t)> &job) {
        struct Latch { std::mutex m; std::condition_variable c; size_t left; } latch;
        latch.left = n;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (size_t s = 0; s < n; s++)
                q_[s].push_back([&job, &latch, s] {
                    job(s);
                    std::lock_guard<std::mutex> g2(latch.m);
                    if (--latch.left == 0) latch.c.notify_one();
                });
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> l(latch.m);
        latch.c.wait(l, [&] { return latch.left == 0; });
    }
private:
    void loop(size_t s) {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [&] { return stop_ || !q_[s].empty(); });
                if (q_[s].empty()) return;
                f = std::move(q_[s].front());
                q_[s].pop_front();
            }
            f();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::deque<std::function<void()>>> q_;
    std::vector<std::thread> th_;
    bool stop_ = false;
};

}  // namespace

struct cx_sharded {
    uint32_t dim = 0;
    std::vector<cx_index *> shards;
    std::vector<int> devices;
    uint64_t block = PLACEMENT_BLOCK;   // consecutive new ids per shard (CX_SHARD_PLACEMENT_BLOCK overrides: tests interleave finer)
    int root = 0;                 // device of shard 0: gather + merge happen there
    bool p2p = true;              // every shard device can store into the root device's memory
    // global rows: sequence number -> id, location, liveness
    std::vector<uint8_t> seq_ids;
    std::vector<uint32_t> seq_shard, seq_row;
    std::vector<uint8_t> seq_alive;
    uint64_t n_alive = 0, n_fresh = 0;
    std::unordered_map<IdKey, uint32_t, IdHash> map;                          // id -> global row (live ids)
    std::unordered_map<IdKey, std::pair<uint32_t, uint32_t>, IdHash> pending_meta;   // as cx_index::pending_meta
    std::vector<NodeStats> h_stats;              // per global row: what apply_score_decay reads (decay.cpp); empty until set
    std::vector<std::vector<uint32_t>> h_gseq;   // [shard][local row] -> global row
    std::vector<uint32_t *> d_gseq;              // the same on the shard's device
    std::vector<size_t> c_gseq, n_gseq_up;       // capacity / rows uploaded
    mutable std::mutex mu;
    mutable std::vector<RootCtx *> pool;
    mutable std::vector<LinkSet *> link_pool;   // streams + scratch of the linker passes (as many sets as passes ever ran at once)
    mutable std::unique_ptr<ShardWorkers> workers;   // per-shard enqueue threads (created with the handle when it has more than one shard)
};

namespace {

struct RootLease {
    const cx_sharded *h;
    RootCtx *c = nullptr;
    explicit RootLease(const cx_sharded *hh) : h(hh) {
        {
            std::lock_guard<std::mutex> g(h->mu);
            if (!h->pool.empty()) { c = h->pool.back(); h->pool.pop_back(); }
        }
        if (c) return;
        std::unique_ptr<RootCtx> n(new RootCtx());
        n->device = h->root;
        if (hipSetDevice(h->root) != hipSuccess || hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking) != hipSuccess) {
            set_err(CX_ERR_DEVICE, "sharded: cannot create the root stream");
            return;
        }
        n->ev.assign(h->shards.size(), nullptr);
        for (size_t s = 0; s < h->shards.size(); s++) {
            if (hipSetDevice(h->devices[s]) != hipSuccess ||
                hipEventCreateWithFlags(&n->ev[s], hipEventDisableTiming) != hipSuccess) {
                set_err(CX_ERR_DEVICE, "sharded: cannot create an event on device %d", h->devices[s]);
                return;
            }
        }
        c = n.release();
    }
    ~RootLease() {
        if (!c) return;
        std::lock_guard<std::mutex> g(h->mu);
        h->pool.push_back(c);
    }
    RootLease(const RootLease &) = delete;
    RootLease &operator=(const RootLease &) = delete;
};

int upload_gseq(cx_sharded *h, size_t s) {
    const std::vector<uint32_t> &g = h->h_gseq[s];
    if (h->n_gseq_up[s] >= g.size()) return CX_OK;
    CX_HIP(hipSetDevice(h->devices[s]));
    if (h->c_gseq[s] < g.size()) {
        const size_t cap = std::max<size_t>(g.size(), std::max<size_t>(h->c_gseq[s] * 2, 1024));
        uint32_t *n = nullptr;
        CX_HIP(hipMalloc((void **)&n, cap * 4));
        if (h->d_gseq[s]) CX_HIP(hipFree(h->d_gseq[s]));
        h->d_gseq[s] = n;
        h->c_gseq[s] = cap;
        h->n_gseq_up[s] = 0;
    }
    const size_t lo = h->n_gseq_up[s];
    CX_HIP(hipMemcpy(h->d_gseq[s] + lo, g.data() + lo, (g.size() - lo) * 4, hipMemcpyHostToDevice));
    h->n_gseq_up[s] = g.size();
    return CX_OK;
}

// vector/index.rs:298-314 over the shards.  The input is cut into runs that go to one shard with one call: a run of
// fresh ids inside one placement block (their embeddings are contiguous in the caller's buffer: no staging copy), or a
// single known id (replaced in place where it lives).
int upsert_impl(cx_sharded *h, uint64_t n, const uint8_t *ids, const float *embs, uint64_t len, bool on_device) {
    if (!h) return set_err(CX_ERR_VALIDATION, "null index");
    if (len != h->dim)
        return set_err(CX_ERR_VALIDATION, "Embedding dimension mismatch: expected %u, got %llu", h->dim, (unsigned long long)len);
    if (!n) return CX_OK;
    if (!ids || !embs) return set_err(CX_ERR_VALIDATION, "null ids/embeddings");
    if (h->seq_shard.size() + n >= 0xFFFFFFF0ull) return set_err(CX_ERR_VALIDATION, "a sharded index holds at most 2^32-16 global rows");
    const size_t P = h->shards.size();
    int rc = CX_OK;
    uint64_t i = 0;
    while (i < n && rc == CX_OK) {
        const IdKey key = id_key(ids + 16 * i);
        auto it = h->map.find(key);
        if (it != h->map.end()) {
            const uint32_t s = h->seq_shard[it->second];
            rc = on_device ? cx_upsert_batch_dev(h->shards[s], 1, ids + 16 * i, embs + i * len, len)
                           : cx_upsert_batch(h->shards[s], 1, ids + 16 * i, embs + i * len, len);
            i++;
            continue;
        }
        const size_t s = (size_t)((h->n_fresh / h->block) % P);
        const uint64_t room = h->block - h->n_fresh % h->block;
        const uint64_t run_start = i;
        const uint64_t first_row = cx_row_count(h->shards[s]);
        // The run of new ids that goes to shard s: found WITHOUT touching the handle, committed only for the rows the shard
        // then holds.  (Round-2 ADVICE: bookkeeping written before a shard call that failed — a hipMalloc in grow_rows is
        // enough — left ids live in the handle but absent from the shard, and the retry appended rows d_gseq did not cover.)
        std::unordered_map<IdKey, uint64_t, IdHash> in_run;
        while (i < n && i - run_start < room) {
            const IdKey kk = id_key(ids + 16 * i);
            if (h->map.find(kk) != h->map.end() || in_run.count(kk)) break;    // a known id, or a repeat inside this run
            in_run.emplace(kk, i);
            if (!h->pending_meta.empty()) {     // metadata that arrived before the vector: the shard keeps it pending until the row exists
                auto pm = h->pending_meta.find(kk);
                if (pm != h->pending_meta.end()) (void)cx_set_metadata(h->shards[s], ids + 16 * i, pm->second.first, pm->second.second);
            }
            i++;
        }
        const uint64_t m = i - run_start;
#ifdef CX_TEST_HOOKS   // fault injection, compiled into libcortex_hip_testhooks.so only (csrc/Makefile): the N-th shard append fails
        static const long fail_at = getenv("CX_SHARD_FAIL_UPSERT") ? atol(getenv("CX_SHARD_FAIL_UPSERT")) : -1;
        static std::atomic<long> appends{0};
        if (fail_at >= 0 && appends.fetch_add(1) == fail_at) rc = set_err(CX_ERR_DEVICE, "injected failure of a shard append (CX_SHARD_FAIL_UPSERT)");
        else
#endif
            rc = on_device ? cx_upsert_batch_dev(h->shards[s], m, ids + 16 * run_start, embs + run_start * len, len)
                           : cx_upsert_batch(h->shards[s], m, ids + 16 * run_start, embs + run_start * len, len);
        const uint64_t now = cx_row_count(h->shards[s]);
        const uint64_t held = now >= first_row ? std::min<uint64_t>(now - first_row, m) : 0;   // all of them, or what a failed call got as far as
        for (uint64_t t = 0; t < held; t++) {
            const uint64_t src = run_start + t;
            const IdKey kk = id_key(ids + 16 * src);
            const uint32_t seq = (uint32_t)h->seq_shard.size();
            h->map.emplace(kk, seq);
            h->seq_ids.insert(h->seq_ids.end(), ids + 16 * src, ids + 16 * src + 16);
            h->seq_shard.push_back((uint32_t)s);
            h->seq_row.push_back((uint32_t)(first_row + t));
            h->seq_alive.push_back(1);
            h->h_gseq[s].push_back(seq);
            h->n_alive++;
            h->n_fresh++;
            h->pending_meta.erase(kk);
        }
        if (rc == CX_OK && now != first_row + m)
            rc = set_err(CX_ERR_DEVICE, "sharded: shard %zu holds %llu rows after an append of %llu to %llu", s, (unsigned long long)now,
                         (unsigned long long)m, (unsigned long long)first_row);
    }
    // the device copies of the row maps cover every row a shard holds, also when the call fails half way
    for (size_t s = 0; s < P; s++) {
        const int rc2 = upload_gseq(h, s);
        if (rc == CX_OK) rc = rc2;
    }
    return rc;
}

struct Hit {
    float score, dist;
    uint32_t seq;
};
inline bool hit_better(const Hit &a, const Hit &b) {   // score descending, NaN last, then global row ascending
    const bool an = a.score != a.score, bn = b.score != b.score;
    if (an != bn) return bn;
    if (!an && a.score != b.score) return a.score > b.score;
    return a.seq < b.seq;
}

// Merged block in pinned host memory.
struct HostBlock {
    uint32_t *counts, *rows;
    float *scores, *dists;
};
int ensure_host_block(RootCtx *c, uint64_t nq, uint64_t k, HostBlock &b) {
    const size_t cpad = (size_t)((nq + 3) / 4 * 4), entries = (size_t)std::max<uint64_t>(nq * k, 1);
    if (int rc = ensure_pinned(c->h_out, c->c_hout, cpad + 3 * entries)) return rc;
    b.counts = c->h_out;
    b.rows = c->h_out + cpad;
    b.scores = reinterpret_cast<float *>(c->h_out + cpad + entries);
    b.dists = reinterpret_cast<float *>(c->h_out + cpad + 2 * entries);
    return CX_OK;
}

// nq queries against every shard, k <= TOPK_MAX: the device path described at the top of the file.
int search_device_path(const cx_sharded *h, RootCtx *root, uint64_t nq, const float *queries, uint64_t len, uint32_t k,
                       const cx_filter *filter, HostBlock &hb) {
    const size_t P = h->shards.size();
    const size_t words = part_words(nq, k);
    CX_HIP(hipSetDevice(h->root));
    if (int rc = ensure_dev(root->d_gather, root->c_gather, P * words)) return rc;
    if (!h->p2p)
        if (int rc = ensure_pinned(root->h_gather, root->c_hgather, P * words)) return rc;
    uint32_t *gather = h->p2p ? root->d_gather : root->h_gather;
    if (int rc = ensure_host_block(root, nq, k, hb)) return rc;
    std::vector<std::unique_ptr<CtxLease>> leases(P);
    // whatever a failed call left queued on the shard streams writes into the root's pooled gather buffer: it must have
    // finished before the leases (and the root's) go back to their pools (round-2 ADVICE)
    struct DrainOnError {
        std::vector<std::unique_ptr<CtxLease>> &l;
        bool armed = true;
        ~DrainOnError() {
            if (!armed) return;
            for (auto &x : l)
                if (x && x->c) { (void)hipSetDevice(x->ix->device); (void)hipStreamSynchronize(x->c->stream); }
        }
    } drain{leases};
    std::vector<int> rcs(P, CX_OK);
    std::vector<std::string> msgs(P);
    auto enqueue = [&](size_t s) {   // everything one shard contributes, queued on its own stream
        auto body = [&]() -> int {
            const cx_index *ix = h->shards[s];
            uint32_t *part = gather + s * words;
            if (ix->n_rows == 0) {   // an empty shard contributes empty lists
                if (!h->p2p) { memset(part + 3 * nq * k, 0, nq * 4); return CX_OK; }
                CX_HIP(hipSetDevice(h->root));
                CX_HIP(hipMemsetAsync(part + 3 * nq * k, 0, nq * 4, root->stream));
                return CX_OK;
            }
            if (int rc = use_device(ix)) return rc;
            leases[s].reset(new CtxLease(ix));
            Ctx *c = leases[s]->c;
            if (!c) return CX_ERR_DEVICE;
            const uint32_t k_eff = (uint32_t)std::min<uint64_t>(k, ix->n_rows);
            std::vector<float> tails;
            if (int rc = stage_queries(ix, c, nq, queries, len, tails)) return rc;
            FilterUpload fu;
            if (int rc = build_filter(ix, c, filter, c->stream, fu)) return rc;
            const size_t entries = (size_t)nq * k_eff, cpad = (size_t)((nq + 3) / 4 * 4);
            if (int rc = ensure_dev(c->d_out_rows, c->or_cap, cpad + 3 * entries)) return rc;
            uint32_t *l_counts = c->d_out_rows, *l_rows = c->d_out_rows + cpad;
            float *l_scores = reinterpret_cast<float *>(l_rows + entries), *l_dists = l_scores + entries;
            if (int rc = search_core(ix, c, c->d_query, tails.data(), nq, k_eff, fu.f, 0.0f, false, l_rows, l_scores, l_dists, l_counts,
                                     c->stream))
                return rc;
            if (int rc = launch_publish_part(l_rows, l_scores, l_dists, l_counts, h->d_gseq[s], (uint32_t)nq, k_eff, k,
                                             (uint32_t)ix->n_rows, part, c->stream))
                return rc;
            CX_HIP(hipEventRecord(root->ev[s], c->stream));
            return CX_OK;
        };
        try {
            rcs[s] = body();
            if (rcs[s] != CX_OK) msgs[s] = err_buf();   // the message is thread-local: carry it out
        } catch (...) { rcs[s] = on_exception(); msgs[s] = err_buf(); }
    };
    // empty shards touch the root stream: keep those on this thread; the others go to their shards' enqueue threads
    if (h->workers) {
        h->workers->run_all(P, [&](size_t s) { if (h->shards[s]->n_rows) enqueue(s); });
        for (size_t s = 0; s < P; s++)
            if (!h->shards[s]->n_rows) enqueue(s);
    } else {
        for (size_t s = 0; s < P; s++) enqueue(s);
    }
    for (size_t s = 0; s < P; s++)
        if (rcs[s] != CX_OK) return set_err(rcs[s], "%s", msgs[s].c_str());
    CX_HIP(hipSetDevice(h->root));
    for (size_t s = 0; s < P; s++)
        if (leases[s]) CX_HIP(hipStreamWaitEvent(root->stream, root->ev[s], 0));
    const size_t n = (size_t)nq * k;
    if (int rc = launch_merge_parts_seq((uint32_t)P, (uint32_t)nq, k, words, gather, reinterpret_cast<float *>(gather + n),
                                        reinterpret_cast<float *>(gather + 2 * n), gather + 3 * n, hb.rows, hb.scores, hb.dists,
                                        hb.counts, root->stream))
        return rc;
    CX_HIP(hipStreamSynchronize(root->stream));   // the shard streams are behind it (event waits): their leases may go back
    drain.armed = false;
    return check_result_block(hb.counts, hb.rows, nq, k, k, h->seq_shard.size());
}

// any k (and the threshold search): per-shard host API on one thread per shard, merged on the host
template <class F>
int fan_out_host(const cx_sharded *h, std::vector<std::vector<Hit>> &parts, F &&per_shard) {
    const size_t P = h->shards.size();
    parts.assign(P, {});
    std::vector<int> rcs(P, CX_OK);
    std::vector<std::string> msgs(P);
    std::vector<std::thread> pool;
    for (size_t s = 0; s < P; s++)
        pool.emplace_back([&, s] {
            try {
                rcs[s] = per_shard(s, parts[s]);
                if (rcs[s] != CX_OK) msgs[s] = err_buf();   // the message is thread-local: carry it out
            } catch (...) { rcs[s] = on_exception(); msgs[s] = err_buf(); }
        });
    for (auto &t : pool) t.join();
    for (size_t s = 0; s < P; s++)
        if (rcs[s] != CX_OK) return set_err(rcs[s], "%s", msgs[s].c_str());
    return CX_OK;
}

int hits_of(const cx_sharded *h, size_t s, const uint8_t *ids, const float *sc, const float *di, uint64_t n, std::vector<Hit> &out) {
    (void)s;
    for (uint64_t i = 0; i < n; i++) {
        auto it = h->map.find(id_key(ids + 16 * i));
        if (it == h->map.end()) return set_err(CX_ERR_DEVICE, "sharded: a shard returned an id the index does not know");
        out.push_back(Hit{sc[i], di[i], it->second});
    }
    return CX_OK;
}

}  // namespace

namespace {
// every neighbour of the query with score >= threshold over all shards, best first (score desc, global row asc): the
// shards' own threshold searches (count-then-fill, like any caller of cx_search_threshold) merged on the host
int threshold_hits(const cx_sharded *h, const float *query, uint64_t len, float threshold, const cx_filter *filter, std::vector<Hit> &all) {
    std::vector<std::vector<Hit>> parts;
    if (int rc = fan_out_host(h, parts, [&](size_t s, std::vector<Hit> &out) -> int {
            const cx_index *ix = h->shards[s];
            if (cx_len(ix) == 0) return CX_OK;
            uint64_t c = 256;
            for (;;) {
                std::vector<uint8_t> ids(16 * c);
                std::vector<float> sc(c), di(c);
                uint64_t n = 0, need = 0;
                const int rc = cx_search_threshold(ix, query, len, threshold, filter, c, ids.data(), sc.data(), di.data(), &n, &need);
                if (rc == CX_ERR_CAPACITY && need > c) { c = need; continue; }
                if (rc) return rc;
                return hits_of(h, s, ids.data(), sc.data(), di.data(), n, out);
            }
        }))
        return rc;
    all.clear();
    for (auto &p : parts) all.insert(all.end(), p.begin(), p.end());
    std::sort(all.begin(), all.end(), hit_better);
    return CX_OK;
}
}  // namespace

extern "C" {

cx_sharded *cx_sharded_create(uint32_t dimension, uint32_t n_shards, const int *device_ids) {
    return cx_sharded_create_ex(dimension, n_shards, device_ids, CX_DTYPE_F32);
}

cx_sharded *cx_sharded_create_ex(uint32_t dimension, uint32_t n_shards, const int *device_ids, int dtype) try {
    if (!n_shards || !device_ids) { set_err(CX_ERR_VALIDATION, "sharded: no devices"); return nullptr; }
    if (n_shards > MAX_PARTS) { set_err(CX_ERR_VALIDATION, "sharded: at most %u shards", MAX_PARTS); return nullptr; }
    std::unique_ptr<cx_sharded> h(new cx_sharded());
    h->dim = dimension;
    if (const char *e = getenv("CX_SHARD_PLACEMENT_BLOCK")) h->block = std::max<long long>(1, atoll(e));
    for (uint32_t s = 0; s < n_shards; s++) {
        cx_index *ix = cx_create_ex(dimension, device_ids[s], dtype);
        if (!ix) { for (cx_index *p : h->shards) cx_destroy(p); return nullptr; }
        h->shards.push_back(ix);
        h->devices.push_back(device_ids[s]);
    }
    h->root = device_ids[0];
    h->h_gseq.resize(n_shards);
    h->d_gseq.assign(n_shards, nullptr);
    h->c_gseq.assign(n_shards, 0);
    h->n_gseq_up.assign(n_shards, 0);
    // peer access in both directions between every pair of distinct devices: shards store into the root's gather buffer,
    // owners scatter scanned vectors into every shard's query block
    for (uint32_t a = 0; a < n_shards; a++)
        for (uint32_t b = 0; b < n_shards; b++) {
            const int da = device_ids[a], db = device_ids[b];
            if (da == db) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) { h->p2p = false; continue; }
            if (hipSetDevice(da) != hipSuccess) { h->p2p = false; continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) h->p2p = false;
            (void)hipGetLastError();
        }
    if (h->shards.size() > 1 && !(getenv("CX_SHARD_ENQUEUE_THREADS") && atoi(getenv("CX_SHARD_ENQUEUE_THREADS")) == 0))
        h->workers.reset(new ShardWorkers(h->shards.size()));
    return h.release();
} catch (...) { cx::on_exception(); return nullptr; }

/* VectorIndex::save / load for the sharded handle (vector/index.rs:437-473): ONE file in the reference's layout — the
 * file of a single index over the same calls, entries in global row order — so an index saved sharded loads on one GPU,
 * and the other way round. */
int cx_sharded_save(const cx_sharded *h, const char *path) try {
    if (!h || !path) return set_err(CX_ERR_VALIDATION, "null argument");
    const uint32_t dim = h->dim;
    std::vector<IndexFileMeta> metas;
    {   // metadata in global row order, then the ids that only have metadata so far (handle and shards)
        std::vector<std::vector<IndexFileMeta>> per(h->shards.size());
        std::vector<std::unordered_map<IdKey, size_t, IdHash>> at(h->shards.size());
        for (size_t s = 0; s < h->shards.size(); s++) {
            collect_metas(h->shards[s], per[s]);
            for (size_t i = 0; i < per[s].size(); i++) at[s].emplace(id_key(per[s][i].id), i);
        }
        std::vector<std::vector<char>> used(h->shards.size());
        for (size_t s = 0; s < per.size(); s++) used[s].assign(per[s].size(), 0);
        for (size_t q = 0; q < h->seq_shard.size(); q++) {
            if (!h->seq_alive[q]) continue;
            const uint32_t s = h->seq_shard[q];
            auto it = at[s].find(id_key(&h->seq_ids[16 * q]));
            if (it == at[s].end()) continue;
            metas.push_back(per[s][it->second]);
            used[s][it->second] = 1;
        }
        for (size_t s = 0; s < per.size(); s++)
            for (size_t i = 0; i < per[s].size(); i++)
                if (!used[s][i]) metas.push_back(per[s][i]);
        if (!h->pending_meta.empty()) {
            const cx_index *ix0 = h->shards[0];   // every shard interns every string in the same order: shard 0's table names the codes
            std::lock_guard<std::mutex> g(ix0->intern_mu);
            std::vector<const std::string *> names(ix0->interned.size() + 1, nullptr);
            for (auto &kv : ix0->interned) names[kv.second] = &kv.first;
            for (auto &kv : h->pending_meta) {
                IndexFileMeta e;
                memcpy(e.id, &kv.first.a, 8);
                memcpy(e.id + 8, &kv.first.b, 8);
                const uint32_t kc = kv.second.first, ac = kv.second.second;
                if (kc < names.size() && names[kc]) e.kind = *names[kc];
                if (ac < names.size() && names[ac]) e.agent = *names[ac];
                metas.push_back(std::move(e));
            }
        }
    }
    return save_index_file(path, dim, h->n_alive, [&](IndexFileWriter &w) -> int {
        // global rows are dealt to the shards in placement blocks: runs of consecutive global rows are runs of consecutive
        // local rows of one shard, read back with one copy each (32 MiB at most)
        const uint64_t slab_rows = std::max<uint64_t>(1, (32ull << 20) / std::max<uint64_t>(1, (uint64_t)dim * 4));
        std::vector<float> host((size_t)slab_rows * std::max(dim, 1u));
        std::vector<uint16_t> tmp16;
        const size_t n_seq = h->seq_shard.size();
        size_t q = 0;
        while (q < n_seq) {
            const uint32_t s = h->seq_shard[q], r0 = h->seq_row[q];
            size_t e = q + 1;
            while (e < n_seq && e - q < slab_rows && h->seq_shard[e] == s && h->seq_row[e] == r0 + (e - q)) e++;
            const cx_index *ix = h->shards[s];
            if (use_device(ix) != CX_OK || read_rows_host(ix, r0, e - q, host.data(), tmp16) != CX_OK)
                return set_err(CX_ERR_DEVICE, "Failed to write index file: device read failed");
            for (size_t t = q; t < e; t++) {
                if (!h->seq_alive[t]) continue;
                w.uuid(&h->seq_ids[16 * t]);
                w.u64(dim);
                w.bytes(host.data() + (t - q) * dim, (size_t)dim * 4);
            }
            q = e;
        }
        return CX_OK;
    }, metas);
} catch (...) { return cx::on_exception(); }

cx_sharded *cx_sharded_load_ex(const char *path, uint32_t n_shards, const int *device_ids, int dtype) try {
    if (!path) {
        set_err(CX_ERR_VALIDATION, "null path");
        return nullptr;
    }
    cx_sharded *h = nullptr;
    IndexFileSink sink;
    sink.create = [&](uint64_t dim, uint64_t) -> int {
        h = cx_sharded_create_ex((uint32_t)dim, n_shards, device_ids, dtype);
        return h ? CX_OK : CX_ERR_DEVICE;
    };
    sink.upsert = [&](uint64_t n, const uint8_t *ids, const float *rows, uint64_t dim) { return cx_sharded_upsert_batch(h, n, ids, rows, dim); };
    sink.intern = [&](const char *s, uint64_t n) { return cx_sharded_intern(h, s, n); };
    sink.set_meta = [&](const uint8_t *id, uint32_t kc, uint32_t ac) { return cx_sharded_set_metadata(h, id, kc, ac); };
    if (const int rc = load_index_file(path, sink)) {
        if (h) {
            const std::string msg = err_buf();
            cx_sharded_destroy(h);
            set_err(rc, "%s", msg.c_str());
        }
        return nullptr;
    }
    return h;
} catch (...) { cx::on_exception(); return nullptr; }

void cx_sharded_destroy(cx_sharded *h) {
    if (!h) return;
    h->workers.reset();
    for (RootCtx *c : h->pool) delete c;
    for (LinkSet *l : h->link_pool) delete l;
    for (size_t s = 0; s < h->shards.size(); s++) {
        (void)hipSetDevice(h->devices[s]);
        (void)hipFree(h->d_gseq[s]);
        cx_destroy(h->shards[s]);
    }
    delete h;
}

uint32_t cx_sharded_n_shards(const cx_sharded *h) { return h ? (uint32_t)h->shards.size() : 0; }
const cx_index *cx_sharded_shard(const cx_sharded *h, uint32_t i) { return (h && i < h->shards.size()) ? h->shards[i] : nullptr; }
int cx_sharded_peer_to_peer(const cx_sharded *h) { return (h && h->p2p) ? 1 : 0; }
uint64_t cx_sharded_len(const cx_sharded *h) { return h ? h->n_alive : 0; }
uint32_t cx_sharded_dimension(const cx_sharded *h) { return h ? h->dim : 0; }
uint64_t cx_sharded_row_count(const cx_sharded *h) { return h ? h->seq_shard.size() : 0; }

int cx_sharded_upsert(cx_sharded *h, const uint8_t id[16], const float *embedding, uint64_t len) try {
    return upsert_impl(h, 1, id, embedding, len, false);
} catch (...) { return cx::on_exception(); }
int cx_sharded_upsert_batch(cx_sharded *h, uint64_t n, const uint8_t *ids, const float *embeddings, uint64_t len) try {
    return upsert_impl(h, n, ids, embeddings, len, false);
} catch (...) { return cx::on_exception(); }
int cx_sharded_upsert_batch_dev(cx_sharded *h, uint64_t n, const uint8_t *ids, const float *d_embeddings, uint64_t len) try {
    return upsert_impl(h, n, ids, d_embeddings, len, true);
} catch (...) { return cx::on_exception(); }

int cx_sharded_remove(cx_sharded *h, const uint8_t id[16]) try {
    if (!h || !id) return set_err(CX_ERR_VALIDATION, "null argument");
    h->pending_meta.erase(id_key(id));
    auto it = h->map.find(id_key(id));
    if (it == h->map.end()) return CX_OK;
    const uint32_t seq = it->second;
    if (int rc = cx_remove(h->shards[h->seq_shard[seq]], id)) return rc;   // the shard first: a failure leaves the handle as it was
    h->map.erase(it);
    h->seq_alive[seq] = 0;
    h->n_alive--;
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_sharded_set_metadata(cx_sharded *h, const uint8_t id[16], uint32_t kind_code, uint32_t agent_code) try {
    if (!h || !id) return set_err(CX_ERR_VALIDATION, "null argument");
    if (kind_code >= (1u << 24)) return set_err(CX_ERR_VALIDATION, "kind code out of range");
    auto it = h->map.find(id_key(id));
    if (it == h->map.end()) {    // no vector yet: the shard is not known until the insert places the id
        h->pending_meta[id_key(id)] = {kind_code, agent_code};
        return CX_OK;
    }
    return cx_set_metadata(h->shards[h->seq_shard[it->second]], id, kind_code, agent_code);
} catch (...) { return cx::on_exception(); }

uint32_t cx_sharded_intern(cx_sharded *h, const char *utf8, uint64_t len) try {
    if (!h) return 0;
    uint32_t code = 0;
    for (size_t s = 0; s < h->shards.size(); s++) {   // every shard interns every string in the same order: same codes
        const uint32_t c = cx_intern(h->shards[s], utf8, len);
        if (s == 0) code = c;
        else if (c != code) { set_err(CX_ERR_DEVICE, "sharded: intern tables diverged"); return 0; }
    }
    return code;
} catch (...) { cx::on_exception(); return 0; }
uint32_t cx_sharded_lookup(const cx_sharded *h, const char *utf8, uint64_t len) {
    return h ? cx_lookup(h->shards[0], utf8, len) : 0;
}

int cx_sharded_row_id(const cx_sharded *h, uint64_t global_row, uint8_t out_id[16]) try {
    if (!h || !out_id) return set_err(CX_ERR_VALIDATION, "null argument");
    if (global_row >= h->seq_shard.size()) return set_err(CX_ERR_VALIDATION, "row %llu out of range", (unsigned long long)global_row);
    memcpy(out_id, &h->seq_ids[16 * (size_t)global_row], 16);
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_sharded_rows_of(const cx_sharded *h, uint64_t n, const uint8_t *ids, uint32_t *out_rows) try {
    if (!h || (n && (!ids || !out_rows))) return set_err(CX_ERR_VALIDATION, "null argument");
    for (uint64_t i = 0; i < n; i++) {
        auto it = h->map.find(id_key(ids + 16 * i));
        out_rows[i] = it == h->map.end() ? 0xFFFFFFFFu : it->second;
    }
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_sharded_rebuild(cx_sharded *h) try {
    if (!h) return set_err(CX_ERR_VALIDATION, "null index");
    if (h->n_alive == h->seq_shard.size()) return CX_OK;
    for (cx_index *ix : h->shards)
        if (int rc = cx_rebuild(ix)) return rc;
    // compaction keeps the order of the rows inside a shard, so the live global rows, renumbered in order, land on
    // consecutive local rows of their shard
    const size_t P = h->shards.size(), n_old = h->seq_shard.size();
    std::vector<uint8_t> ids;
    std::vector<uint32_t> shard, row;
    std::vector<std::vector<uint32_t>> gseq(P);
    ids.reserve(16 * h->n_alive);
    h->map.clear();
    for (size_t q = 0; q < n_old; q++) {
        if (!h->seq_alive[q]) continue;
        const uint32_t s = h->seq_shard[q], nq = (uint32_t)shard.size();
        ids.insert(ids.end(), &h->seq_ids[16 * q], &h->seq_ids[16 * q] + 16);
        shard.push_back(s);
        row.push_back((uint32_t)gseq[s].size());
        gseq[s].push_back(nq);
        h->map.emplace(id_key(&h->seq_ids[16 * q]), nq);
    }
    if (!h->h_stats.empty()) {   // the stats move with their rows
        std::vector<NodeStats> st;
        st.reserve(h->n_alive);
        for (size_t q = 0; q < n_old; q++)
            if (h->seq_alive[q]) st.push_back(q < h->h_stats.size() ? h->h_stats[q] : NodeStats{});
        h->h_stats.swap(st);
    }
    h->seq_ids.swap(ids);
    h->seq_shard.swap(shard);
    h->seq_row.swap(row);
    h->seq_alive.assign(h->seq_shard.size(), 1);
    h->h_gseq.swap(gseq);
    for (size_t s = 0; s < P; s++) {
        if (h->h_gseq[s].size() != cx_row_count(h->shards[s]))
            return set_err(CX_ERR_DEVICE, "sharded: shard %zu holds %llu rows after compaction, %zu expected", s,
                           (unsigned long long)cx_row_count(h->shards[s]), h->h_gseq[s].size());
        h->n_gseq_up[s] = 0;
        if (int rc = upload_gseq(h, s)) return rc;
    }
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_sharded_search_batch(const cx_sharded *h, uint64_t nq, const float *queries, uint64_t len, uint64_t k,
                            const cx_filter *filter, uint8_t *out_ids, float *out_scores, float *out_distances,
                            uint64_t *out_counts) try {
    if (!h) return set_err(CX_ERR_VALIDATION, "null index");
    if (nq && (!queries || !out_counts)) return set_err(CX_ERR_VALIDATION, "null queries/out_counts");
    for (uint64_t i = 0; i < nq; i++) out_counts[i] = 0;
    if (!nq || h->n_alive == 0 || k == 0) return CX_OK;   // vector/index.rs:331-333
    if (!out_ids || !out_scores || !out_distances) return set_err(CX_ERR_VALIDATION, "null output buffer");
    const uint64_t k_all = std::min<uint64_t>(k, h->seq_shard.size());
    if (k_all <= TOPK_MAX) {
        RootLease root(h);
        if (!root.c) return CX_ERR_DEVICE;
        HostBlock hb;
        if (int rc = search_device_path(h, root.c, nq, queries, len, (uint32_t)k_all, filter, hb)) return rc;
        for (uint64_t i = 0; i < nq; i++) {
            const uint32_t cnt = hb.counts[i];
            out_counts[i] = cnt;
            for (uint32_t j = 0; j < cnt; j++) {
                const size_t src = (size_t)i * k_all + j, dst = (size_t)i * k + j;
                memcpy(out_ids + 16 * dst, &h->seq_ids[16 * (size_t)hb.rows[src]], 16);
                out_scores[dst] = hb.scores[src];
                out_distances[dst] = hb.dists[src];
            }
        }
        return CX_OK;
    }
    // k > 256: every shard's own large-k path (dense keys + radix sort), merged on the host
    for (uint64_t qi = 0; qi < nq; qi++) {
        std::vector<std::vector<Hit>> parts;
        if (int rc = fan_out_host(h, parts, [&](size_t s, std::vector<Hit> &out) -> int {
                const cx_index *ix = h->shards[s];
                const uint64_t cap = std::min<uint64_t>(k, cx_row_count(ix));
                if (!cap) return CX_OK;
                std::vector<uint8_t> ids(16 * cap);
                std::vector<float> sc(cap), di(cap);
                uint64_t n = 0;
                if (int rc = cx_search(ix, queries + qi * len, len, k, filter, ids.data(), sc.data(), di.data(), &n)) return rc;
                return hits_of(h, s, ids.data(), sc.data(), di.data(), n, out);
            }))
            return rc;
        std::vector<Hit> all;
        for (auto &p : parts) all.insert(all.end(), p.begin(), p.end());
        std::sort(all.begin(), all.end(), hit_better);
        const uint64_t take = std::min<uint64_t>(all.size(), k);
        out_counts[qi] = take;
        for (uint64_t j = 0; j < take; j++) {
            const size_t dst = (size_t)qi * k + j;
            memcpy(out_ids + 16 * dst, &h->seq_ids[16 * (size_t)all[j].seq], 16);
            out_scores[dst] = all[j].score;
            out_distances[dst] = all[j].dist;
        }
    }
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_sharded_search(const cx_sharded *h, const float *query, uint64_t len, uint64_t k, const cx_filter *filter,
                      uint8_t *out_ids, float *out_scores, float *out_distances, uint64_t *n_out) try {
    if (!n_out) return set_err(CX_ERR_VALIDATION, "null n_out");
    return cx_sharded_search_batch(h, 1, query, len, k, filter, out_ids, out_scores, out_distances, n_out);
} catch (...) { return cx::on_exception(); }

int cx_sharded_search_threshold(const cx_sharded *h, const float *query, uint64_t len, float threshold,
                                const cx_filter *filter, uint64_t cap, uint8_t *out_ids, float *out_scores,
                                float *out_distances, uint64_t *n_out, uint64_t *n_needed) try {
    if (!h || !query || !n_out) return set_err(CX_ERR_VALIDATION, "null argument");
    *n_out = 0;
    if (n_needed) *n_needed = 0;
    if (h->n_alive == 0) return CX_OK;
    std::vector<Hit> all;
    if (int rc = threshold_hits(h, query, len, threshold, filter, all)) return rc;
    if (n_needed) *n_needed = all.size();
    const uint64_t take = std::min<uint64_t>(all.size(), cap);
    if (take && (!out_ids || !out_scores || !out_distances)) return set_err(CX_ERR_VALIDATION, "null output buffer");
    for (uint64_t j = 0; j < take; j++) {
        memcpy(out_ids + 16 * j, &h->seq_ids[16 * (size_t)all[j].seq], 16);
        out_scores[j] = all[j].score;
        out_distances[j] = all[j].dist;
    }
    *n_out = take;
    if (all.size() > cap)
        return set_err(CX_ERR_CAPACITY, "search_threshold: %llu results, buffer holds %llu", (unsigned long long)all.size(),
                       (unsigned long long)cap);
    return CX_OK;
} catch (...) { return cx::on_exception(); }

}  // extern "C"

// ------------------------------------------------------------------------------------------------ all-pairs passes

namespace {

// AutoLinker::run_cycle's kNN loop (auto_linker.rs:215-264) / DedupScanner::scan (dedup.rs:65-127) over the shards,
// in global rows.  scan: the scanned global rows in scan order (live rows only).
// lists_out != null: no rule walk — the merged ordered top-k lists of the scanned rows themselves (cx_sharded_topk_lists_rows:
// `search(&emb, topk, None)` per scanned node, auto_linker.rs:221) go to lists_out->{rows, scores, counts} at the scan positions
// in lists_out->pos; every shard then runs its batched search over the block instead of the thresholded filter pass.
struct ListsOut { uint32_t *rows; float *scores; uint32_t *counts; const uint32_t *pos; };
int link_pass_sharded(const cx_sharded *h, const std::vector<uint32_t> &scan, const std::vector<uint64_t> &ex_off,
                      const uint32_t *ex_to, uint32_t topk, float threshold, uint32_t max_edges, uint64_t max_cycle,
                      const uint8_t *deleted, bool dedup, std::vector<uint32_t> &o_from, std::vector<uint32_t> &o_to,
                      std::vector<float> &o_w, const ListsOut *lists_out = nullptr) {
    const size_t P = h->shards.size();
    const uint32_t dim = h->dim;
    const uint64_t n_seq = h->seq_shard.size();
    if (topk == 0 || topk > TOPK_MAX) return set_err(CX_ERR_VALIDATION, "autolink: topk must be in 1..%u", TOPK_MAX);
    if (scan.empty()) return CX_OK;
    if (!h->p2p) return set_err(CX_ERR_DEVICE, "sharded all-pairs pass needs peer access between the shard devices");
    RootLease rl(h);
    RootCtx *root = rl.c;
    if (!root) return CX_ERR_DEVICE;
    const uint32_t blk = (uint32_t)std::min<size_t>(LINK_BLOCK, scan.size());
    const size_t words = part_words(blk, topk);
    // per-shard streams and scratch: a set from the handle's pool (grow-only)
    struct LinkLease {
        const cx_sharded *h;
        LinkSet *set = nullptr;
        explicit LinkLease(const cx_sharded *hh) : h(hh) {
            std::lock_guard<std::mutex> g(h->mu);
            if (!h->link_pool.empty()) { set = h->link_pool.back(); h->link_pool.pop_back(); }
        }
        ~LinkLease() {
            if (!set) return;
            for (auto &l : set->ls)   // whatever an error path left queued must not outlive the lease
                if (l && l->stream) { (void)hipSetDevice(l->device); (void)hipStreamSynchronize(l->stream); }
            std::lock_guard<std::mutex> g(h->mu);
            h->link_pool.push_back(set);
        }
    } lease(h);
    if (!lease.set) {
        lease.set = new LinkSet();
        lease.set->ls.resize(P);
    }
    std::vector<std::unique_ptr<LinkShard>> &ls = lease.set->ls;
    for (size_t s = 0; s < P; s++) {
        if (!ls[s]) {
            ls[s].reset(new LinkShard());
            LinkShard &l = *ls[s];
            l.device = h->devices[s];
            CX_HIP(hipSetDevice(l.device));
            CX_HIP(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
            CX_HIP(hipEventCreateWithFlags(&l.scattered, hipEventDisableTiming));
            CX_HIP(hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
        }
        LinkShard &l = *ls[s];
        CX_HIP(hipSetDevice(l.device));
        if (int rc = ensure_dev(l.d_q, l.c_q, (size_t)blk * std::max(dim, 1u))) return rc;
        if (int rc = ensure_dev(l.d_rows, l.c_rows, (size_t)blk * topk)) return rc;
        if (int rc = ensure_dev(l.d_scores, l.c_scores, (size_t)blk * topk)) return rc;
        if (int rc = ensure_dev(l.d_dists, l.c_dists, (size_t)blk * topk)) return rc;
        if (int rc = ensure_dev(l.d_cnt, l.c_cnt, (size_t)blk)) return rc;
        if (int rc = ensure_dev(l.d_src, l.c_src, (size_t)blk)) return rc;
        if (int rc = ensure_dev(l.d_pos, l.c_pos, (size_t)blk)) return rc;
    }
    CX_HIP(hipSetDevice(h->root));
    if (int rc = ensure_dev(root->d_gather, root->c_gather, P * words)) return rc;
    if (int rc = ensure_dev(root->d_lists, root->c_lists, words)) return rc;
    if (int rc = ensure_dev(root->d_scan, root->c_scan, (size_t)blk)) return rc;
    if (int rc = ensure_dev(root->d_counts, root->c_counts, (size_t)blk)) return rc;
    if (int rc = ensure_dev(root->d_offsets, root->c_offsets, (size_t)blk)) return rc;
    const size_t tb = scan_temp_bytes(blk);
    if (int rc = ensure_dev(root->d_temp, root->c_temp, tb)) return rc;
    if (deleted) {
        if (int rc = ensure_dev(root->d_deleted, root->c_deleted, (size_t)n_seq)) return rc;
        CX_HIP(hipMemcpyAsync(root->d_deleted, deleted, n_seq, hipMemcpyHostToDevice, root->stream));
    }
    const bool has_ex = !ex_off.empty() && !dedup;
    if (has_ex)
        if (int rc = ensure_dev(root->d_exist_off, root->c_exist_off, (size_t)blk + 1)) return rc;

    uint64_t emitted = 0;
    std::vector<uint32_t> src, pos, ex_sorted;
    std::vector<uint64_t> off_blk;
    for (size_t lo = 0; lo < scan.size() && emitted < max_cycle; lo += blk) {
        const uint32_t m = (uint32_t)std::min<size_t>(blk, scan.size() - lo);
        // 1. owners scatter the block's vectors into every shard's query block (peer stores)
        for (size_t o = 0; o < P; o++) {
            src.clear(); pos.clear();
            for (uint32_t i = 0; i < m; i++) {
                const uint32_t q = scan[lo + i];
                if (h->seq_shard[q] == o) { src.push_back(h->seq_row[q]); pos.push_back(i); }
            }
            LinkShard &l = *ls[o];
            CX_HIP(hipSetDevice(l.device));
            if (!src.empty()) {
                CX_HIP(hipMemcpyAsync(l.d_src, src.data(), src.size() * 4, hipMemcpyHostToDevice, l.stream));
                CX_HIP(hipMemcpyAsync(l.d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, l.stream));
                CX_HIP(hipStreamSynchronize(l.stream));   // src / pos are reused for the next owner
                for (size_t t = 0; t < P; t++)
                    if (int rc = (h->shards[o]->dtype == 1 ? launch_scatter_rows(h->shards[o]->rows16(), ls[t]->d_q, l.d_src, l.d_pos, (uint32_t)src.size(), dim, l.stream) : launch_scatter_rows(h->shards[o]->d_rows, ls[t]->d_q, l.d_src, l.d_pos, (uint32_t)src.size(), dim, l.stream)))
                        return rc;
            }
            CX_HIP(hipEventRecord(l.scattered, l.stream));
        }
        // 2. every shard: the block's ordered neighbour lists against its own rows, published to the root
        std::vector<int> rcs(P, CX_OK);
        std::vector<std::string> msgs(P);
        std::vector<std::thread> pool;
        for (size_t t = 0; t < P; t++)
            pool.emplace_back([&, t] {
                try {
                    LinkShard &l = *ls[t];
                    const cx_index *ix = h->shards[t];
                    uint32_t *part = root->d_gather + t * words;
                    int rc = CX_OK;
                    if (hipSetDevice(l.device) != hipSuccess) rc = set_err(CX_ERR_DEVICE, "hipSetDevice failed");
                    for (size_t o = 0; o < P && rc == CX_OK; o++)
                        if (hipStreamWaitEvent(l.stream, ls[o]->scattered, 0) != hipSuccess) rc = set_err(CX_ERR_DEVICE, "hipStreamWaitEvent failed");
                    if (rc == CX_OK) {
                        if (ix->n_rows == 0) {
                            if (hipMemsetAsync(part + 3 * (size_t)m * topk, 0, (size_t)m * 4, l.stream) != hipSuccess) rc = set_err(CX_ERR_DEVICE, "memset failed");
                        } else {
                            uint32_t k_src = topk;
                            if (lists_out) {   // this shard's own top-k of the block's vectors (lists k_src wide)
                                k_src = (uint32_t)std::min<uint64_t>(topk, ix->n_rows);
                                rc = cx_search_batch_dev(ix, m, l.d_q, k_src, nullptr, l.d_rows, l.d_scores, l.d_dists, l.d_cnt, l.stream);
                            } else {
                                rc = cx_autolink_lists_dev(ix, m, l.d_q, topk, threshold, l.d_rows, l.d_scores, l.d_dists, l.d_cnt, l.stream);
                            }
                            if (rc == CX_OK)
                                rc = launch_publish_part(l.d_rows, l.d_scores, l.d_dists, l.d_cnt, h->d_gseq[t], m, k_src, topk,
                                                         (uint32_t)ix->n_rows, part, l.stream);
                        }
                    }
                    if (rc == CX_OK && hipEventRecord(l.done, l.stream) != hipSuccess) rc = set_err(CX_ERR_DEVICE, "hipEventRecord failed");
                    if (rc == CX_OK && hipStreamSynchronize(l.stream) != hipSuccess) rc = set_err(CX_ERR_DEVICE, "shard stream failed");
                    rcs[t] = rc;
                    if (rc != CX_OK) msgs[t] = err_buf();
                } catch (...) { rcs[t] = on_exception(); msgs[t] = err_buf(); }
            });
        for (auto &t : pool) t.join();
        for (size_t t = 0; t < P; t++)
            if (rcs[t] != CX_OK) return set_err(rcs[t], "%s", msgs[t].c_str());
        // 3. root: merge the parts (ties by global row), then the reference's walk over the merged lists
        CX_HIP(hipSetDevice(h->root));
        const size_t n = (size_t)m * topk;
        const uint32_t *g = root->d_gather;
        // parts were written with this block's m: their strides are part_words(blk) apart but laid out for m queries
        uint32_t *L = root->d_lists;
        if (int rc = launch_merge_parts_seq((uint32_t)P, m, topk, words, g, reinterpret_cast<const float *>(g + n),
                                            reinterpret_cast<const float *>(g + 2 * n), g + 3 * n, L, reinterpret_cast<float *>(L + n),
                                            reinterpret_cast<float *>(L + 2 * n), L + 3 * n, root->stream))
            return rc;
        if (lists_out) {
            std::vector<uint32_t> hr(n), hc(m);
            std::vector<float> hs(n);
            CX_HIP(hipMemcpyAsync(hr.data(), L, n * 4, hipMemcpyDeviceToHost, root->stream));
            CX_HIP(hipMemcpyAsync(hs.data(), L + n, n * 4, hipMemcpyDeviceToHost, root->stream));
            CX_HIP(hipMemcpyAsync(hc.data(), L + 3 * n, (size_t)m * 4, hipMemcpyDeviceToHost, root->stream));
            CX_HIP(hipStreamSynchronize(root->stream));
            if (int rc = check_result_block(hc.data(), hr.data(), m, topk, topk, n_seq)) return rc;
            for (uint32_t i = 0; i < m; i++) {
                const size_t p = lists_out->pos[lo + i];
                lists_out->counts[p] = hc[i];
                memcpy(lists_out->rows + p * topk, hr.data() + (size_t)i * topk, (size_t)hc[i] * 4);
                memcpy(lists_out->scores + p * topk, hs.data() + (size_t)i * topk, (size_t)hc[i] * 4);
            }
            continue;
        }
        CX_HIP(hipMemcpyAsync(root->d_scan, scan.data() + lo, (size_t)m * 4, hipMemcpyHostToDevice, root->stream));
        LinkArgs a;
        memset(&a, 0, sizeof a);
        a.scan_rows = root->d_scan;
        a.list_rows = L;
        a.list_scores = reinterpret_cast<float *>(L + n);
        a.list_cnt = L + 3 * n;
        a.deleted = deleted ? root->d_deleted : nullptr;
        a.meta = nullptr;   // removed rows never enter `scan`
        a.n_scan = m;
        a.topk = topk;
        a.max_edges = dedup ? 0xFFFFFFFFu : max_edges;
        a.max_total = dedup ? ~0ull : max_cycle - emitted;
        a.dedup = dedup ? 1u : 0u;
        a.threshold = threshold;
        a.counts = root->d_counts;
        if (has_ex) {
            off_blk.resize((size_t)m + 1);
            const uint64_t base = ex_off[lo];
            for (uint32_t i = 0; i <= m; i++) off_blk[i] = ex_off[lo + i] - base;
            const uint64_t n_ex = off_blk[m];
            ex_sorted.assign(ex_to + base, ex_to + base + n_ex);
            for (uint32_t i = 0; i < m; i++)
                if (off_blk[i + 1] - off_blk[i] > 1) std::sort(ex_sorted.begin() + off_blk[i], ex_sorted.begin() + off_blk[i + 1]);
            if (int rc = ensure_dev(root->d_exist_to, root->c_exist_to, (size_t)std::max<uint64_t>(n_ex, 1))) return rc;
            CX_HIP(hipMemcpyAsync(root->d_exist_off, off_blk.data(), ((size_t)m + 1) * 8, hipMemcpyHostToDevice, root->stream));
            if (n_ex) CX_HIP(hipMemcpyAsync(root->d_exist_to, ex_sorted.data(), (size_t)n_ex * 4, hipMemcpyHostToDevice, root->stream));
            a.existing_offsets = root->d_exist_off;
            a.existing_to = root->d_exist_to;
        }
        // dedup.rs:85-87 is search_threshold — no k: a merged list that is full at topk with its tail still at or above the
        // threshold hides neighbours.  Such rows ("dense") emit nothing on the device; their complete lists come from the
        // shards' threshold searches and are walked here (autolink.cpp does the same for one index).
        std::vector<uint32_t> dense;   // positions in the block
        if (dedup) {
            std::vector<uint32_t> cnt(m);
            std::vector<float> tail(m);
            CX_HIP(hipMemcpyAsync(cnt.data(), L + 3 * n, (size_t)m * 4, hipMemcpyDeviceToHost, root->stream));
            CX_HIP(hipMemcpy2DAsync(tail.data(), 4, reinterpret_cast<const float *>(L + n) + (topk - 1), (size_t)topk * 4, 4, m, hipMemcpyDeviceToHost,
                                    root->stream));
            CX_HIP(hipStreamSynchronize(root->stream));
            for (uint32_t i = 0; i < m; i++)
                if (cnt[i] >= topk && tail[i] >= threshold) dense.push_back(i);
            if (!dense.empty()) {
                if (int rc = ensure_dev(root->d_exist_to, root->c_exist_to, dense.size())) return rc;   // free in a dedup pass: the dense positions
                CX_HIP(hipMemcpyAsync(root->d_exist_to, dense.data(), dense.size() * 4, hipMemcpyHostToDevice, root->stream));
                if (int rc = launch_patch_u32(L + 3 * n, root->d_exist_to, nullptr, 0u, (uint32_t)dense.size(), root->stream)) return rc;
            }
        }
        if (int rc = launch_link_rules(a, false, root->stream)) return rc;
        if (int rc = launch_exclusive_scan(root->d_counts, root->d_offsets, m, root->d_temp, tb, root->stream)) return rc;
        uint64_t last_off = 0;
        uint32_t last_cnt = 0;
        CX_HIP(hipMemcpyAsync(&last_off, root->d_offsets + (m - 1), 8, hipMemcpyDeviceToHost, root->stream));
        CX_HIP(hipMemcpyAsync(&last_cnt, root->d_counts + (m - 1), 4, hipMemcpyDeviceToHost, root->stream));
        CX_HIP(hipStreamSynchronize(root->stream));
        const uint64_t n_edges = std::min<uint64_t>(last_off + last_cnt, a.max_total);
        const size_t at = o_from.size();
        if (n_edges) {
            if (int rc = ensure_dev(root->d_from, root->c_from, (size_t)n_edges)) return rc;
            if (int rc = ensure_dev(root->d_to, root->c_to, (size_t)n_edges)) return rc;
            if (int rc = ensure_dev(root->d_w, root->c_w, (size_t)n_edges)) return rc;
            a.offsets = root->d_offsets;
            a.out_from = root->d_from;
            a.out_to = root->d_to;
            a.out_weight = root->d_w;
            if (int rc = launch_link_rules(a, true, root->stream)) return rc;
            o_from.resize(at + n_edges); o_to.resize(at + n_edges); o_w.resize(at + n_edges);
            CX_HIP(hipMemcpyAsync(o_from.data() + at, root->d_from, n_edges * 4, hipMemcpyDeviceToHost, root->stream));
            CX_HIP(hipMemcpyAsync(o_to.data() + at, root->d_to, n_edges * 4, hipMemcpyDeviceToHost, root->stream));
            CX_HIP(hipMemcpyAsync(o_w.data() + at, root->d_w, n_edges * 4, hipMemcpyDeviceToHost, root->stream));
            CX_HIP(hipStreamSynchronize(root->stream));
            for (size_t i = at; i < o_from.size(); i++)
                if (o_from[i] >= n_seq || o_to[i] >= n_seq)
                    return set_err(CX_ERR_DEVICE, "device edge list is corrupt: %u -> %u in an index of %llu rows", o_from[i], o_to[i],
                                   (unsigned long long)n_seq);
            emitted += n_edges;
        }
        if (!dense.empty()) {   // splice the dense rows' pairs into the block's edges, in scan order
            std::vector<uint64_t> off(m);
            CX_HIP(hipMemcpy(off.data(), root->d_offsets, (size_t)m * 8, hipMemcpyDeviceToHost));
            std::vector<uint32_t> b_from(o_from.begin() + at, o_from.end()), b_to(o_to.begin() + at, o_to.end());
            std::vector<float> b_w(o_w.begin() + at, o_w.end());
            o_from.resize(at); o_to.resize(at); o_w.resize(at);
            std::vector<float> vec(dim);
            std::vector<Hit> all;
            size_t di = 0;
            for (uint32_t i = 0; i < m; i++) {
                const uint64_t lo_e = off[i], hi_e = i + 1 < m ? off[i + 1] : b_from.size();
                if (di < dense.size() && dense[di] == i) {
                    di++;
                    const uint32_t self = scan[lo + i], sh = h->seq_shard[self];
                    const cx_index *ix = h->shards[sh];
                    LinkShard &l = *ls[sh];
                    CX_HIP(hipSetDevice(l.device));
                    if (int rc = cx_copy_rows_dev(ix, h->seq_row[self], 1, l.d_q, l.stream)) return rc;
                    CX_HIP(hipMemcpyAsync(vec.data(), l.d_q, (size_t)dim * 4, hipMemcpyDeviceToHost, l.stream));
                    CX_HIP(hipStreamSynchronize(l.stream));
                    CX_HIP(hipSetDevice(h->root));
                    if (int rc = threshold_hits(h, vec.data(), dim, threshold, nullptr, all)) return rc;
                    for (const Hit &x : all) {   // dedup.rs:89-113 as link_rules_kernel walks a list
                        if (x.seq == self) continue;
                        if (x.seq < self && !(deleted && deleted[x.seq])) continue;
                        if (!(x.score >= threshold)) continue;
                        o_from.push_back(self); o_to.push_back(x.seq); o_w.push_back(x.score);
                    }
                } else {
                    o_from.insert(o_from.end(), b_from.begin() + lo_e, b_from.begin() + hi_e);
                    o_to.insert(o_to.end(), b_to.begin() + lo_e, b_to.begin() + hi_e);
                    o_w.insert(o_w.end(), b_w.begin() + lo_e, b_w.begin() + hi_e);
                }
            }
        }
    }
    return CX_OK;
}

int hand_over(const std::vector<uint32_t> &f, const std::vector<uint32_t> &t, const std::vector<float> &w, uint64_t cap,
              uint32_t *out_from, uint32_t *out_to, float *out_w, uint64_t *n_out, uint64_t *n_needed) {
    const uint64_t total = f.size(), take = std::min<uint64_t>(total, cap);
    if (n_needed) *n_needed = total;
    if (take) {
        if (!out_from || !out_to || !out_w) return set_err(CX_ERR_VALIDATION, "null output buffer");
        memcpy(out_from, f.data(), take * 4);
        memcpy(out_to, t.data(), take * 4);
        memcpy(out_w, w.data(), take * 4);
    }
    *n_out = take;
    if (total > cap) return set_err(CX_ERR_CAPACITY, "%llu edges, buffer holds %llu", (unsigned long long)total, (unsigned long long)cap);
    return CX_OK;
}

}  // namespace

extern "C" {

int cx_sharded_autolink_pass_rows(const cx_sharded *h, uint64_t n_scan, const uint32_t *scan_rows, uint64_t topk,
                                  float threshold, uint64_t max_edges_per_node, uint64_t max_edges_per_cycle,
                                  const uint8_t *deleted, const uint64_t *existing_offsets, const uint32_t *existing_to,
                                  uint64_t cap, uint32_t *out_from, uint32_t *out_to, float *out_weight, uint64_t *n_out,
                                  uint64_t *n_needed) try {
    if (!h || !n_out) return set_err(CX_ERR_VALIDATION, "null argument");
    *n_out = 0;
    if (n_needed) *n_needed = 0;
    const uint64_t n_seq = h->seq_shard.size();
    if (!scan_rows) n_scan = n_seq;
    if (existing_offsets) {
        if (existing_offsets[0] != 0) return set_err(CX_ERR_VALIDATION, "autolink: existing_offsets[0] must be 0");
        for (uint64_t i = 0; i < n_scan; i++)
            if (existing_offsets[i + 1] < existing_offsets[i]) return set_err(CX_ERR_VALIDATION, "autolink: existing_offsets decrease at node %llu", (unsigned long long)i);
        if (existing_offsets[n_scan] && !existing_to) return set_err(CX_ERR_VALIDATION, "autolink: existing_to is null");
    }
    // a removed row has no embedding and proposes nothing (auto_linker.rs:217-218): it leaves the scan set, its
    // existing-edge segment with it
    std::vector<uint32_t> scan;
    std::vector<uint64_t> ex_off;
    std::vector<uint32_t> ex_to;
    scan.reserve(n_scan);
    if (existing_offsets) ex_off.push_back(0);
    for (uint64_t i = 0; i < n_scan; i++) {
        const uint64_t q = scan_rows ? scan_rows[i] : i;
        if (q >= n_seq) return set_err(CX_ERR_VALIDATION, "autolink: scan row %llu out of range", (unsigned long long)q);
        if (!h->seq_alive[q]) continue;
        scan.push_back((uint32_t)q);
        if (existing_offsets) {
            ex_to.insert(ex_to.end(), existing_to + existing_offsets[i], existing_to + existing_offsets[i + 1]);
            ex_off.push_back(ex_to.size());
        }
    }
    std::vector<uint32_t> f, t;
    std::vector<float> w;
    if (int rc = link_pass_sharded(h, scan, ex_off, ex_to.data(), (uint32_t)std::min<uint64_t>(topk, 0xFFFFFFFFull), threshold,
                                   (uint32_t)std::min<uint64_t>(max_edges_per_node, 0xFFFFFFFFull), max_edges_per_cycle, deleted, false,
                                   f, t, w))
        return rc;
    return hand_over(f, t, w, cap, out_from, out_to, out_weight, n_out, n_needed);
} catch (...) { return cx::on_exception(); }

int cx_sharded_dedup_scan_rows(const cx_sharded *h, float dedup_threshold, const uint8_t *deleted, uint64_t cap,
                               uint32_t *out_a, uint32_t *out_b, float *out_similarity, uint64_t *n_out,
                               uint64_t *n_needed) try {
    if (!h || !n_out) return set_err(CX_ERR_VALIDATION, "null argument");
    *n_out = 0;
    if (n_needed) *n_needed = 0;
    std::vector<uint32_t> scan;   // every indexed, non-deleted node in (global) row order (dedup.rs:70-81)
    for (uint64_t q = 0; q < h->seq_shard.size(); q++)
        if (h->seq_alive[q] && !(deleted && deleted[q])) scan.push_back((uint32_t)q);
    std::vector<uint32_t> f, t;
    std::vector<float> w;
    if (int rc = link_pass_sharded(h, scan, {}, nullptr, TOPK_MAX, dedup_threshold, 0, ~0ull, deleted, true, f, t, w)) return rc;
    return hand_over(f, t, w, cap, out_a, out_b, out_similarity, n_out, n_needed);
} catch (...) { return cx::on_exception(); }

}  // extern "C"

// ------------------------------------------------------------------------ the rest of the single-index surface, sharded

extern "C" {

int cx_sharded_topk_lists_rows(const cx_sharded *h, uint64_t n_scan, const uint32_t *scan_rows, uint64_t topk64,
                               uint32_t *out_rows, float *out_scores, uint32_t *out_counts) try {
    if (!h || !out_rows || !out_scores || !out_counts) return set_err(CX_ERR_VALIDATION, "null argument");
    if (topk64 == 0 || topk64 > TOPK_MAX) return set_err(CX_ERR_VALIDATION, "topk lists: topk must be in 1..%u", TOPK_MAX);
    const uint64_t n_seq = h->seq_shard.size();
    if (!scan_rows) n_scan = n_seq;
    std::vector<uint32_t> scan, pos;
    for (uint64_t i = 0; i < n_scan; i++) {
        const uint64_t q = scan_rows ? scan_rows[i] : i;
        if (q >= n_seq) return set_err(CX_ERR_VALIDATION, "scan row %llu out of range", (unsigned long long)q);
        out_counts[i] = 0;                    // a removed row has no embedding: no list (auto_linker.rs:217-218)
        if (!h->seq_alive[q]) continue;
        scan.push_back((uint32_t)q);
        pos.push_back((uint32_t)i);
    }
    ListsOut lo{out_rows, out_scores, out_counts, pos.data()};
    std::vector<uint32_t> f, t;
    std::vector<float> w;
    return link_pass_sharded(h, scan, {}, nullptr, (uint32_t)topk64, 0.0f, 0, ~0ull, nullptr, false, f, t, w, &lo);
} catch (...) { return cx::on_exception(); }

int cx_sharded_set_metadata_batch(cx_sharded *h, uint64_t n, const uint8_t *ids, const uint32_t *kind_codes,
                                  const uint32_t *agent_codes) try {
    if (!h) return set_err(CX_ERR_VALIDATION, "null index");
    if (n && (!ids || !kind_codes || !agent_codes)) return set_err(CX_ERR_VALIDATION, "null argument");
    for (uint64_t i = 0; i < n; i++)
        if (int rc = cx_sharded_set_metadata(h, ids + 16 * i, kind_codes[i], agent_codes[i])) return rc;
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_sharded_set_node_stats_batch(cx_sharded *h, uint64_t n, const uint8_t *ids, const uint32_t *kind_codes,
                                    const int64_t *last_accessed_s, const uint32_t *last_accessed_ns,
                                    const uint64_t *access_counts) try {
    if (!h) return set_err(CX_ERR_VALIDATION, "null index");
    if (!n) return CX_OK;
    if (!ids || !kind_codes || !last_accessed_s || !access_counts) return set_err(CX_ERR_VALIDATION, "null argument");
    if (h->h_stats.size() < h->seq_shard.size()) h->h_stats.resize(h->seq_shard.size());
    for (uint64_t i = 0; i < n; i++) {
        auto it = h->map.find(id_key(ids + 16 * i));
        if (it == h->map.end()) continue;   // a node without a vector never shows up in a search
        NodeStats &s = h->h_stats[it->second];
        s.last_s = last_accessed_s[i];
        s.last_ns = last_accessed_ns ? last_accessed_ns[i] : 0u;
        s.kind = kind_codes[i];
        s.access = access_counts[i];
    }
    return CX_OK;
} catch (...) { return cx::on_exception(); }

int cx_sharded_bulk_load_nodes(cx_sharded *h, uint64_t n, const uint8_t *blob, const uint64_t *offsets, uint32_t flags,
                               cx_bulk_stats *stats) try {
    if (stats) *stats = cx_bulk_stats{};
    if (!h) return set_err(CX_ERR_VALIDATION, "null index");
    cx::BulkSink sink;
    sink.dim = h->dim;
    sink.device = h->root;
    sink.upsert = [h](uint64_t m, const uint8_t *ids, const float *embs) { return cx_sharded_upsert_batch(h, m, ids, embs, h->dim); };
    sink.intern = [h](const char *p, uint64_t len) { return cx_sharded_intern(h, p, len); };
    sink.set_stats = [h](uint64_t m, const uint8_t *ids, const uint32_t *kc, const int64_t *ls, const uint32_t *lns, const uint64_t *ac) {
        return cx_sharded_set_node_stats_batch(h, m, ids, kc, ls, lns, ac);
    };
    sink.set_meta = [h](uint64_t m, const uint8_t *ids, const uint32_t *k, const uint32_t *a) { return cx_sharded_set_metadata_batch(h, m, ids, k, a); };
    return cx::bulk_load_impl(sink, n, blob, offsets, flags, stats);
} catch (...) { return cx::on_exception(); }

/* cx_search_decayed over the shards: the handler's candidates -> apply_score_decay -> stable re-rank -> truncate
 * (routes.rs:889-947), with the candidates from cx_sharded_search and the node stats kept per global row. */
int cx_sharded_search_decayed(const cx_sharded *h, const float *query, uint64_t len, uint64_t limit, uint64_t candidate_limit,
                              const cx_filter *filter, const cx_decay_config *cfg, float recency_bias, int64_t now_s,
                              uint32_t now_ns, uint8_t *out_ids, float *out_scores, float *out_raw_scores, uint64_t *n_out) try {
    if (!h || !query || !cfg || !n_out) return set_err(CX_ERR_VALIDATION, "null argument");
    *n_out = 0;
    if (cfg->n_by_kind && (!cfg->kind_codes || !cfg->kind_rates)) return set_err(CX_ERR_VALIDATION, "null by_kind table");
    if (candidate_limit < limit) candidate_limit = limit;
    const uint64_t cap = std::max<uint64_t>(1, std::min<uint64_t>(candidate_limit, h->seq_shard.size()));
    std::vector<uint8_t> ids(16 * (size_t)cap);
    std::vector<float> raw((size_t)cap), dist((size_t)cap);
    uint64_t n = 0;
    if (int rc = cx_sharded_search(h, query, len, candidate_limit, filter, ids.data(), raw.data(), dist.data(), &n)) return rc;
    if (n && limit && (!out_ids || !out_scores || !out_raw_scores)) return set_err(CX_ERR_VALIDATION, "null output buffer");
    std::vector<float> fin((size_t)n);
    std::vector<uint32_t> order((size_t)n);
    for (uint64_t i = 0; i < n; i++) {
        auto it = h->map.find(id_key(&ids[16 * (size_t)i]));
        int64_t ls = 0; uint32_t lns = 0, kind = 0; uint64_t ac = 0;   // nodes nobody described: the epoch, never accessed (types.rs:56)
        if (it != h->map.end() && it->second < h->h_stats.size()) {
            const NodeStats &s = h->h_stats[it->second];
            ls = s.last_s; lns = s.last_ns; kind = s.kind; ac = s.access;
        }
        fin[i] = cx_apply_score_decay(cfg, raw[i], recency_bias, now_s, now_ns, kind, ls, lns, ac);
        order[i] = (uint32_t)i;
    }
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
