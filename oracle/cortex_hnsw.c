/*
 * cortex_hnsw.c — CPU HNSW baseline.  TEST INFRASTRUCTURE / REPORTED BASELINE ONLY.
 *
 * The reference's approximate path is the third-party crate instant-distance 0.6.1 (Cargo.lock:2066-2069),
 * whose source is NOT in /root/reference; it is called at vector/index.rs:345-346 (search) and :430
 * (Builder::default().build).  This file is a from-the-paper restatement (Malkov & Yashunin 2018,
 * Algorithms 1-5: layered greedy search, ef-bounded best-first layer search, heuristic neighbour
 * selection, bidirectional links with shrinking) using the parameters recalled for that crate —
 * M = 32, M0 = 64, ef_construction = 100, ef_search = 100, mL = 1/ln(M) — and the reference's own
 * point distance (EmbeddingPoint::distance, index.rs:169-179, via cxo_distance).  It reproduces neither the
 * crate's RNG nor its parallel insertion order: HNSW result sets and recall are PARITY UNPINNED; the
 * numbers it yields are labelled "restatement" wherever they are reported.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "cortex_oracle.h"

typedef struct { float d; uint32_t id; } cand;

typedef struct {
    uint32_t n, dim;
    const float *rows;        /* [n][dim], borrowed */
    int M, M0, efc;
    int max_level;
    uint32_t entry;
    uint8_t *level;           /* [n] */
    uint32_t **links;         /* [n] -> per node: for each level l<=level: [cap+1] (count first) */
    uint32_t *visited;        /* epoch marks */
    uint32_t epoch;
    uint64_t rng;
    uint64_t dist_evals;
    uint8_t *locks;           /* [n] spin locks of the neighbour lists: the multi-threaded build and batch search only */
} hnsw;

static inline int cap_of(const hnsw *h, int lvl) { return lvl == 0 ? h->M0 : h->M; }
static inline uint32_t *list_of(const hnsw *h, uint32_t id, int lvl) {
    uint32_t *p = h->links[id];
    int off = 0;
    for (int l = 0; l < lvl; l++) off += cap_of(h, l) + 1;
    return p + off;
}
static inline float dist(hnsw *h, const float *q, uint32_t id) {
    h->dist_evals++;
    return cxo_distance(q, h->rows + (size_t)id * h->dim, h->dim);
}
static double uniform(hnsw *h) { /* xorshift64* */
    h->rng ^= h->rng >> 12; h->rng ^= h->rng << 25; h->rng ^= h->rng >> 27;
    return (double)((h->rng * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}

/* binary heaps over cand: min-heap (closest first) and max-heap (farthest first) */
static void heap_push(cand *a, int *n, cand c, int max) {
    int i = (*n)++;
    a[i] = c;
    while (i > 0) {
        int p = (i - 1) / 2;
        int swap = max ? a[p].d < a[i].d : a[p].d > a[i].d;
        if (!swap) break;
        cand t = a[p]; a[p] = a[i]; a[i] = t; i = p;
    }
}
static cand heap_pop(cand *a, int *n, int max) {
    cand top = a[0];
    a[0] = a[--(*n)];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, b = i;
        if (l < *n && (max ? a[l].d > a[b].d : a[l].d < a[b].d)) b = l;
        if (r < *n && (max ? a[r].d > a[b].d : a[r].d < a[b].d)) b = r;
        if (b == i) break;
        cand t = a[b]; a[b] = a[i]; a[i] = t; i = b;
    }
    return top;
}

/* Algorithm 2: ef-bounded best-first search on one layer; result (unsorted) in res[0..*nres) */
static void search_layer(hnsw *h, const float *q, uint32_t ep, int ef, int lvl, cand *res, int *nres, cand *cq, int qcap) {
    int ncq = 0;
    *nres = 0;
    h->epoch++;
    cand e = { dist(h, q, ep), ep };
    h->visited[ep] = h->epoch;
    heap_push(cq, &ncq, e, 0);
    heap_push(res, nres, e, 1);
    while (ncq > 0) {
        cand c = heap_pop(cq, &ncq, 0);
        if (*nres >= ef && c.d > res[0].d) break;
        uint32_t *nb = list_of(h, c.id, lvl);
        for (uint32_t i = 1; i <= nb[0]; i++) {
            uint32_t v = nb[i];
            if (h->visited[v] == h->epoch) continue;
            h->visited[v] = h->epoch;
            float dv = dist(h, q, v);
            if (*nres < ef || dv < res[0].d) {
                cand cv = { dv, v };
                if (ncq < qcap) heap_push(cq, &ncq, cv, 0);
                heap_push(res, nres, cv, 1);
                if (*nres > ef) heap_pop(res, nres, 1);
            }
        }
    }
}

static int cmp_cand(const void *a, const void *b) {
    float x = ((const cand *)a)->d, y = ((const cand *)b)->d;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* Algorithm 4: heuristic selection of up to m neighbours from sorted candidates */
static int select_heuristic_c(hnsw *h, cand *c, int nc, int m, uint32_t *out, uint64_t *evals) {
    qsort(c, (size_t)nc, sizeof(cand), cmp_cand);
    int no = 0;
    for (int i = 0; i < nc && no < m; i++) {
        int good = 1;
        const float *vi = h->rows + (size_t)c[i].id * h->dim;
        for (int j = 0; j < no; j++) {
            (*evals)++;
            if (cxo_distance(vi, h->rows + (size_t)out[j] * h->dim, h->dim) < c[i].d) { good = 0; break; }
        }
        if (good) out[no++] = c[i].id;
    }
    return no;
}
static int select_heuristic(hnsw *h, cand *c, int nc, int m, uint32_t *out) { return select_heuristic_c(h, c, nc, m, out, &h->dist_evals); }

void *cxo_hnsw_build(const float *rows, uint32_t n, uint32_t dim, int M, int M0, int efc, uint64_t seed) {
    hnsw *h = (hnsw *)calloc(1, sizeof *h);
    h->n = n; h->dim = dim; h->rows = rows; h->M = M; h->M0 = M0; h->efc = efc;
    h->rng = seed ? seed : 0x9E3779B97F4A7C15ull;
    h->level = (uint8_t *)calloc(n ? n : 1, 1);
    h->links = (uint32_t **)calloc(n ? n : 1, sizeof(uint32_t *));
    h->visited = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    h->max_level = -1;
    const double mL = 1.0 / log((double)M);
    const int qcap = 4 * efc + 4 * M0 + 64;
    cand *res = (cand *)malloc((size_t)(efc + 2) * sizeof(cand));
    cand *cq = (cand *)malloc((size_t)qcap * sizeof(cand));
    cand *tmp = (cand *)malloc((size_t)(M0 + 2) * sizeof(cand));
    uint32_t *sel = (uint32_t *)malloc((size_t)(M0 + 1) * sizeof(uint32_t));
    for (uint32_t id = 0; id < n; id++) {
        double u = uniform(h);
        if (u < 1e-300) u = 1e-300;
        int lvl = (int)floor(-log(u) * mL);
        if (lvl > 15) lvl = 15;
        h->level[id] = (uint8_t)lvl;
        size_t words = 0;
        for (int l = 0; l <= lvl; l++) words += (size_t)cap_of(h, l) + 1;
        h->links[id] = (uint32_t *)calloc(words, sizeof(uint32_t));
        if (h->max_level < 0) { h->max_level = lvl; h->entry = id; continue; }
        const float *q = rows + (size_t)id * dim;
        uint32_t ep = h->entry;
        int nres;
        for (int l = h->max_level; l > lvl; l--) {          /* greedy descent, ef = 1 */
            search_layer(h, q, ep, 1, l, res, &nres, cq, qcap);
            ep = res[0].id;
        }
        for (int l = lvl < h->max_level ? lvl : h->max_level; l >= 0; l--) {
            search_layer(h, q, ep, efc, l, res, &nres, cq, qcap);
            int best = 0;
            for (int i = 1; i < nres; i++) if (res[i].d < res[best].d) best = i;
            ep = res[best].id;
            const int m = l == 0 ? M0 : M;   /* this restatement links up to the layer's capacity */
            int ns = select_heuristic(h, res, nres, m, sel);
            uint32_t *mine = list_of(h, id, l);
            mine[0] = (uint32_t)ns;
            memcpy(mine + 1, sel, (size_t)ns * sizeof(uint32_t));
            for (int s = 0; s < ns; s++) {                    /* bidirectional link, shrink if full */
                uint32_t *nb = list_of(h, sel[s], l);
                const int capn = cap_of(h, l);
                if ((int)nb[0] < capn) { nb[++nb[0]] = id; continue; }
                const float *vn = rows + (size_t)sel[s] * dim;
                int nt = 0;
                for (uint32_t i = 1; i <= nb[0]; i++) { tmp[nt].id = nb[i]; tmp[nt].d = dist(h, vn, nb[i]); nt++; }
                tmp[nt].id = id; tmp[nt].d = dist(h, vn, id); nt++;
                uint32_t keep[65];
                int nk = select_heuristic(h, tmp, nt, capn, keep);
                nb[0] = (uint32_t)nk;
                memcpy(nb + 1, keep, (size_t)nk * sizeof(uint32_t));
            }
        }
        if (lvl > h->max_level) { h->max_level = lvl; h->entry = id; }
    }
    free(res); free(cq); free(tmp); free(sel);
    return h;
}

/* Algorithm 5; out_rows/out_dist get up to k results closest-first.  Not thread-safe (visited marks). */
size_t cxo_hnsw_search(void *hv, const float *q, size_t k, int ef, uint32_t *out_rows, float *out_dist) {
    hnsw *h = (hnsw *)hv;
    if (!h->n) return 0;
    if (ef < (int)k) ef = (int)k;
    const int qcap = 4 * ef + 4 * h->M0 + 64;
    cand *res = (cand *)malloc((size_t)(ef + 2) * sizeof(cand));
    cand *cq = (cand *)malloc((size_t)qcap * sizeof(cand));
    uint32_t ep = h->entry;
    int nres;
    for (int l = h->max_level; l > 0; l--) {
        search_layer(h, q, ep, 1, l, res, &nres, cq, qcap);
        ep = res[0].id;
    }
    search_layer(h, q, ep, ef, 0, res, &nres, cq, qcap);
    qsort(res, (size_t)nres, sizeof(cand), cmp_cand);
    size_t n = (size_t)nres < k ? (size_t)nres : k;
    for (size_t i = 0; i < n; i++) { out_rows[i] = res[i].id; out_dist[i] = res[i].d; }
    free(res); free(cq);
    return n;
}

/* ---- multi-threaded build and batch search --------------------------------------------------------------------------
 * The reference builds with rayon (Builder::build inserts in parallel, index.rs:430) and searches batches with
 * par_iter (:390-410); a baseline timed "on the same box's host cores" has to use them too.  Same algorithm as above,
 * the usual concurrent form: every node's neighbour lists are guarded by a spin lock (a reader copies the list out
 * under it), a node whose level exceeds the current top holds a global lock for its whole insertion, every thread has
 * its own visited marks and heaps.  The insertion order — and so the graph — depends on the schedule, as it does in
 * the reference: parity unpinned either way. */
typedef struct { uint32_t *visited; uint32_t epoch; uint64_t evals; cand *res, *cq, *tmp; uint32_t *sel; uint32_t nbuf[80]; } tctx;

static inline void lock_node(hnsw *h, uint32_t id) {
    while (__atomic_test_and_set(&h->locks[id], __ATOMIC_ACQUIRE))
        while (__atomic_load_n(&h->locks[id], __ATOMIC_RELAXED)) { }
}
static inline void unlock_node(hnsw *h, uint32_t id) { __atomic_clear(&h->locks[id], __ATOMIC_RELEASE); }
static inline float dist_t(hnsw *h, tctx *t, const float *q, uint32_t id) {
    t->evals++;
    return cxo_distance(q, h->rows + (size_t)id * h->dim, h->dim);
}
static void search_layer_mt(hnsw *h, tctx *t, const float *q, uint32_t ep, int ef, int lvl, int *nres, int qcap) {
    int ncq = 0;
    *nres = 0;
    if (++t->epoch == 0) { memset(t->visited, 0, (size_t)h->n * sizeof(uint32_t)); t->epoch = 1; }
    cand e = { dist_t(h, t, q, ep), ep };
    t->visited[ep] = t->epoch;
    heap_push(t->cq, &ncq, e, 0);
    heap_push(t->res, nres, e, 1);
    while (ncq > 0) {
        cand c = heap_pop(t->cq, &ncq, 0);
        if (*nres >= ef && c.d > t->res[0].d) break;
        lock_node(h, c.id);
        const uint32_t *nb = list_of(h, c.id, lvl);
        const uint32_t nn = nb[0];
        memcpy(t->nbuf, nb + 1, (size_t)nn * sizeof(uint32_t));
        unlock_node(h, c.id);
        for (uint32_t i = 0; i < nn; i++) {
            uint32_t v = t->nbuf[i];
            if (t->visited[v] == t->epoch) continue;
            t->visited[v] = t->epoch;
            float dv = dist_t(h, t, q, v);
            if (*nres < ef || dv < t->res[0].d) {
                cand cv = { dv, v };
                if (ncq < qcap) heap_push(t->cq, &ncq, cv, 0);
                heap_push(t->res, nres, cv, 1);
                if (*nres > ef) heap_pop(t->res, nres, 1);
            }
        }
    }
}
static void tctx_init(tctx *t, const hnsw *h, int ef) {
    const int qcap = 4 * ef + 4 * h->M0 + 64;
    t->visited = (uint32_t *)calloc(h->n ? h->n : 1, sizeof(uint32_t));
    t->epoch = 0;
    t->evals = 0;
    t->res = (cand *)malloc((size_t)(ef + 2) * sizeof(cand));
    t->cq = (cand *)malloc((size_t)qcap * sizeof(cand));
    t->tmp = (cand *)malloc((size_t)(h->M0 + 2) * sizeof(cand));
    t->sel = (uint32_t *)malloc((size_t)(h->M0 + 1) * sizeof(uint32_t));
}
static void tctx_free(tctx *t) { free(t->visited); free(t->res); free(t->cq); free(t->tmp); free(t->sel); }

void *cxo_hnsw_build_mt(const float *rows, uint32_t n, uint32_t dim, int M, int M0, int efc, uint64_t seed, int n_threads) {
    hnsw *h = (hnsw *)calloc(1, sizeof *h);
    h->n = n; h->dim = dim; h->rows = rows; h->M = M; h->M0 = M0; h->efc = efc;
    h->rng = seed ? seed : 0x9E3779B97F4A7C15ull;
    h->level = (uint8_t *)calloc(n ? n : 1, 1);
    h->links = (uint32_t **)calloc(n ? n : 1, sizeof(uint32_t *));
    h->visited = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    h->locks = (uint8_t *)calloc(n ? n : 1, 1);
    h->max_level = -1;
    if (!n) return h;
    const double mL = 1.0 / log((double)M);
    for (uint32_t id = 0; id < n; id++) {   /* levels and list storage first: the same level sequence as the serial build */
        double u = uniform(h);
        if (u < 1e-300) u = 1e-300;
        int lvl = (int)floor(-log(u) * mL);
        if (lvl > 15) lvl = 15;
        h->level[id] = (uint8_t)lvl;
        size_t words = 0;
        for (int l = 0; l <= lvl; l++) words += (size_t)cap_of(h, l) + 1;
        h->links[id] = (uint32_t *)calloc(words, sizeof(uint32_t));
    }
    h->max_level = h->level[0];
    h->entry = 0;
    const int qcap = 4 * efc + 4 * M0 + 64;
    uint8_t glock = 0;
    uint64_t evals = 0;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads) reduction(+ : evals)
    {
        tctx t;
        tctx_init(&t, h, efc);
#pragma omp for schedule(dynamic, 8)
        for (uint32_t id = 1; id < n; id++) {
            const int lvl = h->level[id];
            int top = __atomic_load_n(&h->max_level, __ATOMIC_ACQUIRE);   /* the level first, then the entry: the entry read is at least that high */
            int have_g = 0;
            if (lvl > top) {
                while (__atomic_test_and_set(&glock, __ATOMIC_ACQUIRE)) { }
                have_g = 1;
                top = __atomic_load_n(&h->max_level, __ATOMIC_ACQUIRE);
            }
            uint32_t ep = __atomic_load_n(&h->entry, __ATOMIC_ACQUIRE);
            const float *q = rows + (size_t)id * dim;
            int nres;
            for (int l = top; l > lvl; l--) {
                search_layer_mt(h, &t, q, ep, 1, l, &nres, qcap);
                ep = t.res[0].id;
            }
            for (int l = lvl < top ? lvl : top; l >= 0; l--) {
                search_layer_mt(h, &t, q, ep, efc, l, &nres, qcap);
                int best = 0;
                for (int i = 1; i < nres; i++) if (t.res[i].d < t.res[best].d) best = i;
                ep = t.res[best].id;
                const int m = l == 0 ? M0 : M;
                int ns = select_heuristic_c(h, t.res, nres, m, t.sel, &t.evals);
                lock_node(h, id);
                uint32_t *mine = list_of(h, id, l);
                /* a neighbour linked back to this node already (it found the node through another thread's link): keep those */
                uint32_t have = mine[0];
                for (int s = 0; s < ns && (int)have < m; s++) {
                    int dup = 0;
                    for (uint32_t i = 1; i <= have; i++) if (mine[i] == t.sel[s]) { dup = 1; break; }
                    if (!dup) mine[++have] = t.sel[s];
                }
                mine[0] = have;
                unlock_node(h, id);
                for (int s = 0; s < ns; s++) {
                    const uint32_t nbid = t.sel[s];
                    lock_node(h, nbid);
                    uint32_t *nb = list_of(h, nbid, l);
                    const int capn = cap_of(h, l);
                    int dup = 0;
                    for (uint32_t i = 1; i <= nb[0]; i++) if (nb[i] == id) { dup = 1; break; }
                    if (!dup) {
                        if ((int)nb[0] < capn) nb[++nb[0]] = id;
                        else {
                            const float *vn = rows + (size_t)nbid * dim;
                            int nt = 0;
                            for (uint32_t i = 1; i <= nb[0]; i++) { t.tmp[nt].id = nb[i]; t.tmp[nt].d = dist_t(h, &t, vn, nb[i]); nt++; }
                            t.tmp[nt].id = id; t.tmp[nt].d = dist_t(h, &t, vn, id); nt++;
                            uint32_t keep[65];
                            int nk = select_heuristic_c(h, t.tmp, nt, capn, keep, &t.evals);
                            nb[0] = (uint32_t)nk;
                            memcpy(nb + 1, keep, (size_t)nk * sizeof(uint32_t));
                        }
                    }
                    unlock_node(h, nbid);
                }
            }
            if (have_g) {
                if (lvl > h->max_level) {
                    __atomic_store_n(&h->entry, id, __ATOMIC_RELEASE);
                    __atomic_store_n(&h->max_level, lvl, __ATOMIC_RELEASE);
                }
                __atomic_clear(&glock, __ATOMIC_RELEASE);
            }
        }
        evals += t.evals;
        tctx_free(&t);
    }
    h->dist_evals = evals;
    return h;
}

/* nq queries, n_threads at a time (HnswIndex::search_batch, index.rs:390-410); out_rows / out_dist [nq][k], out_counts [nq] */
void cxo_hnsw_search_batch(void *hv, const float *queries, size_t nq, size_t k, int ef, int n_threads, uint32_t *out_rows, float *out_dist,
                           size_t *out_counts) {
    hnsw *h = (hnsw *)hv;
    if (ef < (int)k) ef = (int)k;
    if (!h->locks) h->locks = (uint8_t *)calloc(h->n ? h->n : 1, 1);
    const int qcap = 4 * ef + 4 * h->M0 + 64;
    if (n_threads < 1) n_threads = 1;
    uint64_t evals = 0;
#pragma omp parallel num_threads(n_threads) reduction(+ : evals)
    {
        tctx t;
        tctx_init(&t, h, ef);
#pragma omp for schedule(dynamic, 4)
        for (size_t qi = 0; qi < nq; qi++) {
            out_counts[qi] = 0;
            if (!h->n) continue;
            const float *q = queries + qi * h->dim;
            uint32_t ep = h->entry;
            int nres;
            for (int l = h->max_level; l > 0; l--) {
                search_layer_mt(h, &t, q, ep, 1, l, &nres, qcap);
                ep = t.res[0].id;
            }
            search_layer_mt(h, &t, q, ep, ef, 0, &nres, qcap);
            qsort(t.res, (size_t)nres, sizeof(cand), cmp_cand);
            const size_t m = (size_t)nres < k ? (size_t)nres : k;
            for (size_t i = 0; i < m; i++) { out_rows[qi * k + i] = t.res[i].id; out_dist[qi * k + i] = t.res[i].d; }
            out_counts[qi] = m;
        }
        evals += t.evals;
        tctx_free(&t);
    }
    __atomic_fetch_add(&h->dist_evals, evals, __ATOMIC_RELAXED);
}

uint64_t cxo_hnsw_dist_evals(void *hv) { return ((hnsw *)hv)->dist_evals; }

void cxo_hnsw_free(void *hv) {
    hnsw *h = (hnsw *)hv;
    if (!h) return;
    for (uint32_t i = 0; i < h->n; i++) free(h->links[i]);
    free(h->links); free(h->level); free(h->visited); free(h->locks); free(h);
}
