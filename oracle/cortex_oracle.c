/*
 * cortex_oracle.c — CPU restatement of the reference's exact similarity path.
 * TEST INFRASTRUCTURE ONLY (see cortex_oracle.h for the rules and the pinning
 * status).  Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile);
 * sequential left-to-right f32 accumulation with separately rounded products
 * is what `iter().zip().map(|(a,b)| a*b).sum::<f32>()` computes in the
 * reference, so no FMA contraction and no reassociation are allowed here.
 *
 * Declared order where the reference leaves it open: results
 * are ordered by score descending, ties by insertion row ascending (a stable
 * sort over rows visited in insertion order); a NaN score sorts after every
 * number.  The reference iterates a HashMap (random order) into a stable sort
 * with NaN comparing Equal (vector/index.rs:266-292), which leaves both open.
 */
#include "cortex_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[256];
const char *cxo_last_error(void) { return g_err; }

/* ---------------------------------------------------------------- storage */

typedef struct {
    uint8_t id[16];
    float *vec;     /* own allocation per row: HashMap<Uuid, Vec<f32>> (index.rs:187) */
    char *kind;     /* NodeMetadata (index.rs:196-200); NULL = no metadata */
    char *agent;
    int alive;
} cxo_row;

struct cxo_index {
    size_t dim;
    cxo_row *rows;
    size_t n_rows, cap_rows;
    size_t n_alive;
    /* open-addressing id -> row+1 */
    uint32_t *slots;
    size_t n_slots;
    /* the part of `metadata: HashMap<NodeId, NodeMetadata>` (index.rs:189) whose ids have no vector (yet): the
     * two maps are independent (set_metadata :219-222 never looks at `vectors`; the reference's own integration
     * test sets metadata BEFORE insert, vector/tests.rs:65-66).  A plain list: test infrastructure, small. */
    struct cxo_pending { uint8_t id[16]; char *kind, *agent; } *pending;
    size_t n_pending, cap_pending;
};

static long pending_find(const cxo_index *ix, const uint8_t id[16]) {
    for (size_t i = 0; i < ix->n_pending; i++)
        if (memcmp(ix->pending[i].id, id, 16) == 0) return (long)i;
    return -1;
}
static void pending_drop(cxo_index *ix, long i, int free_strings) {
    if (free_strings) { free(ix->pending[i].kind); free(ix->pending[i].agent); }
    ix->pending[i] = ix->pending[ix->n_pending - 1];
    ix->n_pending--;
}

static uint64_t id_hash(const uint8_t id[16]) {
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 16; i++) { h ^= id[i]; h *= 1099511628211ull; }
    return h;
}

static void map_rebuild(cxo_index *ix, size_t n_slots) {
    free(ix->slots);
    ix->slots = (uint32_t *)calloc(n_slots, sizeof(uint32_t));
    ix->n_slots = n_slots;
    for (size_t r = 0; r < ix->n_rows; r++) {
        if (!ix->rows[r].alive) continue;
        size_t s = id_hash(ix->rows[r].id) & (n_slots - 1);
        while (ix->slots[s]) s = (s + 1) & (n_slots - 1);
        ix->slots[s] = (uint32_t)r + 1;
    }
}

static long map_find(const cxo_index *ix, const uint8_t id[16]) {
    if (!ix->n_slots) return -1;
    size_t s = id_hash(id) & (ix->n_slots - 1);
    while (ix->slots[s]) {
        uint32_t r = ix->slots[s] - 1;
        if (ix->rows[r].alive && memcmp(ix->rows[r].id, id, 16) == 0) return (long)r;
        s = (s + 1) & (ix->n_slots - 1);
    }
    return -1;
}

cxo_index *cxo_index_new(size_t dimension) {
    cxo_index *ix = (cxo_index *)calloc(1, sizeof(*ix));
    ix->dim = dimension;
    map_rebuild(ix, 1024);
    return ix;
}

void cxo_index_free(cxo_index *ix) {
    if (!ix) return;
    for (size_t r = 0; r < ix->n_rows; r++) {
        free(ix->rows[r].vec); free(ix->rows[r].kind); free(ix->rows[r].agent);
    }
    for (size_t i = 0; i < ix->n_pending; i++) { free(ix->pending[i].kind); free(ix->pending[i].agent); }
    free(ix->pending);
    free(ix->rows); free(ix->slots); free(ix);
}

size_t cxo_len(const cxo_index *ix) { return ix->n_alive; }
size_t cxo_dimension(const cxo_index *ix) { return ix->dim; }
size_t cxo_row_count(const cxo_index *ix) { return ix->n_rows; }
const float *cxo_row_ptr(const cxo_index *ix, uint32_t row) {
    return (row < ix->n_rows && ix->rows[row].alive) ? ix->rows[row].vec : NULL;
}

/* vector/index.rs:298-314 — dimension check, then HashMap upsert */
int cxo_insert(cxo_index *ix, const uint8_t id[16], const float *emb, size_t len) {
    if (len != ix->dim) {
        snprintf(g_err, sizeof g_err, "Embedding dimension mismatch: expected %zu, got %zu",
                 ix->dim, len);
        return -1;
    }
    long r = map_find(ix, id);
    if (r >= 0) { /* replace value, keep row position */
        memcpy(ix->rows[r].vec, emb, len * sizeof(float));
        return 0;
    }
    if (ix->n_rows == ix->cap_rows) {
        ix->cap_rows = ix->cap_rows ? ix->cap_rows * 2 : 1024;
        ix->rows = (cxo_row *)realloc(ix->rows, ix->cap_rows * sizeof(cxo_row));
    }
    cxo_row *row = &ix->rows[ix->n_rows];
    memset(row, 0, sizeof *row);
    memcpy(row->id, id, 16);
    row->vec = (float *)malloc((len ? len : 1) * sizeof(float));
    memcpy(row->vec, emb, len * sizeof(float));
    row->alive = 1;
    long pm = pending_find(ix, id);   /* metadata.get(id) now finds what set_metadata stored earlier */
    if (pm >= 0) { row->kind = ix->pending[pm].kind; row->agent = ix->pending[pm].agent; pending_drop(ix, pm, 0); }
    ix->n_rows++; ix->n_alive++;
    if ((ix->n_rows + 1) * 2 > ix->n_slots) map_rebuild(ix, ix->n_slots * 2);
    else {
        size_t s = id_hash(id) & (ix->n_slots - 1);
        while (ix->slots[s]) s = (s + 1) & (ix->n_slots - 1);
        ix->slots[s] = (uint32_t)(ix->n_rows - 1) + 1;
    }
    return 0;
}

int cxo_insert_batch(cxo_index *ix, size_t n, const uint8_t *ids, const float *embs, size_t len) {
    for (size_t i = 0; i < n; i++)
        if (cxo_insert(ix, ids + 16 * i, embs + i * len, len)) return -1;
    return 0;
}

/* vector/index.rs:316-323 — drop vector and metadata; never an error */
int cxo_remove(cxo_index *ix, const uint8_t id[16]) {
    long pm = pending_find(ix, id);   /* self.metadata.remove(&id) :318, vector or not */
    if (pm >= 0) pending_drop(ix, pm, 1);
    long r = map_find(ix, id);
    if (r < 0) return 0;
    cxo_row *row = &ix->rows[r];
    row->alive = 0;
    free(row->vec); row->vec = NULL;
    free(row->kind); row->kind = NULL;
    free(row->agent); row->agent = NULL;
    ix->n_alive--;
    map_rebuild(ix, ix->n_slots); /* keep probing chains valid */
    return 0;
}

/* vector/index.rs:219-222 — metadata map is independent of the vector map;
 * matches_filter only ever looks it up for ids that are in `vectors`, so an
 * entry for an id without a vector waits in `pending` until the insert. */
void cxo_set_metadata(cxo_index *ix, const uint8_t id[16], const char *kind, const char *agent) {
    long r = map_find(ix, id);
    if (r < 0) {
        long pm = pending_find(ix, id);
        if (pm >= 0) { free(ix->pending[pm].kind); free(ix->pending[pm].agent); }
        else {
            if (ix->n_pending == ix->cap_pending) {
                ix->cap_pending = ix->cap_pending ? ix->cap_pending * 2 : 16;
                ix->pending = (struct cxo_pending *)realloc(ix->pending, ix->cap_pending * sizeof *ix->pending);
            }
            pm = (long)ix->n_pending++;
            memcpy(ix->pending[pm].id, id, 16);
        }
        ix->pending[pm].kind = strdup(kind);
        ix->pending[pm].agent = strdup(agent);
        return;
    }
    free(ix->rows[r].kind); free(ix->rows[r].agent);
    ix->rows[r].kind = strdup(kind);
    ix->rows[r].agent = strdup(agent);
}

/* ------------------------------------------------------------- arithmetic */

/* vector/index.rs:169-179: three sequential f32 sums, two sqrt, one divide */
float cxo_distance(const float *a, const float *b, size_t d) {
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (size_t i = 0; i < d; i++) { float p = a[i] * b[i]; dot = dot + p; }
    for (size_t i = 0; i < d; i++) { float p = a[i] * a[i]; na = na + p; }
    for (size_t i = 0; i < d; i++) { float p = b[i] * b[i]; nb = nb + p; }
    float norm_a = sqrtf(na);
    float norm_b = sqrtf(nb);
    float similarity = dot / (norm_a * norm_b);
    return 1.0f - similarity;
}

/* vector/index.rs:254-256: (1.0 - distance).clamp(0.0, 1.0); f32::clamp
 * returns NaN for NaN */
float cxo_distance_to_similarity(float distance) {
    float s = 1.0f - distance;
    if (s < 0.0f) s = 0.0f;
    if (s > 1.0f) s = 1.0f;
    return s;
}

/* ----------------------------------------------------------------- filter */

/* vector/index.rs:225-251 */
static int matches_filter(const cxo_index *ix, const cxo_row *row, const cxo_filter *f) {
    (void)ix;
    if (f->has_exclude) {
        for (size_t i = 0; i < f->n_exclude; i++)
            if (memcmp(f->exclude + 16 * i, row->id, 16) == 0) return 0;
    }
    if (row->kind) { /* only when metadata exists for this id (:234) */
        if (f->has_kinds) {
            int found = 0;
            for (size_t i = 0; i < f->n_kinds; i++)
                if (strcmp(f->kinds[i], row->kind) == 0) { found = 1; break; }
            if (!found) return 0;
        }
        if (f->has_agent) {
            if (strcmp(f->source_agent, row->agent) != 0) return 0;
        }
    }
    return 1;
}

/* ------------------------------------------------------------------- sort */

/* "a sorts strictly before b": score descending, NaN after every number.
 * Equal scores return 0 both ways, so the stable merge keeps row order. */
static int before(const cxo_result *a, const cxo_result *b) {
    int an = isnan(a->score), bn = isnan(b->score);
    if (an || bn) return (!an) && bn;
    return a->score > b->score;
}

static void merge_sort(cxo_result *v, cxo_result *tmp, size_t n) {
    if (n < 2) return;
    if (n <= 16) { /* insertion sort, stable */
        for (size_t i = 1; i < n; i++) {
            cxo_result x = v[i];
            size_t j = i;
            while (j > 0 && before(&x, &v[j - 1])) { v[j] = v[j - 1]; j--; }
            v[j] = x;
        }
        return;
    }
    size_t h = n / 2;
    merge_sort(v, tmp, h);
    merge_sort(v + h, tmp, n - h);
    memcpy(tmp, v, h * sizeof *v);
    size_t i = 0, j = h, o = 0;
    while (i < h && j < n) {
        if (before(&v[j], &tmp[i])) v[o++] = v[j++]; /* right wins only if strictly better */
        else v[o++] = tmp[i++];
    }
    while (i < h) v[o++] = tmp[i++];
}

/* ----------------------------------------------------------------- search */

/* vector/index.rs:259-294 brute_force_search.  Mirrors the reference's data
 * movement too: the query is cloned once (:265) and every row is cloned
 * before its distance (:270), every surviving row yields a result, the whole
 * list is stably sorted, then truncated. Returns a malloc'd array. */
static cxo_result *brute_force_all(const cxo_index *ix, const float *query,
                                   const cxo_filter *filter, size_t *n_out) {
    size_t d = ix->dim;
    float *q = (float *)malloc((d ? d : 1) * sizeof(float));
    memcpy(q, query, d * sizeof(float));
    cxo_result *res = (cxo_result *)malloc((ix->n_alive ? ix->n_alive : 1) * sizeof *res);
    size_t n = 0;
    for (size_t r = 0; r < ix->n_rows; r++) {
        const cxo_row *row = &ix->rows[r];
        if (!row->alive) continue;
        float *clone = (float *)malloc((d ? d : 1) * sizeof(float));
        memcpy(clone, row->vec, d * sizeof(float));
        float distance = cxo_distance(q, clone, d);
        free(clone);
        if (filter && !matches_filter(ix, row, filter)) continue;
        memcpy(res[n].node_id, row->id, 16);
        res[n].score = cxo_distance_to_similarity(distance);
        res[n].distance = distance;
        res[n].row = (uint32_t)r;
        n++;
    }
    free(q);
    cxo_result *tmp = (cxo_result *)malloc((n ? n : 1) * sizeof *tmp);
    merge_sort(res, tmp, n);
    free(tmp);
    *n_out = n;
    return res;
}

/* vector/index.rs:325-374 on the exact path: empty -> [], else brute force
 * (:338-340; the HNSW branch :342-371 is third-party and not restated). */
size_t cxo_search(const cxo_index *ix, const float *query, size_t k,
                  const cxo_filter *filter, cxo_result *out) {
    if (ix->n_alive == 0) return 0;
    size_t n;
    cxo_result *all = brute_force_all(ix, query, filter, &n);
    if (n > k) n = k; /* truncate(k) :292 */
    memcpy(out, all, n * sizeof *all);
    free(all);
    return n;
}

/* vector/index.rs:376-388: search(query, len.max(1)) then score >= threshold */
size_t cxo_search_threshold(const cxo_index *ix, const float *query, float threshold,
                            const cxo_filter *filter, cxo_result *out) {
    if (ix->n_alive == 0) return 0;
    size_t n;
    cxo_result *all = brute_force_all(ix, query, filter, &n);
    size_t o = 0;
    for (size_t i = 0; i < n; i++)
        if (all[i].score >= threshold) out[o++] = all[i]; /* NaN >= t is false */
    free(all);
    return o;
}

/* vector/index.rs:390-410 */
void cxo_search_batch(const cxo_index *ix, size_t nq, const float *queries, size_t k,
                      const cxo_filter *filter, int n_threads,
                      cxo_result *out, size_t *counts) {
    long i;
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
    for (i = 0; i < (long)nq; i++)
        counts[i] = cxo_search(ix, queries + (size_t)i * ix->dim, k, filter, out + (size_t)i * k);
}

/* ----------------------------------------------------------------- config */

void cxo_config_default(cxo_similarity_config *c) { /* vector/config.rs:24-33 */
    c->auto_link_threshold = 0.75f;
    c->dedup_threshold = 0.92f;
    c->contradiction_threshold = 0.80f;
    c->auto_link_k = 20;
}
float cxo_config_clamp(float t) { /* vector/config.rs:42-57 */
    if (t < 0.0f) t = 0.0f;
    if (t > 1.0f) t = 1.0f;
    return t;
}
int cxo_config_validate(const cxo_similarity_config *c) { /* vector/config.rs:66-87 */
    if (c->auto_link_threshold >= c->dedup_threshold) {
        snprintf(g_err, sizeof g_err, "auto_link_threshold must be less than dedup_threshold");
        return 1;
    }
    if (c->contradiction_threshold >= c->dedup_threshold) {
        snprintf(g_err, sizeof g_err, "contradiction_threshold must be less than dedup_threshold");
        return 2;
    }
    if (c->auto_link_k == 0) {
        snprintf(g_err, sizeof g_err, "auto_link_k must be greater than 0");
        return 3;
    }
    return 0;
}

/* ------------------------------------------------------------ auto-linker */

/* linker/auto_linker.rs:215-264 with linker/rules.rs:42-62 as the only rule, then the per-cycle
 * truncation of :284-287.  existing_offsets/existing_to: CSR over the scanned nodes (scan order) of the
 * rows each already has an outgoing related_to edge to — the `existing_set` of :226-231 restricted to the
 * one relation this rule proposes; such neighbours are skipped WITHOUT counting (:249-258) and the walk
 * goes on down the top-k list.  NULL = no edges yet. */
size_t cxo_autolink_pass(const cxo_index *ix, size_t n_scan, const uint32_t *scan_rows,
                         size_t topk, float threshold, size_t max_edges_per_node, size_t max_edges_per_cycle,
                         const uint8_t *deleted, const uint64_t *existing_offsets, const uint32_t *existing_to,
                         int n_threads, cxo_edge *out, size_t cap, size_t *n_needed) {
    /* the cap is tested after a neighbour's edges were pushed (:259-262): a node proposes at most
     * max(max_edges_per_node, 1) edges with this one rule */
    const size_t per_node = max_edges_per_node ? max_edges_per_node : 1;
    size_t *cnt = (size_t *)calloc(n_scan ? n_scan : 1, sizeof(size_t));
    cxo_edge *tmp = (cxo_edge *)malloc((n_scan ? n_scan : 1) * per_node * sizeof *tmp);
    long i;
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
    for (i = 0; i < (long)n_scan; i++) {
        uint32_t r = scan_rows[i];
        const float *emb = cxo_row_ptr(ix, r);
        if (!emb) continue;
        size_t kk = topk < ix->n_alive ? topk : ix->n_alive;
        cxo_result *similar = (cxo_result *)malloc((kk ? kk : 1) * sizeof *similar);
        size_t n = cxo_search(ix, emb, topk, NULL, similar);          /* :220-221 */
        size_t node_edge_count = 0;                                   /* :224 */
        const uint32_t *have = existing_offsets ? existing_to + existing_offsets[i] : NULL;   /* :226-231 */
        const size_t n_have = existing_offsets ? (size_t)(existing_offsets[i + 1] - existing_offsets[i]) : 0;
        for (size_t j = 0; j < n; j++) {
            if (similar[j].row == r) continue;                         /* :235-237 skip self */
            if (deleted && deleted[similar[j].row]) continue;          /* :240-243 */
            if (similar[j].score >= threshold) {                       /* rules.rs:50 */
                int exists = 0;                                        /* :253-254 existing_set.contains(&key) */
                for (size_t e = 0; e < n_have; e++)
                    if (have[e] == similar[j].row) { exists = 1; break; }
                if (!exists) {
                    cxo_edge *e = &tmp[(size_t)i * per_node + node_edge_count];
                    e->from_row = r; e->to_row = similar[j].row; e->weight = similar[j].score;
                    node_edge_count++;                                 /* :255-256 */
                }
            }
            if (node_edge_count >= max_edges_per_node) break;          /* :261-263 */
        }
        cnt[i] = node_edge_count;
        free(similar);
    }
    size_t total = 0;
    for (size_t s = 0; s < n_scan; s++)
        for (size_t j = 0; j < cnt[s]; j++) {
            if (total >= max_edges_per_cycle) break;                   /* :284-287 take(max_edges_per_cycle) */
            if (total < cap) out[total] = tmp[s * per_node + j];
            total++;
        }
    free(cnt); free(tmp);
    if (n_needed) *n_needed = total;
    return total < cap ? total : cap;
}

/* ------------------------------------------------------------------ dedup */

typedef struct { uint64_t *keys; size_t n_slots, n; } u64set;
static void set_init(u64set *s, size_t n_slots) {
    s->keys = (uint64_t *)calloc(n_slots, sizeof(uint64_t)); s->n_slots = n_slots; s->n = 0;
}
static int set_insert(u64set *s, uint64_t key) { /* returns 1 if newly inserted; key != 0 */
    if ((s->n + 1) * 2 > s->n_slots) {
        u64set b; set_init(&b, s->n_slots * 2);
        for (size_t i = 0; i < s->n_slots; i++) if (s->keys[i]) set_insert(&b, s->keys[i]);
        free(s->keys); *s = b;
    }
    size_t p = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 17) & (s->n_slots - 1);
    while (s->keys[p]) { if (s->keys[p] == key) return 0; p = (p + 1) & (s->n_slots - 1); }
    s->keys[p] = key; s->n++;
    return 1;
}

/* linker/dedup.rs:65-127.  Nodes are scanned in row order; the pair key is
 * the unordered pair (dedup.rs:93-97 orders by Uuid; any canonical unordered
 * key gives the same seen-set). */
size_t cxo_dedup_scan(const cxo_index *ix, float dedup_threshold, const uint8_t *deleted,
                      cxo_edge *out, size_t cap, size_t *n_needed) {
    u64set seen; set_init(&seen, 1024);
    cxo_result *buf = (cxo_result *)malloc((ix->n_alive ? ix->n_alive : 1) * sizeof *buf);
    size_t total = 0;
    for (size_t r = 0; r < ix->n_rows; r++) {
        if (!ix->rows[r].alive) continue;                /* no embedding in the index */
        if (deleted && deleted[r]) continue;             /* :72-74 */
        size_t n = cxo_search_threshold(ix, ix->rows[r].vec, dedup_threshold, NULL, buf); /* :85-87 */
        for (size_t j = 0; j < n; j++) {
            uint32_t o = buf[j].row;
            if (o == r) continue;                        /* :91-93 */
            uint64_t lo = o < r ? o : r, hi = o < r ? r : o;
            uint64_t key = ((hi + 1) << 32) | (lo + 1);
            if (!set_insert(&seen, key)) continue;       /* :99-102 */
            if (total < cap) { out[total].from_row = (uint32_t)r; out[total].to_row = o; out[total].weight = buf[j].score; }
            total++;
        }
    }
    free(buf); free(seen.keys);
    if (n_needed) *n_needed = total;
    return total < cap ? total : cap;
}
