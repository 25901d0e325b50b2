/*
 * cortex_oracle.h — CPU restatement of the reference's exact similarity path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cortex_amd/ (the product) may
 * include, link, import or execute anything from oracle/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there
 * only as the checker / reported baseline, never as the thing shipped.
 *
 * Pinning status: the reference (Rust, crates/cortex-core) cannot be compiled
 * in this image (no cargo/rustc, dependencies not vendored), and it holds no
 * numeric golden vectors for this path.  The restatement is pinned by the
 * reference's own known-answer tests (vector/index.rs:484-566, :579-728,
 * vector/config.rs:93-135, linker/rules.rs:403-421), re-expressed in
 * tests/test_oracle_reference_kat.py, and cross-checked by an independent
 * numpy restatement (oracle/np_twin.py).  HNSW (instant-distance 0.6.1,
 * third-party, not in /root/reference) result sets: parity unpinned.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/crates/cortex-core/src/).
 */
#ifndef CORTEX_ORACLE_H
#define CORTEX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SimilarityResult — vector/index.rs:11-15 */
typedef struct cxo_result {
    uint8_t node_id[16];
    float score;    /* clamp(1 - distance, 0, 1); NaN stays NaN */
    float distance; /* 1 - cosine, unclamped */
    uint32_t row;   /* oracle-only: insertion row of the hit (declared tie order) */
} cxo_result;

/* VectorFilter — vector/index.rs:18-47.  has_* model Option<..>::Some */
typedef struct cxo_filter {
    int has_kinds;
    size_t n_kinds;
    const char *const *kinds;
    int has_exclude;
    size_t n_exclude;
    const uint8_t *exclude; /* n_exclude * 16 bytes */
    int has_agent;
    const char *source_agent;
} cxo_filter;

typedef struct cxo_index cxo_index;

/* EmbeddingPoint::distance — vector/index.rs:169-179 */
float cxo_distance(const float *a, const float *b, size_t d);
/* HnswIndex::distance_to_similarity — vector/index.rs:254-256 */
float cxo_distance_to_similarity(float distance);

/* HnswIndex::new — vector/index.rs:204-211 */
cxo_index *cxo_index_new(size_t dimension);
void cxo_index_free(cxo_index *ix);
/* insert — vector/index.rs:298-314; returns 0, or -1 (message via cxo_last_error) */
int cxo_insert(cxo_index *ix, const uint8_t id[16], const float *emb, size_t len);
/* bulk helper: n rows, ids n*16 bytes, row-major embeddings */
int cxo_insert_batch(cxo_index *ix, size_t n, const uint8_t *ids, const float *embs, size_t len);
/* remove — vector/index.rs:316-323 */
int cxo_remove(cxo_index *ix, const uint8_t id[16]);
/* set_metadata — vector/index.rs:219-222 */
void cxo_set_metadata(cxo_index *ix, const uint8_t id[16], const char *kind, const char *agent);
/* len — vector/index.rs:412-414 */
size_t cxo_len(const cxo_index *ix);
size_t cxo_dimension(const cxo_index *ix);

/* search — vector/index.rs:325-374 on the exact path (:338-340 -> :259-294).
 * Writes at most k results into out (caller provides room for min(k, len)),
 * returns the number written. */
size_t cxo_search(const cxo_index *ix, const float *query, size_t k,
                  const cxo_filter *filter, cxo_result *out);
/* search_threshold — vector/index.rs:376-388.  out must hold cxo_len() entries. */
size_t cxo_search_threshold(const cxo_index *ix, const float *query, float threshold,
                            const cxo_filter *filter, cxo_result *out);
/* search_batch — vector/index.rs:390-410 (rayon par_iter over queries; here
 * OpenMP over queries with n_threads workers, 1 = sequential).
 * out: nq * k entries, counts: nq entries. */
void cxo_search_batch(const cxo_index *ix, size_t nq, const float *queries, size_t k,
                      const cxo_filter *filter, int n_threads,
                      cxo_result *out, size_t *counts);

/* SimilarityConfig — vector/config.rs:3-87 */
typedef struct cxo_similarity_config {
    float auto_link_threshold;
    float dedup_threshold;
    float contradiction_threshold;
    size_t auto_link_k;
} cxo_similarity_config;
void cxo_config_default(cxo_similarity_config *c);              /* :24-33 */
float cxo_config_clamp(float threshold);                        /* :42-57 */
int cxo_config_validate(const cxo_similarity_config *c);        /* :66-87; 0 ok, else 1/2/3 = which rule failed */

/* Proposed edge from SimilarityLinkRule — linker/rules.rs:42-62 */
typedef struct cxo_edge {
    uint32_t from_row;
    uint32_t to_row;
    float weight; /* = score */
} cxo_edge;

/* The auto-linker's per-node kNN loop — linker/auto_linker.rs:215-264 with
 * only SimilarityLinkRule active (legacy structural rules off, no config
 * rules): search(emb, topk=100), skip self, skip deleted neighbours, keep
 * score >= threshold, drop edges that already exist WITHOUT counting them
 * (:226-231, :249-258), count, stop once max_edges_per_node reached; then
 * the first max_edges_per_cycle proposals of the cycle (:284-287).
 * rows: the scanned nodes, in scan order.  deleted: optional per-row flags
 * (nodes tombstoned in storage but still indexed, quirk Q2).
 * existing_offsets [n_scan+1] / existing_to: CSR, per scanned node the rows it
 * already has a related_to edge to (NULL = none).
 * Returns the number of edges written (at most cap; *n_needed gets the total). */
size_t cxo_autolink_pass(const cxo_index *ix, size_t n_scan, const uint32_t *scan_rows,
                         size_t topk, float threshold, size_t max_edges_per_node, size_t max_edges_per_cycle,
                         const uint8_t *deleted, const uint64_t *existing_offsets, const uint32_t *existing_to,
                         int n_threads, cxo_edge *out, size_t cap, size_t *n_needed);

/* DedupScanner::scan pair emission — linker/dedup.rs:65-127: per live row,
 * search_threshold(dedup_threshold), skip self, canonical unordered pair,
 * first-seen wins.  Emits (a=scanned row, b=other row, similarity). */
size_t cxo_dedup_scan(const cxo_index *ix, float dedup_threshold, const uint8_t *deleted,
                      cxo_edge *out, size_t cap, size_t *n_needed);

/* raw access for tests / baselines */
const float *cxo_row_ptr(const cxo_index *ix, uint32_t row);
size_t cxo_row_count(const cxo_index *ix); /* rows ever allocated, incl. removed */
const char *cxo_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
