/*
 * cortex_hip.h — C ABI of the MI355X-native similarity engine that replaces
 * cortex-core's vector layer (L2+L1 in SURVEY.md §1) behind the reference's
 * own seam, `trait VectorIndex` (crates/cortex-core/src/vector/index.rs:50-99).
 *
 * Every entry point names the reference interface it replaces (file:line,
 * relative to /root/reference/crates/cortex-core/src/).  The binding a
 * maintainer adds on the Rust side (`impl VectorIndex for HipIndex`) is shown
 * in INTEGRATION.md.
 *
 * Conventions
 *  - Plain C: pointers and sizes only, no C++/torch types.
 *  - Status return: 0 = ok, non-zero = CX_ERR_*; the message is read with
 *    cx_last_error() (thread-local) and maps to CortexError::Validation(msg)
 *    (error.rs:7-50), the only error kind the reference's vector layer
 *    produces.  Nothing aborts or unwinds across the boundary.
 *  - Ownership: inputs are borrowed for the call; outputs are written into
 *    caller-allocated buffers; the engine never returns memory to free.
 *  - Ids are the reference's NodeId = Uuid (types.rs:9): 16 raw bytes.
 *  - Threading mirrors `Send + Sync` under the callers' RwLock
 *    (RwLockVectorIndex, vector/index.rs:104-163): functions taking
 *    `const cx_index*` (&self) are re-entrant and may run concurrently from
 *    many threads; functions taking `cx_index*` (&mut self) need exclusive
 *    access, which the caller's write lock already provides.
 *  - Order of results where the reference leaves it open (HashMap iteration
 *    into a stable sort, vector/index.rs:266-292): score descending, ties by
 *    insertion row ascending, NaN scores last.
 *  - There is no CPU fallback: without a usable gfx950 device cx_create
 *    fails with CX_ERR_DEVICE.
 */
#ifndef CORTEX_HIP_H
#define CORTEX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CX_OK 0
#define CX_ERR_VALIDATION 1 /* bad argument, dimension mismatch */
#define CX_ERR_DEVICE 2     /* no device / HIP runtime failure */
#define CX_ERR_IO 3         /* save/load file errors */
#define CX_ERR_CAPACITY 4   /* caller buffer too small; see n_needed */

typedef struct cx_index cx_index;

/* VectorFilter (vector/index.rs:18-47).  has_* = Option::Some.  Kinds and
 * agents are interned strings (cx_intern). */
typedef struct cx_filter {
    int32_t has_exclude;
    uint64_t n_exclude;
    const uint8_t *exclude_ids; /* n_exclude * 16 bytes */
    int32_t has_kinds;
    uint64_t n_kinds;
    const uint32_t *kind_codes;
    int32_t has_agent;
    uint32_t agent_code;
} cx_filter;

/* thread-local message of the last failing call on this thread */
const char *cx_last_error(void);
/* number of visible HIP devices (0 when none; never fails) */
int cx_device_count(void);

/* ---- lifecycle ------------------------------------------------------- */

/* HnswIndex::new(dimension) / with_metadata — vector/index.rs:204-216.
 * device: HIP device ordinal this shard lives on.  NULL on failure. */
cx_index *cx_create(uint32_t dimension, int device);
/* The same with a storage dtype for the HBM row store (cx_create(dim, dtype, ...); BASELINE config 5 is a
 * bf16 store).  CX_DTYPE_F32: cx_create.  CX_DTYPE_BF16: every vector is rounded to bf16 (nearest even) ONCE, when it is
 * inserted, and kept as 2 bytes per element — half the HBM footprint and half the bytes of every scan; all arithmetic
 * stays f32 (products of bf16 values are exact in f32), so every result is what the reference's distance path
 * (vector/index.rs:169-179, :254-294) returns for the rounded vectors, and that is what cx_save writes.  Every entry point
 * of this header works on either kind of handle; cx_device_rows is NULL for a bf16 store. */
#define CX_DTYPE_F32 0
#define CX_DTYPE_BF16 1
cx_index *cx_create_ex(uint32_t dimension, int device, int dtype);
int cx_dtype(const cx_index *ix);   /* CX_DTYPE_*; -1 for NULL */
void cx_destroy(cx_index *ix);
/* pre-size the HBM row store (rows, not bytes); optional */
int cx_reserve(cx_index *ix, uint64_t rows);

/* ---- mutation: &mut self --------------------------------------------- */

/* VectorIndex::insert — vector/index.rs:298-314.  len != dimension ->
 * CX_ERR_VALIDATION "Embedding dimension mismatch: expected D, got L".
 * Upsert: an existing id keeps its row, the vector is replaced.  Visible to
 * the next search (the reference's exact-path semantics). */
int cx_upsert(cx_index *ix, const uint8_t id[16], const float *embedding, uint64_t len);
/* n inserts in one call: ids n*16 bytes, embeddings row-major n*len (host) */
int cx_upsert_batch(cx_index *ix, uint64_t n, const uint8_t *ids, const float *embeddings, uint64_t len);
/* same, embeddings already resident in HBM on the index's device */
int cx_upsert_batch_dev(cx_index *ix, uint64_t n, const uint8_t *ids, const float *d_embeddings, uint64_t len);
/* VectorIndex::remove — vector/index.rs:316-323; unknown id is not an error */
int cx_remove(cx_index *ix, const uint8_t id[16]);
/* HnswIndex::set_metadata — vector/index.rs:219-222 (kind, source_agent interned) */
int cx_set_metadata(cx_index *ix, const uint8_t id[16], uint32_t kind_code, uint32_t agent_code);
/* n set_metadata calls in one (ids n*16 bytes).  Like the reference's separate metadata map, metadata set for an id
 * that has no vector yet is kept and binds when the vector is inserted (vector/tests.rs:65-66 does exactly that);
 * cx_remove drops it (:318). */
int cx_set_metadata_batch(cx_index *ix, uint64_t n, const uint8_t *ids, const uint32_t *kind_codes,
                          const uint32_t *agent_codes);
/* string -> stable code for cx_set_metadata (NodeKind / agent names); adds the string if it is new (&mut self) */
uint32_t cx_intern(cx_index *ix, const char *utf8, uint64_t len);
/* &self: the code of a string for a cx_filter, 0 if it was never interned — no row carries code 0, so a kind or
 * agent nobody was tagged with matches no row that has metadata, as in matches_filter (vector/index.rs:225-251).
 * Safe to call concurrently with searches and with cx_intern. */
uint32_t cx_lookup(const cx_index *ix, const char *utf8, uint64_t len);
/* VectorIndex::rebuild — vector/index.rs:416-435.  The exact engine needs no
 * graph: this compacts removed rows out of HBM (order preserved). Never
 * required for correctness. */
int cx_rebuild(cx_index *ix);

/* VectorIndex::save / load — vector/index.rs:437-473, same file format: bincode of
 * (HashMap<Uuid, Vec<f32>>, HashMap<Uuid, NodeMetadata>, usize), so files written by
 * HnswIndex::save load here and vice versa.  Errors carry the reference's messages
 * ("Failed to write index file: ...", "Failed to read index file: ...",
 * "Failed to deserialize index: ...").  cx_load returns NULL on failure. */
int cx_save(const cx_index *ix, const char *path);
cx_index *cx_load(const char *path, int device);
cx_index *cx_load_ex(const char *path, int device, int dtype);   /* the file's f32 vectors into a store of that dtype (cx_create_ex) */

/* ---- bulk load from stored nodes ----------------------- */

/* One stored `Node` (types.rs:26-68) as the reference keeps it in its nodes table: bincode 1.3 with
 * `bincode::deserialize` options (little endian, fixed-width integers, u64 lengths, trailing bytes
 * allowed), layout pinned by the reference's golden bytes (storage/redb_storage.rs:1827-1857).
 * Pointers point INTO the record; strings are validated UTF-8, not NUL-terminated. */
typedef struct cx_node_view {
    uint8_t id[16];
    const char *kind;          uint64_t kind_len;   /* NodeKind */
    const char *title;         uint64_t title_len;  /* data.title */
    const char *body;          uint64_t body_len;   /* data.body */
    uint64_t n_tags;                                 /* data.tags.len() */
    const char *agent;         uint64_t agent_len;  /* source.agent */
    const uint8_t *embedding;  /* embedding_len f32 LE values, not necessarily 4-byte aligned; NULL = None */
    uint64_t embedding_len;
    int32_t has_embedding;
    float importance;
    uint64_t access_count;
    int64_t last_accessed_at_s; uint32_t last_accessed_at_ns; /* DateTime<Utc>: seconds since the epoch + ns */
    int64_t created_at_s;       uint32_t created_at_ns;
    int64_t updated_at_s;       uint32_t updated_at_ns;
    uint8_t deleted;
    uint64_t bytes_used;       /* bytes of the record the node occupies */
} cx_node_view;

/* RedbStorage::deserialize_node (storage/redb_storage.rs:230-232).  Host only — needs no device.
 * Failure = CX_ERR_VALIDATION "Failed to deserialize node: ..." (the record list_nodes would skip,
 * :709-712): truncated input, bad Option/bool byte, invalid UTF-8 or timestamp, and any record whose
 * data.metadata map is non-empty (bincode cannot deserialize serde_json::Value, so the reference
 * cannot read those either). */
int cx_node_decode(const uint8_t *record, uint64_t len, cx_node_view *out);

typedef struct cx_bulk_stats {
    uint64_t records;      /* given */
    uint64_t undecodable;  /* skipped like list_nodes skips corrupt records */
    uint64_t deleted;      /* tombstoned nodes the default NodeFilter hides (redb_storage.rs:345-349) */
    uint64_t no_embedding; /* embedding == None */
    uint64_t dim_mismatch; /* insert would fail: wrong length */
    uint64_t indexed;      /* successful inserts (the reference's `indexed`, serve.rs:108-114) */
} cx_bulk_stats;

#define CX_BULK_STRICT 1u          /* a dimension mismatch fails the load (Cortex::open, api.rs:59-63) instead
                                      of skipping the node (serve.rs:111-115) */
#define CX_BULK_INCLUDE_DELETED 2u /* NodeFilter::include_deleted() */
#define CX_BULK_SET_METADATA 4u    /* also set_metadata(id, kind, source.agent) per node (the reference does not) */
#define CX_BULK_KEEP_ORDER 8u      /* insert in the order given instead of list_nodes' newest-first order */
#define CX_BULK_SET_STATS 16u      /* also record kind / last_accessed_at / access_count per node for
                                      cx_search_decayed (cx_set_node_stats_batch) */

/* The start-up loop `for node in list_nodes(NodeFilter::new()) { if let Some(e) = &node.embedding
 * { index.insert(node.id, e) } }` (serve.rs:105-123, api.rs:56-70) over n raw table values:
 * record i = blob[offsets[i] .. offsets[i+1]).  Nodes are inserted newest `created_at` first, stable
 * (list_nodes' order, redb_storage.rs:727-728), so rows — and with them the tie order of searches —
 * come out as in the reference.  stats may be NULL.  The `rebuild()` that follows in the reference
 * is a no-op here. */
int cx_bulk_load_nodes(cx_index *ix, uint64_t n, const uint8_t *blob, const uint64_t *offsets,
                       uint32_t flags, cx_bulk_stats *stats);

/* ---- query-time score decay and re-rank ----------------- */

/* ScoreDecayConfig (vector/scoring.rs:22-78); by_kind keys are interned NodeKind codes (cx_intern).
 * Defaults of the reference: enabled, daily_rate 0.02, max_age_days 365, min_factor 0.1,
 * echo_weight 0.05, echo_cap 2.0, recency_weight 0.15, by_kind {event .05, observation .04,
 * decision .005, pattern .005, fact .01, preference .005}. */
typedef struct cx_decay_config {
    int32_t enabled;
    double daily_rate, max_age_days, min_factor, echo_weight, echo_cap;
    float recency_weight;
    uint32_t n_by_kind;
    const uint32_t *kind_codes; /* n_by_kind */
    const double *kind_rates;   /* n_by_kind */
} cx_decay_config;

/* What apply_score_decay reads of a node — kind, last_accessed_at, access_count (types.rs:31,50,57) —
 * for n ids (n*16 bytes); kept per row on the host.  Ids without a vector are ignored; rows never
 * described read as last_accessed_at = the epoch, access_count 0, kind 0.  last_accessed_ns may be NULL. */
int cx_set_node_stats_batch(cx_index *ix, uint64_t n, const uint8_t *ids, const uint32_t *kind_codes,
                            const int64_t *last_accessed_s, const uint32_t *last_accessed_ns,
                            const uint64_t *access_counts);

/* apply_score_decay — vector/scoring.rs:84-114, with `now` passed in (the reference reads Utc::now()).
 * Host only.  f64 for the two factors, then f32 left to right without FMA, like the reference. */
float cx_apply_score_decay(const cx_decay_config *cfg, float raw_score, float recency_bias, int64_t now_s,
                           uint32_t now_ns, uint32_t kind_code, int64_t last_accessed_s,
                           uint32_t last_accessed_ns, uint64_t access_count);

/* The HTTP search handler's sequence (cortex-server/src/http/routes.rs:889-947): search(query,
 * candidate_limit, filter) -> final = apply_score_decay(node, raw, cfg, recency_bias) per candidate ->
 * stable sort by final descending (NaN compares equal) -> truncate(limit).  The handler uses
 * candidate_limit = max(3 limit, 30) when cfg.enabled && recency_bias > 0, else limit (:899-903);
 * a smaller value than limit is raised to limit.  Writes n_out <= limit ids, decayed scores and
 * the raw scores ("score" / "raw_score" of the response).  &self: re-entrant. */
int cx_search_decayed(const cx_index *ix, const float *query, uint64_t len, uint64_t limit,
                      uint64_t candidate_limit, const cx_filter *filter, const cx_decay_config *cfg,
                      float recency_bias, int64_t now_s, uint32_t now_ns, uint8_t *out_ids,
                      float *out_scores, float *out_raw_scores, uint64_t *n_out);

/* ---- queries: &self --------------------------------------------------- */

/* VectorIndex::len — vector/index.rs:412-414 */
uint64_t cx_len(const cx_index *ix);
uint32_t cx_dimension(const cx_index *ix);
/* rows resident in HBM including removed-but-not-compacted ones */
uint64_t cx_row_count(const cx_index *ix);
/* id of a row (for callers that work with row indices); 0 ok */
int cx_row_id(const cx_index *ix, uint64_t row, uint8_t out_id[16]);
/* out_alive[i] = 1 if row row_lo + i holds a vector, 0 if it was removed (a tombstone until cx_rebuild compacts).  The
 * linker passes skip removed rows when they scan (auto_linker.rs:217-218: a node without an embedding) and
 * cx_topk_lists_rows returns an empty list for them.  0 ok. */
int cx_rows_alive(const cx_index *ix, uint64_t row_lo, uint64_t n, uint8_t *out_alive);
/* rows of n ids (the linker passes take row indices): out_rows[i] = row of ids[16 i ..], or UINT32_MAX if the id
 * is not in the index (a node without an embedding: auto_linker.rs:217-218 skips it).  0 ok. */
int cx_rows_of(const cx_index *ix, uint64_t n, const uint8_t *ids, uint32_t *out_rows);

/* VectorIndex::search — vector/index.rs:325-374 on its exact path
 * (:338-340 -> brute_force_search :259-294).  Writes n_out <= k results,
 * best first: out_ids n*16 bytes, out_scores = clamp(1-distance,0,1),
 * out_distances = 1 - cosine.  An empty index yields n_out = 0 (:331-333).
 * len is the query length; like the reference (zip, :172) a query longer or
 * shorter than the dimension is not an error. */
int cx_search(const cx_index *ix, const float *query, uint64_t len, uint64_t k,
              const cx_filter *filter, uint8_t *out_ids, float *out_scores,
              float *out_distances, uint64_t *n_out);

/* VectorIndex::search_threshold — vector/index.rs:376-388: every row with
 * score >= threshold, best first.  Writes min(cap, n) results; *n_needed = n.
 * Returns CX_ERR_CAPACITY when n > cap (call again with a larger buffer). */
int cx_search_threshold(const cx_index *ix, const float *query, uint64_t len, float threshold,
                        const cx_filter *filter, uint64_t cap, uint8_t *out_ids,
                        float *out_scores, float *out_distances, uint64_t *n_out,
                        uint64_t *n_needed);

/* VectorIndex::search_batch — vector/index.rs:390-410.  queries row-major
 * nq*len; out arrays hold nq*k entries, query i's results start at i*k and
 * out_counts[i] of them are valid.  The corpus is read once per batch. */
int cx_search_batch(const cx_index *ix, uint64_t nq, const float *queries, uint64_t len, uint64_t k,
                    const cx_filter *filter, uint8_t *out_ids, float *out_scores,
                    float *out_distances, uint64_t *out_counts);

/* ---- the auto-linker's similarity pass, batched -------------------------- */

/* AutoLinker::run_cycle's per-node kNN loop (linker/auto_linker.rs:215-264) with
 * SimilarityLinkRule (linker/rules.rs:42-62) as the only rule, for n_scan nodes
 * in one pass: for each scanned node, in scan order, take its `topk` nearest
 * rows in score order (search(emb, topk), self included in the ranks), skip
 * self, skip rows flagged in `deleted` (nodes the storage has tombstoned but
 * that are still indexed), propose an edge for score >=
 * threshold with weight = score UNLESS the node already has that edge
 * (`existing_set`, :226-231: dropped without counting, :249-258, and the walk
 * goes on down the list), stop once max_edges_per_node were proposed (:261-263;
 * the test follows the push, so 0 still lets a node's first neighbour through,
 * as in the reference); of all proposals, in scan order, the first
 * max_edges_per_cycle are kept (:284-287; UINT64_MAX = no truncation).
 * scan_rows = row index of each scanned node (NULL = every row, n_scan
 * ignored); deleted = cx_row_count() flags or NULL.  existing_offsets
 * [n_scan + 1] / existing_to: CSR over the scanned nodes, in scan order, of
 * the rows each already has an outgoing related_to edge to (cx_rows_of maps
 * ids; targets without a row can be left out); NULL = no edges yet.  Rows are
 * insertion rows (cx_row_id maps them to ids).  Edges come out in scan order,
 * then score order.  Writes min(cap, n) edges, *n_needed = n; CX_ERR_CAPACITY
 * if n > cap.  topk <= 256.  Structural / config rules stay on the host: they
 * need the ordered neighbour lists only (cx_topk_lists_rows). */
int cx_autolink_pass_rows(const cx_index *ix, uint64_t n_scan, const uint32_t *scan_rows, uint64_t topk,
                          float threshold, uint64_t max_edges_per_node, uint64_t max_edges_per_cycle,
                          const uint8_t *deleted, const uint64_t *existing_offsets,
                          const uint32_t *existing_to, uint64_t cap, uint32_t *out_from, uint32_t *out_to,
                          float *out_weight, uint64_t *n_out, uint64_t *n_needed);

/* The auto-linker's neighbour query for a set of nodes at once (auto_linker.rs:221: `search(&emb, 100, None)`
 * per scanned node; SURVEY §8 a14': "ordered top-100 (j, score) per i" is the engine's contract, every rule —
 * structural and config rules included — walks these lists on the host in reference order).  For each scanned
 * row, in scan order: its `topk` nearest rows, best first, self included (the reference's walk skips it),
 * removed rows excluded, no filter.  scan_rows = NULL: every row.  out_rows / out_scores: [n_scan][topk],
 * out_counts: [n_scan] (= min(topk, live rows)).  One batched search per <= 16384 scanned rows (cx_search_batch's
 * engine: rows are read once per 32 or 64 queries). */
int cx_topk_lists_rows(const cx_index *ix, uint64_t n_scan, const uint32_t *scan_rows, uint64_t topk,
                       uint32_t *out_rows, float *out_scores, uint32_t *out_counts);

/* DedupScanner::scan's pair emission (linker/dedup.rs:65-127): every indexed,
 * non-deleted node in row order, its neighbours with score >= dedup_threshold
 * in score order, self skipped, each unordered pair reported once by the node
 * scanned first.  out_a = scanned row, out_b = other row. */
int cx_dedup_scan_rows(const cx_index *ix, float dedup_threshold, const uint8_t *deleted, uint64_t cap,
                       uint32_t *out_a, uint32_t *out_b, float *out_similarity, uint64_t *n_out,
                       uint64_t *n_needed);

/* cx_autolink_pass_rows without the copy-out: edges stay in HBM, only their
 * number and the device time of the four phases (ms: shadow refresh, MFMA
 * filter, exact rescore, link rules) are returned.  bench.py's auto-link leg. */
int cx_autolink_pass_timed(const cx_index *ix, uint64_t n_scan, const uint32_t *scan_rows, uint64_t topk,
                           float threshold, uint64_t max_edges_per_node, uint64_t max_edges_per_cycle,
                           const uint64_t *existing_offsets, const uint32_t *existing_to,
                           uint64_t *n_edges, double *phase_ms);

/* Multi-GPU building block of the all-pairs pass: the ordered neighbour lists of nq
 * EXTERNAL vectors (d_queries: nq x dimension f32 in HBM, e.g. a block of another shard's rows) against this
 * shard: per query the rows with score >= threshold, best first, at most topk (<= 256).  Outputs in HBM:
 * d_out_rows / d_out_scores / d_out_dists [nq][topk] (local row indices), d_out_counts [nq] — the layout
 * cx_merge_topk_dev folds after the all-gather.  Lists that overflow the internal candidate cap come from
 * the exact scan path and are not thresholded (the rule walk applies the threshold).  Synchronous. */
int cx_autolink_lists_dev(const cx_index *ix, uint64_t nq, const float *d_queries, uint64_t topk, float threshold,
                          uint32_t *d_out_rows, float *d_out_scores, float *d_out_dists,
                          uint32_t *d_out_counts, void *stream);

/* Copies rows [row_lo, row_lo + n) of the shard into a caller buffer in HBM (to broadcast a scanned block). */
int cx_copy_rows_dev(const cx_index *ix, uint64_t row_lo, uint64_t n, float *d_dst, void *stream);

/* ---- HBM-resident variants (multi-GPU shards, benchmarking) ----------- */

/* As cx_search / cx_search_batch with the queries (dimension floats each)
 * and the results in HBM on the index's device.  d_rows are local row
 * indices; d_counts[i] valid entries per query.  Enqueued on `stream` (a
 * hipStream_t; NULL = the legacy default stream); returns without waiting. */
int cx_search_dev(const cx_index *ix, const float *d_query, uint64_t k, const cx_filter *filter,
                  uint32_t *d_rows, float *d_scores, float *d_distances, uint32_t *d_count,
                  void *stream);
int cx_search_batch_dev(const cx_index *ix, uint64_t nq, const float *d_queries, uint64_t k,
                        const cx_filter *filter, uint32_t *d_rows, float *d_scores,
                        float *d_distances, uint32_t *d_counts, void *stream);

/* A caller that runs a STREAM of cx_search_batch_dev calls of nq queries each (a rank of the sharded search: vector/index.rs:397-403
 * per shard, every batch followed by its exchange and merge) gains by rotating over several HIP streams: a batch's head, tail,
 * re-score and selection run under its neighbours' streaming.  How many is worth it depends on what a call of this size runs on
 * this index as it stands: 4 for passes of 64 queries, 2 for passes of 128 (row widths up to 512: two of those in each other's
 * way lose), 1 (no rotation) for single queries and for stores too small for the screening pass.  Measured:
 * profiles/r04/tuning.md 1.10.  Never fails (NULL index: 1). */
uint32_t cx_search_batch_streams_hint(const cx_index *ix, uint64_t nq);

/* Merge per-shard partial top-k lists after the all-gather.
 * Shard p's lists live at d_rows/d_scores/d_distances + p*part_stride, each
 * [nq][k], and its counts at d_counts + p*part_stride, [nq] (part_stride in
 * 4-byte elements: the size of one rank's packed all-gather chunk; 0 = dense
 * arrays [n_parts][nq][k] and [n_parts][nq]).  Shard p's rows are offset by
 * part_base[p] (host array) so ties resolve by global row.  Outputs [nq][k]
 * global rows (u64), scores, distances, and counts [nq]. */
int cx_merge_topk_dev(int device, uint64_t n_parts, uint64_t nq, uint64_t k, uint64_t part_stride,
                      const uint64_t *part_base, const uint32_t *d_rows, const float *d_scores,
                      const float *d_distances, const uint32_t *d_counts, uint64_t *d_out_rows,
                      float *d_out_scores, float *d_out_distances, uint32_t *d_out_counts,
                      void *stream);

/* ---- one index over several GPUs of the node, behind the same boundary ------------------
 *
 * `cx_sharded` is the multi-device form of `cx_index` for a host that is ONE process (the reference's shape:
 * serve.rs:101 holds one Arc<RwLock<index>>): one shard (a cx_index) per entry of device_ids, one stream per
 * shard, nothing for the caller to orchestrate.  New ids are placed block-round-robin (4096 consecutive new ids per
 * shard, SURVEY §8e "append-only ingest round-robins by block"), a known id keeps its shard and row (HashMap insert
 * semantics, vector/index.rs:307).  Every row carries its GLOBAL insertion sequence number ("global row"): result
 * order, tie order (score desc, global row asc) and every row-indexed interface below are those of ONE index that
 * saw the same sequence of calls — the sharded index is checked against exactly that.
 * A search runs the shard scans concurrently; each shard publishes its partial top-k straight into the gather
 * buffer on the first shard's device (peer-to-peer writes over xGMI: single hop, no ring — the payload is KBs and
 * latency-bound), the root merges the parts (merge_parts_kernel) and hands the result to the host.
 * A device may be listed more than once (several shards on one GPU; how the tests run on a one-GPU box).
 * Threading: as for cx_index — `const cx_sharded*` entry points are re-entrant, the others need exclusivity. */
typedef struct cx_sharded cx_sharded;

cx_sharded *cx_sharded_create(uint32_t dimension, uint32_t n_shards, const int *device_ids);   /* NULL on failure */
cx_sharded *cx_sharded_create_ex(uint32_t dimension, uint32_t n_shards, const int *device_ids, int dtype);   /* dtype: cx_create_ex */
void cx_sharded_destroy(cx_sharded *h);
uint32_t cx_sharded_n_shards(const cx_sharded *h);
/* shard i as a plain index (read-only uses: cx_len, cx_row_count, cx_device_rows ...) */
const cx_index *cx_sharded_shard(const cx_sharded *h, uint32_t i);
/* 1 if every shard can write the root's memory directly (xGMI / same device), 0 if parts travel through pinned host memory */
int cx_sharded_peer_to_peer(const cx_sharded *h);

/* VectorIndex::insert / remove / set_metadata / len / rebuild (vector/index.rs:298-323, :219-222, :412-435) */
int cx_sharded_upsert(cx_sharded *h, const uint8_t id[16], const float *embedding, uint64_t len);
int cx_sharded_upsert_batch(cx_sharded *h, uint64_t n, const uint8_t *ids, const float *embeddings, uint64_t len);
/* same, embeddings resident in HBM of any device of the node (copied device to device into the owning shards) */
int cx_sharded_upsert_batch_dev(cx_sharded *h, uint64_t n, const uint8_t *ids, const float *d_embeddings, uint64_t len);
int cx_sharded_remove(cx_sharded *h, const uint8_t id[16]);
int cx_sharded_set_metadata(cx_sharded *h, const uint8_t id[16], uint32_t kind_code, uint32_t agent_code);
uint32_t cx_sharded_intern(cx_sharded *h, const char *utf8, uint64_t len);          /* same code on every shard */
uint32_t cx_sharded_lookup(const cx_sharded *h, const char *utf8, uint64_t len);
uint64_t cx_sharded_len(const cx_sharded *h);
uint32_t cx_sharded_dimension(const cx_sharded *h);
/* global rows ever assigned (removed ones included until cx_sharded_rebuild renumbers) */
uint64_t cx_sharded_row_count(const cx_sharded *h);
int cx_sharded_row_id(const cx_sharded *h, uint64_t global_row, uint8_t out_id[16]);
/* out_rows[i] = global row of ids[16 i ..] or UINT32_MAX */
int cx_sharded_rows_of(const cx_sharded *h, uint64_t n, const uint8_t *ids, uint32_t *out_rows);
/* compacts every shard and renumbers the global rows (order preserved) */
int cx_sharded_rebuild(cx_sharded *h);
/* VectorIndex::save / load (vector/index.rs:437-473) for the sharded handle: ONE bincode file in the reference's layout,
 * entries in global row order — the file cx_save writes for a single index over the same calls, so either kind of index
 * loads what the other saved (and what the reference saved).  cx_sharded_load_ex places the file's vectors like
 * cx_sharded_upsert_batch would (dtype: CX_DTYPE_F32 / CX_DTYPE_BF16 row stores). */
int cx_sharded_save(const cx_sharded *h, const char *path);
cx_sharded *cx_sharded_load_ex(const char *path, uint32_t n_shards, const int *device_ids, int dtype);

/* VectorIndex::search / search_batch / search_threshold over all shards — same contracts as cx_search,
 * cx_search_batch, cx_search_threshold */
int cx_sharded_search(const cx_sharded *h, const float *query, uint64_t len, uint64_t k, const cx_filter *filter,
                      uint8_t *out_ids, float *out_scores, float *out_distances, uint64_t *n_out);
int cx_sharded_search_batch(const cx_sharded *h, uint64_t nq, const float *queries, uint64_t len, uint64_t k,
                            const cx_filter *filter, uint8_t *out_ids, float *out_scores, float *out_distances,
                            uint64_t *out_counts);
int cx_sharded_search_threshold(const cx_sharded *h, const float *query, uint64_t len, float threshold,
                                const cx_filter *filter, uint64_t cap, uint8_t *out_ids, float *out_scores,
                                float *out_distances, uint64_t *n_out, uint64_t *n_needed);

/* cx_autolink_pass_rows / cx_dedup_scan_rows over all shards; every row argument and result is a GLOBAL row
 * (scan_rows, deleted[cx_sharded_row_count()], existing_to, out_from / out_to).  Per block of scanned nodes the
 * owners scatter the nodes' vectors into every shard's query block (peer writes), every shard produces the block's
 * ordered neighbour lists against its own rows (cx_autolink_lists_dev: MFMA filter + exact rescore), the lists are
 * published to the root and merged like partial top-k lists, and the root walks the reference's rules. */
int cx_sharded_autolink_pass_rows(const cx_sharded *h, uint64_t n_scan, const uint32_t *scan_rows, uint64_t topk,
                                  float threshold, uint64_t max_edges_per_node, uint64_t max_edges_per_cycle,
                                  const uint8_t *deleted, const uint64_t *existing_offsets,
                                  const uint32_t *existing_to, uint64_t cap, uint32_t *out_from, uint32_t *out_to,
                                  float *out_weight, uint64_t *n_out, uint64_t *n_needed);
int cx_sharded_dedup_scan_rows(const cx_sharded *h, float dedup_threshold, const uint8_t *deleted, uint64_t cap,
                               uint32_t *out_a, uint32_t *out_b, float *out_similarity, uint64_t *n_out,
                               uint64_t *n_needed);
/* cx_topk_lists_rows over all shards (the linker's `search(&emb, topk, None)` per scanned node when the reference's
 * structural rules are on, SURVEY a14'): global rows in, global rows out; every shard runs its batched search over
 * each block of scanned vectors, the lists are merged on the root. */
int cx_sharded_topk_lists_rows(const cx_sharded *h, uint64_t n_scan, const uint32_t *scan_rows, uint64_t topk,
                               uint32_t *out_rows, float *out_scores, uint32_t *out_counts);
/* cx_set_metadata_batch / cx_bulk_load_nodes / cx_set_node_stats_batch / cx_search_decayed, sharded: the start-up
 * load from the nodes table (serve.rs:105-123) and the HTTP handler's decayed search (routes.rs:889-947) work on the
 * multi-GPU index unchanged; node stats are kept per global row. */
int cx_sharded_set_metadata_batch(cx_sharded *h, uint64_t n, const uint8_t *ids, const uint32_t *kind_codes,
                                  const uint32_t *agent_codes);
int cx_sharded_bulk_load_nodes(cx_sharded *h, uint64_t n, const uint8_t *blob, const uint64_t *offsets, uint32_t flags,
                               cx_bulk_stats *stats);
int cx_sharded_set_node_stats_batch(cx_sharded *h, uint64_t n, const uint8_t *ids, const uint32_t *kind_codes,
                                    const int64_t *last_accessed_s, const uint32_t *last_accessed_ns,
                                    const uint64_t *access_counts);
int cx_sharded_search_decayed(const cx_sharded *h, const float *query, uint64_t len, uint64_t limit,
                              uint64_t candidate_limit, const cx_filter *filter, const cx_decay_config *cfg,
                              float recency_bias, int64_t now_s, uint32_t now_ns, uint8_t *out_ids,
                              float *out_scores, float *out_raw_scores, uint64_t *n_out);

/* ---- measurement and diagnostics -------------------------------------- */

/* The check every host entry point applies to a result block it read back from the device before using any of it
 * as an index or a length ("nothing aborts across the boundary" also when a kernel misbehaves): counts[i] <= k_max
 * and rows[i * stride + j] < n_rows for j < counts[i], else CX_ERR_DEVICE with a message.  Exported so the guard
 * itself can be tested without a GPU. */
int cx_debug_check_result_block(const uint32_t *counts, const uint32_t *rows, uint64_t nq, uint64_t stride,
                                uint64_t k_max, uint64_t n_rows);

/* When on, every launch of the dominant scan kernel is bracketed by HIP
 * events on the stream it runs on (bench.py's roofline leg). */
int cx_profile_enable(cx_index *ix, int on);
/* Sum of the bracketed kernel durations (ms) and their count since the last
 * reset; waits for the recorded events. */
int cx_profile_read(cx_index *ix, double *kernel_ms_sum, uint64_t *launches, int reset);

/* The MFMA filter GEMM of the most recent TIMED all-pairs pass on this index (cx_autolink_pass_timed), for bench.py's
 * MFMA roofline: out[0] = duration of the GEMM kernel alone in ms (HIP events around that one launch, on its stream),
 * out[1] = flops its MFMAs executed (2 x 256 x 256 x dim per tile launched: symmetric passes launch tiles tj >= ti only),
 * out[2] = tiles, out[3] = which kernel ran (0 pair_filter256_kernel, 1 pair_filter_p_kernel, 2 the 128-tile / stream
 * kernels: out[1] = 0 then), out[4] = shader clock in GHz a block of the persistent kernel measured over its life
 * (s_memtime against the 100 MHz reference; 0 if not measured).  All zero before the first timed pass. */
int cx_autolink_filter_profile(const cx_index *ix, double out[5]);

/* raw device pointer to the f32 row store (row-major, cx_dimension floats per
 * row) — read-only view for tools and tests */
const float *cx_device_rows(const cx_index *ix);

#ifdef __cplusplus
}
#endif
#endif /* CORTEX_HIP_H */
