// kernels.hpp — host-callable launchers of the HIP kernels (internal API).
#pragma once

#include <vector>

#include "common.hpp"

namespace cx {

constexpr uint32_t TOPK_MAX = 256;  // largest k served by the in-register top-k lists

struct MergeArgs {
    const uint64_t *part_keys;  // [n_lists][k]
    const float *part_sims;
    uint32_t n_lists;
    uint32_t k;
    uint32_t *out_rows;   // [k]
    float *out_scores;    // [k]
    float *out_dists;     // [k]
    uint32_t *out_count;  // [1]
    const uint32_t *run_if = nullptr;   // non-null: the launch does nothing unless *run_if != 0
    const uint32_t *seg_counts = nullptr;   // radix merge of unsorted lists only: [nq][n_lists * k / seg_len] valid entries per segment
    uint32_t seg_len = 0;
    // merge_small_kernel only (merge_batch_writes_bound): also write score_ord of query q's k-th result to bound_out[q] (0 = fewer
    // than k results: no bound) and clear *clear_word — batchg's bound and overflow flag without two more stream operations
    uint32_t *bound_out = nullptr;
    uint32_t *clear_word = nullptr;
};
bool merge_batch_writes_bound(uint32_t k, uint32_t n_lists);

// One single-query scan over the row store.
struct ScanArgs {
    const float *rows;     // [n_rows][dim] f32 row-major, 16-byte aligned when dim % 4 == 0; null for a bf16 store
    const uint16_t *rows16 = nullptr;   // bf16 store (cx_create_ex): [n_rows][dim] bf16 row-major
    const float *query;    // [dim] f32 in HBM
    float q_tail_sumsq;    // sum of squares of query elements beyond dim (host queries longer than dim)
    uint32_t n_rows;
    uint32_t dim;
    uint32_t k;
    DevFilter flt;
    // MODE 0: per-block partial lists [grid][k]
    uint64_t *part_keys;
    float *part_sims;
    // MODE 1: dense per-row keys/sims [n_rows] (0 = filtered out / below threshold)
    uint64_t *dense_keys;
    float *dense_sims;
    float threshold;
    uint32_t has_threshold;
};

// number of blocks launch_scan_topk will use for this shape (scratch sizing)
uint32_t scan_grid_blocks(uint32_t n_rows, uint32_t dim, bool rows16 = false);
// k <= TOPK_MAX.  scan + merge, results sorted best-first.
// ev0/ev1 (optional) are recorded on the stream right before / after the scan kernel.
int launch_scan_topk(const ScanArgs &a, const MergeArgs &m, bool nontemporal, hipStream_t stream,
                     hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// dense keys for the sort path (large k, threshold search)
int launch_scan_dense(const ScanArgs &a, bool nontemporal, hipStream_t stream);

// sort path: sorts n (key, sim) pairs best-first, counts non-empty keys, and
// writes the first min(count, k) as rows/scores/dists.
size_t sort_temp_bytes(uint32_t n);
int launch_sort_select(uint64_t *keys_in, float *sims_in, uint64_t *keys_tmp, float *sims_tmp, uint32_t n,
                       uint32_t k, void *temp, size_t temp_bytes, uint32_t *out_rows, float *out_scores,
                       float *out_dists, uint32_t *out_count, hipStream_t stream);

// cross-shard merge of [n_parts][nq][k] partial lists
constexpr uint32_t MAX_PARTS = 64;
struct PartBase { uint64_t base[MAX_PARTS]; };  // by-value kernel argument: first global row of each shard
int launch_merge_parts(uint32_t n_parts, uint32_t nq, uint32_t k, uint64_t part_stride, const PartBase &part_base,
                       const uint32_t *d_rows, const float *d_scores, const float *d_dists,
                       const uint32_t *d_counts, uint64_t *out_rows, float *out_scores, float *out_dists,
                       uint32_t *out_counts, hipStream_t stream);

// single-process sharded index (sharded.cpp): rows are global insertion sequence numbers, ties resolve by them
int launch_merge_parts_seq(uint32_t n_parts, uint32_t nq, uint32_t k, uint64_t part_stride, const uint32_t *d_seq_rows,
                           const float *d_scores, const float *d_dists, const uint32_t *d_counts, uint32_t *out_seq_rows,
                           float *out_scores, float *out_dists, uint32_t *out_counts, hipStream_t stream);
// a shard's lists -> its part of the root's gather buffer (peer writes), local rows -> sequence numbers
int launch_publish_part(const uint32_t *rows, const float *scores, const float *dists, const uint32_t *counts,
                        const uint32_t *gseq, uint32_t nq, uint32_t k_src, uint32_t k_dst, uint32_t n_rows, uint32_t *dst,
                        hipStream_t stream);
// rows of this shard -> given vectors of a query block (possibly on a peer device)
int launch_scatter_rows(const float *src, float *dst, const uint32_t *d_src_rows, const uint32_t *d_dst_pos, uint32_t n,
                        uint32_t dim, hipStream_t stream);
int launch_scatter_rows(const uint16_t *src, float *dst, const uint32_t *d_src_rows, const uint32_t *d_dst_pos, uint32_t n,
                        uint32_t dim, hipStream_t stream);   // bf16 store

// ---- batched search (batch.hip) ----
struct BatchArgs {
    const float *rows;      // [n_rows][dim]
    const float *queries;   // [nq][dim] in HBM
    const float *norms;     // [n_rows (+16 readable)] |row|^2 (cx_index::d_norms)
    const char *split;      // bf16 hi/lo copy of the rows, tile by tile as MFMA fragments (cx_index::d_split, batch_common.hpp)
    uint32_t n_rows, nq, dim, k, capq;
    uint32_t n_groups;      // query groups in this launch (gridDim.y); nq covers all of them
    uint32_t qpp;           // queries per group: batch_queries_per_pass(dim, k, queries of the whole call)
    DevFilter flt;
    uint64_t *part_keys;    // [nq][grid][k]
    float *part_sims;
    unsigned long long *diag;  // diagnostic build only (CX_BATCH_DIAG=1): [grid*4 waves][5] cycle sums
    uint32_t *gslots;       // [n_groups][64][128] zeroed before the launch: cross-block score bound (batch.hip, "global slots")
    uint32_t arm;           // CX_BATCH_ARM: measurement arms (results invalid): 1 no appends, 2 no shrink check, 4 consumers only keep the barriers, 8 no compaction, 16 no slot refresh
};
bool batch_supported(uint32_t dim, uint32_t k);
uint32_t batch_grid_blocks(uint32_t n_rows);
// rows [row_lo, row_hi) -> the split store the batched search reads (dim 384 / 768)
int launch_build_split(const float *rows, char *split, uint32_t row_lo, uint32_t row_hi, uint32_t dim, hipStream_t stream);
// 64; 32 for the wide lists (32 < k <= 104) when the call has no more than 32 queries
uint32_t batch_queries_per_pass(uint32_t dim, uint32_t k, uint64_t nq);
void batch_launch_shape(uint32_t n_rows, uint32_t dim, uint64_t nq, uint32_t k, uint32_t *chunks, uint32_t *groups);
// one pass of <= 64 queries; per-block lists, to be folded by launch_merge_batch
int launch_batch_scan(BatchArgs a, uint32_t grid, hipStream_t stream);
// second stage for nq queries at once: query q's lists are part[q*n_lists*k ...], outputs at [q*k]; sorted_lists = false:
// the lists are unordered sets with empty (0) slots anywhere (batchg's candidate lists)
int launch_merge_batch(const MergeArgs &m, uint32_t nq, hipStream_t stream, bool sorted_lists = true);
// segmented candidate lists (m.seg_counts / m.seg_len set) -> top k, live entries held in registers; *d_redo is set (never
// cleared) when a query has more live entries than that path holds — batchg passes its overflow flag: the dense pass redoes it
int launch_cand_select(const MergeArgs &m, uint32_t nq, uint32_t *d_redo, hipStream_t stream);

// ---- batched search as a bf16 screening pass + exact re-score of the survivors (batchs.hip) ----
constexpr uint32_t BS_SL = 2048;                      // bound slots per query in the control block (tile t -> slot t mod BS_SL)
constexpr uint32_t BS_MAXQ = 128;                     // queries of one pass: two banks of 64 at row widths up to 512 (batchs.hip: NB), one bank above
constexpr uint32_t BS_CTL_BOUND = BS_MAXQ * BS_SL;    // word offsets inside the control block
constexpr uint32_t BS_CTL_CNT = BS_CTL_BOUND + BS_MAXQ;
constexpr uint32_t BS_CTL_NEXT = BS_CTL_CNT + BS_MAXQ; // the next unclaimed tile beyond the statically dealt first ones

constexpr uint32_t BS_CTL_MRG = BS_CTL_NEXT + 16;     // 64 margins (2 eps of each query, f32 bits) of the last pass, for the re-score's second look
constexpr uint32_t BS_CTL_REDO = BS_CTL_MRG + BS_MAXQ;     // 64 flags: the query is irregular (|q|^2 is zero, non-finite or outside [BS_REG_LO, BS_REG_HI]) — screened not at all, redone exactly
constexpr uint32_t BS_CTL_FAIL = BS_CTL_REDO + BS_MAXQ;    // sticky: a worker gave up waiting for room in its hit ring (hits dropped): every query of the pass is redone exactly
constexpr uint32_t BS_CTL_ARR = BS_CTL_FAIL + 2;   // blocks whose workers are through their first tiles (one add per block): when all are, what the slots hold is all the sample there is (a line of its own kind: not the tile counter's)
constexpr uint32_t BS_CTL_WORDS = BS_CTL_FAIL + 16;
// A vector is REGULAR when its sum of squares, as the reference's f32 arithmetic computes it (vector/index.rs:174-175), is a
// number in [BS_REG_LO, BS_REG_HI]: then no product or partial sum of a pair of regular vectors overflows, what underflows is
// below 1e-8 of |x||y|, and the screening bound's 1e-4 of f32 slack holds.  Anything else — a zero vector, elements scaled by
// 1e+-20 (x^2 overflows to infinity / underflows to 0: the reference divides by infinity or by zero and scores 0, 1 or NaN), an
// Inf or NaN element — is IRREGULAR: it gets a zero shadow row, is never screened, and every pair it is part of is computed with
// the reference's arithmetic (irregular rows: a short list re-scored for every query; irregular queries: the exact redo).
constexpr float BS_REG_LO = 1.0e-30f, BS_REG_HI = 1.0e30f;
__host__ __device__ inline bool bs_regular(float sumsq) { return sumsq >= BS_REG_LO && sumsq <= BS_REG_HI; }   // (a NaN fails both)
constexpr uint32_t BS_IRR_CAP = 1024;                 // irregular rows a store may hold before its screening paths are switched off
constexpr uint32_t BS_REDO_WAVES = 128;               // waves that redo one query exactly (the re-score kernel's 32 slices x 4; 16 slices measured: k = 100 loses 4 us, k = 256 gains 8)
struct BatchSArgs {
    const uint16_t *shadow_t; // cx_index::d_shadow_t: rows L2-normalised, bf16, the all-pairs filter's tiled layout (tiled_shadow_off below)
    const uint32_t *shadow_err; // cx_index::d_shadow_err: the largest rounding error of a shadow row (f32 bits); null = the worst case 2^-8
    const float *queries;   // [nq][dim] f32 in HBM
    const float *rows;      // the f32 store (the re-score reads it); null for a bf16 store
    const uint16_t *rows16; // the bf16 store
    uint32_t n_rows, nq, dim, k;
    DevFilter flt;
    uint32_t *ctl;          // [BS_CTL_WORDS]: BS_MAXQ x BS_SL bound slots | published bounds | list lengths | tile counter | margins | flags; zero between
                            // passes (the select kernel clears what a pass used)
    uint32_t *cand_rows;    // [nq][cap] candidate lists: row; approximate, then exact cosine (written by the re-score kernel)
    float *cand_cos;
    uint32_t cap;           // entries per query (>= BS_REDO_WAVES x k): a query whose list runs over is redone exactly by the re-score kernel
    const uint32_t *irr_rows; // [irr_n] the store's irregular rows (zero shadow rows: never hits), re-scored for every query
    uint32_t irr_n;
    uint32_t arm;           // CX_BATCHS_ARM: measurement arms (results invalid): 1 workers drop their hits, 4 the service wave drops them
    uint32_t pub_min;       // slots a query's first publisher waits for (CX_BATCHS_PUB_MIN, 64; at least k)
    uint32_t claim;         // tiles a service wave claims at a time (set by the launcher: 21, CX_BATCHS_CLAIM)
    uint32_t claim_tail;    // tiles per claim once fewer than a full claim per block are left (7; 0 = the same size to the end; CX_BATCHS_CLAIM_TAIL)
    uint32_t loc_min_rows;  // fewest rows of a pass whose blocks run ahead on block-local first bounds (set by the launcher; CX_BATCHS_LOC_MIN)
    unsigned long long *tl; // CX_BATCHS_TL=1: [grid][32] s_memrealtime stamps of a pass (diagnostic; null otherwise)
    // threshold mode (launch_batchs_thr: the all-pairs filter of <= 64 scanned rows, allpairs_stream.hip's contract); nq = n_scan
    float thr_lo;               // threshold - eps
    const uint32_t *scan_rows;  // [n_scan] row of each scanned node, or null = identity
    const uint16_t *shadow_q;   // scanned vectors that are not rows of this shard ([n_scan][dim] normalised bf16), or null
    uint32_t *thr_cand_cnt;     // [n_scan], zeroed by the caller
    uint32_t *thr_cand;         // [n_scan][thr_cap]
    uint32_t thr_cap;
    uint32_t *thr_next;         // [2] the pass's tile counter, and a flag a worker raises when it gave up on its hit ring (hits dropped: the caller
                                // redoes every scanned row exactly); both zeroed by the launcher
};
bool batchs_supported(uint32_t dim, uint32_t k);   // dim % 128 == 0, dim <= 1024, k <= 256
bool batchs_thr_supported(uint32_t n_rows, uint32_t dim, uint32_t n_scan);
int launch_batchs_thr(const BatchSArgs &a, hipStream_t stream);
uint32_t batchs_min_rows();   // fewest rows that take this path (CX_BATCHS_MIN_ROWS)
uint32_t batchs_queries_per_pass(uint32_t dim, uint32_t n_rows, uint64_t nq);   // 128 for calls of more than 64 queries at widths up to 512 (CX_BATCHS_QPP), else 64
uint32_t batchs_cand_cap(uint32_t n_rows, uint32_t k);   // entries per query of the candidate lists (CX_BATCHS_CAND_CAP, 65,536; never fewer than an exact redo writes)
int launch_batchs_pass(const BatchSArgs &a, hipStream_t stream);
int launch_batchs_select(const BatchSArgs &a, uint32_t *out_rows, float *out_scores, float *out_dists, uint32_t *out_count, hipStream_t stream);

// ---- batched search for the other row widths (batchg.hip): dense cosines for <= 64 queries, then top-k ----
bool batchg_supported(uint32_t dim, uint32_t k);       // dim % 128 == 0, dim <= 4096, k <= 256
size_t batchg_qimg_bytes(uint32_t dim);                 // scratch for the split query images
int launch_batchg_scores(const float *rows, const float *norms, uint32_t n_rows, uint32_t dim, const float *d_queries, uint32_t nq,
                         char *d_qimg, float *d_qq, float *d_dense, uint32_t stride, hipStream_t stream);   // split + one dense pass
// The pass in pieces.  Filter mode: a dense pass over every tile_step-th row tile gives each query a bound (the k-th best
// score of the sample: the k-th best of all rows can only be higher), the pass over all rows then writes only the rows that
// reach it — (key, cosine) into per-block lists that launch_merge_batch folds like any other partial lists — instead of
// 4 bytes per row and query.  A block whose list of some query runs over sets *overflow; the caller then runs the dense
// pass with run_if = overflow (kernels that return at once when the flag is 0), which is exact whatever the data.
struct BatchGFilter {
    const uint32_t *tau_ord;   // [64] score_ord of each query's bound
    uint64_t *cand_keys;       // [64][batchg_grid(n_rows)][cb]
    float *cand_sims;
    uint32_t *cand_counts;     // [64][batchg_grid(n_rows)] entries each block wrote (<= cb)
    uint32_t *overflow;        // [1], zeroed
    uint32_t cb;               // entries per block and query: a multiple of k
    DevFilter flt;             // rows that fail it are no candidates
};
uint32_t batchg_tile_rows(bool rows16 = false);
uint32_t batchg_grid(uint32_t n_rows, bool rows16 = false);
uint32_t batchg_sample_rows(uint32_t n_rows, uint32_t tile_step, uint32_t *n_tiles_out, bool rows16 = false);   // dense columns a sampled pass fills
int launch_batchg_split(const float *d_queries, uint32_t nq, uint32_t dim, char *d_qimg, float *d_qq, hipStream_t stream, bool rows16 = false);
int launch_batchg_pass(const float *rows, const float *norms, uint32_t n_rows, uint32_t dim, uint32_t nq, const char *d_qimg, const float *d_qq,
                       float *d_dense, uint32_t stride, uint32_t tile_step, const BatchGFilter *f, const uint32_t *run_if, hipStream_t stream,
                       const uint16_t *rows16 = nullptr);   // rows16: a bf16 store (rows == null; the split must have been made with rows16 = true)
// tau_ord[q] = score_ord of the k-th best score among dense[q][0 .. n) over the columns whose row passes flt (0 when fewer
// than k do); column e stands for row (e / tile_rows) * tile_step * tile_rows + e % tile_rows
int launch_bound_select(const float *d_dense, uint32_t stride, uint32_t n, uint32_t nq, uint32_t k, uint32_t *tau_ord, const DevFilter &flt,
                        uint32_t tile_rows, uint32_t tile_step, hipStream_t stream);
// tau_ord[q] = score_ord of the k-th entry of query q's ordered list (0: fewer than k entries)
int launch_tau_from_lists(const float *d_scores, const uint32_t *d_counts, uint32_t nq, uint32_t k, uint32_t *tau_ord, hipStream_t stream);
uint32_t dense_topk_chunks(uint32_t n_rows);
int launch_dense_topk(const float *d_dense, uint32_t stride, uint32_t n_rows, uint32_t nq, uint32_t k, const DevFilter &flt,
                      uint64_t *part_keys, float *part_sims, uint32_t chunks, hipStream_t stream, const uint32_t *run_if = nullptr);

// ---- all-pairs auto-link pass (allpairs.hip) ----
int launch_build_shadow(const float *rows, uint16_t *shadow, uint32_t row_lo, uint32_t row_hi, uint32_t dim,
                        hipStream_t stream);
// the index's own shadow (autolink.cpp: ensure_shadow), tiled (dim % 32 == 0) or row-major: err_max = the largest || bf16(x) - x ||
// of a normalised row (f32 bits, atomic max; tiled only); irregular rows (bs_regular above) get zero shadow rows and are
// counted in *irr_cnt / listed in irr_rows[BS_IRR_CAP] unless entries [0, irr_n_before) already hold them
int launch_build_shadow_index(const float *rows, const uint16_t *rows16, uint16_t *shadow, bool tiled, uint32_t row_lo, uint32_t row_hi, uint32_t dim,
                              hipStream_t stream, uint32_t *err_max, uint32_t *irr_cnt, uint32_t *irr_rows, uint32_t irr_n_before);
int launch_build_shadow(const uint16_t *rows16, uint16_t *shadow, uint32_t row_lo, uint32_t row_hi, uint32_t dim,
                        hipStream_t stream);   // bf16 store
// a pass over a store with irregular rows.  After the filter: the irregular rows (those that still are: irr_ok, [irr_n] scratch,
// filled here) become candidates of every scanned row.  After the rescore: scanned vectors that are irregular themselves are flagged
// in `overflow` (the exact path lists them); ext_vecs = the scanned vectors when they are not rows of this shard (f32 [n_scan][dim])
int launch_irr_append(const float *rows, const uint16_t *rows16, uint32_t dim, uint32_t n_rows, const uint32_t *irr_rows, uint32_t irr_n, uint32_t *irr_ok,
                      uint32_t n_scan, uint32_t *cand_cnt, uint32_t *cand, uint32_t cap, hipStream_t stream);
int launch_irr_mark(const uint32_t *irr_rows, const uint32_t *irr_ok, uint32_t irr_n, const uint32_t *scan_rows, const float *ext_vecs, uint32_t dim, uint32_t n_scan,
                    uint32_t *overflow, hipStream_t stream);


// Tiled shadow (dim % 32 == 0): [16-row block][K-step of 32 elements][16 rows x 64 B], the four 16-byte pieces of a row's
// K-step XOR-permuted with bits 3-4 of the row — one LDS-DMA instruction of the 256-tile kernels = 1 KiB of contiguous
// memory.  Offset, in bf16 elements, of 16-byte piece `piece` (8 elements) of row `row`; kt32 = dim / 32.
__host__ __device__ inline size_t tiled_shadow_off(uint32_t row, uint32_t piece, uint32_t kt32) {
    return ((size_t)(row >> 4) * kt32 + (piece >> 2)) * 512u + (row & 15u) * 32u + (((piece & 3u) ^ ((row >> 3) & 3u)) << 3);
}

struct PairFilterArgs {
    const uint16_t *shadow;     // [n_rows][dim] bf16, L2-normalised rows (the J operand), row-major: only kept when dim % 32 != 0
    const uint16_t *shadow_t;   // the J operand in the tiled layout above (every kernel reads it when it is there); null = read `shadow`
    const uint16_t *shadow_q;   // I operand if the scanned vectors are not rows of this shard ([n_scan][dim]); null = shadow
    const uint32_t *scan_rows;  // [n_scan] row of each scanned node, or null = identity
    uint32_t n_scan, n_rows, dim;
    float thr_lo;               // threshold - eps (bf16 error bound)
    uint32_t *cand_cnt;         // [n_scan], zeroed by the caller
    uint32_t *cand;             // [n_scan][cap]
    uint32_t cap;
    uint32_t symmetric;         // scan set == all rows in order: compute tiles tj >= ti only, emit (i,j) and (j,i)
    const uint32_t *tile_list;  // symmetric only: (ti << 16) | tj of every live tile, in launch order
    uint32_t n_tiles;           // entries in tile_list
    unsigned long long *diag;   // CX_PAIR_DIAG=1 only: [tiles][4] cycle stamps (prologue, main loop, epilogue)
    // persistent kernel only (allpairs_p.hip): hits leave the GEMM as (i | j << 32) pairs, pair_scatter_kernel fills cand
    uint64_t *pairs;            // [pair_cap]
    uint32_t *pair_ctl;         // [32] zeroed by the launcher: [0] pairs written, [1] pairs lost (pair_cap too small), [8 + x] tile tickets of XCD x, [16..20] block 0's clock stamps
    uint32_t pair_cap;
    uint32_t scan_lo;           // persistent kernel: the scanned rows are the shard's rows scan_lo .. scan_lo + n_scan in order (0 with
                                // scan_rows == null; with scan_contig set, scan_rows[i] == scan_lo + i and the kernel ignores the array)
    uint32_t scan_contig;
    const uint16_t *shadow_i;   // persistent kernel: the I operand when the scanned vectors are NOT a run of the shard's rows — a staged panel in the
                                // tiled layout (scanned vector i at tiled row i, whole 256-row tiles, zero beyond n_scan: launch_stage_scan_rows /
                                // launch_build_shadow_index); scan_lo = 0, scan_contig = 1 then.  null = the shard's own shadow_t
    uint32_t block_rows;        // persistent kernel: scanned rows per tile, 256 (default) or 128 (pair_filter_p_block_rows)
    void *ev_begin, *ev_end;    // optional hipEvent_t pair recorded around the GEMM kernel alone (256-tile and persistent kernels)
};
// live tiles of the symmetric pass in L2-friendly order (host side); tile = 128 rows
void pair_filter_tile_list(uint32_t n_rows, std::vector<uint32_t> &out);
// persistent blocks, LDS ring running through the tile boundaries, hits handed over as pairs (allpairs_p.hip); scanned
// rows = the shard's rows in order (tiled shadow both sides), dim % 64 == 0, dim >= 384
bool pair_filter_p_supported(const PairFilterArgs &a);
// the scanned rows of a LIST (arbitrary rows of the shard) as a staged I panel: their shadow pieces copied out of the tiled shadow into
// `out` (tiled layout, scanned row i at row i, n_pad = n_scan rounded up to 256 rows, zero beyond n_scan); bytes = n_pad x dim x 2
int launch_stage_scan_rows(const uint16_t *shadow_t, const uint32_t *d_scan_rows, uint32_t n_scan, uint32_t dim, uint16_t *out, hipStream_t stream);
uint32_t pair_filter_p_block_rows();
void pair_filter_p_tile_list(uint32_t n_rows, uint32_t bm, std::vector<uint32_t> &out);
int launch_pair_filter_p(const PairFilterArgs &a, hipStream_t stream);
int launch_pair_filter(const PairFilterArgs &a, hipStream_t stream);
// scan sets of <= 64 rows (streaming ingest): scanned rows in registers, the shard's shadow streamed tile by tile
bool pair_filter_stream_supported(const PairFilterArgs &a);
int launch_pair_filter_stream(const PairFilterArgs &a, hipStream_t stream);

struct RescoreArgs {
    const float *rows;     // f32 store; null for a bf16 store
    const uint16_t *rows16 = nullptr;   // bf16 store
    const float *q_rows;   // scanned vectors when they are not rows of this shard ([n_scan][dim]); null = rows
    float *out_dists;      // optional [n_scan][topk]
    const uint32_t *meta;
    const uint32_t *scan_rows;
    const uint32_t *cand_cnt;
    const uint32_t *cand;
    uint32_t n_scan, dim, cap, topk;
    float threshold;
    uint32_t *out_rows;    // [n_scan][topk] ordered best-first
    float *out_scores;     // [n_scan][topk]
    uint32_t *out_cnt;     // [n_scan]
    uint32_t *overflow;    // [n_scan] 1 = candidate list overflowed, redo on the exact path
    // Symmetric pass only (scan set == all rows in order, every list present): scratch [n_scan][cap] for the exact
    // cosine of each list entry.  score(i, j) == score(j, i) bit for bit, so each pair is summed once — by the row
    // with the lower index — and written into both lists; null = every list scores its own entries.
    float *pair_sims;
};
int launch_rescore(const RescoreArgs &a, hipStream_t stream);

struct LinkArgs {
    const uint32_t *scan_rows;
    const uint32_t *list_rows;
    const float *list_scores;
    const uint32_t *list_cnt;
    const uint8_t *deleted;   // [n_rows] storage tombstones (quirk Q2) or null
    const uint32_t *meta;     // [n_rows] index metadata: a scanned row that was removed from the index has no embedding and proposes nothing
    // auto_linker.rs:226-231: per scanned node (scan order) the rows it already has a related_to edge to, CSR, each
    // node's segment sorted ascending (the host sorts its copy); null = no edges yet.  Such neighbours are skipped
    // WITHOUT counting towards max_edges (:249-258).
    const uint64_t *existing_offsets;   // [n_scan + 1]
    const uint32_t *existing_to;
    uint32_t n_scan, topk, max_edges, dedup;
    uint64_t max_total;       // emit pass: edges at positions >= max_total are dropped (:284-287 take(max_edges_per_cycle))
    float threshold;
    uint32_t *counts;         // [n_scan]  (count pass)
    const uint64_t *offsets;  // [n_scan]  (emit pass)
    uint32_t *out_from, *out_to;
    float *out_weight;
};
int launch_link_rules(const LinkArgs &a, bool emit, hipStream_t stream);

// exclusive prefix sum u32 -> u64 (edge offsets); temp sized by scan_temp_bytes
size_t scan_temp_bytes(uint32_t n);
int launch_exclusive_scan(const uint32_t *in, uint64_t *out, uint32_t n, void *temp, size_t temp_bytes,
                          hipStream_t stream);

// lists of the redo rows (contiguous, k_src wide) -> their places in the pass's list arrays (k_dst wide)
int launch_scatter_lists(const uint32_t *src_rows, const float *src_scores, const float *src_dists, const uint32_t *src_cnt,
                         const uint32_t *d_pos, uint32_t n, uint32_t k_src, uint32_t k_dst, uint32_t *dst_rows,
                         float *dst_scores, float *dst_dists, uint32_t *dst_cnt, hipStream_t stream);

// dst[d_idx[i]] = d_val ? d_val[i] : value, i < n
int launch_patch_u32(uint32_t *dst, const uint32_t *d_idx, const uint32_t *d_val, uint32_t value, uint32_t n, hipStream_t stream);
// edge segments t = 0 .. n_seg: (to, w)[src_off[t] .. src_off[t + 1]) -> out_* at dst_off[seg_pos[t]], out_from = from_row[t]
int launch_copy_edge_segments(const uint64_t *d_dst_off, const uint32_t *d_seg_pos, const uint64_t *d_src_off, const uint32_t *d_from_row,
                              const uint32_t *d_to, const float *d_w, uint32_t n_seg, uint32_t *out_from, uint32_t *out_to, float *out_w,
                              hipStream_t stream);

// |row|^2 of rows [row_lo, row_hi) -> norms[row] (one wave per row)
// (lossy: one word, set to 1 when a row's squares sum to exactly 0 although it has a non-zero element — cx_index::d_norms_lossy)
int launch_row_norms(const float *rows, float *norms, uint32_t row_lo, uint32_t row_hi, uint32_t dim, uint32_t *lossy, hipStream_t stream);
int launch_row_norms(const uint16_t *rows, float *norms, uint32_t row_lo, uint32_t row_hi, uint32_t dim, uint32_t *lossy, hipStream_t stream);

// row maintenance
int launch_gather_rows(const float *src, float *dst, const uint32_t *d_src_rows, uint32_t n_dst, uint32_t dim,
                       hipStream_t stream);
// bf16 stores (d_src_rows == null: rows 0 .. n_dst in order): rows out as f32, rows moved inside the store, f32 rows in (RNE)
int launch_gather_rows(const uint16_t *src, float *dst, const uint32_t *d_src_rows, uint32_t n_dst, uint32_t dim, hipStream_t stream);
int launch_gather_rows(const uint16_t *src, uint16_t *dst, const uint32_t *d_src_rows, uint32_t n_dst, uint32_t dim, hipStream_t stream);
int launch_gather_rows(const float *src, uint16_t *dst, const uint32_t *d_src_rows, uint32_t n_dst, uint32_t dim, hipStream_t stream);

}  // namespace cx
