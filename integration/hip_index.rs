//! hip_index.rs — the reference-side binding: `impl VectorIndex for HipIndex` over libcortex_hip.so.
//!
//! Drop this file into crates/cortex-core/src/vector/ (mod hip_index; pub use hip_index::HipIndex;)
//! and swap `HnswIndex` for `HipIndex` at the construction sites listed in INTEGRATION.md.
//! NOT COMPILED in this repository's image (no Rust toolchain); every semantic decision lives on the
//! C side, this file only marshals.  The identical ABI is exercised by cortex_amd/index.py + tests/.
use crate::error::{CortexError, Result};
use crate::types::{Embedding, NodeId, NodeKind};
use crate::vector::{SimilarityResult, VectorFilter, VectorIndex};
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::path::Path;

#[repr(C)]
struct CxFilter {
    has_exclude: i32,
    n_exclude: u64,
    exclude_ids: *const u8,
    has_kinds: i32,
    n_kinds: u64,
    kind_codes: *const u32,
    has_agent: i32,
    agent_code: u32,
}

#[link(name = "cortex_hip")]
extern "C" {
    fn cx_last_error() -> *const c_char;
    fn cx_device_count() -> c_int;
    fn cx_create(dimension: u32, device: c_int) -> *mut c_void;
    fn cx_create_ex(dimension: u32, device: c_int, dtype: c_int) -> *mut c_void;
    fn cx_dtype(h: *const c_void) -> c_int;
    fn cx_destroy(ix: *mut c_void);
    fn cx_upsert(ix: *mut c_void, id: *const u8, emb: *const f32, len: u64) -> c_int;
    fn cx_remove(ix: *mut c_void, id: *const u8) -> c_int;
    fn cx_set_metadata(ix: *mut c_void, id: *const u8, kind: u32, agent: u32) -> c_int;
    fn cx_intern(ix: *mut c_void, s: *const c_char, len: u64) -> u32;
    fn cx_lookup(ix: *const c_void, s: *const c_char, len: u64) -> u32;
    fn cx_rebuild(ix: *mut c_void) -> c_int;
    fn cx_len(ix: *const c_void) -> u64;
    fn cx_row_count(ix: *const c_void) -> u64;
    fn cx_search(ix: *const c_void, q: *const f32, len: u64, k: u64, f: *const CxFilter,
                 ids: *mut u8, scores: *mut f32, dists: *mut f32, n_out: *mut u64) -> c_int;
    fn cx_search_threshold(ix: *const c_void, q: *const f32, len: u64, thr: f32, f: *const CxFilter,
                           cap: u64, ids: *mut u8, scores: *mut f32, dists: *mut f32,
                           n_out: *mut u64, n_needed: *mut u64) -> c_int;
    fn cx_search_batch(ix: *const c_void, nq: u64, qs: *const f32, len: u64, k: u64, f: *const CxFilter,
                       ids: *mut u8, scores: *mut f32, dists: *mut f32, counts: *mut u64) -> c_int;
    // auto-linker / dedup passes
    fn cx_autolink_pass_rows(ix: *const c_void, n_scan: u64, scan_rows: *const u32, topk: u64, threshold: f32,
                             max_edges_per_node: u64, max_edges_per_cycle: u64, deleted: *const u8,
                             existing_offsets: *const u64, existing_to: *const u32, cap: u64, out_from: *mut u32,
                             out_to: *mut u32, out_weight: *mut f32, n_out: *mut u64, n_needed: *mut u64) -> c_int;
    fn cx_rows_of(ix: *const c_void, n: u64, ids: *const u8, out_rows: *mut u32) -> c_int;
    fn cx_topk_lists_rows(ix: *const c_void, n_scan: u64, scan_rows: *const u32, topk: u64, out_rows: *mut u32,
                          out_scores: *mut f32, out_counts: *mut u32) -> c_int;
    fn cx_dedup_scan_rows(ix: *const c_void, dedup_threshold: f32, deleted: *const u8, cap: u64, out_a: *mut u32,
                          out_b: *mut u32, out_similarity: *mut f32, n_out: *mut u64, n_needed: *mut u64) -> c_int;
    fn cx_row_id(ix: *const c_void, row: u64, out_id16: *mut u8) -> c_int;
    fn cx_rows_alive(ix: *const c_void, row_lo: u64, n: u64, out_alive: *mut u8) -> c_int;
    fn cx_save(ix: *const c_void, path: *const c_char) -> c_int;
    fn cx_load(path: *const c_char, device: c_int) -> *mut c_void;
    fn cx_load_ex(path: *const c_char, device: c_int, dtype: c_int) -> *mut c_void;
    // query-time score decay + re-rank
    fn cx_set_node_stats_batch(ix: *mut c_void, n: u64, ids: *const u8, kind_codes: *const u32, last_accessed_s: *const i64,
                               last_accessed_ns: *const u32, access_counts: *const u64) -> c_int;
    fn cx_search_decayed(ix: *const c_void, q: *const f32, len: u64, limit: u64, candidate_limit: u64, f: *const CxFilter,
                         cfg: *const CxDecayConfig, recency_bias: f32, now_s: i64, now_ns: u32, ids: *mut u8,
                         scores: *mut f32, raw_scores: *mut f32, n_out: *mut u64) -> c_int;
    // one index over several GPUs of the node: same contracts, `cx_sharded` handle
    fn cx_sharded_create(dimension: u32, n_shards: u32, device_ids: *const c_int) -> *mut c_void;
    fn cx_sharded_create_ex(dimension: u32, n_shards: u32, device_ids: *const c_int, dtype: c_int) -> *mut c_void;
    fn cx_sharded_destroy(h: *mut c_void);
    fn cx_sharded_upsert(h: *mut c_void, id: *const u8, emb: *const f32, len: u64) -> c_int;
    fn cx_sharded_upsert_batch(h: *mut c_void, n: u64, ids: *const u8, embs: *const f32, len: u64) -> c_int;
    fn cx_sharded_remove(h: *mut c_void, id: *const u8) -> c_int;
    fn cx_sharded_set_metadata(h: *mut c_void, id: *const u8, kind: u32, agent: u32) -> c_int;
    fn cx_sharded_intern(h: *mut c_void, s: *const c_char, len: u64) -> u32;
    fn cx_sharded_lookup(h: *const c_void, s: *const c_char, len: u64) -> u32;
    fn cx_sharded_len(h: *const c_void) -> u64;
    fn cx_sharded_row_count(h: *const c_void) -> u64;
    fn cx_sharded_rebuild(h: *mut c_void) -> c_int;
    fn cx_sharded_save(h: *const c_void, path: *const c_char) -> c_int;
    fn cx_sharded_load_ex(path: *const c_char, n_shards: u32, device_ids: *const c_int, dtype: c_int) -> *mut c_void;
    fn cx_sharded_search(h: *const c_void, q: *const f32, len: u64, k: u64, f: *const CxFilter,
                         ids: *mut u8, scores: *mut f32, dists: *mut f32, n_out: *mut u64) -> c_int;
    fn cx_sharded_search_threshold(h: *const c_void, q: *const f32, len: u64, thr: f32, f: *const CxFilter,
                                   cap: u64, ids: *mut u8, scores: *mut f32, dists: *mut f32,
                                   n_out: *mut u64, n_needed: *mut u64) -> c_int;
    fn cx_sharded_search_batch(h: *const c_void, nq: u64, qs: *const f32, len: u64, k: u64, f: *const CxFilter,
                               ids: *mut u8, scores: *mut f32, dists: *mut f32, counts: *mut u64) -> c_int;
    fn cx_sharded_rows_of(h: *const c_void, n: u64, ids: *const u8, out_rows: *mut u32) -> c_int;
    fn cx_sharded_row_id(h: *const c_void, global_row: u64, out_id16: *mut u8) -> c_int;
    fn cx_sharded_autolink_pass_rows(h: *const c_void, n_scan: u64, scan_rows: *const u32, topk: u64, threshold: f32,
                                     max_edges_per_node: u64, max_edges_per_cycle: u64, deleted: *const u8,
                                     existing_offsets: *const u64, existing_to: *const u32, cap: u64, out_from: *mut u32,
                                     out_to: *mut u32, out_weight: *mut f32, n_out: *mut u64, n_needed: *mut u64) -> c_int;
    fn cx_sharded_dedup_scan_rows(h: *const c_void, dedup_threshold: f32, deleted: *const u8, cap: u64, out_a: *mut u32,
                                  out_b: *mut u32, out_similarity: *mut f32, n_out: *mut u64, n_needed: *mut u64) -> c_int;
    fn cx_sharded_topk_lists_rows(h: *const c_void, n_scan: u64, scan_rows: *const u32, topk: u64, out_rows: *mut u32,
                                  out_scores: *mut f32, out_counts: *mut u32) -> c_int;
    fn cx_sharded_bulk_load_nodes(h: *mut c_void, n: u64, blob: *const u8, offsets: *const u64, flags: u32,
                                  stats: *mut CxBulkStats) -> c_int;
    fn cx_sharded_set_node_stats_batch(h: *mut c_void, n: u64, ids: *const u8, kind_codes: *const u32, last_accessed_s: *const i64,
                                       last_accessed_ns: *const u32, access_counts: *const u64) -> c_int;
    fn cx_sharded_search_decayed(h: *const c_void, q: *const f32, len: u64, limit: u64, candidate_limit: u64, f: *const CxFilter,
                                 cfg: *const CxDecayConfig, recency_bias: f32, now_s: i64, now_ns: u32, ids: *mut u8,
                                 scores: *mut f32, raw_scores: *mut f32, n_out: *mut u64) -> c_int;
    // start-up bulk load from the nodes table
    fn cx_bulk_load_nodes(ix: *mut c_void, n: u64, blob: *const u8, offsets: *const u64, flags: u32,
                          stats: *mut CxBulkStats) -> c_int;
}

#[repr(C)]
pub struct CxDecayConfig {   // ScoreDecayConfig (vector/scoring.rs:22-78) with by_kind keys interned
    pub enabled: i32,
    pub daily_rate: f64,
    pub max_age_days: f64,
    pub min_factor: f64,
    pub echo_weight: f64,
    pub echo_cap: f64,
    pub recency_weight: f32,
    pub n_by_kind: u32,
    pub kind_codes: *const u32,
    pub kind_rates: *const f64,
}

#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct CxBulkStats {
    pub records: u64,
    pub undecodable: u64,
    pub deleted: u64,
    pub no_embedding: u64,
    pub dim_mismatch: u64,
    pub indexed: u64,
}
pub const CX_BULK_STRICT: u32 = 1;

const CX_ERR_CAPACITY: c_int = 4;

fn last_error() -> CortexError {
    // the vector layer only ever produces CortexError::Validation(String) (index.rs:299-305, :438-459)
    let msg = unsafe { CStr::from_ptr(cx_last_error()) }.to_string_lossy().into_owned();
    CortexError::Validation(msg)
}
fn check(rc: c_int) -> Result<()> { if rc == 0 { Ok(()) } else { Err(last_error()) } }

/// Exact cosine index resident in the HBM of one MI355X.
pub struct HipIndex { h: *mut c_void, dimension: usize }
// &self entry points are re-entrant in the library; &mut self ones need exclusivity,
// which Arc<RwLock<HipIndex>> at every call site already provides.
unsafe impl Send for HipIndex {}
unsafe impl Sync for HipIndex {}
impl Drop for HipIndex { fn drop(&mut self) { unsafe { cx_destroy(self.h) } } }

/// Owns the marshalled arrays a CxFilter points into.
struct FilterBuf { ex: Vec<u8>, kinds: Vec<u32>, c: CxFilter }

impl HipIndex {
    pub fn new(dimension: usize) -> Self { Self::on_device(dimension, 0).expect("no MI355X available") }
    pub fn with_metadata(dimension: usize) -> Self { Self::new(dimension) }
    pub fn on_device(dimension: usize, device: i32) -> Result<Self> {
        let h = unsafe { cx_create(dimension as u32, device) };
        if h.is_null() { Err(last_error()) } else { Ok(Self { h, dimension }) }
    }
    /// A bf16 row store (BASELINE config 5): vectors rounded to bf16 once at insert, 2 bytes per element in HBM; results
    /// are the reference's for the rounded vectors.  dtype: 0 = f32, 1 = bf16.
    pub fn with_dtype(dimension: usize, device: i32, dtype: i32) -> Result<Self> {
        let h = unsafe { cx_create_ex(dimension as u32, device, dtype) };
        if h.is_null() { Err(last_error()) } else { Ok(Self { h, dimension }) }
    }
    pub fn is_bf16(&self) -> bool { unsafe { cx_dtype(self.h) == 1 } }
    // adds the string if new: only from &mut self (set_metadata, bulk load)
    fn intern(&mut self, s: &str) -> u32 { unsafe { cx_intern(self.h, s.as_ptr() as *const c_char, s.len() as u64) } }
    // read-only: filters on the concurrent &self path; 0 = never interned = matches no row that has metadata
    fn lookup(&self, s: &str) -> u32 { unsafe { cx_lookup(self.h, s.as_ptr() as *const c_char, s.len() as u64) } }
    /// serve.rs:105-123 / api.rs:56-70 in one call: `values` are the raw bincode values of the nodes table
    /// (a `RedbStorage::raw_node_values()` iterator a maintainer adds next to `list_nodes`), handed over
    /// without deserialising a `Node` per row.  `strict` = Cortex::open's `?` on a wrong-length embedding.
    pub fn bulk_load_nodes<'a>(&mut self, values: impl Iterator<Item = &'a [u8]>, strict: bool) -> Result<CxBulkStats> {
        let mut blob: Vec<u8> = Vec::new();
        let mut offsets: Vec<u64> = vec![0];
        for v in values {
            blob.extend_from_slice(v);
            offsets.push(blob.len() as u64);
        }
        let mut st = CxBulkStats::default();
        let rc = unsafe {
            cx_bulk_load_nodes(self.h, (offsets.len() - 1) as u64, blob.as_ptr(), offsets.as_ptr(),
                               if strict { CX_BULK_STRICT } else { 0 }, &mut st)
        };
        if rc != 0 { return Err(last_error()); }
        Ok(st)
    }

    /// routes.rs:889-947 in one call: candidates, apply_score_decay, stable re-rank, truncate.
    pub fn search_decayed(&self, q: &Vec<f32>, limit: usize, cfg: &ScoreDecayConfig, recency_bias: f32,
                          now: DateTime<Utc>) -> Result<Vec<(NodeId, f32, f32)>> {
        let (codes, rates): (Vec<u32>, Vec<f64>) = cfg.by_kind.iter()
            .map(|(k, r)| (self.lookup(k), *r)).unzip();
        let c = CxDecayConfig { enabled: cfg.enabled as i32, daily_rate: cfg.daily_rate, max_age_days: cfg.max_age_days,
            min_factor: cfg.min_factor, echo_weight: cfg.echo_weight, echo_cap: cfg.echo_cap, recency_weight: cfg.recency_weight,
            n_by_kind: codes.len() as u32, kind_codes: codes.as_ptr(), kind_rates: rates.as_ptr() };
        let cand = if cfg.enabled && recency_bias > 0.0 { (limit * 3).max(30) } else { limit };
        let cap = limit.max(1);
        let (mut ids, mut sc, mut raw, mut n) = (vec![0u8; 16 * cap], vec![0f32; cap], vec![0f32; cap], 0u64);
        let rc = unsafe { cx_search_decayed(self.h, q.as_ptr(), q.len() as u64, limit as u64, cand as u64, std::ptr::null(), &c,
            recency_bias, now.timestamp(), now.timestamp_subsec_nanos(), ids.as_mut_ptr(), sc.as_mut_ptr(), raw.as_mut_ptr(), &mut n) };
        if rc != 0 { return Err(last_error()); }
        Ok((0..n as usize).map(|i| (Uuid::from_slice(&ids[16 * i..16 * i + 16]).unwrap(), sc[i], raw[i])).collect())
    }

    fn rows_of(&self, ids: &[NodeId]) -> Result<Vec<u32>> {
        let mut flat = Vec::with_capacity(16 * ids.len());
        for id in ids { flat.extend_from_slice(id.as_bytes()); }
        let mut rows = vec![0u32; ids.len()];
        check(unsafe { cx_rows_of(self.h, ids.len() as u64, flat.as_ptr(), rows.as_mut_ptr()) })?;
        Ok(rows)
    }

    /// The kNN loop of `AutoLinker::run_cycle` (auto_linker.rs:215-264) for SimilarityLinkRule, one call per cycle:
    /// `nodes` = the cycle's batch in scan order (nodes without a row in the index are skipped like :217-218 would
    /// after a failed `ensure_embedding`), `existing[i]` = targets of `storage.edges_from(nodes[i].id)` whose relation
    /// is related_to (:226-231), `deleted` = tombstoned-but-indexed neighbours (:240-243).  Returns (from, to, score)
    /// in proposal order, already truncated to `max_edges_per_cycle` (:284-287).
    pub fn similarity_edges(&self, nodes: &[NodeId], existing: &[Vec<NodeId>], cfg: &SimilarityConfig,
                            max_edges_per_node: usize, max_edges_per_cycle: usize, deleted: &[NodeId])
        -> Result<Vec<(NodeId, NodeId, f32)>> {
        const NO_ROW: u32 = u32::MAX;
        let all = self.rows_of(nodes)?;
        let (mut scan, mut ex_off, mut ex_to): (Vec<u32>, Vec<u64>, Vec<u32>) = (Vec::new(), vec![0], Vec::new());
        for (i, &r) in all.iter().enumerate() {
            if r == NO_ROW { continue; }
            scan.push(r);
            if let Some(t) = existing.get(i) { ex_to.extend(self.rows_of(t)?.into_iter().filter(|&x| x != NO_ROW)); }
            ex_off.push(ex_to.len() as u64);
        }
        let n_rows = unsafe { cx_row_count(self.h) } as usize;
        let mut del = vec![0u8; if deleted.is_empty() { 0 } else { n_rows }];
        for r in self.rows_of(deleted)? { if r != NO_ROW { del[r as usize] = 1; } }
        let mut cap = (scan.len() * 4).max(1024);
        loop { // count-then-fill
            let (mut fr, mut to, mut w) = (vec![0u32; cap], vec![0u32; cap], vec![0f32; cap]);
            let (mut n, mut need) = (0u64, 0u64);
            let rc = unsafe { cx_autolink_pass_rows(self.h, scan.len() as u64, scan.as_ptr(), 100, cfg.auto_link_threshold,
                max_edges_per_node as u64, max_edges_per_cycle as u64, if del.is_empty() { std::ptr::null() } else { del.as_ptr() },
                ex_off.as_ptr(), ex_to.as_ptr(), cap as u64, fr.as_mut_ptr(), to.as_mut_ptr(), w.as_mut_ptr(), &mut n, &mut need) };
            if rc == CX_ERR_CAPACITY && need as usize > cap { cap = need as usize; continue; }
            check(rc)?;
            let mut id = [0u8; 16];
            let mut out = Vec::with_capacity(n as usize);
            for i in 0..n as usize {
                check(unsafe { cx_row_id(self.h, fr[i] as u64, id.as_mut_ptr()) })?;
                let a = NodeId::from_slice(&id).unwrap();
                check(unsafe { cx_row_id(self.h, to[i] as u64, id.as_mut_ptr()) })?;
                out.push((a, NodeId::from_slice(&id).unwrap(), w[i]));
            }
            return Ok(out);
        }
    }

    pub fn set_metadata(&mut self, id: NodeId, kind: NodeKind, source_agent: String) {
        let (k, a) = (self.intern(kind.as_str()), self.intern(&source_agent));
        unsafe { cx_set_metadata(self.h, id.as_bytes().as_ptr(), k, a) };
    }
    fn filter(&self, f: Option<&VectorFilter>) -> Option<Box<FilterBuf>> {
        let f = f?;
        let mut b = Box::new(FilterBuf { ex: Vec::new(), kinds: Vec::new(), c: unsafe { std::mem::zeroed() } });
        if let Some(ex) = &f.exclude {
            for id in ex { b.ex.extend_from_slice(id.as_bytes()); }
            b.c.has_exclude = 1; b.c.n_exclude = ex.len() as u64; b.c.exclude_ids = b.ex.as_ptr();
        }
        if let Some(kinds) = &f.kinds {
            b.kinds = kinds.iter().map(|k| self.lookup(k.as_str())).collect();
            b.c.has_kinds = 1; b.c.n_kinds = kinds.len() as u64; b.c.kind_codes = b.kinds.as_ptr();
        }
        if let Some(agent) = &f.source_agent { b.c.has_agent = 1; b.c.agent_code = self.lookup(agent); }
        Some(b)
    }
    fn collect(ids: &[u8], scores: &[f32], dists: &[f32], n: usize) -> Vec<SimilarityResult> {
        (0..n).map(|i| SimilarityResult {
            node_id: NodeId::from_slice(&ids[16 * i..16 * i + 16]).unwrap(),
            score: scores[i], distance: dists[i],
        }).collect()
    }
}

impl VectorIndex for HipIndex {
    fn insert(&mut self, id: NodeId, embedding: &Embedding) -> Result<()> {
        check(unsafe { cx_upsert(self.h, id.as_bytes().as_ptr(), embedding.as_ptr(), embedding.len() as u64) })
    }
    fn remove(&mut self, id: NodeId) -> Result<()> { check(unsafe { cx_remove(self.h, id.as_bytes().as_ptr()) }) }

    fn search(&self, query: &Embedding, k: usize, filter: Option<&VectorFilter>) -> Result<Vec<SimilarityResult>> {
        let cap = k.min(unsafe { cx_row_count(self.h) } as usize).max(1);
        let (mut ids, mut sc, mut di) = (vec![0u8; 16 * cap], vec![0f32; cap], vec![0f32; cap]);
        let mut n = 0u64;
        let fb = self.filter(filter);
        let fp = fb.as_ref().map_or(std::ptr::null(), |b| &b.c as *const CxFilter);
        check(unsafe { cx_search(self.h, query.as_ptr(), query.len() as u64, k as u64, fp,
                                 ids.as_mut_ptr(), sc.as_mut_ptr(), di.as_mut_ptr(), &mut n) })?;
        Ok(Self::collect(&ids, &sc, &di, n as usize))
    }

    fn search_threshold(&self, query: &Embedding, threshold: f32, filter: Option<&VectorFilter>)
        -> Result<Vec<SimilarityResult>> {
        let fb = self.filter(filter);
        let fp = fb.as_ref().map_or(std::ptr::null(), |b| &b.c as *const CxFilter);
        let mut cap = 256usize;
        loop { // count-then-fill: CX_ERR_CAPACITY reports the size needed
            let (mut ids, mut sc, mut di) = (vec![0u8; 16 * cap], vec![0f32; cap], vec![0f32; cap]);
            let (mut n, mut need) = (0u64, 0u64);
            let rc = unsafe { cx_search_threshold(self.h, query.as_ptr(), query.len() as u64, threshold, fp,
                                                  cap as u64, ids.as_mut_ptr(), sc.as_mut_ptr(), di.as_mut_ptr(),
                                                  &mut n, &mut need) };
            if rc == CX_ERR_CAPACITY { cap = need as usize; continue; }
            check(rc)?;
            return Ok(Self::collect(&ids, &sc, &di, n as usize));
        }
    }

    fn search_batch(&self, queries: &[(NodeId, Embedding)], k: usize, filter: Option<&VectorFilter>)
        -> Result<HashMap<NodeId, Vec<SimilarityResult>>> {
        let nq = queries.len();
        if nq == 0 { return Ok(HashMap::new()); }
        let len = queries[0].1.len();
        let mut flat = Vec::with_capacity(nq * len);
        for (_, e) in queries { flat.extend_from_slice(e); } // one upload, corpus read once per batch
        let kk = k.max(1);
        let (mut ids, mut sc, mut di) = (vec![0u8; 16 * nq * kk], vec![0f32; nq * kk], vec![0f32; nq * kk]);
        let mut counts = vec![0u64; nq];
        let fb = self.filter(filter);
        let fp = fb.as_ref().map_or(std::ptr::null(), |b| &b.c as *const CxFilter);
        check(unsafe { cx_search_batch(self.h, nq as u64, flat.as_ptr(), len as u64, k as u64, fp,
                                       ids.as_mut_ptr(), sc.as_mut_ptr(), di.as_mut_ptr(), counts.as_mut_ptr()) })?;
        let mut map = HashMap::with_capacity(nq);
        for (i, (qid, _)) in queries.iter().enumerate() {
            let o = i * kk;
            map.insert(*qid, Self::collect(&ids[16 * o..], &sc[o..], &di[o..], counts[i] as usize));
        }
        Ok(map)
    }

    fn len(&self) -> usize { unsafe { cx_len(self.h) as usize } }
    fn rebuild(&mut self) -> Result<()> { check(unsafe { cx_rebuild(self.h) }) }

    fn save(&self, path: &Path) -> Result<()> {
        let p = std::ffi::CString::new(path.to_string_lossy().as_bytes()).map_err(|e| CortexError::Validation(e.to_string()))?;
        check(unsafe { cx_save(self.h, p.as_ptr()) })
    }
    fn load(path: &Path) -> Result<Self> {
        let p = std::ffi::CString::new(path.to_string_lossy().as_bytes()).map_err(|e| CortexError::Validation(e.to_string()))?;
        // CORTEX_HIP_DTYPE=bf16 loads the file's f32 vectors into a bf16 store (cx_load_ex)
        let bf16 = std::env::var("CORTEX_HIP_DTYPE").map(|v| v == "bf16").unwrap_or(false);
        let h = unsafe { if bf16 { cx_load_ex(p.as_ptr(), 0, 1) } else { cx_load(p.as_ptr(), 0) } };
        if h.is_null() { return Err(last_error()); }
        Ok(Self { h, dimension: 0 })
    }
}


/// The same index over several MI355X of the node (`cx_sharded`, cortex_hip.h): one shard per device, shard scans
/// concurrent, partial top-k lists merged on the first device; results, tie order and rows are those of ONE index.
/// serve.rs:101 becomes `Arc::new(RwLock::new(ShardedHipIndex::new(dim, &[0, 1, 2, 3, 4, 5, 6, 7])?))`.
pub struct ShardedHipIndex { h: *mut c_void }
unsafe impl Send for ShardedHipIndex {}
unsafe impl Sync for ShardedHipIndex {}
impl Drop for ShardedHipIndex { fn drop(&mut self) { unsafe { cx_sharded_destroy(self.h) } } }

impl ShardedHipIndex {
    pub fn new(dimension: usize, devices: &[i32]) -> Result<Self> {
        let h = unsafe { cx_sharded_create(dimension as u32, devices.len() as u32, devices.as_ptr()) };
        if h.is_null() { Err(last_error()) } else { Ok(Self { h }) }
    }
    pub fn with_dtype(dimension: usize, devices: &[i32], dtype: i32) -> Result<Self> {
        let h = unsafe { cx_sharded_create_ex(dimension as u32, devices.len() as u32, devices.as_ptr(), dtype) };
        if h.is_null() { Err(last_error()) } else { Ok(Self { h }) }
    }
    pub fn set_metadata(&mut self, id: NodeId, kind: NodeKind, source_agent: String) {
        let k = unsafe { cx_sharded_intern(self.h, kind.as_str().as_ptr() as *const c_char, kind.as_str().len() as u64) };
        let a = unsafe { cx_sharded_intern(self.h, source_agent.as_ptr() as *const c_char, source_agent.len() as u64) };
        unsafe { cx_sharded_set_metadata(self.h, id.as_bytes().as_ptr(), k, a) };
    }
    /// n inserts in one call (the start-up loop of serve.rs:105-123 without N FFI round trips)
    pub fn insert_batch(&mut self, ids: &[NodeId], embeddings: &[f32], len: usize) -> Result<()> {
        let mut flat = Vec::with_capacity(16 * ids.len());
        for id in ids { flat.extend_from_slice(id.as_bytes()); }
        check(unsafe { cx_sharded_upsert_batch(self.h, ids.len() as u64, flat.as_ptr(), embeddings.as_ptr(), len as u64) })
    }
    fn lookup(&self, s: &str) -> u32 { unsafe { cx_sharded_lookup(self.h, s.as_ptr() as *const c_char, s.len() as u64) } }
    /// serve.rs:105-123 in one call, as HipIndex::bulk_load_nodes: the records are decoded once on the host and the
    /// embeddings dealt to the shards block by block.
    pub fn bulk_load_nodes<'a>(&mut self, values: impl Iterator<Item = &'a [u8]>, strict: bool) -> Result<CxBulkStats> {
        let (mut blob, mut offsets): (Vec<u8>, Vec<u64>) = (Vec::new(), vec![0]);
        for v in values { blob.extend_from_slice(v); offsets.push(blob.len() as u64); }
        let mut st = CxBulkStats::default();
        check(unsafe { cx_sharded_bulk_load_nodes(self.h, (offsets.len() - 1) as u64, blob.as_ptr(), offsets.as_ptr(),
                                                  if strict { CX_BULK_STRICT } else { 0 }, &mut st) })?;
        Ok(st)
    }
    fn filter(&self, f: Option<&VectorFilter>) -> Option<Box<FilterBuf>> {
        let f = f?;
        let mut b = Box::new(FilterBuf { ex: Vec::new(), kinds: Vec::new(), c: unsafe { std::mem::zeroed() } });
        if let Some(ex) = &f.exclude {
            for id in ex { b.ex.extend_from_slice(id.as_bytes()); }
            b.c.has_exclude = 1; b.c.n_exclude = ex.len() as u64; b.c.exclude_ids = b.ex.as_ptr();
        }
        if let Some(kinds) = &f.kinds {
            b.kinds = kinds.iter().map(|k| self.lookup(k.as_str())).collect();
            b.c.has_kinds = 1; b.c.n_kinds = kinds.len() as u64; b.c.kind_codes = b.kinds.as_ptr();
        }
        if let Some(agent) = &f.source_agent { b.c.has_agent = 1; b.c.agent_code = self.lookup(agent); }
        Some(b)
    }
    /// AutoLinker::run_cycle's kNN loop for SimilarityLinkRule over all shards — HipIndex::similarity_edges with
    /// global rows underneath (cx_sharded_rows_of / cx_sharded_row_id instead of cx_rows_of / cx_row_id).
    pub fn similarity_edges(&self, nodes: &[NodeId], existing: &[Vec<NodeId>], cfg: &SimilarityConfig,
                            max_edges_per_node: usize, max_edges_per_cycle: usize) -> Result<Vec<(NodeId, NodeId, f32)>> {
        const NO_ROW: u32 = u32::MAX;
        let rows_of = |ids: &[NodeId]| -> Result<Vec<u32>> {
            let mut flat = Vec::with_capacity(16 * ids.len());
            for id in ids { flat.extend_from_slice(id.as_bytes()); }
            let mut rows = vec![0u32; ids.len()];
            check(unsafe { cx_sharded_rows_of(self.h, ids.len() as u64, flat.as_ptr(), rows.as_mut_ptr()) })?;
            Ok(rows)
        };
        let all = rows_of(nodes)?;
        let (mut scan, mut ex_off, mut ex_to): (Vec<u32>, Vec<u64>, Vec<u32>) = (Vec::new(), vec![0], Vec::new());
        for (i, &r) in all.iter().enumerate() {
            if r == NO_ROW { continue; }
            scan.push(r);
            if let Some(t) = existing.get(i) { ex_to.extend(rows_of(t)?.into_iter().filter(|&x| x != NO_ROW)); }
            ex_off.push(ex_to.len() as u64);
        }
        let mut cap = (scan.len() * 4).max(1024);
        loop {
            let (mut fr, mut to, mut w) = (vec![0u32; cap], vec![0u32; cap], vec![0f32; cap]);
            let (mut n, mut need) = (0u64, 0u64);
            let rc = unsafe { cx_sharded_autolink_pass_rows(self.h, scan.len() as u64, scan.as_ptr(), 100, cfg.auto_link_threshold,
                max_edges_per_node as u64, max_edges_per_cycle as u64, std::ptr::null(), ex_off.as_ptr(), ex_to.as_ptr(),
                cap as u64, fr.as_mut_ptr(), to.as_mut_ptr(), w.as_mut_ptr(), &mut n, &mut need) };
            if rc == CX_ERR_CAPACITY && need as usize > cap { cap = need as usize; continue; }
            check(rc)?;
            let mut id = [0u8; 16];
            let mut out = Vec::with_capacity(n as usize);
            for i in 0..n as usize {
                check(unsafe { cx_sharded_row_id(self.h, fr[i] as u64, id.as_mut_ptr()) })?;
                let a = NodeId::from_slice(&id).unwrap();
                check(unsafe { cx_sharded_row_id(self.h, to[i] as u64, id.as_mut_ptr()) })?;
                out.push((a, NodeId::from_slice(&id).unwrap(), w[i]));
            }
            return Ok(out);
        }
    }
}

impl VectorIndex for ShardedHipIndex {
    fn insert(&mut self, id: NodeId, embedding: &Embedding) -> Result<()> {
        check(unsafe { cx_sharded_upsert(self.h, id.as_bytes().as_ptr(), embedding.as_ptr(), embedding.len() as u64) })
    }
    fn remove(&mut self, id: NodeId) -> Result<()> { check(unsafe { cx_sharded_remove(self.h, id.as_bytes().as_ptr()) }) }
    fn search(&self, query: &Embedding, k: usize, filter: Option<&VectorFilter>) -> Result<Vec<SimilarityResult>> {
        let cap = k.min(unsafe { cx_sharded_row_count(self.h) } as usize).max(1);
        let (mut ids, mut sc, mut di) = (vec![0u8; 16 * cap], vec![0f32; cap], vec![0f32; cap]);
        let mut n = 0u64;
        let fb = self.filter(filter);
        let fp = fb.as_ref().map_or(std::ptr::null(), |b| &b.c as *const CxFilter);
        check(unsafe { cx_sharded_search(self.h, query.as_ptr(), query.len() as u64, k as u64, fp,
                                         ids.as_mut_ptr(), sc.as_mut_ptr(), di.as_mut_ptr(), &mut n) })?;
        Ok(HipIndex::collect(&ids, &sc, &di, n as usize))
    }
    fn search_threshold(&self, query: &Embedding, threshold: f32, filter: Option<&VectorFilter>)
        -> Result<Vec<SimilarityResult>> {
        let fb = self.filter(filter);
        let fp = fb.as_ref().map_or(std::ptr::null(), |b| &b.c as *const CxFilter);
        let mut cap = 256usize;
        loop {
            let (mut ids, mut sc, mut di) = (vec![0u8; 16 * cap], vec![0f32; cap], vec![0f32; cap]);
            let (mut n, mut need) = (0u64, 0u64);
            let rc = unsafe { cx_sharded_search_threshold(self.h, query.as_ptr(), query.len() as u64, threshold, fp,
                                                          cap as u64, ids.as_mut_ptr(), sc.as_mut_ptr(), di.as_mut_ptr(),
                                                          &mut n, &mut need) };
            if rc == CX_ERR_CAPACITY { cap = need as usize; continue; }
            check(rc)?;
            return Ok(HipIndex::collect(&ids, &sc, &di, n as usize));
        }
    }
    fn search_batch(&self, queries: &[(NodeId, Embedding)], k: usize, filter: Option<&VectorFilter>)
        -> Result<HashMap<NodeId, Vec<SimilarityResult>>> {
        let nq = queries.len();
        if nq == 0 { return Ok(HashMap::new()); }
        let len = queries[0].1.len();
        let mut flat = Vec::with_capacity(nq * len);
        for (_, e) in queries { flat.extend_from_slice(e); }
        let kk = k.max(1);
        let (mut ids, mut sc, mut di) = (vec![0u8; 16 * nq * kk], vec![0f32; nq * kk], vec![0f32; nq * kk]);
        let mut counts = vec![0u64; nq];
        let fb = self.filter(filter);
        let fp = fb.as_ref().map_or(std::ptr::null(), |b| &b.c as *const CxFilter);
        check(unsafe { cx_sharded_search_batch(self.h, nq as u64, flat.as_ptr(), len as u64, k as u64, fp,
                                               ids.as_mut_ptr(), sc.as_mut_ptr(), di.as_mut_ptr(), counts.as_mut_ptr()) })?;
        let mut map = HashMap::with_capacity(nq);
        for (i, (qid, _)) in queries.iter().enumerate() {
            let o = i * kk;
            map.insert(*qid, HipIndex::collect(&ids[16 * o..], &sc[o..], &di[o..], counts[i] as usize));
        }
        Ok(map)
    }
    fn len(&self) -> usize { unsafe { cx_sharded_len(self.h) as usize } }
    fn rebuild(&mut self) -> Result<()> { check(unsafe { cx_sharded_rebuild(self.h) }) }
    /// vector/index.rs:437-445 — ONE file in the reference's own layout (entries in global row order): what `HipIndex::save`
    /// writes for a single index over the same calls, and what `HnswIndex::load` reads.
    fn save(&self, path: &Path) -> Result<()> {
        let p = std::ffi::CString::new(path.to_string_lossy().as_bytes()).map_err(|e| CortexError::Validation(e.to_string()))?;
        check(unsafe { cx_sharded_save(self.h, p.as_ptr()) })
    }
    /// vector/index.rs:447-473 — over the GPUs named in CORTEX_HIP_DEVICES ("0,1,2,3,4,5,6,7"); unset or empty: every device
    /// cx_device_count() reports.  An entry that is not a device number is an error — a typo must not shrink the shard set silently.
    fn load(path: &Path) -> Result<Self> {
        let p = std::ffi::CString::new(path.to_string_lossy().as_bytes()).map_err(|e| CortexError::Validation(e.to_string()))?;
        let n_dev = unsafe { cx_device_count() };
        let devs: Vec<c_int> = match std::env::var("CORTEX_HIP_DEVICES").ok().filter(|v| !v.trim().is_empty()) {
            Some(v) => {
                let mut out = Vec::new();
                for x in v.split(',') {
                    let d: c_int = x.trim().parse().map_err(|_| CortexError::Validation(format!("CORTEX_HIP_DEVICES: '{}' is not a device number", x.trim())))?;
                    if d < 0 || d >= n_dev { return Err(CortexError::Validation(format!("CORTEX_HIP_DEVICES: device {} of {} visible", d, n_dev))); }
                    out.push(d);
                }
                out
            }
            None => (0..n_dev.max(1)).collect(),
        };
        let bf16 = std::env::var("CORTEX_HIP_DTYPE").map(|v| v == "bf16").unwrap_or(false);
        let h = unsafe { cx_sharded_load_ex(p.as_ptr(), devs.len() as u32, devs.as_ptr(), if bf16 { 1 } else { 0 }) };
        if h.is_null() { return Err(last_error()); }
        Ok(Self { h })
    }
}
