#!/usr/bin/env python3
"""bench.py — the hot path's headline benchmark (BASELINE.json: kNN queries/s at 1M x 768-d).

A step = one query against the HBM-resident corpus: the single-query cosine top-k scan
(libcortex_hip.so through the C ABI) over this rank's 1M x 768 f32 shard, plus — for N > 1 —
the RCCL all-gather of the packed partial top-k lists and the merge kernel.  `value` is WEAK
scaling: every GPU holds 1M rows, the corpus is N x 1M rows, and `value` counts 1M-row shard
scans per second over all ranks (= queries/s at N = 1).  The STRONG-scaling line BASELINE
configs[3] asks for (ONE 10M x 768 corpus, batch 64, rows = 10M / N per rank) is
`extra.config4_10Mx768_sharded_batch64_k10`, printed for every N.  Inputs (corpus, queries)
are generated in HBM and stay there; nothing crosses PCIe inside the timed region.  `value`
is timed exactly as asked (W warm-up + K steps, nothing in front); the same region again
behind 64 untimed scans is `extra.steady_state`, 20 scans after a 1 s idle gap `extra.cold_burst`.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches itself: the parent — which has not
touched the GPU — starts `torch.distributed.run` with N ranks as a CHILD process (never an exec), relays rank 0's JSON
line and exits with the child's code.  CX_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse it.

Rank 0 prints ONE JSON line (contract in the task description) carrying `roofline`
(live HIP-event timing of the scan kernel) and, at N = 1, `cpu_baseline` (the CPU
restatement of the reference's brute-force path timed on this box's host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def _baseline_metric() -> str:
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "kNN queries/sec + auto-link pairs/sec at 1M\u00d7768-d; recall@10 vs exact"


BASELINE_METRIC = _baseline_metric()
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SEED_CORPUS, SEED_QUERIES, SEED_DUP = 20260313, 20260314, 20260315
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def synth_ids(row_lo: int, n: int) -> np.ndarray:
    ids = np.zeros((n, 16), dtype=np.uint8)
    ids[:, 0] = 0xC0
    ids[:, 8:] = (np.arange(n, dtype=np.uint64) + np.uint64(row_lo)).astype(">u8").view(np.uint8).reshape(n, 8)
    return ids


def spawn_ranks(n: int) -> int:
    """--gpus N > 1 without a launcher: run this script under torch.distributed.run as a child process, one rank per GPU
    (127.0.0.1 rendezvous, a free port), pass its output through, return its exit code.  Nothing here initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in child.stdout:          # rank 0 prints the one JSON line; anything else (launcher notes) goes to stderr
        if out.lstrip().startswith("{") and '"metric"' in out:
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc else (0 if line is not None else 1)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--rows", type=int, default=1_000_000, help="rows per GPU")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1,
                    help="queries per step (default 1 = the headline single-query workload; 64 = BASELINE config 4's batch)")
    ap.add_argument("--preroll", type=int, default=64,
                    help="untimed scans in front of the SECOND timed region (extra.steady_state); the first — `value` — has none")
    ap.add_argument("--no-config4", action="store_true", help="skip the 10M-row sharded batch-64 leg (extra)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autolink", action="store_true", help="skip the auto-link all-pairs leg (extra)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the 1-thread CPU baseline leg")
    ap.add_argument("--recall-queries", type=int, default=1024,
                    help="held-out queries whose GPU answers are checked against the CPU oracle (SURVEY 8d: all 1,024; ~80 s on 16 threads)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            raise SystemExit(spawn_ranks(args.gpus))
        args.gpus = world
    # one rank per GPU; CX_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 path
    backend = os.environ.get("CX_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import cortex_amd
    from cortex_amd import _lib
    from cortex_amd.sharded import ShardedKnn, hip_local_fn
    L = _lib.load()

    n, d, k = args.rows, args.dim, args.k
    total_rows = n * world
    n_centres = max(1, total_rows // 50)
    row_lo = rank * n

    # corpus shard: generated in HBM, handed to the index device-to-device
    gen = torch.empty((n, d), dtype=torch.float32, device=dev)
    rc = L.cx_synth_fill_dev(local_rank, gen.data_ptr(), SEED_CORPUS, SEED_CORPUS, SEED_DUP, n_centres, row_lo, n, d, 1)
    assert rc == 0, L.cx_last_error()
    ix = cortex_amd.HipIndex(d, device=local_rank)
    ix.reserve(n)
    ix.insert_batch_dev(synth_ids(row_lo, n), gen.data_ptr(), n, d)
    nq_pool = 1024   # 1,024 held-out queries from the corpus' generator
    queries = torch.empty((nq_pool, d), dtype=torch.float32, device=dev)
    rc = L.cx_synth_fill_dev(local_rank, queries.data_ptr(), SEED_CORPUS, SEED_QUERIES, SEED_DUP, n_centres, 0, nq_pool, d, 0)
    assert rc == 0, L.cx_last_error()
    torch.cuda.synchronize()

    bases = [r * n for r in range(world)]
    B = max(1, min(args.batch, nq_pool))
    knn = ShardedKnn(rank, world, bases, B, k, dev, hip_local_fn(ix))
    qptr = queries.data_ptr()

    def step(i: int) -> None:
        # stream of queries: the all-gather of query i is hidden under the scan of query i+1 (N > 1)
        knn.submit(qptr + ((i * B) % (nq_pool - B + 1)) * d * 4)

    # Two timed regions of the same W warm-up + K steps (round-2 ADVICE: both figures, always):
    #  (1) exactly as asked, nothing in front — `value`, `ms_per_step`, `roofline`.  After ANY idle gap of the device (0.2 s is
    #      enough) a stream of identical scans starts at the steady 0.437 ms, climbs to ~0.50 ms around the 10th launch and is
    #      back by the 40th (scripts/cold_probe.py, profiles/r04/cold_start_probe.json): the board's power management settling
    #      under a memory-bound load, not first touch; a trickle of work during the gap or dummy launches in front do not
    #      remove it.  `--steps 20 --warmup 5` sits on that hump, and that is what a server sees on the first scans of a burst.
    #  (2) the same again behind `--preroll` untimed scans — `extra.steady_state`: the rate of a stream that keeps coming.
    def timed_region(first: int):
        for i in range(args.warmup):
            step(first + i)
        knn.flush()
        torch.cuda.synchronize()
        ix.profile_read(reset=True)
        ix.profile_enable(True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(first + args.warmup + i)
        knn.flush()                      # every query's merged result is complete inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        ix.profile_enable(False)
        km, kn = ix.profile_read(reset=True)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, km, kn

    elapsed, kern_ms, kern_n = timed_region(0)
    for i in range(args.preroll):
        step(args.warmup + args.steps + i)
    elapsed_ss, kern_ms_ss, kern_n_ss = timed_region(args.warmup + args.steps + args.preroll)

    value = args.steps * B * world / elapsed
    algo_bytes = float(n) * d * 4.0  # N*d*sizeof(f32) per query per shard; norms recomputed in-scan
    if B >= 3:   # (--batch: the batched pass streams the 2-byte normalised shadow once per launch — batch_roofline below)
        algo_bytes = float(n) * d * 2.0
    avg_ms = kern_ms / max(1, kern_n)
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; gfx950 x2 read
    # correction) whose summary is committed under profiles/; reported only for the shape it was taken on
    traffic, traffic_src = None, None
    for rnd in ("r04", "r03", "r02", "r01"):
        pmc = os.path.join(ROOT, "profiles", rnd, "knn_1Mx768_pmc_final.json")
        if B == 1 and n == 1_000_000 and d == 768 and os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("scan_kernel_hbm_bytes_per_launch")
            traffic_src = f"profiles/{rnd}/knn_1Mx768_pmc_final.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
            break
    out = {
        # BASELINE.json's metric, verbatim; `value` is its first component (kNN queries/s), the auto-link pairs/s
        # and recall@10 components are extra.autolink_allpairs* and extra.recall_at_k_vs_exact
        "metric": BASELINE_METRIC,
        "metric_component": "knn_queries_per_sec_1Mx768",
        "value": value,
        "unit": "queries/s (x 1M-row shards scanned per query)",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "preroll_untimed": 0,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"cosine kNN k={k}, {'single query' if B == 1 else f'batch of {B} queries'} per step, "
                        f"{n} x {d} f32 rows per GPU (exact brute force, HBM-resident)",
            "rows_per_gpu": n, "dim": d, "k": k, "batch": B, "total_rows": total_rows,
            "sharding": "row-range, RCCL all-gather of partial top-k + merge" if world > 1 else "single shard",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": float(traffic) if traffic else None, "traffic_source": traffic_src,
            "kernel": "cx::scan_kernel" if B < 3 else f"cx::batchs_kernel<{d}> (streams the 2-byte normalised shadow: store-equivalent figures)", "avg_kernel_ms": avg_ms, "launches": kern_n,
            "algorithmic_bytes_per_launch": algo_bytes,
            "step_achieved": algo_bytes / (elapsed / args.steps) / 1e9, "frac_step": algo_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
        },
    }

    avg_ss = kern_ms_ss / max(1, kern_n_ss)
    out.setdefault("extra", {})["steady_state"] = {
        "what": f"the same {args.warmup} warm-up + {args.steps} timed steps again, behind {args.preroll} untimed scans run back to back with them "
                "(the post-idle transient of the board has passed: profiles/r04/cold_start_probe.json)",
        "value": args.steps * B * world / elapsed_ss, "ms_per_step": elapsed_ss / args.steps * 1e3, "preroll_untimed": args.preroll,
        "roofline": {"bound": "hbm", "achieved": algo_bytes / (avg_ss * 1e-3) / 1e9 if avg_ss > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (algo_bytes / (avg_ss * 1e-3) / 1e9 / HBM_PEAK_GBS) if avg_ss > 0 else 0.0, "avg_kernel_ms": avg_ss, "launches": kern_n_ss},
    }
    if rank == 0 and world == 1 and B == 1:
        # what a server whose GPU idles between queries sees: 20 scans right after a 1 s gap, each timed by its own HIP events
        time.sleep(1.0)
        burst = []
        for i in range(20):
            ix.profile_read(reset=True)
            ix.profile_enable(True)
            step(i)
            knn.flush()
            torch.cuda.synchronize()
            ix.profile_enable(False)
            km, kn = ix.profile_read(reset=True)
            burst.append(km / max(1, kn))
        out["extra"]["cold_burst"] = {"what": "20 single scans right after a 1 s idle gap (host sync after each), kernel ms of each",
                                      "kernel_ms": [round(x, 4) for x in burst], "mean_kernel_ms": float(np.mean(burst)),
                                      "worst_kernel_ms": float(np.max(burst)),
                                      "frac_of_hbm_peak_mean": algo_bytes / (float(np.mean(burst)) * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if rank == 0 and world == 1 and B == 1 and d % 128 == 0 and d <= 1024 and n >= 131072:
        out["extra"]["single_query_screened"] = single_screened_leg(ix, knn, step, queries, n, d, k, args.steps, args.warmup)
    if rank == 0 and world == 1:
        # the box's own peaks: a plain streaming-read kernel and a register-only MFMA loop; the nominal
        # peaks stay the contract's denominators, these say how much of what THIS board delivers the kernels reach
        import ctypes as C
        v = C.c_double(0)
        if L.cx_probe_read_bw(local_rank, 3 << 30, 4, C.byref(v)) == 0 and v.value > 0:
            out["roofline"]["measured_stream_read_GBs"] = v.value
            out["roofline"]["frac_of_measured"] = achieved / v.value
        w = C.c_double(0)
        if L.cx_probe_mfma_tflops(local_rank, 50.0, C.byref(w)) == 0 and w.value > 0:
            out.setdefault("extra", {})["measured_mfma_bf16_TFLOPs"] = w.value
        # the same loop fed from LDS at the filter GEMM's ratio, and with that kernel's LDS-DMA traffic on top: the
        # ceilings of its main loop on this board
        for mode, key in ((0, "measured_mfma_bf16_lds_fed_TFLOPs"), (1, "measured_mfma_bf16_lds_fed_with_dma_TFLOPs")):
            w = C.c_double(0)
            if L.cx_probe_mfma_lds_tflops(local_rank, 50.0, mode, C.byref(w)) == 0 and w.value > 0:
                out.setdefault("extra", {})[key] = w.value
    if not args.no_config4 and B == 1:
        # BASELINE configs[3] as stated: ONE 10M x 768 corpus row-sharded over the N ranks (strong scaling), batch 64, k=10
        c4 = config4_sharded_leg(L, local_rank, dev, rank, world)
        if rank == 0:
            out.setdefault("extra", {})["config4_10Mx768_sharded_batch64_k10"] = c4
        if rank == 0 and world == 1:   # the one-process shape over every GPU this process sees (8 on the driver's scaling node)
            out["extra"]["config4_10Mx768_one_process_cx_sharded_batch64_k10"] = config4_one_process_leg(L, torch.cuda.device_count())
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], extra = cpu_baseline(ix, gen, queries, n, d, k, args.cpu_seconds, args.recall_queries)
        out.setdefault("extra", {}).update(extra)
    del gen
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out.setdefault("extra", {})["config1_10k_x_384_k5"] = config1_leg(L, local_rank)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and (n, d) != (1_000_000, 384):
        out.setdefault("extra", {})["config2_1M_x_384_k10"] = config2_leg(L, local_rank, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_autolink and B == 1:
        out.setdefault("extra", {})["config4_shard_1.25Mx768_batch64_k10"] = config4_leg(L, local_rank, dev)
        # the same shard at the reference's default embedding width (384-d, embedding.rs:43-50) and with the linker's lists
        out["extra"]["batch64_1.25Mx384_k10"] = config4_leg(L, local_rank, dev, d=384, steps=100,
                                                            shard_note="the reference's default width; one of 8 shards of 10M rows")
        out["extra"]["batch128_1.25Mx384_k10"] = config4_leg(L, local_rank, dev, d=384, B=128, steps=100,
                                                             shard_note="the reference's default width; 128 queries per call = ONE pass of two 64-query banks (row widths up to 512)")
        out["extra"]["batch64_1.25Mx384_k100"] = config4_leg(L, local_rank, dev, d=384, k=100, steps=100,
                                                             shard_note="the reference's default width, the linker's list length")
        out["extra"]["batch64_1.25Mx768_k100"] = config4_leg(L, local_rank, dev, k=100, steps=60,
                                                             shard_note="one of 8 shards of 10M rows, the linker's list length")
    if rank == 0 and world == 1 and not args.no_autolink:
        out.setdefault("extra", {})["autolink_allpairs"] = autolink_leg(L, local_rank, d, args.no_cpu_baseline)
        # the same pass over the bench corpus itself: BASELINE.json's metric names "auto-link pairs/sec at 1Mx768"
        out["extra"]["autolink_allpairs_bench_corpus"] = autolink_on_index(ix, n, d)
        mp = out["extra"].get("measured_mfma_bf16_TFLOPs")
        for leg in ("autolink_allpairs", "autolink_allpairs_bench_corpus"):
            r = out["extra"][leg]["roofline"]
            cl = out["extra"].get("measured_mfma_bf16_lds_fed_with_dma_TFLOPs")
            if cl:
                r["main_loop_ceiling_TFLOPs"] = cl
                r["frac_of_main_loop_ceiling"] = r["achieved"] / cl
            if mp:
                r["measured_peak_TFLOPs"] = mp
                r["frac_of_measured"] = r["achieved"] / mp
    if rank == 0 and world == 1 and not args.no_autolink and not args.no_cpu_baseline and B == 1:
        out["extra"]["config5_shard_6.25Mx1024_streaming_ingest"] = config5_leg(L, local_rank, dev)
        # config 5's row width through search_batch: the batched kernel for the widths batch2 has no instance for
        out["extra"]["config5_shard_6.25Mx1024_batch64_k10"] = config4_leg(L, local_rank, dev, n=6_250_000, d=1024, steps=10,
                                                                           shard_note="one of 8 shards of config 5's 50M rows, f32 store")
        out["extra"]["config5_shard_6.25Mx1024_bf16_batch64_k10"] = config4_leg(L, local_rank, dev, n=6_250_000, d=1024, steps=10, dtype="bf16",
                                                                                shard_note="one of 8 shards of config 5's 50M rows, bf16 store")
    if rank == 0:
        print(json.dumps(out), flush=True)
    ix.close()
    if world > 1:
        dist.destroy_process_group()


def single_screened_leg(ix, knn, step, queries: torch.Tensor, n: int, d: int, k: int, steps: int, warmup: int) -> dict:
    """The routed option CX_SINGLE_SCREENED=1 (index.cpp: search_core): a single query through the batched search's screening
    pass — the 2-byte normalised shadow streamed once (n x d x 2 bytes: half the f32 rows), survivors re-scored exactly — for
    callers that search node by node (the linker's per-node loop, auto_linker.rs:215-222; the HTTP handler).  NOT `value`: the
    headline stays on scan_kernel and the contract's bytes.  Checked here: every held-out query's list against the scan's."""
    os.environ["CX_SINGLE_SCREENED"] = "1"
    try:
        for i in range(max(64, warmup)):      # (the first call builds the shadow)
            step(i)
        knn.flush()
        torch.cuda.synchronize()
        ix.profile_read(reset=True)
        ix.profile_enable(True)
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        knn.flush()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ix.profile_enable(False)
        km, kn = ix.profile_read(reset=True)
        qs_h = queries.cpu().numpy()
        scr = [ix.search_arrays(q, k) for q in qs_h]
        os.environ["CX_SINGLE_SCREENED"] = "0"
        same = near = 0
        worst = 0.0
        for q, (gi, gs, gd) in zip(qs_h, scr):
            si, ss, sd = ix.search_arrays(q, k)
            worst = max(worst, float(np.max(np.abs(gs.astype(np.float64) - ss.astype(np.float64)))) if len(gs) == len(ss) and len(gs) else 1.0)
            if len(gs) == len(ss) and np.array_equal(gi, si):
                same += 1
            elif len(gs) == len(ss) and all(abs(float(a) - float(b)) <= 5e-5 for a, b in zip(gs, ss)):
                near += 1
    finally:
        os.environ["CX_SINGLE_SCREENED"] = "0"
    by = float(n) * d * 2.0
    avg = km / max(1, kn)
    return {"what": "CX_SINGLE_SCREENED=1: one query per step through batchs_kernel (bf16 screening of the normalised shadow + exact f32 re-score); "
                    "a routed option, off by default", "queries_per_s": steps / el, "ms_per_step": el / steps * 1e3,
            "bytes_streamed_per_query": by, "stored_row_bytes_per_query": float(n) * d * 4.0,
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "kernel": f"cx::batchs_kernel<{d}>", "avg_kernel_ms": avg, "launches": kn,
                         "achieved": by / (avg * 1e-3) / 1e9 if avg else 0.0, "frac": by / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS if avg else 0.0,
                         "frac_step": by / (el / steps) / 1e9 / HBM_PEAK_GBS},
            "lists_vs_scan": {"queries": len(qs_h), "identical_ids": same, "near_tie_only": near, "max_abs_score_diff": worst}}


def cpu_baseline(ix, gen: torch.Tensor, queries: torch.Tensor, n: int, d: int, k: int, budget_s: float, recall_queries: int = 128):
    """The reference's brute-force path (vector/index.rs:259-294) as restated in oracle/, timed on
    this box's host cores on the same corpus and queries; also the recall/parity of the GPU answers."""
    from oracle import oracle as O
    rows_h = gen.cpu().numpy()
    qs_h = queries.cpu().numpy()
    o = O.OracleIndex(d)
    o.insert_batch(synth_ids(0, n), rows_h)
    del rows_h
    # 1 thread: what HnswIndex::search does per call when no HNSW graph exists
    t0 = time.perf_counter()
    done, res = 0, []
    while done < 64 and (time.perf_counter() - t0 < budget_s or done < 2):
        res.append(o.search(qs_h[done], k))
        done += 1
    t1 = time.perf_counter() - t0
    # GPU answers for the same queries: recall@k vs the exact oracle list, near-ties (5e-5) interchangeable
    hits, tot, max_ds = 0, 0, 0.0
    for i in range(done):
        gi, gs, gd = ix.search_arrays(qs_h[i], k)
        g_rows = gi[:, 8:].copy().view(">u8").reshape(-1).astype(np.int64)
        e = res[i]
        kth = float(e["score"][-1])
        for r, s in zip(g_rows, gs):
            tot += 1
            if r in set(int(x) for x in e["row"]) or abs(float(s) - kth) <= 5e-5:
                hits += 1
        max_ds = max(max_ds, float(np.max(np.abs(gs.astype(np.float64) - e["score"].astype(np.float64)))))
    # the boundary as the reference calls it: host query in, host results out (3 KB H2D + k results D2H + sync)
    t4 = time.perf_counter()
    reps = 200
    for i in range(reps):
        ix.search_arrays(qs_h[i % len(qs_h)], k)
    host_api_qps = reps / (time.perf_counter() - t4)
    # all host cores across queries: the reference's search_batch (rayon par_iter, :390-410)
    # the GPU box's CPU share is 16 hardware threads per GPU (os.cpu_count() reports the whole host)
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    nb = max(1, min(recall_queries, len(qs_h)))
    t2 = time.perf_counter()
    exact_b = o.search_batch(qs_h[:nb], k, n_threads=cores)
    t3 = time.perf_counter() - t2
    # recall@k over that larger sample, GPU answers from ONE search_batch call (the batched kernel)
    bi, bs, bd, bc = ix.search_batch_arrays(qs_h[:nb], k)
    bh = bt = 0
    for i in range(nb):
        want = set(int(x) for x in exact_b[i]["row"])
        kth = float(exact_b[i]["score"][-1]) if len(exact_b[i]) else 0.0
        g_rows = bi[i, :int(bc[i]), 8:].copy().view(">u8").reshape(-1).astype(np.int64)
        for r, sc in zip(g_rows, bs[i, :int(bc[i])]):
            bt += 1
            if int(r) in want or abs(float(sc) - kth) <= 5e-5:
                bh += 1
    # ... and over ALL the held-out queries the engine checks itself: the batched kernel's lists (16 calls of 64) against the
    # single-query scan's, id for id (two different kernels and merges; near-ties within 5e-5 interchangeable)
    agree = checked = 0
    worst = 0.0
    for q0 in range(0, len(qs_h), 64):
        qb = qs_h[q0:q0 + 64]
        bi2, bs2, bd2, bc2 = ix.search_batch_arrays(qb, k)
        for i in range(len(qb)):
            si, ss, sd = ix.search_arrays(qb[i], k)
            m = int(bc2[i])
            rows_b = bi2[i, :m, 8:].copy().view(">u8").reshape(-1).astype(np.int64)
            rows_s = si[:, 8:].copy().view(">u8").reshape(-1).astype(np.int64)
            checked += 1
            worst = max(worst, float(np.max(np.abs(bs2[i, :m].astype(np.float64) - ss[:m].astype(np.float64)))) if m == len(ss) and m else 0.0)
            if m == len(ss) and (np.array_equal(rows_b, rows_s) or all(abs(float(a) - float(b)) <= 5e-5 for a, b in zip(bs2[i, :m], ss))):
                agree += 1
    # the reference's approximate path (index.rs:342-371 over instant-distance 0.6.1) at the headline's width: the first
    # 50k rows of the corpus, built on all host cores like the reference's rayon build (index.rs:430; 100k rows took 99 s
    # on the GPU box's 16 cores, the full 1M would take half an hour), queried one at a time (Cortex::search) and as one parallel batch
    # (search_batch, index.rs:390-410); recall against the exact oracle on the same slice
    hn = min(n, 50_000)
    rows_s = gen[:hn].cpu().numpy()
    t5 = time.perf_counter()
    hidx = O.HnswBaseline(rows_s, n_threads=cores)
    t_build = time.perf_counter() - t5
    os_ = O.OracleIndex(d)
    os_.insert_batch(synth_ids(0, hn), rows_s)
    nqh = min(128, len(qs_h))
    t6 = time.perf_counter()
    ann = [hidx.search(qs_h[i], k, 100) for i in range(nqh)]
    t_ann = time.perf_counter() - t6
    t7 = time.perf_counter()
    ann_rows, _, ann_cnt = hidx.search_batch(qs_h[:nqh], k, 100, n_threads=cores)
    t_ann_mt = time.perf_counter() - t7
    ex_s = os_.search_batch(qs_h[:nqh], k, n_threads=cores)
    rec_h = sum(len(set(ann[i][0].tolist()) & set(ex_s[i]["row"].tolist())) for i in range(nqh)) / float(nqh * k)
    del hidx, os_, rows_s
    base = {
        "value": done / t1, "unit": "queries/s", "cores": 1, "kind": "port",
        "sample": f"{done} queries, full {n} x {d} corpus, oracle brute force (-O2, no FMA, sequential f32), 1 thread",
    }
    extra = {
        "recall_at_k_vs_exact": hits / max(1, tot), "max_abs_score_diff_vs_oracle": max_ds,
        "recall_at_k_vs_exact_search_batch": {"value": bh / max(1, bt), "queries": nb,
                                              "note": f"GPU search_batch against the CPU oracle's exact lists on {nb} of the {len(qs_h)} held-out queries (--recall-queries 1024: all)"},
        "single_vs_batch_self_check": {"queries": checked, "identical_or_near_tie_lists": agree, "max_abs_score_diff": worst,
                                       "note": "every held-out query through the single-query scan and through the batched kernel: the two paths' top-k lists"},
        "host_api_pcie_inclusive_qps": host_api_qps,
        "cpu_all_cores": {"value": nb / t3, "unit": "queries/s", "cores": cores,
                          "sample": f"{nb} queries in one search_batch, {cores} threads"},
        "cpu_hnsw_restatement": {"value": nqh / t_ann, "unit": "queries/s", "cores": 1, "recall_at_k_vs_exact": rec_h,
                                 "queries_per_s_all_cores": nqh / t_ann_mt, "build_s": t_build, "build_cores": cores,
                                 "sample": f"first {hn} rows of the corpus x {d}, {nqh} queries, k={k}",
                                 "params": "M=32 M0=64 ef_construction=100 ef_search=100",
                                 "note": "restatement of instant-distance 0.6.1 from the HNSW paper, concurrent build; parity unpinned; "
                                         "a slice of the corpus: build time grows a little faster than linearly"},
    }
    return base, extra


def config1_leg(L, device: int, n: int = 10_000, d: int = 384, k: int = 5, nq: int = 200):
    """BASELINE config 1 (10k x 384, Cortex::search k=5): the GPU engine through the host API next to the
    CPU paths the reference can take — exact brute force (the oracle) and HNSW (from-the-paper restatement of
    instant-distance 0.6.1 with M=32, M0=64, ef_construction=100, ef_search=100; parity unpinned)."""
    import cortex_amd
    from oracle import oracle as O
    rows = O.synth_rows(n, d)
    qs = O.synth_queries(n, d, nq)
    ids = synth_ids(0, n)
    ix = cortex_amd.HipIndex(d, device=device)
    ix.insert_batch(ids, rows)
    ix.search_arrays(qs[0], k)
    t0 = time.perf_counter()
    got = [ix.search_arrays(q, k) for q in qs]
    t_gpu = time.perf_counter() - t0
    # the same engine through search_batch (index.rs:390-410): 1024 queries in one call, host buffers both ways
    qb = O.synth_queries(n, d, 1024)
    ix.search_batch_arrays(qb, k)
    t0 = time.perf_counter()
    for _ in range(5):
        bi, bs, bd, bc = ix.search_batch_arrays(qb, k)
    t_batch = (time.perf_counter() - t0) / 5
    same = sum(int(np.array_equal(bi[i, :k], ix.search_arrays(qb[i], k)[0])) for i in range(32))
    o = O.OracleIndex(d)
    o.insert_batch(ids, rows)
    t0 = time.perf_counter()
    exact = [o.search(q, k) for q in qs[:50]]
    t_bf = time.perf_counter() - t0
    t0 = time.perf_counter()
    h = O.HnswBaseline(rows)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    ann = [h.search(q, k, 100) for q in qs]
    t_ann = time.perf_counter() - t0
    rec_h = sum(len(set(ann[i][0].tolist()) & set(exact[i]["row"].tolist())) for i in range(50)) / (50.0 * k)
    g_rows = [g[0][:, 8:].copy().view(">u8").reshape(-1).astype(np.int64) for g in got[:50]]
    rec_g = sum(len(set(g_rows[i].tolist()) & set(exact[i]["row"].tolist())) for i in range(50)) / (50.0 * k)
    ix.close()
    return {"gpu_host_api_qps": nq / t_gpu, "gpu_recall_at_5_vs_exact": rec_g,
            "gpu_search_batch_host_api_qps": 1024 / t_batch, "gpu_search_batch_agrees_with_single": f"{same}/32 id lists",
            "cpu_brute_force_qps_1thread": 50 / t_bf,
            "cpu_hnsw_restatement": {"qps_1thread": nq / t_ann, "recall_at_5_vs_exact": rec_h, "build_s": t_build,
                                     "params": "M=32 M0=64 ef_construction=100 ef_search=100",
                                     "note": "restatement of instant-distance 0.6.1 from the HNSW paper; parity unpinned"}}


def batch_roofline(n: int, d: int, store_elem_bytes: float, avg_kernel_ms: float, step_s: float, kern_n: int, note: str = "") -> dict:
    """HBM roofline object of a batched-search leg.  Stores of >= 131,072 rows go through batchs.hip: one launch streams
    the index's normalised bf16 shadow of the shard once — the all-pairs filter's tiled copy: n x d x 2 bytes — and the
    survivors (a few hundred per query) are re-scored exactly from the stored rows.  `achieved` / `frac` are those bytes
    over the screening kernel's duration (HIP events around the launch), `frac_step` the same bytes over the whole step
    (re-score, selection, gaps).  The contract's figure (the shard's stored bytes per batch) over the same
    durations is given beside it as store_equivalent_*: it can exceed the peak because the pass reads half of them (f32
    stores)."""
    algo = float(n) * d * 2.0
    store = float(n) * d * store_elem_bytes
    avg = avg_kernel_ms
    return {"bound": "hbm", "achieved": algo / (avg * 1e-3) / 1e9 if avg else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": algo / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS if avg else 0.0, "kernel": f"cx::batchs_kernel<{d}>", "avg_kernel_ms": avg,
            "launches": kern_n, "algorithmic_bytes_per_launch": algo,
            "step_achieved": algo / step_s / 1e9, "frac_step": algo / step_s / 1e9 / HBM_PEAK_GBS,
            "store_bytes_per_launch": store, "store_equivalent_GBs": store / (avg * 1e-3) / 1e9 if avg else 0.0,
            "store_equivalent_step_GBs": store / step_s / 1e9,
            "frac_note": "frac: the screening kernel's launches alone over the bytes it streams (the 2-byte normalised shadow); frac_step: the same "
                         "bytes over the whole step (exact re-score of the survivors, selection, gaps); store_equivalent_*: the stored rows' "
                         "bytes (the contract's per-batch figure) over the same durations" + (("; " + note) if note else "")}


def config4_leg(L, device: int, dev, n: int = 1_250_000, d: int = 768, k: int = 10, B: int = 64, steps: int = 40,
                shard_note: str = "one of 8 shards of 10M rows", dtype: str = "f32"):
    """BASELINE configs[3], one GPU's share: 64 queries per step over a 1.25M x 768 f32 shard (10M rows / 8 GPUs),
    k=10 — the batched MFMA kernel the sharded search runs before its all-gather; same measurement as the headline
    (cx_search_batch_dev, HIP events around the kernel)."""
    import cortex_amd
    gen = torch.empty((n, d), dtype=torch.float32, device=dev)
    assert L.cx_synth_fill_dev(device, gen.data_ptr(), SEED_CORPUS, SEED_CORPUS, SEED_DUP, n // 50, 0, n, d, 1) == 0
    ix = cortex_amd.HipIndex(d, device=device, dtype=dtype)
    ix.reserve(n)
    ix.insert_batch_dev(synth_ids(0, n), gen.data_ptr(), n, d)
    del gen
    qs = torch.empty((256, d), dtype=torch.float32, device=dev)
    assert L.cx_synth_fill_dev(device, qs.data_ptr(), SEED_CORPUS, SEED_QUERIES, SEED_DUP, n // 50, 0, 256, d, 0) == 0
    o_rows = torch.empty((B, k), dtype=torch.int32, device=dev)
    o_sc = torch.empty((B, k), dtype=torch.float32, device=dev)
    o_di = torch.empty((B, k), dtype=torch.float32, device=dev)
    o_cnt = torch.empty(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def one(i):
        ix.search_batch_dev(qs.data_ptr() + ((i * B) % (256 - B + 1)) * d * 4, B, k, o_rows.data_ptr(), o_sc.data_ptr(),
                            o_di.data_ptr(), o_cnt.data_ptr(), stream)
    for i in range(5):
        one(i)
    torch.cuda.synchronize()
    ix.profile_read(reset=True)
    ix.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(steps):
        one(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ix.profile_enable(False)
    kern_ms, kern_n = ix.profile_read(reset=True)
    # the same shard the way a rank of the sharded search runs it (ShardedKnn.submit, world = 1 here: no all-gather): a stream of
    # batches on rotating HIP streams, each batch's re-score / selection and the next one's pass free to overlap — `frac_step`
    # above is ONE stream, every kernel of a step behind the previous one
    from cortex_amd.sharded import ShardedKnn, hip_local_fn
    knn = ShardedKnn(0, 1, [0], B, k, dev, hip_local_fn(ix))
    for i in range(8):
        knn.submit(qs.data_ptr() + ((i * B) % (256 - B + 1)) * d * 4)
    knn.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        knn.submit(qs.data_ptr() + ((i * B) % (256 - B + 1)) * d * 4)
    knn.flush()
    torch.cuda.synchronize()
    el_s = time.perf_counter() - t0
    ix.close()
    avg = kern_ms / max(1, kern_n)
    roof = batch_roofline(n, d, 2.0 if dtype == "bf16" else 4.0, avg, el / steps, kern_n)
    roof["frac_step_stream_of_batches"] = roof["algorithmic_bytes_per_launch"] / (el_s / steps) / 1e9 / HBM_PEAK_GBS
    if B > 64:
        roof["frac_note"] += ("; a pass of 128 queries (two 64-query banks, row widths up to 512) streams the same bytes as a pass of 64 and takes "
                              "~1.3x as long (16 MFMAs per K-step and wave instead of 8): the fractions are per launch — compare queries_per_s "
                              "with the 64-query leg of the same shard (two one-bank passes for 128 queries: 0.427 ms against 0.263, "
                              "profiles/r04/tuning.md 1.7)")
    roof["frac_note"] += ("; frac_step_stream_of_batches: the same bytes over a step of a STREAM of batches on rotating HIP streams "
                          f"(ShardedKnn.submit: what a rank of the sharded search runs; {len(knn._streams) if knn._streams else 1} streams — "
                          "cx_search_batch_streams_hint)")
    return {"workload": f"cosine kNN k={k}, batch of {B} queries per step, {n} x {d} {dtype} rows ({shard_note})",
            "queries_per_s": steps * B / el, "ms_per_step": el / steps * 1e3,
            "stream_of_batches": {"queries_per_s": steps * B / el_s, "ms_per_step": el_s / steps * 1e3},
            "roofline": roof}


def config4_sharded_leg(L, device: int, dev, rank: int, world: int, total: int = 10_000_000, d: int = 768, k: int = 10,
                        B: int = 64, steps: int = 20, warmup: int = 5):
    """BASELINE configs[3] as it is stated: a FIXED 10M x 768 f32 corpus row-sharded over the N ranks (rank r owns rows
    [r*10M/N, (r+1)*10M/N)), batches of 64 queries, k = 10, one all-gather of the packed partial top-k lists + merge per
    batch — strong scaling, so the >= 6x target from 1 to 8 GPUs can be read off the per-N lines.  Same protocol as the
    headline: barrier + synchronize on both sides, max over ranks.  At N = 1 the whole corpus (30.7 GB + the batched
    kernel's split store of the same size) sits on the one GPU."""
    import cortex_amd
    from cortex_amd.sharded import ShardedKnn, hip_local_fn
    per = total // world
    n = per if rank < world - 1 else total - per * (world - 1)
    row_lo = rank * per
    ix = cortex_amd.HipIndex(d, device=device)
    ix.reserve(n)
    chunk = 1_000_000
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        gen = torch.empty((m, d), dtype=torch.float32, device=dev)
        assert L.cx_synth_fill_dev(device, gen.data_ptr(), SEED_CORPUS, SEED_CORPUS, SEED_DUP, total // 50, row_lo + lo, m, d, 1) == 0
        ix.insert_batch_dev(synth_ids(row_lo + lo, m), gen.data_ptr(), m, d)
        del gen
    qs = torch.empty((256, d), dtype=torch.float32, device=dev)
    assert L.cx_synth_fill_dev(device, qs.data_ptr(), SEED_CORPUS, SEED_QUERIES, SEED_DUP, total // 50, 0, 256, d, 0) == 0
    knn = ShardedKnn(rank, world, [r * per for r in range(world)], B, k, dev, hip_local_fn(ix))

    def step(i):
        knn.submit(qs.data_ptr() + ((i * B) % (256 - B + 1)) * d * 4)
    for i in range(8 + warmup):      # the first call also builds the norm cache and the split store
        step(i)
    knn.flush()
    torch.cuda.synchronize()
    # the pass's own duration (the kernel-level `frac`): a few batches one at a time — in the timed region below the batches of
    # the stream overlap on several HIP streams (ShardedKnn.submit) and an event pair around one pass spans its neighbours too
    ix.profile_read(reset=True)
    ix.profile_enable(True)
    for i in range(6):
        step(i)
        knn.flush()
        torch.cuda.synchronize()
    ix.profile_enable(False)
    kern_ms, kern_n = ix.profile_read(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    knn.flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ix.close()
    torch.cuda.empty_cache()
    avg = kern_ms / max(1, kern_n)
    return {"workload": f"cosine kNN k={k}, batches of {B} queries, ONE {total} x {d} f32 corpus row-sharded over {world} GPU(s) "
                        f"({n} rows on this rank), all-gather of partial top-k + merge per batch",
            "scaling": "strong", "n_gpus": world, "queries_per_s": steps * B / el, "ms_per_step": el / steps * 1e3, "steps": steps,
            "roofline": batch_roofline(n, d, 4.0, avg, el / steps, kern_n,
                                       "rank 0's shard; one launch reads the rank's shadow once for 64 queries; frac: passes run one at a time "
                                       "before the timed region; frac_step: the timed stream of batches (several in flight on "
                                       "rotating HIP streams), all-gather and merge included")}


def config4_one_process_leg(L, n_dev: int, total: int = 10_000_000, d: int = 768, k: int = 10, B: int = 64, steps: int = 20, warmup: int = 5):
    """BASELINE configs[3] in the OTHER deployment shape: the reference's host is ONE process holding ONE index
    (serve.rs:101), so the drop-in shards under the extern-"C" boundary — cx_sharded over every GPU this process sees
    (one shard per device: per-shard enqueue threads, peer-to-peer publish of the partial lists, merge on the first
    device), called like the trait's search_batch: host queries in, host ids out (PCIe inclusive).  With one visible
    GPU it is one shard holding all 10M rows."""
    import cortex_amd
    devs = list(range(max(1, n_dev)))
    sh = cortex_amd.ShardedHipIndex(d, devs)
    dev0 = torch.device("cuda", 0)
    chunk = 500_000
    for lo in range(0, total, chunk):
        m = min(chunk, total - lo)
        gen = torch.empty((m, d), dtype=torch.float32, device=dev0)
        assert L.cx_synth_fill_dev(0, gen.data_ptr(), SEED_CORPUS, SEED_CORPUS, SEED_DUP, total // 50, lo, m, d, 1) == 0
        try:
            sh.insert_batch_dev(synth_ids(lo, m), gen.data_ptr(), m, d)
        except cortex_amd.CortexError:
            sh.insert_batch(synth_ids(lo, m), gen.cpu().numpy())   # no peer access from device 0: through the host
        del gen
    qd = torch.empty((256, d), dtype=torch.float32, device=dev0)
    assert L.cx_synth_fill_dev(0, qd.data_ptr(), SEED_CORPUS, SEED_QUERIES, SEED_DUP, total // 50, 0, 256, d, 0) == 0
    qs = qd.cpu().numpy()
    for i in range(3 + warmup):
        sh.search_batch_arrays(qs[(i * B) % (256 - B + 1):][:B], k)
    t0 = time.perf_counter()
    for i in range(steps):
        sh.search_batch_arrays(qs[(i * B) % (256 - B + 1):][:B], k)
    el = time.perf_counter() - t0
    p2p = bool(sh.peer_to_peer)
    sh.close()
    torch.cuda.empty_cache()
    by = float(total) * d * 2.0   # the shards' normalised bf16 shadows (batchs.hip), streamed once per batch
    return {"workload": f"cosine kNN k={k}, batches of {B} queries, ONE {total} x {d} f32 corpus behind cx_sharded_search_batch in one process, "
                        f"one shard on each of {len(devs)} visible GPU(s); host queries in, host ids out",
            "scaling": "strong", "n_gpus": len(devs), "peer_to_peer": p2p, "queries_per_s": steps * B / el, "ms_per_step": el / steps * 1e3, "steps": steps,
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS * len(devs), "achieved": by / (el / steps) / 1e9,
                         "frac": by / (el / steps) / 1e9 / (HBM_PEAK_GBS * len(devs)),
                         "store_equivalent_step_GBs": 2.0 * by / (el / steps) / 1e9,
                         "note": "whole step (host API, PCIe and merge included): the bytes the screening passes stream (total x d x 2) over the "
                                 "aggregate HBM peak of the GPUs used; store_equivalent: the f32 rows' bytes over the same time"}}


def config5_leg(L, device: int, dev, n: int = 6_250_000, d: int = 1024, thr: float = 0.85):
    """BASELINE configs[4], one GPU's share: a 6.25M x 1024 bf16 shard (50M rows / 8 GPUs: 12.8 GB of rows), streaming
    auto-link ingest as the reference runs it (auto_linker.rs:378-398 insert, then :220-221 search per new node): every
    TICK inserts a batch of NEW rows (cx_upsert_batch_dev: rounded to bf16 once, on the device), extends the normalised
    shadow and its tiled copy by exactly those rows, and links exactly those rows against the whole shard
    (cx_autolink_pass_timed).  Reported per tick: upsert, shadow extension, filter, exact rescore, rules, and the
    roofline fraction of the WHOLE tick — HBM for batches of 64 (the filter streams the shard's shadow once: n x d x 2
    bytes), MFMA for batches of 500 (two 256-row panels against the shard)."""
    import cortex_amd
    ticks = 7
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    mem_before = torch.cuda.mem_get_info(dev)[0]
    ix = cortex_amd.HipIndex(d, device=device, dtype="bf16")
    ix.reserve(n + ticks * (64 + 500) + 1024)
    chunk = 1_000_000
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        gen = torch.empty((m, d), dtype=torch.float32, device=dev)
        assert L.cx_synth_fill_dev(device, gen.data_ptr(), SEED_CORPUS, SEED_CORPUS, SEED_DUP, n // 50, lo, m, d, 1) == 0
        ix.insert_batch_dev(synth_ids(lo, m), gen.data_ptr(), m, d)
        del gen
    t = float(np.float32(thr))
    ix.autolink_pass_timed(100, t, 50, np.arange(n - 64, n, dtype=np.uint32))   # builds the shadow of the resident shard
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    shard_gb = (mem_before - torch.cuda.mem_get_info(dev)[0]) / 1e9
    out = {"workload": f"streaming auto-link ingest into a {n} x {d} bf16 shard ({n * d * 2 / 1e9:.1f} GB of rows + the normalised bf16 shadow the "
                       f"filter streams), threshold {thr}, top-100, 50 edges/node; a tick = insert a batch of new rows + extend the shadow + link the batch",
           "storage_dtype": "bf16", "ticks_per_batch_size": ticks,
           "shard_device_memory_GB": shard_gb, "shard_device_memory_note": "rows (bf16) + ONE normalised shadow in the filter kernels' tiled layout + norms, ids and scratch; "
                                                                           "rounds 1-2 kept a row-major shadow as well (+ 12.8 GB)"}
    cur = n
    for b in (64, 500):
        rec = []
        for tick in range(ticks):
            gen = torch.empty((b, d), dtype=torch.float32, device=dev)
            assert L.cx_synth_fill_dev(device, gen.data_ptr(), SEED_CORPUS, SEED_CORPUS, SEED_DUP, n // 50, cur, b, d, 1) == 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ix.insert_batch_dev(synth_ids(cur, b), gen.data_ptr(), b, d)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ne, ph = ix.autolink_pass_timed(100, t, 50, np.arange(cur, cur + b, dtype=np.uint32))
            t2 = time.perf_counter()
            prof = ix.autolink_filter_profile()
            cur += b
            rec.append({"upsert_ms": (t1 - t0) * 1e3, "pass_wall_ms": (t2 - t1) * 1e3, "tick_ms": (t2 - t0) * 1e3, "shadow_extend_ms": ph[0],
                        "filter_ms": ph[1], "rescore_ms": ph[2], "rules_ms": ph[3], "edges": int(ne), "rows_in_shard": cur, "prof": prof})
        # the MEDIAN tick (the first tick of a batch size pays its scratch allocations and is left out); every tick's phases
        # are in all_ticks
        later = sorted(rec[1:], key=lambda r: r["tick_ms"])
        best = dict(later[len(later) // 2])
        prof = best.pop("prof")
        rows_now = best["rows_in_shard"]
        tick_s = best["tick_ms"] * 1e-3
        if b <= 128:
            by = rows_now * d * 2.0
            roof = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "kernel": "cx::batchs_kernel<1024, true> (threshold mode; shards below 131,072 rows: cx::pair_filter_stream_kernel)",
                    "achieved": by / (best["filter_ms"] * 1e-3) / 1e9, "frac": by / (best["filter_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "whole_tick_achieved": by / tick_s / 1e9, "frac_of_whole_tick": by / tick_s / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_tick": by}
        else:
            fl = prof["executed_flops"] if prof["executed_flops"] else 2.0 * 512 * rows_now * d
            ksec = (prof["kernel_ms"] or best["filter_ms"]) * 1e-3
            roof = {"bound": "mfma", "unit": "TFLOP/s", "peak": 2500.0, "kernel": prof["kernel"], "achieved": fl / ksec / 1e12, "frac": fl / ksec / 2.5e15,
                    "avg_kernel_ms": ksec * 1e3, "whole_tick_achieved": fl / tick_s / 1e12, "frac_of_whole_tick": fl / tick_s / 2.5e15,
                    "executed_flops_per_tick": fl}
        best["pairs_per_s"] = b * float(rows_now) / tick_s
        best["roofline"] = roof
        best["reported_tick"] = "median of ticks 2.." + str(len(rec))
        best["all_ticks_ms"] = [round(r["tick_ms"], 3) for r in rec]
        best["all_ticks"] = [{kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in r.items() if kk != "prof"} for r in rec]
        out[f"batch_{b}"] = best
    ix.close()
    return out


def config2_leg(L, device: int, dev, n: int = 1_000_000, d: int = 384, k: int = 10, steps: int = 300):
    """BASELINE configs[1] (1M x 384 f32, cosine kNN k=10, single query per step, HBM-resident): the headline
    workload at the other embedding width, same measurement (cx_search_dev, HIP events around every scan)."""
    import cortex_amd
    gen = torch.empty((n, d), dtype=torch.float32, device=dev)
    assert L.cx_synth_fill_dev(device, gen.data_ptr(), SEED_CORPUS, SEED_CORPUS, SEED_DUP, n // 50, 0, n, d, 1) == 0
    ix = cortex_amd.HipIndex(d, device=device)
    ix.reserve(n)
    ix.insert_batch_dev(synth_ids(0, n), gen.data_ptr(), n, d)
    del gen
    qs = torch.empty((64, d), dtype=torch.float32, device=dev)
    assert L.cx_synth_fill_dev(device, qs.data_ptr(), SEED_CORPUS, SEED_QUERIES, SEED_DUP, n // 50, 0, 64, d, 0) == 0
    o_rows = torch.empty(k, dtype=torch.int32, device=dev)
    o_sc = torch.empty(k, dtype=torch.float32, device=dev)
    o_di = torch.empty(k, dtype=torch.float32, device=dev)
    o_cnt = torch.empty(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def one(i):
        ix.search_batch_dev(qs.data_ptr() + (i % 64) * d * 4, 1, k, o_rows.data_ptr(), o_sc.data_ptr(), o_di.data_ptr(),
                            o_cnt.data_ptr(), stream)
    for i in range(30):
        one(i)
    torch.cuda.synchronize()
    ix.profile_read(reset=True)
    ix.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(steps):
        one(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ix.profile_enable(False)
    kern_ms, kern_n = ix.profile_read(reset=True)
    ix.close()
    avg = kern_ms / max(1, kern_n)
    algo = float(n) * d * 4.0
    return {"workload": f"cosine kNN k={k}, single query per step, {n} x {d} f32 rows (exact brute force, HBM-resident)",
            "queries_per_s": steps / el, "ms_per_step": el / steps * 1e3,
            "roofline": {"bound": "hbm", "achieved": algo / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algo / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": "cx::scan_kernel", "avg_kernel_ms": avg,
                         "launches": kern_n, "algorithmic_bytes_per_launch": algo,
                         "step_achieved": algo / (el / steps) / 1e9, "frac_step": algo / (el / steps) / 1e9 / HBM_PEAK_GBS}}


def mfma_roofline(contract_flops: float, prof: dict, phase_ms: float) -> dict:
    """MFMA roofline object of the filter GEMM.  `achieved` / `frac` are the EXECUTED bf16 MFMA rate against the 2.5 PF
    dense peak — matrix-core utilisation, what north_star's ">= 50 % MFMA utilisation" asks about — over the GEMM kernel's
    own duration: HIP events around that one launch on its stream (cx_autolink_filter_profile), the figure rocprofv3's
    kernel trace reports for the same kernel.  `phase_ms` is the whole filter phase of the pass (that kernel + the memsets
    in front of it + pair_scatter_kernel behind it) and `frac_of_phase` the same flops over it.  Cosine is symmetric: only
    tiles with tj >= ti run and each emits both directions, so the contract's figure (2 N^2 d, no symmetry
    credit) per second is ~2x the executed rate; it is a throughput equivalent and is reported as such, not as `frac`."""
    sec = prof["kernel_ms"] * 1e-3
    executed_flops = prof["executed_flops"]
    return {"bound": "mfma", "achieved": executed_flops / sec / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
            "frac": executed_flops / sec / 2.5e15, "kernel": prof["kernel"], "dtype": "bf16 in, f32 accumulate",
            "executed_flops_per_launch": executed_flops, "tiles_per_launch": prof["tiles"], "algorithmic_flops_per_launch": contract_flops,
            "contract_equivalent_TFLOPs": contract_flops / sec / 1e12,
            "contract_equivalent_frac": contract_flops / sec / 2.5e15,
            "avg_kernel_ms": prof["kernel_ms"], "best_kernel_ms": prof.get("best_kernel_ms"), "kernel_ms_every_pass": prof.get("kernel_ms_every_pass"),
            "shader_clock_ghz_in_kernel": prof["shader_clock_ghz"],
            "filter_phase_ms": phase_ms, "frac_of_phase": executed_flops / (phase_ms * 1e-3) / 2.5e15}


def autolink_on_index(ix, n: int, d: int, thr: float = 0.85, reps: int = 2):
    """All-pairs auto-link pass (threshold 0.85, top-100, cap 50) over an index that is already resident;
    first call builds the bf16 shadow and the tile list, the best of the following `reps` is reported."""
    thr32 = float(np.float32(thr))
    best = None
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        ne, ph = ix.autolink_pass_timed(100, thr32, 50)
        wall = time.perf_counter() - t0
        if rep and (best is None or wall < best[0]):
            best = (wall, ph, ne, ix.autolink_filter_profile())
    wall, ph, ne, prof = best
    flops = 2.0 * n * n * d
    return {"workload": f"auto-link all-pairs {n} x {d}, threshold {thr}, top-100, 50 edges/node, similarity rule only",
            "pairs_per_s": n * float(n) / wall, "wall_ms": wall * 1e3, "edges": ne,
            "phase_ms": {"shadow_refresh": ph[0], "mfma_filter_gemm": ph[1], "exact_rescore": ph[2], "link_rules": ph[3]},
            "mode": "first pass over a graph without edges (no existing_set); similarity rule only",
            "roofline": mfma_roofline(flops, prof, ph[1])}


def autolink_leg(L, device: int, d: int, skip_cpu: bool, n: int = 100_000, thr: float = 0.85):
    """BASELINE config 3: auto-link all-pairs over n x d rows, threshold 0.85, top-100, cap 50 edges per
    node (cx_autolink_pass_timed: edges stay in HBM).  MFMA roofline for the filter GEMM; the CPU figure is
    the oracle's restatement of the reference loop (auto_linker.rs:215-264) on a slice of scanned nodes,
    extrapolated linearly in the number of scanned nodes (each is one O(N*d) search)."""
    import cortex_amd
    gen = torch.empty((n, d), dtype=torch.float32, device=torch.device("cuda", device))
    rc = L.cx_synth_fill_dev(device, gen.data_ptr(), SEED_CORPUS, SEED_CORPUS, SEED_DUP, n // 50, 0, n, d, 1)
    assert rc == 0, L.cx_last_error()
    ix = cortex_amd.HipIndex(d, device=device)
    ix.insert_batch_dev(synth_ids(0, n), gen.data_ptr(), n, d)
    thr32 = float(np.float32(thr))
    best = None
    # 12 passes; wall = the best of the last 11, the GEMM kernel's duration = the MEAN of the last 8: after an idle gap the
    # chip's clock ramps over the first ~4 launches of this kernel (1.52 -> 1.76 GHz inside the kernel, tuning.md §1)
    kms = []
    for rep in range(12):
        t0 = time.perf_counter()
        ne, ph = ix.autolink_pass_timed(100, thr32, 50)
        wall = time.perf_counter() - t0
        pf = ix.autolink_filter_profile()
        kms.append(pf["kernel_ms"])
        if rep and (best is None or wall < best[0]):
            best = (wall, ph, ne, pf)
    wall, ph, ne, prof = best
    prof = dict(prof, kernel_ms=float(np.mean(kms[4:])), best_kernel_ms=float(np.min(kms)), kernel_ms_every_pass=[round(x, 4) for x in kms])   # the AVERAGE launch behind the ramp: what a kernel trace averages to
    flops = 2.0 * n * n * d           # the full ordered matrix, no symmetry credit
    res = {
        "workload": f"auto-link all-pairs {n} x {d}, threshold {thr}, top-100, 50 edges/node, similarity rule only",
        "pairs_per_s": n * float(n) / wall, "wall_ms": wall * 1e3, "edges": ne,
        "phase_ms": {"shadow_refresh": ph[0], "mfma_filter_gemm": ph[1], "exact_rescore": ph[2], "link_rules": ph[3]},
        "mode": "first pass over a graph without edges (no existing_set); similarity rule only",
        "roofline": mfma_roofline(flops, prof, ph[1]),
    }
    # The all-pairs case the north star names is the RESCAN after a threshold / model change (auto_linker.rs:137-182):
    # edges exist, the reference drops them without counting (:226-231, :249-258) and walks deeper into each top-100
    # list.  Same pass with the existing-edge CSR of ~30 % of the nodes (what the first pass created for them at the
    # old threshold 0.85), new threshold 0.75, and the reference's per-cycle cap lifted (Q8: the GPU pass does all N).
    fr, to, w = ix.autolink_pass_rows(None, 100, thr32, 50)
    rng = np.random.default_rng(5)
    keep_node = rng.random(n) < 0.3
    sel = keep_node[fr]
    cnt = np.bincount(fr[sel].astype(np.int64), minlength=n).astype(np.uint64)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(cnt)
    existing = (off, to[sel].astype(np.uint32))          # edges come out grouped by from-row in scan order
    thr_new = float(np.float32(0.75))
    best2 = None
    for rep in range(5):
        t0 = time.perf_counter()
        ne2, ph2 = ix.autolink_pass_timed(100, thr_new, 50, None, existing=existing)
        wall2 = time.perf_counter() - t0
        if rep and (best2 is None or wall2 < best2[0]):
            best2 = (wall2, ph2, ne2, ix.autolink_filter_profile())
    wall2, ph2, ne2, prof2 = best2
    res["rescan_with_existing_edges"] = {
        "mode": f"rescan at threshold 0.75 of a graph linked at 0.85: {int(keep_node.sum())} of {n} nodes carry {int(sel.sum())} "
                "existing related_to edges (skipped without counting, auto_linker.rs:249-258); similarity rule only",
        "pairs_per_s": n * float(n) / wall2, "wall_ms": wall2 * 1e3, "edges": int(ne2),
        "phase_ms": {"shadow_refresh": ph2[0], "mfma_filter_gemm": ph2[1], "exact_rescore": ph2[2], "link_rules": ph2[3]},
        "roofline": mfma_roofline(flops, prof2, ph2[1]),
    }
    del fr, to, w
    # the same pass with the scanned nodes given as a LIST (a cycle's batch in the order the storage returns it; the sharded pass's
    # external-query blocks take the same route): the rows' shadow pieces are gathered into a staged panel and the persistent kernel
    # runs on it (round 3: pair_filter256_kernel for every list).  No symmetry credit here: every ordered pair's tile is computed.
    perm = np.random.default_rng(9).permutation(n).astype(np.uint32)
    best3 = None
    for rep in range(4):
        t0 = time.perf_counter()
        ne3, ph3 = ix.autolink_pass_timed(100, thr32, 50, perm)
        wall3 = time.perf_counter() - t0
        if rep and (best3 is None or wall3 < best3[0]):
            best3 = (wall3, ph3, ne3, ix.autolink_filter_profile())
    wall3, ph3, ne3, prof3 = best3
    res["row_list_scan"] = {
        "mode": f"all {n} rows scanned as a shuffled LIST (staged I panel + cx::pair_filter_p_kernel; no symmetry: 2 N^2 d flops executed)",
        "pairs_per_s": n * float(n) / wall3, "wall_ms": wall3 * 1e3, "edges": int(ne3),
        "phase_ms": {"shadow_refresh": ph3[0], "mfma_filter_gemm": ph3[1], "exact_rescore": ph3[2], "link_rules": ph3[3]},
        "roofline": mfma_roofline(flops, prof3, ph3[1]),
    }
    # the ordered top-100 neighbour lists of every row (SURVEY a14': what the linker needs when the reference's legacy
    # structural rules are on — its default): the batched search in its wide mode over the same corpus, host API
    ix.topk_lists_rows(100, np.arange(2048, dtype=np.uint32))
    t0 = time.perf_counter()
    lr, ls, lc = ix.topk_lists_rows(100, None)
    t_lists = time.perf_counter() - t0
    res["top100_lists_all_rows"] = {"seconds": t_lists, "lists_per_s": n / t_lists, "full_lists": int((lc == 100).sum()),
                                    "kernel": "cx::pair_filter_p_kernel at a sampled k-th-best threshold + exact rescore; short / "
                                              "overflowed lists redone by cx::batch2_kernel (wide lists) — autolink.cpp: lists_by_filter",
                                    "round1_seconds_same_call": 0.196}
    del lr, ls, lc
    if not skip_cpu:
        from oracle import oracle as O
        o = O.OracleIndex(d)
        o.insert_batch(synth_ids(0, n), gen.cpu().numpy())
        cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
        m = 4 * cores
        t0 = time.perf_counter()
        e = o.autolink_pass(np.arange(m), 100, thr32, 50, n_threads=cores)
        t = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": m * float(n) / t, "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": f"{m} scanned nodes of {n} (each one brute-force search + rule walk), "
                                         f"{cores} threads; a full pass is {n / m:.0f}x this slice"}
    ix.close()
    return res


if __name__ == "__main__":
    main()
