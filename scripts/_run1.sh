set -e
python -m pytest tests/test_hip_parity.py -q -x -k "search_batch or batched_search" 2>&1 | tail -5
H=scripts/probes/_batchg_harness
for p in 0 2; do echo "v4 dense probe $p"; CX_BATCHG_PROBE=$p $H 1000000 1024 40; done
echo filter; python scripts/bench_batch_dim.py --rows 1000000 --dim 1024 --steps 40 2>/dev/null
echo filter k100; python scripts/bench_batch_dim.py --rows 1000000 --dim 1024 --k 100 --steps 40 2>/dev/null
echo filter 4M; python scripts/bench_batch_dim.py --rows 4000000 --dim 1024 --steps 10 2>/dev/null
echo filter 512; python scripts/bench_batch_dim.py --rows 2000000 --dim 512 --steps 40 2>/dev/null
