for b in 1 2; do echo "blocks/CU $b"; CX_BG_BLOCKS_PER_CU=$b python scripts/bench_batch_dim.py --rows 4000000 --dim 1024 --steps 10 2>/dev/null; done
