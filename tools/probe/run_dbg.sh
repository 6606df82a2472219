set -e
for t in 0 1 2; do for d in 0; do
D2T_BF16X3_TILE=$t D2T_DBG=$d python bench.py --no-pipeline --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "tile$t-dbg$d"
done; done
D2T_BF16X3_TILE=1 D2T_DBG=4 python bench.py --no-pipeline --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "tile1-dbg4"
D2T_BF16X3_TILE=1 D2T_DBG=8 python bench.py --no-pipeline --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "tile1-dbg8"
D2T_BF16X3_TILE=1 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "bf16x3 or split" 2>&1 | tail -2
D2T_BF16X3_TILE=2 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "bf16x3" 2>&1 | tail -2
