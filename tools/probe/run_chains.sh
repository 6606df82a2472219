set -e
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "pipelined or headline" 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>/dev/null | python tools/bench_line.py default
python bench.py --steps 20 --warmup 6 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py steps20
