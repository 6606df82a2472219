set -e
for cfg in "2 0" "2 16" "2 32" "2 48" "2 64"; do set -- $cfg
python bench.py --steps 12 --warmup 6 --no-cpu-baseline --chains $1 --reserve $2 2>/dev/null | python tools/bench_line.py "chains$1-reserve$2"
done
