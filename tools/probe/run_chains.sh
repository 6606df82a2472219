set -e
for cfg in "2 32" "2 48" "2 64" "2 80" "2 96"; do set -- $cfg
python bench.py --steps 20 --warmup 6 --no-cpu-baseline --chains $1 --reserve $2 2>/dev/null | python tools/bench_line.py "chains$1-reserve$2"
done
