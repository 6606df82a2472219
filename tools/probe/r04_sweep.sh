#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_sweep
mkdir -p $out
cd $R
for cfg in "3 6" "3 8" "2 6" "4 6" "3 5" "2 8" "3 6"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --steps 48 --warmup 6 --chains $1 --group $2 --no-cpu-baseline --no-secondary > $out/c$1_g$2.log 2>&1
  echo "== chains $1 group $2: $(grep '^{' $out/c$1_g$2.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])')"
done
