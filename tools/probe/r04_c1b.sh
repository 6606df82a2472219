#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_c1b
mkdir -p $out
cd $R
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -k "c1 or t1 or two_row or resnet" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
for g in 4 6 8 12; do
  timeout -k 10 200 python3 bench.py --config C1 --steps 48 --warmup 4 --group $g --no-cpu-baseline --no-secondary > $out/c1_g$g.log 2>&1
  echo "== C1 group $g: $(grep '^{' $out/c1_g$g.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"].get("decode_loops"))')"
done
for ch in 2 4; do
  timeout -k 10 200 python3 bench.py --config C1 --steps 48 --warmup 4 --group 8 --chains $ch --no-cpu-baseline --no-secondary > $out/c1_ch$ch.log 2>&1
  echo "== C1 group 8 chains $ch: $(grep '^{' $out/c1_ch$ch.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
