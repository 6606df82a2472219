#!/bin/bash
# Round 4: the 'mixed' precision on the GPU box -- its tests, its margin against the oracle on fresh seeds, its rate.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_mixed
mkdir -p $out
cd $R
timeout -k 10 600 python3 -m pytest tests/test_fp16x2_gpu.py -x -q -k "mixed" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -5 $out/tests.log
MARGIN_PRECS=${MARGIN_PRECS:-bf16x3,mixed:4,mixed:6,mixed:8,fp16x2} timeout -k 10 900 python3 tools/probe/fp16x2_margin.py ${MARGIN_SPECS:-T2:2:48:64:12:6:3 C2:2:128:512:40:5:2 C4:1:160:640:40:3:2} > $out/margin.log 2>&1; echo "margin rc=$?"
grep -v "^  " $out/margin.log
for cfg in "bf16x3" "mixed --mixed-units 4" "mixed --mixed-units 6" "mixed --mixed-units 8" "bf16x3"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --precision $cfg > $out/bench_$tag.log 2>&1
  echo "== $cfg: $(grep '^{' $out/bench_$tag.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("parity"))')"
done
