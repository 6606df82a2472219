#!/bin/bash
# Round 4: the prefetching two-row decode kernel -- parity tests, A/B against the round-3 issue order, kernel durations;
# then the 26-seed margin run of the 'mixed' precision at 2 / 3 / 4 units.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_row
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py -x -q -k "greedy or headline or 64_rows or shard or pipelined" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
timeout -k 10 300 python3 -m pytest tests/test_fp16x2_gpu.py -x -q -k "mixed" > $out/tests_mixed.log 2>&1; echo "mixed tests rc=$?"; tail -3 $out/tests_mixed.log
for arm in new old new old; do
  if [ $arm = old ]; then export D2T_DECODE_ROW2_NO_PREFETCH=1; else unset D2T_DECODE_ROW2_NO_PREFETCH; fi
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $out/bench_$arm.log 2>&1
  echo "== $arm: $(grep '^{' $out/bench_$arm.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"].get("decode_loops"))')"
done
unset D2T_DECODE_ROW2_NO_PREFETCH
D2T_DECODE_TRACE=1 python3 tools/decode_trace.py 6 pipelined16 0 > $out/trace_new.log 2>&1; grep "loop\|L0\." $out/trace_new.log
D2T_DECODE_ROW2_NO_PREFETCH=1 D2T_DECODE_TRACE=1 python3 tools/decode_trace.py 6 pipelined16 0 > $out/trace_old.log 2>&1; grep "loop\|L0\." $out/trace_old.log
MARGIN_PRECS=bf16x3,mixed:2,mixed:3,mixed:4 timeout -k 10 900 python3 tools/probe/fp16x2_margin.py T2:2:48:64:12:9:3 C2:2:128:512:40:13:2 C4:1:160:640:40:7:2 S0:2:128:512:40:7:2 > $out/margin26.log 2>&1; echo "margin rc=$?"
grep -v "^  " $out/margin26.log
