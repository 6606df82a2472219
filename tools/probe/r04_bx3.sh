#!/bin/bash
# Round 4: the greedy row kernel's cross-attention on split-bf16 MFMAs -- parity, margin against the oracle, phases, A/B.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_bx3
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py -x -q -k "greedy or headline or 64_rows or shard or pipelined or full_tensors" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
MARGIN_PRECS=bf16x3 timeout -k 10 600 python3 tools/probe/fp16x2_margin.py T2:2:48:64:12:6:3 C2:2:128:512:40:5:2 C2:2:128:512:150:2:1 S0:2:128:512:40:3:1 > $out/margin.log 2>&1; grep -v "^  " $out/margin.log
for m in 0 1; do D2T_DECODE_CROSS_FP32=$m D2T_PROBE_LIB=doc2tex_amd/csrc/libd2t_probe.so timeout -k 10 200 python3 tools/probe/row_phases.py 6 2>&1 | tail -23 > $out/phases_fp32_$m.log; cat $out/phases_fp32_$m.log; done
for arm in 0 1 0 1; do  # (D2T_DECODE_CROSS_FP32: probe build only)
  D2T_PROBE_LIB=doc2tex_amd/csrc/libd2t_probe.so D2T_DECODE_CROSS_FP32=$arm timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $out/ab$arm.log 2>&1
  echo "== cross-attention on fp32 MFMA=$arm: $(grep '^{' $out/ab$arm.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"].get("decode_loops"))')"
done
