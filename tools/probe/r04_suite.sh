#!/bin/bash
# Round 4: the whole GPU suite after the prune, the row kernel's phase timeline (probe build), tail-kernel A/B on C1 / C2.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_suite
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; echo "gpu suite rc=$?"; tail -6 $out/tests.log
D2T_PROBE_LIB=doc2tex_amd/csrc/libd2t_probe.so timeout -k 10 200 python3 tools/probe/row_phases.py 6 > $out/phases.log 2>&1; cat $out/phases.log | tail -18
for arm in 0 1 0 1; do
  D2T_PROBE_LIB=doc2tex_amd/csrc/libd2t_probe.so D2T_CONV_TAIL_ALWAYS=$arm timeout -k 10 300 python3 bench.py --config C1 --steps 40 --warmup 5 --no-cpu-baseline --no-secondary > $out/c1_tail$arm.log 2>&1
  echo "== C1 tail_always=$arm: $(grep '^{' $out/c1_tail$arm.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["gemm_MNK"], d["roofline"]["avg_launch_ms"])')"
done
for arm in 0 1; do
  D2T_PROBE_LIB=doc2tex_amd/csrc/libd2t_probe.so D2T_CONV_TAIL_ALWAYS=$arm timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $out/c2_tail$arm.log 2>&1
  echo "== C2 tail_always=$arm: $(grep '^{' $out/c2_tail$arm.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])')"
done
