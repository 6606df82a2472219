#!/bin/bash
# Round-4 data gathering on the GPU box: baseline line, per-layer encoder table, decode kernels' HBM traffic / L2 behaviour.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_gather
mkdir -p $out
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $out/bench_base.log 2>&1 && grep '^{' $out/bench_base.log | cut -c1-400
python3 tools/probe/encoder_only.py 20 > $out/encoder_only.log 2>&1 && cat $out/encoder_only.log
D2T_DECODE_TRACE=1 python3 tools/decode_trace.py 6 pipelined16 0 > $out/decode_trace_alone.log 2>&1; tail -32 $out/decode_trace_alone.log
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass -d $out/pmc_$tag -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $out/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $out/pmc_$tag.log; }
  python3 $R/tools/probe/pmc_by_kernel.py $out/pmc_$tag > $out/pmc_$tag.txt 2>&1
  echo "== $pass"; cat $out/pmc_$tag.txt
  rm -rf $out/pmc_$tag
done
