#!/bin/bash
# End of round 4: per-kernel PMC averages of the driver's serving command on the final kernels (separate --pmc passes with
# --kernel-trace only; per-kernel table by tools/probe/pmc_by_kernel.py)
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_pmc_final
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $pass -d $out/pmc_$tag -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $out/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $out/pmc_$tag.log; exit 1; }
  python3 $R/tools/probe/pmc_by_kernel.py $out/pmc_$tag > $out/pmc_$tag.txt 2>&1
  echo "== pass $pass"; cat $out/pmc_$tag.txt
  rm -rf $out/pmc_$tag
done
