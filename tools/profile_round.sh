#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench command, then separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / MFMA + clock counters; never combined with runtime/sys tracing).
# usage (inside gpurun): bash tools/profile_round.sh <tag> [bench args...]; outputs under gpurun_out/prof_<tag>/
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline "$@" > $out/trace.log 2>&1
grep '^{' $out/trace.log > $out/bench_line_profiled.json || true
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/pmc_write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -d $out/pmc_mfma -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/pmc_mfma.log 2>&1
echo "mfma done"
# keep only what the aggregator needs from the PMC passes (the raw CSVs are large)
python3 $R/tools/pmc_aggregate.py $out > $out/pmc_dominant.json
cat $out/pmc_dominant.json
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_mfma
