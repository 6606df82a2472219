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
# The driver's own command (20 timed steps after 5 warm-up steps).  Most forward passes of the process are the graph-priming
# ones before the warm-up, during which no decode loop is in flight and the convolution would hand its last round of tiles
# to the small kernel; D2T_CONV_TAIL=0 pins the serving form of the kernel (what the timed region runs) for every launch,
# so that the per-kernel average is comparable with bench.py's live timing.
# (round 4: the shipped library reads no environment; launches made while no decode loop is in flight hand their last tile round to the 64x128 build)
rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "$@" > $out/trace.log 2>&1
grep '^{' $out/trace.log > $out/bench_line_profiled.json || true
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/pmc_write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $out/pmc_mfma -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/pmc_mfma.log 2>&1
echo "mfma done"
# keep only what the aggregator needs from the PMC passes (the raw CSVs are large)
python3 $R/tools/pmc_aggregate.py $out > $out/pmc_dominant.json
cat $out/pmc_dominant.json
cat $out/pmc_stem.json 2>/dev/null || true
# the per-kernel table of the trace (top rows) for profiles/
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in acc.values())
with open(out + "/kernel_stats.csv", "w") as fh:
    fh.write("kernel,calls,total_us,avg_us,percent\n")
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        fh.write('"%s",%d,%.1f,%.2f,%.2f\n' % (k.replace('"', "'"), n, t, t / n, 100 * t / tot))
PY
head -25 $out/kernel_stats.csv
# rocprofv3's own --stats table, then drop the raw traces (gpurun merges at most 64 MiB back)
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/rocprofv3_kernel_stats.csv 2>/dev/null || true
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_mfma $out/trace
