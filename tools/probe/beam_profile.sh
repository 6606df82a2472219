#!/bin/bash
# kernel-time breakdown of the batched beam search (config C4) under rocprofv3; usage (inside gpurun): bash tools/probe/beam_profile.sh <tag> [n] [beam]
R=${GRAFT_REPO_ROOT:-$PWD}
tag=${1:-beam}; n=${2:-128}; beam=${3:-5}; mode=${4:-both}; shared=${5:-row}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o bm --output-format csv -- python3 $R/tools/beam_bench.py $n $beam $mode $shared > $R/gpurun_out/prof_$tag.log 2>&1
grep "C4 beam" $R/gpurun_out/prof_$tag.log
python3 - "$R/gpurun_out/prof_$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"].split("(")[0][:70]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in acc.values())
print(f"total kernel time {tot / 1e3:.1f} ms")
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{k:72s} {c:7d} {t / 1e3:9.1f} ms {t / c:8.1f} us {100 * t / tot:5.1f}%")
PY
rm -rf $R/gpurun_out/prof_$tag
