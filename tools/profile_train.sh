#!/bin/bash
# Kernel-trace stats of the training step on the GPU box: bash tools/profile_train.sh <tag> [batch [steps [C2|S0]]]
# -> gpurun_out/prof_<tag>/{trace/, kernel_stats_train.csv}
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- python3 $R/tools/train_bench.py ${1:-32} ${2:-3} bf16x3 ${3:-C2} > $out/trace.log 2>&1
tail -1 $out/trace.log
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
byg = collections.defaultdict(lambda: [0, 0.0])  # weight-gradient launches by grid
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a[0] += 1; a[1] += d; a[2] = max(a[2], d)
        if "wgrad_rec" in r["Kernel_Name"] or "wgrad_bf16x3" in r["Kernel_Name"]:
            g = byg[(r["Kernel_Name"][5:22], r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))]
            g[0] += 1; g[1] += d
tot = sum(v[1] for v in acc.values())
with open(out + "/kernel_stats_train.csv", "w") as fh:
    fh.write("kernel,calls,total_us,avg_us,max_us,pct\n")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        fh.write('"%s",%d,%.1f,%.2f,%.1f,%.2f\n' % (k, v[0], v[1], v[1] / v[0], v[2], 100 * v[1] / tot))
for k, v in sorted(byg.items(), key=lambda kv: -kv[1][1])[:12]:
    print("wgrad grid", k, v[0], "launches, avg %.1f us" % (v[1] / v[0]))
for i, l in enumerate(open(out + "/kernel_stats_train.csv")):
    if i < 16: print(l.rstrip()[:170])
PY
rm -rf $out/trace
