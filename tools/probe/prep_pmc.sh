#!/bin/bash
# HBM traffic of the pre-processing kernels (separate PMC passes, kernel-trace only)
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_prep_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o pmc --output-format csv -- python3 $R/tools/prep_bench.py --steps 4 --cpu-sample 4 > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o pmc --output-format csv -- python3 $R/tools/prep_bench.py --steps 4 --cpu-sample 4 > $out/write.log 2>&1
python3 - <<PY
import csv, glob, collections
res = {}
for kind in ("fetch", "write"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % kind, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "prep_" in n:
                acc[n.split("(")[0][-40:]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res.setdefault(k, {})[kind] = (sum(v) / len(v), len(v))
for k, v in res.items():
    f = v.get("fetch", (0, 0)); w = v.get("write", (0, 0))
    print(k, "FETCH_SIZE KB/launch %.0f (x2 corrected %.1f MB)" % (f[0], 2 * f[0] * 1024 / 1e6), "WRITE_SIZE KB/launch %.0f (%.1f MB)" % (w[0], w[0] * 1024 / 1e6), "launches", f[1])
PY
