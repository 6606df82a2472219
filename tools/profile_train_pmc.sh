#!/bin/bash
# PMC passes on the training step's dominant forward / data-gradient convolution (inside gpurun): bash tools/profile_train_pmc.sh <tag>
# Separate --pmc passes with --kernel-trace only (FETCH_SIZE; WRITE_SIZE; SQ / GRBM) of tools/train_bench.py 32 1;
# output: gpurun_out/prof_<tag>/pmc_train_dominant.json for profiles/rNN_pmc_train_dominant_bf16x3.json (bench.py's
# secondary.train_c3.roofline.traffic reads it).  The dominant GEMM of the step is the 512 -> 512 3x3 layer at B = 32:
# M = 66048, N = 512, K = 4608, the `_k4608` symbol of the pipelined kernel (forward and data gradient share it).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc --output-format csv -- python3 $R/tools/train_bench.py 32 1 > $out/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o pmc --output-format csv -- python3 $R/tools/train_bench.py 32 1 > $out/pmc_write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -d $out/pmc_mfma -o pmc --output-format csv -- python3 $R/tools/train_bench.py 32 1 > $out/pmc_mfma.log 2>&1
echo "mfma done"
python3 - "$out" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
def rows(sub):
    r = []
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        r += list(csv.DictReader(open(f)))
    return r
def avg(sub, names):
    acc = collections.defaultdict(list); durs = {}; kern = None
    for r in rows(sub):
        if "k4608" not in r["Kernel_Name"] or "conv_bf16x3" not in r["Kernel_Name"]: continue
        kern = r["Kernel_Name"]
        durs[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if r["Counter_Name"] in names: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: sum(v) / len(v) for k, v in acc.items() if v}
    res["launches"] = len(durs); res["avg_us"] = sum(durs.values()) / max(1, len(durs)); res["kernel"] = kern
    return res
f = avg("pmc_fetch", ["FETCH_SIZE"]); w = avg("pmc_write", ["WRITE_SIZE"])
m = avg("pmc_mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT"])
res = {"kernel": m.get("kernel"), "gemm_MNK": [66048, 512, 4608],
       "what": "forward and data-gradient launches of the 512 -> 512 3x3 convolutions in the C3 training step (B = 32, 128x512 crops)",
       "fetch_pass": f, "write_pass": w, "mfma_pass": m}
if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
    res["hbm_read_bytes_per_launch_corrected"] = f["FETCH_SIZE"] * 1024 * 2  # KiB, doubled: gfx950 correction (MI355X_MICROARCH.md)
    res["hbm_write_bytes_per_launch"] = w["WRITE_SIZE"] * 1024
    res["hbm_bytes_per_launch"] = res["hbm_read_bytes_per_launch_corrected"] + res["hbm_write_bytes_per_launch"]
if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
    res["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
    res["effective_clock_GHz"] = m["GRBM_GUI_ACTIVE"] / 8 / (m["avg_us"] * 1e3)
json.dump(res, open(os.path.join(out, "pmc_train_dominant.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_mfma
