#!/bin/bash
# PMC passes on the record weight-gradient kernel (inside gpurun): bash tools/profile_wgrad_pmc.sh <tag>
# Separate --pmc passes with --kernel-trace only; output: gpurun_out/prof_<tag>/pmc_wgrad.json (dominant layer = the
# launches with nine taps and the longest duration class)
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
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $out/pmc_mfma -o pmc --output-format csv -- python3 $R/tools/train_bench.py 32 1 > $out/pmc_mfma.log 2>&1
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
    acc = collections.defaultdict(list); durs = {}
    for r in rows(sub):
        if "wgrad_rec_kernel<4, 4, 0>" not in r["Kernel_Name"]: continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if d < 600: continue  # the dominant layer's launches (512 -> 512, 66048 pixels)
        durs[r["Dispatch_Id"]] = d
        if r["Counter_Name"] in names: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: sum(v) / len(v) for k, v in acc.items() if v}
    res["launches"] = len(durs); res["avg_us"] = sum(durs.values()) / max(1, len(durs))
    return res
f = avg("pmc_fetch", ["FETCH_SIZE"]); w = avg("pmc_write", ["WRITE_SIZE"])
m = avg("pmc_mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS"])
res = {"kernel": "wgrad_rec_kernel<4, 4, 0> on the dominant layer (dW of the 512 -> 512 3x3 convolutions: 66048 pixels, nine taps, B=32)",
       "fetch_pass": f, "write_pass": w, "mfma_pass": m}
if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
    res["hbm_read_GB"] = f["FETCH_SIZE"] * 1024 * 2 / 1e9  # KiB, doubled: gfx950 correction (MI355X_MICROARCH.md)
    res["hbm_write_GB"] = w["WRITE_SIZE"] * 1024 / 1e9
if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
    res["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
    res["clock_GHz"] = m["GRBM_GUI_ACTIVE"] / 8 / (m["avg_us"] * 1e3)
if "SQ_WAVE_CYCLES" in m:
    for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS"):
        if k in m: res[k.lower() + "_frac_of_wave_cycles"] = m[k] / m["SQ_WAVE_CYCLES"]
json.dump(res, open(os.path.join(out, "pmc_wgrad.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_mfma
