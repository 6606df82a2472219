#!/usr/bin/env python3
"""Aggregate the rocprofv3 passes written by tools/profile_round.sh into one JSON for the dominant kernel.

usage: pmc_aggregate.py <gpurun_out/prof_tag>  (prints JSON)
The dominant GEMM (512->512 3x3 convolution, M=132096 N=512 K=4608) is selected by kernel symbol
(`..._k4608` for the split-bf16 path) or, for the fp32 kernel that shares its symbol with other layers, by
duration (> 4.3 ms).  FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 wide-stream
correction, MI355X_MICROARCH.md section HBM)."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]


def rows(sub, pattern):
    out = []
    for f in glob.glob(os.path.join(root, sub, "**", pattern), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


def dominant(name, dur_ms):
    if "k4608" in name:
        return True
    return "conv_mfma_kernel<128, 128>" in name and dur_ms > 4.3


def counter_avg(sub, counters):
    acc = {c: [] for c in counters}
    durs = []
    per = {}
    for r in rows(sub, "*counter_collection.csv"):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if not dominant(r["Kernel_Name"], d):
            continue
        per.setdefault(r["Dispatch_Id"], d)
        if r["Counter_Name"] in acc:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    durs = list(per.values())
    res = {c: (sum(v) / len(v) if v else None) for c, v in acc.items()}
    res["_launches"] = len(durs)
    res["_avg_ms"] = sum(durs) / len(durs) if durs else None
    return res


trace = []
for r in rows("trace", "*kernel_trace.csv"):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if dominant(r["Kernel_Name"], d):
        trace.append((r["Kernel_Name"], d))
out = {"kernel": trace[0][0] if trace else None,
       "gemm": "M=132096 N=512 K=4608 (512->512 3x3 conv @16x129, B=64)",
       "kernel_trace_avg_ms": sum(d for _, d in trace) / len(trace) if trace else None,
       "kernel_trace_launches": len(trace)}
f = counter_avg("pmc_fetch", ["FETCH_SIZE"])
w = counter_avg("pmc_write", ["WRITE_SIZE"])
m = counter_avg("pmc_mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES",
                             "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT"])
out["launches_sampled_pmc"] = f["_launches"]
if f["FETCH_SIZE"] is not None:
    out["FETCH_SIZE_KB_per_launch"] = f["FETCH_SIZE"]
    out["hbm_read_bytes_per_launch_corrected"] = f["FETCH_SIZE"] * 1024 * 2
if w["WRITE_SIZE"] is not None:
    out["WRITE_SIZE_KB_per_launch"] = w["WRITE_SIZE"]
    out["hbm_write_bytes_per_launch"] = w["WRITE_SIZE"] * 1024
if f["FETCH_SIZE"] is not None and w["WRITE_SIZE"] is not None:
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch_corrected"] + out["hbm_write_bytes_per_launch"]
for k, v in m.items():
    if not k.startswith("_") and v is not None:
        out[k] = v
out["pmc_run_avg_ms"] = m["_avg_ms"]
if m.get("GRBM_GUI_ACTIVE") and m["_avg_ms"]:
    clk = m["GRBM_GUI_ACTIVE"] / 8 / (m["_avg_ms"] * 1e-3)
    out["effective_clock_GHz"] = clk / 1e9
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        out["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
# ---- the HBM-bound front of the network: conv0_1 stem, conv0_2 (Cout 64: the 128x64 kernel), first max-pool ----------
def per_kernel(sub, counter, match):
    vals, durs = [], {}
    for r in rows(sub, "*counter_collection.csv"):
        if not match(r["Kernel_Name"], r):
            continue
        durs[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals) if vals else None), len(durs)


def first_pool(name, r):  # the first max-pool of a forward is the largest grid of maxpool_split_kernel
    return "maxpool_split_kernel" in name and int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) >= FIRST_POOL_GRID


pool_grids = [int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) for r in rows("trace", "*kernel_trace.csv")
              if "maxpool_split_kernel" in r["Kernel_Name"]]
FIRST_POOL_GRID = max(pool_grids) if pool_grids else 1 << 60
stem = {}
for key, match in (("stem_split_kernel (conv0_1 + BN + ReLU, 1 -> 32 channels @128x512)", lambda n, r: "stem_split_kernel" in n),
                   ("conv_bf16x3g_128x64 (conv0_2, 32 -> 64 channels @128x512, with the first 2x2 max-pool in its epilogue since round 3)", lambda n, r: "conv_bf16x3g_128x64" in n),
                   ("maxpool_split_kernel, largest pool launch of a forward (round 3: the third pool, k2 s(2,1) p(0,1), 256 channels 32x128 -> 16x129; the first two are fused into conv0_2 / conv1)", first_pool)):
    tr = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows("trace", "*kernel_trace.csv")
          if match(r["Kernel_Name"], r)]
    fv, nf = per_kernel("pmc_fetch", "FETCH_SIZE", match)
    wv, nw = per_kernel("pmc_write", "WRITE_SIZE", match)
    if not tr or fv is None or wv is None:
        continue
    ms = sum(tr) / len(tr)
    rd, wr = fv * 1024 * 2, wv * 1024  # FETCH_SIZE counts half the bytes of wide streaming reads on gfx950 (MI355X_MICROARCH.md, HBM)
    stem[key] = {"kernel_trace_avg_ms": round(ms, 4), "launches": len(tr), "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr,
                 "hbm_bytes": rd + wr, "hbm_GBps": round((rd + wr) / (ms * 1e-3) / 1e9, 1), "frac_of_8TBps": round((rd + wr) / (ms * 1e-3) / 8e12, 3)}
if stem:
    with open(os.path.join(root, "pmc_stem.json"), "w") as fh:
        json.dump({"what": "HBM traffic of the HBM-bound front of the backbone per launch (B = 64, 128x512 crops), from rocprofv3 PMC passes "
                           "of the default bench command: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE; rate = bytes / kernel-trace duration",
                   "kernels": stem}, fh, indent=1)
print(json.dumps(out, indent=1))
