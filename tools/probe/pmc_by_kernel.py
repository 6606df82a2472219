#!/usr/bin/env python3
"""Per-kernel averages of the counters of one rocprofv3 --pmc pass (counter_collection CSVs under <dir>).
usage: pmc_by_kernel.py <dir> [min_total_ms]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0][:70]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = {k: sum(v.values()) for k, v in dur.items()}
for k in sorted(tot, key=lambda k: -tot[k])[:14]:
    n = len(dur[k])
    line = f"{k:70s} n={n:6d} avg_us={tot[k] / n:9.2f}"
    for c, v in sorted(acc[k].items()):
        line += f" {c}={sum(v) / len(v):.4g}"
    print(line)
