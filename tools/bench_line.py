#!/usr/bin/env python3
"""Print the interesting fields of a bench.py JSON line read from stdin."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        print(tag, d["dtype"], "formulas/s", d["value"], "ms/step", d["ms_per_step"], "dom_ms", r["avg_launch_ms"],
              "dom_TF", r["achieved"], "all", r["all_encoder_gemms"])
