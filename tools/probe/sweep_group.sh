#!/bin/bash
# pipelined serving: decode groups x reserved slots, 2 chains (bench.py, 20 timed steps each)
for grp in 1 2; do
 for res in 0 16; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --reserve $res --group $grp --chains 2 > gpurun_out/gr_${grp}_r${res}.log 2>&1
    python - <<PY
import json
for l in open("gpurun_out/gr_${grp}_r${res}.log"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("group=${grp} reserve=${res}", d["value"], "formulas/s", d["ms_per_step"], "ms/step; dominant", r["avg_launch_ms"], "ms; all gemms ms/step", r["all_encoder_gemms"]["ms_per_step"], flush=True)
PY
 done
done
