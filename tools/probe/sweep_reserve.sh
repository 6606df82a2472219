#!/bin/bash
# pipelined serving: sweep the reserved block slots / tile order (bench.py, 12 timed steps each)
for ord in 0 1; do
  for res in 32 64 96 128; do
    D2T_TILE_ORDER=$ord timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --reserve $res > gpurun_out/sw_${ord}_${res}.log 2>&1
    python - <<PY
import json
for l in open("gpurun_out/sw_${ord}_${res}.log"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("order=${ord} reserve=${res}", d["value"], "formulas/s", d["ms_per_step"], "ms/step; dominant", r["avg_launch_ms"], "ms; all gemms ms/step", r["all_encoder_gemms"]["ms_per_step"], flush=True)
PY
  done
done
