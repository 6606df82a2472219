#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_bench
mkdir -p $out
cd $R
t0=$(date +%s)
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
tail -5 $out/bench.err
python3 - <<'PY'
import json,sys,os
d=json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r04_bench/bench_line.json")).read().strip().splitlines()[-1])
print("value",d["value"],"ms",d["ms_per_step"],"parity",d.get("parity"))
print("roofline",{k:d["roofline"][k] for k in ("achieved","frac","avg_launch_ms")})
print("cpu",d["cpu_baseline"]["value"],d["cpu_baseline_cached"]["value"])
for k,v in d["secondary"].items():
    print(k, v.get("value"), v.get("ms_per_step") or v.get("ms_per_batch"), "parity:", v.get("parity"), {kk:v[kk] for kk in ("with_end_bias","per_sample_api","cpu_baseline","speedup_vs_cpu") if kk in v})
PY
