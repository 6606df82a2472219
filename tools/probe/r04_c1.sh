#!/bin/bash
# Round 4: where config C1 (ResNet + TFM-2, d_model 512, 64x256, B = 32) spends its step; decode-group sweep.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_c1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $R/bench.py --config C1 --steps 40 --warmup 4 --no-cpu-baseline --no-secondary > $out/trace.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in acc.values())
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:16]:
    print('%-90s %7d %10.1f us %8.2f avg %5.1f%%' % (k[:90], n, t, t / n, 100 * t / tot))
PY
rm -rf $out/trace
cd $R
for g in 2 4 6 8; do
  timeout -k 10 200 python3 bench.py --config C1 --steps 48 --warmup 4 --group $g --no-cpu-baseline --no-secondary > $out/c1_g$g.log 2>&1
  echo "== C1 group $g: $(grep '^{' $out/c1_g$g.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"].get("decode_loops"))')"
done
python3 tools/probe/encoder_only.py 20 > /dev/null 2>&1
