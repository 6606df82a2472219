#!/bin/bash
# Runs tools/conv_bench.py under rocprofv3 --kernel-trace on the GPU box and prints per-kernel durations.
# usage (inside gpurun): bash tools/run_convbench.sh [reps] [reserved_cus ...]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/convbench
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/convbench -o cb --output-format csv -- python3 $R/tools/conv_bench.py "$@" > $R/gpurun_out/convbench.log 2>&1 || { tail -30 $R/gpurun_out/convbench.log; exit 1; }
grep -E "bit-identical|^ok|Error|error" $R/gpurun_out/convbench.log | cut -c1-200
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
f = glob.glob(R + '/gpurun_out/convbench/**/*kernel_trace.csv', recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'conv_bf16x3' in n:
        d.setdefault((n.split('(')[0].replace('d2t::', ''), r['Grid_Size_X'], r['VGPR_Count']), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    print(k, len(v), ' '.join(f'{x:.0f}' for x in v))
PY
