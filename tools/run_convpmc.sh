#!/bin/bash
# PMC pass over tools/conv_bench.py (dominant shape only when D2T_CONV_ABL is set): clocks, MFMA busy, LDS activity.
# usage (inside gpurun): [D2T_CONV_ABL=n] bash tools/run_convpmc.sh <tag>
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/convpmc_$tag
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/convpmc_$tag -o pmc --output-format csv -- python3 $R/tools/conv_bench.py 3 0 > $R/gpurun_out/convpmc_$tag.log 2>&1 || { tail -20 $R/gpurun_out/convpmc_$tag.log; exit 1; }
python3 - "$tag" <<'PY'
import csv, glob, os, sys, collections
R = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
tag = sys.argv[1]
f = glob.glob(R + f'/gpurun_out/convpmc_{tag}/**/*counter_collection.csv', recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('d2t::', '')
    if 'conv_bf16x3' not in n or 'k4608' not in n and 'probe' not in n:
        continue
    d = acc.setdefault(n, {'disp': {}, 'c': collections.defaultdict(list)})
    d['disp'][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    d['c'][r['Counter_Name']].append(float(r['Counter_Value']))
for n, d in acc.items():
    durs = list(d['disp'].values())
    us = sum(durs) / len(durs)
    c = {k: sum(v) / len(v) for k, v in d['c'].items()}
    clk = c.get('GRBM_GUI_ACTIVE', 0) / 8 / (us * 1e-6) / 1e9
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8
    print(f"{tag} {n}: {us:.0f} us, clock {clk:.2f} GHz, mfma_busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024):.2f}, "
          f"lds_active/cu_cycle {c.get('SQ_LDS_IDX_ACTIVE', 0) / (cyc * 256):.2f}, lds_conflict/cu_cycle {c.get('SQ_LDS_BANK_CONFLICT', 0) / (cyc * 256):.3f}, "
          f"wait_inst_lds/wave_cyc {c.get('SQ_WAIT_INST_LDS', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f}, wait_any/wave_cyc {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f}, "
          f"wait_inst_any/wave_cyc {c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f}")
PY
rm -rf $R/gpurun_out/convpmc_$tag
