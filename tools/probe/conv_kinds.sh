#!/bin/bash
# kernel time of the dominant layer (512 -> 512, 3x3 @16x129, B = 64) per convolution kernel kind (3: split-bf16, 8: fp16x2), product build, under rocprofv3
# usage inside gpurun: [KINDS="3 8"] [REPS=30] bash tools/probe/conv_kinds.sh
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for kind in ${KINDS:-3 8}; do
  rm -rf /tmp/ck_$kind
  rocprofv3 --kernel-trace --stats -d /tmp/ck_$kind -o t --output-format csv -- python3 $R/tools/probe/conv_abl.py ${REPS:-30} $kind > /tmp/ck_$kind.log 2>&1
  f=$(find /tmp/ck_$kind -name "*kernel_stats.csv" | head -1)
  echo "kind=$kind: $(grep -i "conv_\(bf16x3\|f16x2\)[pbw]16" $f | cut -d, -f1-4 | tr '\n' ' ')"
done
