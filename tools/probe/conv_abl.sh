#!/bin/bash
# ablation timings of the 16x16x32 convolution kernels (probe build: see tools/probe/README.md); usage inside gpurun:
#   [KINDS="3 6"] bash tools/probe/conv_abl.sh   -> stdout (kind 3: pipelined, 6: band-resident; ABL 1 no LDS-DMA, 2 no MFMA,
#   4 LDS-DMA never awaited)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export D2T_PROBE_LIB=$R/doc2tex_amd/csrc/libd2t_probe.so
for kind in ${KINDS:-3 6}; do
  for abl in ${ABLS:-0 1 2 4}; do
    rm -rf /tmp/abl_$abl
    D2T_CONV_ABL=$abl rocprofv3 --kernel-trace --stats -d /tmp/abl_$abl -o t --output-format csv -- python3 $R/tools/probe/conv_abl.py ${REPS:-12} $kind > /tmp/abl_$abl.log 2>&1
    f=$(find /tmp/abl_$abl -name "*kernel_stats.csv" | head -1)
    echo "kind=$kind ABL=$abl: $(grep -i "conv_bf16x3[pbw]16" $f | cut -d, -f1-4 | tr '\n' ' ')"
  done
done
