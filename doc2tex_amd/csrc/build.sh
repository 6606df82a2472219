#!/bin/bash
# Build libd2t.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
#   bash build.sh                 the shipped library: objects in obj/, libd2t.so
#   D2T_PROBES=1 bash build.sh    a PROBE build (timing-probe / ablation / A-B switches compiled in, read from the environment):
#                                 objects in obj_probe/, libd2t_probe.so -- never libd2t.so.  Tools select it with
#                                 D2T_PROBE_LIB=doc2tex_amd/csrc/libd2t_probe.so (tools/decode_trace.py, tools/probe/*).
set -e
cd "$(dirname "$0")"
OPT="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -Wno-pass-failed"
OBJ=obj; LIB=libd2t.so
if [ -n "$D2T_PROBES" ]; then OPT="$OPT -DD2T_PROBES"; OBJ=obj_probe; LIB=libd2t_probe.so; fi
mkdir -p $OBJ
SRCS="conv_mfma conv_bf16x3 conv_bf16x3p ops decode recurrent train_kernels train engine prep post posembed"
objs=""
for f in $SRCS; do
  o=$OBJ/$f.o
  objs="$objs $o"
  if [ ! -f $o ] || [ $f.hip -nt $o ] || [ kernels.h -nt $o ] || [ conv_common.h -nt $o ] || [ ctx.h -nt $o ] || [ ../../include/d2t.h -nt $o ] || [ ../../include/d2t_prep.h -nt $o ] || [ unicode_tables.h -nt $o ] || [ build.sh -nt $o ]; then
    (hipcc $OPT -c $f.hip -o $o.tmp && mv $o.tmp $o) &
    pids="$pids $!"
  fi
done
for pid in $pids; do wait $pid; done   # set -e: a failed compile fails the build (no stale object is linked)
hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB $objs
echo "built $(pwd)/$LIB"
