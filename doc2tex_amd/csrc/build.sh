#!/bin/bash
# Build libd2t.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OPT="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value"
# D2T_PROBES=1 bash build.sh: a probe build (timing-probe / ablation switches compiled in; never ship it). Switching needs a clean rebuild.
if [ -n "$D2T_PROBES" ]; then OPT="$OPT -DD2T_PROBES"; fi
for f in conv_mfma conv_bf16x3 conv_bf16x3p conv_winograd ops decode recurrent train_kernels train engine prep post; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ kernels.h -nt $f.o ] || [ conv_common.h -nt $f.o ] || [ ctx.h -nt $f.o ] || [ ../../include/d2t.h -nt $f.o ] || [ ../../include/d2t_prep.h -nt $f.o ] || [ unicode_tables.h -nt $f.o ]; then
    (hipcc $OPT -c $f.hip -o $f.o.tmp && mv $f.o.tmp $f.o) &
    pids="$pids $!"
  fi
done
for pid in $pids; do wait $pid; done   # set -e: a failed compile fails the build (no stale object is linked)
hipcc --offload-arch=gfx950 -shared -fPIC -o libd2t.so conv_mfma.o conv_bf16x3.o conv_bf16x3p.o conv_winograd.o ops.o decode.o recurrent.o train_kernels.o train.o engine.o prep.o post.o
echo "built $(pwd)/libd2t.so"
