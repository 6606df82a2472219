cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_kt -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log 2>&1
grep -h "argmax_embed\|decoder_row\|skinny" $GRAFT_REPO_ROOT/gpurun_out/prof_kt/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/prof_kt/*/*kernel_stats.csv 2>/dev/null | cut -c1-110
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_kt
