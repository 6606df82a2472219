set -e
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_attn
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/a -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $out/a.log 2>&1
grep -i "vit_attention" $out/a/t_kernel_stats.csv
