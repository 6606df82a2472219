set -e
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_train
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- python3 $R/tools/train_bench.py 32 3 > $out/trace.log 2>&1
tail -2 $out/trace.log
