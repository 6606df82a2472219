set -e
R=$PWD
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bf16x3 -o r01 --output-format csv -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
tail -2 $R/gpurun_out/prof_bench.log
ls -R $R/gpurun_out/prof_bf16x3 | head
