#!/bin/bash
# Round 4 final: whole GPU suite, the round profile (kernel stats + PMC passes of the default bench command), the full bench line,
# the training-step profile.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r04_final
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; echo "gpu suite rc=$?"; tail -3 $out/tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
t0=$(date +%s); timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
bash tools/profile_round.sh r04 > $out/profile_round.log 2>&1; echo "profile rc=$?"; tail -12 $out/profile_round.log
bash tools/profile_train.sh r04t 32 7 > $out/profile_train.log 2>&1; echo "train profile rc=$?"; tail -3 $out/profile_train.log
