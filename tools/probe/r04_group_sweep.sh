#!/bin/bash
# decode-group size x chains sweep of the headline serving loop, one box (bench.py --group / --chains)
for cfg in "6 3" "8 3" "8 2" "12 2" "12 3" "6 3"; do
  set -- $cfg
  python3 bench.py --gpus 1 --steps 48 --warmup 6 --group $1 --chains $2 --no-cpu-baseline --no-secondary 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('group $1 chains $2:', d['value'], d['ms_per_step'], d.get('parity'))" || exit 1
done
