#!/bin/bash
# pipelined serving: decode groups at reserve 0, 2 chains (bench.py, 30 timed steps each)
for grp in 2 3 4; do
    timeout -k 10 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --reserve 0 --group $grp --chains 2 2>/dev/null | python tools/bench_line.py "group=$grp"
done
