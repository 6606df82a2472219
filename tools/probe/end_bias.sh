#!/bin/bash
# secondary bench run: realistic lengths through the [s] bias, is_test early exit
for eb in 2.2 2.35 2.5 2.65 2.8; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --end-bias $eb 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('end_bias',$eb, d['value'],'formulas/s', d['ms_per_step'],'ms/step', d.get('early_exit'))"
done
