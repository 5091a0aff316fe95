#!/bin/bash
# config-2 default protocol (5 x 1000 steps) over (substreams, unroll) settings: the per-repeat times show how stable a setting is
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --substreams $1 --unroll $2 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('substreams $1 unroll $2:', round(d['value']), 'env-steps/s  repeats', [round(x,4) for x in d['config']['ms_per_step_repeats']])"
done
