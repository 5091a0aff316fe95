#!/bin/bash
# config-2 bench lines over (substreams, unroll) settings, driver protocol.   bash tools/gpu_bench3.sh <tag> "2 10" "2 1" "1 10" ...
tag=$1; shift; mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline --substreams $1 --unroll $2 > gpurun_out/b3_${tag}_s$1_u$2.json 2> gpurun_out/b3_${tag}_s$1_u$2.err || { echo "bench failed: $cfg"; tail -n 12 gpurun_out/b3_${tag}_s$1_u$2.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/b3_${tag}_s$1_u$2.json').read().strip().splitlines()[-1])
r=d['roofline']
print('substreams $1 unroll $2:', round(d['value']), 'env-steps/s  ms_per_step', round(d['ms_per_step'],4), ' kernel ms per launch', round(r['avg_launch_ms'],4), 'launches', r['launches'], 'achieved', round(r['achieved'],2))"
done
