#!/bin/bash
# config-2 bench lines for a list of "substreams graph" settings (driver protocol, short).   bash tools/gpu_bench2.sh <tag> "2 1" "1 1" "2 0" ...
tag=$1; shift; mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg
  flags="--substreams $1"; [ "$2" = "0" ] && flags="$flags --no-graph"
  timeout -k 10 300 python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline --unroll 1 $flags > gpurun_out/b2_${tag}_s$1_g$2.json 2> gpurun_out/b2_${tag}_s$1_g$2.err || { echo "bench failed: $cfg"; tail -n 12 gpurun_out/b2_${tag}_s$1_g$2.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/b2_${tag}_s$1_g$2.json').read().strip().splitlines()[-1])
print('substreams $1 graph $2:', round(d['value']), 'env-steps/s  ms_per_step', round(d['ms_per_step'],4), ' kernel ms', round(d['roofline']['avg_kernel_ms'],4), 'launches', d['roofline']['launches'])"
done
