#!/bin/bash
# Newton vs CG on the config-2 rollout (driver protocol, short).   bash tools/gpu_newton_bench.sh <tag>
tag=${1:-nb}; mkdir -p gpurun_out
for cfg in "cg 8 8" "newton 4 8" "newton 1 4" "cg 4 4"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --steps 200 --warmup 50 --solver $1 --iterations $2 --ls-iterations $3 --no-cpu-baseline > gpurun_out/nb_${tag}_$1_$2_$3.json 2> gpurun_out/nb_${tag}_$1_$2_$3.err || { echo "bench $cfg failed"; tail -n 5 gpurun_out/nb_${tag}_$1_$2_$3.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/nb_${tag}_$1_$2_$3.json').read().strip().splitlines()[-1])
print('$cfg', round(d['value']), 'env-steps/s  kernel ms', round(d['roofline']['avg_kernel_ms'],4))"
done
