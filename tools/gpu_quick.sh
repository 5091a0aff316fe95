#!/bin/bash
# quick GPU check: selected tests + config-3 bench, fused vs library forward.   bash tools/gpu_quick.sh <tag> [pytest -k expr]
tag=${1:-q}; kexpr=${2:-mlp}
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 600 python -m pytest tests -m gpu -q -k "$kexpr" 2>&1 | tail -n 8
for f in 1 0; do
  RR_FUSED_MLP=$f timeout -k 10 200 python3 bench.py --config 3 --steps 3 --warmup 1 > gpurun_out/cfg3_${tag}_fused$f.json 2> gpurun_out/cfg3_${tag}_fused$f.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/cfg3_${tag}_fused$f.json').read().strip().splitlines()[-1])
print('RR_FUSED_MLP=$f', round(d['value']), 'env-steps/s', d['config']['rollout_s_per_training_step'], d['config']['learner_s_per_training_step'])"
done
