#!/bin/bash
# config-3 bench A/B over an environment switch.   bash tools/gpu_cfg3_ab.sh <tag> <ENVVAR> [values...]
tag=${1:-ab}; var=${2:-RR_FUSED_LOSS}; shift 2; vals=${@:-1 0}
mkdir -p gpurun_out
for f in $vals; do
  env $var=$f timeout -k 10 300 python3 bench.py --config 3 --steps 3 --warmup 1 > gpurun_out/cfg3_${tag}_$var$f.json 2> gpurun_out/cfg3_${tag}_$var$f.err || { echo "bench failed ($var=$f)"; tail -n 8 gpurun_out/cfg3_${tag}_$var$f.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/cfg3_${tag}_$var$f.json').read().strip().splitlines()[-1])
print('$var=$f', round(d['value']), 'env-steps/s  rollout', round(d['config']['rollout_s_per_training_step'],3), 's  learner', round(d['config']['learner_s_per_training_step'],3), 's')"
done
