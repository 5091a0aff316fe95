#!/bin/bash
# config 3 (full PPO loop) for several library builds, interleaved: does a learner-kernel change show up INSIDE the training step?   bash tools/gpu_cfg3_lib_ab.sh lib1.so ...
for rep in 1 2; do
for l in brax-rodent-run_amd/csrc/librodent_hip.so "$@"; do
  RR_LIB=$(pwd)/$l timeout -k 10 300 python3 bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('$l', round(d['value']), 'rollout', round(c['rollout_s_per_training_step'],4), 'learner', round(c['learner_s_per_training_step'],4))"
done
done
