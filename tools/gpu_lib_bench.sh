#!/bin/bash
# config-2 default protocol for several library builds (in-tree first).   bash tools/gpu_lib_bench.sh "<bench flags>" lib1.so lib2.so ...
flags=$1; shift
for l in brax-rodent-run_amd/csrc/librodent_hip.so "$@"; do
  RR_LIB=$(pwd)/$l timeout -k 10 300 python3 bench.py --no-cpu-baseline $flags 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$l', round(d['value']), 'env-steps/s  repeats', [round(x,4) for x in d['config']['ms_per_step_repeats']])"
done
