#!/bin/bash
# interleaved driver-protocol A/B of library builds (the multi-step instance; single-step timings do not predict it).   bash tools/gpu_lib_ab.sh lib1.so lib2.so ...
for rep in 1 2 3; do
for l in brax-rodent-run_amd/csrc/librodent_hip.so "$@"; do
  RR_LIB=$(pwd)/$l timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$l', round(d['value']), round(d['ms_per_step'],4))"
done
done
