#!/bin/bash
# config 5 on the GENERIC one-wave instance (RR_PAIR_WAVES=0) with the wave-priority switches compiled out: what the round-2 verdict's
# 331 k -> 307 k drop is.   bash tools/gpu_pair_prio_ab.sh
for l in brax-rodent-run_amd/csrc/librodent_hip.so build_var/noenvprio.so build_var/nofacprio.so build_var/noprio.so; do
  RR_PAIR_WAVES=0 RR_LIB=$(pwd)/$l timeout -k 10 200 python3 bench.py --config 5 --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$l generic instance', round(d['value']), round(d['ms_per_step'],3))"
done
