#!/bin/bash
# driver-protocol bench lines over pacing settings (RR_PACE, RR_PACE_T, RR_PACE_MODE; csrc/rr_api.hip launch()).   bash tools/gpu_pace_sweep.sh
run() { local fl="$1"; shift; env "$@" timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $fl 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$* $fl', round(d['value']), round(d['ms_per_step'],4))"; }
for rep in 1 2; do
run "" RR_PACE=0
run "" RR_PACE_MODE=0
run "--balance" RR_PACE_MODE=0
run "--balance" RR_PACE=0
run "" RR_PACE_MODE=4
run "" RR_PACE_MODE=4 RR_PACE_T=0.2,0.5,0.9
run "" RR_PACE_MODE=4 RR_PACE_T=0.15,0.35,0.7
run "" RR_PACE_MODE=8
run "" RR_PACE_MODE=12
run "--balance" RR_PACE_MODE=12
done
