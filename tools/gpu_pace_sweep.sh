#!/bin/bash
# driver-protocol bench lines over pacing settings (RR_PACE, RR_PACE_T, RR_PACE_MODE; csrc/rr_api.hip launch()).   bash tools/gpu_pace_sweep.sh
run() { env "$@" timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(d['ms_per_step'],4))"; }
for rep in 1 2; do
run RR_PACE=0
run RR_PACE_T=0.3,0.6,1.0 RR_PACE_MODE=0
run RR_PACE_T=0.1,0.3,0.6 RR_PACE_MODE=0
run RR_PACE_T=0.5,1.0,1.5 RR_PACE_MODE=0
run RR_PACE_T=0.4,0.8,1.2 RR_PACE_MODE=0
run RR_PACE_T=0.7,1.2,2.0 RR_PACE_MODE=0
run RR_PACE_T=0.3,0.5,0.8 RR_PACE_MODE=0
run RR_PACE_T=0.2,0.6,1.2 RR_PACE_MODE=0
done
