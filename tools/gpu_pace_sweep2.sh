#!/bin/bash
run() { local fl="$1"; shift; env "$@" timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $fl 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$* $fl', round(d['value']), round(d['ms_per_step'],4))"; }
for rep in 1 2; do
run "" RR_PACE=0
run "" RR_PACE_MODE=0
run "" RR_PACE_MODE=0 RR_PACE_T=0.2,0.45,0.8
run "" RR_PACE_MODE=0 RR_PACE_T=0.4,0.8,1.3
run "" RR_PACE_MODE=0 RR_PACE_T=0.15,0.3,0.6
run "" RR_PACE_MODE=8
run "--steps 250 --warmup 50" RR_PACE_MODE=0
done
