#!/bin/bash
# selected GPU tests + the default bench line (and the same with SIMD pairing off).  bash tools/gpu_bench_quick.sh <pytest -k expr>
export OMP_NUM_THREADS=16
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -k "${1:-schedule}" 2>&1 | tail -n 6
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/bq.json 2> gpurun_out/bq.err; python3 -c "
import json; d=json.loads(open('gpurun_out/bq.json').read().strip().splitlines()[-1]); print('balanced  ', round(d['value']), d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['config']['ms_per_step_repeats'])"
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/bq0.json 2> gpurun_out/bq0.err; python3 -c "
import json; d=json.loads(open('gpurun_out/bq0.json').read().strip().splitlines()[-1]); print('unbalanced', round(d['value']), d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['config']['ms_per_step_repeats'])"
