#!/bin/bash
# selected GPU tests only.   bash tools/gpu_k.sh <tag> <pytest -k expr>
tag=${1:-k}; kexpr=${2:-newton}
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 900 python -m pytest tests -m gpu -q -k "$kexpr" > gpurun_out/k_$tag.log 2>&1; rc=$?
tail -n 40 gpurun_out/k_$tag.log | cut -c1-400
exit $rc
