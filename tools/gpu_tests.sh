#!/bin/bash
# GPU box: the whole -m gpu suite, then smoke().  usage: bash tools/gpu_tests.sh <tag>   -> gpurun_out/t_<tag>.log
tag=${1:-x}
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 1100 python -m pytest tests -m gpu -q -s --durations=15 > gpurun_out/t_$tag.log 2>&1
rc=$?
tail -n 40 gpurun_out/t_$tag.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/smoke_$tag.log 2>&1
rc=$?
tail -n 3 gpurun_out/smoke_$tag.log
exit $rc
