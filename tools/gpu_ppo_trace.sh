#!/bin/bash
# kernel trace of one config-3 training step (raw trace stays in /tmp on the box; the summary comes back).   bash tools/gpu_ppo_trace.sh <tag>
tag=${1:-t}; out=gpurun_out/ppo_trace_$tag; mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/ppotr_$tag -o kt --output-format csv -- python3 $root/bench.py --config 3 --steps 1 --warmup 1 > $root/$out/bench.log 2>&1 || { echo "trace failed"; tail -n 5 $root/$out/bench.log; exit 1; }
cd $root
python3 tools/prof_summary.py $(find /tmp/ppotr_$tag -name "*kernel_stats.csv" | head -n 1) $out/kernel_stats.csv
head -n 45 $out/kernel_stats.csv | cut -c1-150
