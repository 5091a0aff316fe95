#!/bin/bash
# kernel-trace stats of one PPO training step (config 3): bash tools/prof_ppo.sh <outdir>
out=$PWD/${1:-gpurun_out/prof_ppo}
mkdir -p $out
export TMPDIR=/tmp
root=$PWD
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $root/tools/bench_ppo.py 1 > $out/kt.log 2>&1 || echo "failed"
cd $root
python3 tools/prof_summary.py $(find $out/kt -name "*kernel_stats.csv" | head -n 1) $out/kernel_stats.csv
head -n 14 $out/kernel_stats.csv; tail -n 2 $out/kt.log
