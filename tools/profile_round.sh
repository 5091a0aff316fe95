#!/bin/bash
# Round profile of the default bench: kernel-trace stats + FETCH_SIZE / WRITE_SIZE PMC passes (separate runs), then the bench line.
# usage (GPU box, repo root): bash tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_traffic.json,bench_line.json}
tag=${1:-x}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $out/kt.log 2>&1 || echo "kernel-trace failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o f --output-format csv -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/fetch.log 2>&1 || echo "fetch failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o w --output-format csv -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/write.log 2>&1 || echo "write failed"
cd $root
python3 tools/prof_summary.py $(find $out/kt -name "*kernel_stats.csv" | head -n 1) $out/kernel_stats.csv
python3 tools/pmc_traffic.py $(find $out/fetch -name "*counter_collection.csv" | head -n 1) $(find $out/write -name "*counter_collection.csv" | head -n 1) $out/pmc_traffic.json > /dev/null
timeout -k 10 600 python3 bench.py > $out/bench_line.json 2> $out/bench.err
cat $out/kernel_stats.csv | head -n 4; cat $out/pmc_traffic.json | head -n 12; cat $out/bench_line.json
