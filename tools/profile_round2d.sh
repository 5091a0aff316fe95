#!/bin/bash
# Round-2 final measurement pass (bench default = 2 sub-batches, graph replay; PMC passes on one host-issued 2048-env launch per step) (GPU box): bench cfg 2 / 3 / 5, kernel trace, FETCH / WRITE PMC, SQ counters, phase profile.
tag=${1:-x}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp OMP_NUM_THREADS=16
echo "== bench config 2"; timeout -k 10 300 python3 bench.py > $out/bench_cfg2.json 2> $out/bench_cfg2.err; tail -c 1400 $out/bench_cfg2.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $root/bench.py --no-cpu-baseline > $out/kt.log 2>&1 || echo "kernel-trace failed"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fetch -o f --output-format csv -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --substreams 1 --no-graph --unroll 1 > $out/fetch.log 2>&1 || echo "fetch failed"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/write -o w --output-format csv -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --substreams 1 --no-graph --unroll 1 > $out/write.log 2>&1 || echo "write failed"
cd $root
python3 tools/prof_summary.py $(find /tmp/kt -name "*kernel_stats.csv" | head -n 1) $out/kernel_stats.csv
python3 tools/pmc_traffic.py $(find /tmp/fetch -name "*counter_collection.csv" | head -n 1) $(find /tmp/write -name "*counter_collection.csv" | head -n 1) $out/pmc_traffic.json > /dev/null
head -n 5 $out/kernel_stats.csv; head -n 12 $out/pmc_traffic.json
echo "== phase profile"; timeout -k 10 120 python3 tools/phase_prof.py > $out/phase.txt 2>&1; head -n 30 $out/phase.txt
echo "== SQ"; bash tools/pmc_sq.sh /tmp/sq > $out/sq.txt 2>&1; cat $out/sq.txt
echo "== bench config 3"; timeout -k 10 200 python3 bench.py --config 3 --steps 3 --warmup 1 > $out/bench_cfg3.json 2> $out/bench_cfg3.err; tail -c 700 $out/bench_cfg3.json
echo "== bench config 5"; timeout -k 10 200 python3 bench.py --config 5 --steps 50 --warmup 10 > $out/bench_cfg5.json 2> $out/bench_cfg5.err; tail -c 500 $out/bench_cfg5.json
