#!/bin/bash
# Round-2 measurement pass (GPU box, repo root): bash tools/profile_round2.sh <tag>  -> gpurun_out/prof_<tag>/
tag=${1:-x}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp OMP_NUM_THREADS=16
echo "== bench config 2 (SURVEY 8(d) protocol)"; timeout -k 10 300 python3 bench.py > $out/bench_cfg2.json 2> $out/bench_cfg2.err; tail -c 1500 $out/bench_cfg2.json
echo "== mlp"; timeout -k 10 120 python3 tools/bench_mlp.py > $out/mlp.jsonl 2> $out/mlp.err; cat $out/mlp.jsonl
cd /tmp
echo "== kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $root/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $out/kt.log 2>&1 || echo "kernel-trace failed"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o f --output-format csv -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/fetch.log 2>&1 || echo "fetch failed"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o w --output-format csv -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/write.log 2>&1 || echo "write failed"
cd $root
python3 tools/prof_summary.py $(find $out/kt -name "*kernel_stats.csv" | head -n 1) $out/kernel_stats.csv
python3 tools/pmc_traffic.py $(find $out/fetch -name "*counter_collection.csv" | head -n 1) $(find $out/write -name "*counter_collection.csv" | head -n 1) $out/pmc_traffic.json > /dev/null
head -n 5 $out/kernel_stats.csv; cat $out/pmc_traffic.json | head -n 12
echo "== bench config 3 fused / unfused"
RR_FUSED_MLP=1 timeout -k 10 200 python3 bench.py --config 3 --steps 3 --warmup 1 > $out/bench_cfg3_fused.json 2> $out/bench_cfg3_fused.err; tail -c 900 $out/bench_cfg3_fused.json
RR_FUSED_MLP=0 timeout -k 10 200 python3 bench.py --config 3 --steps 3 --warmup 1 > $out/bench_cfg3_unfused.json 2> $out/bench_cfg3_unfused.err; tail -c 900 $out/bench_cfg3_unfused.json
echo "== bench config 5"; timeout -k 10 200 python3 bench.py --config 5 --steps 50 --warmup 10 > $out/bench_cfg5.json 2> $out/bench_cfg5.err; tail -c 900 $out/bench_cfg5.json
