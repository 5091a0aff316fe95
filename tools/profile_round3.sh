#!/bin/bash
# Round-3 measurement pass (GPU box): every counter is taken on the command the driver times -- `bench.py --steps 20 --warmup 5`, one
# launch of the multi-step (UNROLL) instance per 20 env steps -- not on single-step launches as in round 2.
#   bash tools/profile_round3.sh <tag> ["extra bench flags"]
tag=${1:-x}; flags=${2:-}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp OMP_NUM_THREADS=16
cmd="bench.py --steps 20 --warmup 5 $flags"
echo "== bench (driver protocol)"; timeout -k 10 300 python3 $cmd > $out/bench_cfg2.json 2> $out/bench_cfg2.err; tail -c 600 $out/bench_cfg2.json; echo
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt --output-format csv -- python3 $root/$cmd --no-cpu-baseline > $out/kt.log 2>&1 || echo "kernel-trace failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fetch_$tag -o f --output-format csv -- python3 $root/$cmd --no-cpu-baseline > $out/fetch.log 2>&1 || echo "fetch failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/write_$tag -o w --output-format csv -- python3 $root/$cmd --no-cpu-baseline > $out/write.log 2>&1 || echo "write failed"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/sq_${tag}_$i -o sq --output-format csv -- python3 $root/$cmd --no-cpu-baseline > $out/sq$i.log 2>&1 || echo "sq group $i failed"
done
cd $root
python3 tools/prof_summary.py $(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -n 1) $out/kernel_stats.csv
python3 tools/pmc_round3.py "$cmd" $out /tmp/fetch_$tag /tmp/write_$tag /tmp/sq_${tag}_1 /tmp/sq_${tag}_2 /tmp/sq_${tag}_3 /tmp/sq_${tag}_4
head -n 6 $out/kernel_stats.csv; cat $out/pmc_traffic.json; cat $out/sq.json
