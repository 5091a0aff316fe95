#!/bin/bash
# SQ counters + kernel trace of the two-wave PAIR instance on config 5 (rodent_pair.xml, 4096 envs, one 10-substep launch per step).   bash tools/profile_pair.sh <tag>
tag=${1:-x}; root=$PWD; out=$root/gpurun_out/prof_pair_$tag; mkdir -p $out
export TMPDIR=/tmp OMP_NUM_THREADS=16 RR_PMC_INSTANCE=pair RR_PMC_T=1 RR_PMC_N=4096
cmd="bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ktp_$tag -o kt --output-format csv -- python3 $root/$cmd > $out/kt.log 2>&1 || echo "kernel-trace failed"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/sqp_${tag}_$i -o sq --output-format csv -- python3 $root/$cmd > $out/sq$i.log 2>&1 || echo "sq group $i failed"
done
cd $root
python3 tools/prof_summary.py $(find /tmp/ktp_$tag -name "*kernel_stats.csv" | head -n 1) $out/kernel_stats.csv
python3 tools/pmc_round3.py "$cmd" $out /tmp/none /tmp/none /tmp/sqp_${tag}_1 /tmp/sqp_${tag}_2 /tmp/sqp_${tag}_3 /tmp/sqp_${tag}_4
head -n 4 $out/kernel_stats.csv; cat $out/sq.json
