#!/bin/bash
# SQ instruction-mix counters of rr_step_kernel (one rocprofv3 --pmc pass per counter group; kernel-trace only).
# usage (on the GPU box, repo root): bash tools/pmc_sq.sh <outdir>
out=${1:-gpurun_out/sq}; case $out in /*) ;; *) out=$PWD/$out;; esac
mkdir -p $out
export TMPDIR=/tmp
root=$PWD
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d $out/g$i -o sq --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --substreams 1 --no-graph --unroll 1 > $out/g$i.log 2>&1) || echo "group $i failed"
done
python3 - $out <<'PY'
import csv, glob, sys, statistics, collections
acc = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "rr_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, statistics.median(v[1:] or v), len(v))
PY
