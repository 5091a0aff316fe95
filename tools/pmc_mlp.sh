#!/bin/bash
# SQ counters of the learner's MFMA kernels (one rocprofv3 --pmc pass per counter group).   bash tools/pmc_mlp.sh <outdir> [lib.so]
out=${1:-/tmp/pmc_mlp}; case $out in /*) ;; *) out=$PWD/$out;; esac
lib=${2:-brax-rodent-run_amd/csrc/librodent_hip.so}
mkdir -p $out
export TMPDIR=/tmp RR_LIB=$PWD/$lib
root=$PWD
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $out/g$i -o m --output-format csv -- python3 $root/tools/bench_learner_kernels.py > $out/g$i.log 2>&1) || echo "group $i failed ($grp)"
done
python3 - $out <<'PY'
import csv, glob, sys, statistics, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        for name in ("rr_mlp_forward_kernel", "rr_mlp_value_backward_kernel", "rr_mlp_dw_kernel<2, 2, 2, 2"):
            if name in k:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print("==", name)
    for k, v in sorted(cs.items()):
        print("  ", k, statistics.median(v), len(v))
PY
