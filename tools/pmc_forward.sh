#!/bin/bash
# SQ counters of the learner-shape forward alone (one rocprofv3 --pmc pass per counter group), per dispatch.   bash tools/pmc_forward.sh <outdir>
out=${1:-/tmp/pmc_fwd}; case $out in /*) ;; *) out=$PWD/$out;; esac
mkdir -p $out
export TMPDIR=/tmp
root=$PWD
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" "SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $out/g$i -o m --output-format csv -- python3 $root/tools/pmc_forward.py > $out/g$i.log 2>&1) || echo "group $i failed ($grp)"
done
python3 - $out <<'PY'
import csv, glob, sys, statistics, collections
acc = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "rr_mlp_forward_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: statistics.median(v) for k, v in acc.items()}
for k in sorted(c): print(f"  {k:28s} {c[k]:16.0f}   ({len(acc[k])} dispatches)")
if "SQ_WAVE_CYCLES" in c:
    w = c["SQ_WAVE_CYCLES"]
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY"):
        if k in c: print(f"  {k} / SQ_WAVE_CYCLES = {c[k] / w:.3f}")
    if "SQ_INSTS_MFMA" in c: print(f"  VALU (incl. MFMA) per MFMA = {c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA']:.2f}, LDS per MFMA = {c.get('SQ_INSTS_LDS', 0) / c['SQ_INSTS_MFMA']:.2f}, SALU per MFMA = {c.get('SQ_INSTS_SALU', 0) / c['SQ_INSTS_MFMA']:.2f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c: print(f"  MFMA busy / SQ busy = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f} (units as reported)")
PY
