#!/bin/bash
out=${1:-gpurun_out/ic}
mkdir -p $out
export TMPDIR=/tmp
root=$PWD
(cd /tmp && rocprofv3 -L > $root/$out/counters.txt 2>&1)
grep -i -E "icache|ifetch|INST_LEVEL|SQC_" $out/counters.txt | cut -c1-200 | sort -u | head -n 60 > $out/counters_ic.txt
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_ANY SQ_WAVES" "SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READY SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d $root/$out/g$i -o ic --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $root/$out/g$i.log 2>&1) || echo "group $i failed"
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
