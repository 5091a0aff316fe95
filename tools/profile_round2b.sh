#!/bin/bash
# Round-2 diagnostics (GPU box): SQ counters + per-phase cycles of the step kernel, kernel trace of the PPO learner, and the
# GPU tests added since the last run.   bash tools/profile_round2b.sh <tag>
tag=${1:-x}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp OMP_NUM_THREADS=16
timeout -k 10 300 python -m pytest tests/test_gpu_clip_rollout.py -m gpu -q 2>&1 | tail -n 5
echo "== phase profile"; timeout -k 10 120 python3 tools/phase_prof.py > $out/phase.txt 2>&1; cat $out/phase.txt | head -n 40
echo "== SQ"; bash tools/pmc_sq.sh gpurun_out/prof_$tag/sq > $out/sq.txt 2>&1; cat $out/sq.txt
for f in 0 1; do
  cd /tmp
  RR_FUSED_MLP=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ppo$f -o kt --output-format csv -- python3 $root/bench.py --config 3 --steps 1 --warmup 1 > $out/ppo$f.log 2>&1 || echo "ppo trace failed"
  cd $root
  python3 tools/prof_summary.py $(find /tmp/ppo$f -name "*kernel_stats.csv" | head -n 1) $out/ppo_kernel_stats_fused$f.csv
  echo "== PPO kernels, RR_FUSED_MLP=$f"; python3 - /tmp/ppo$f <<'PY'
import csv, glob, sys
p = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(p)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{int(r["TotalDurationNs"])/1e6:9.1f} ms {100*int(r["TotalDurationNs"])/tot:5.1f}% calls {r["Calls"]:>7s} avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:110]}')
PY
done
