"""Condense a rocprofv3 kernel_stats.csv into a short table (kernel names cut to 130 chars)."""
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
with open(dst, "w") as f:
    f.write("kernel,calls,total_ms,avg_us,percent,min_us,max_us\n")
    for r in rows[:12]:
        name = r["Name"].replace(",", ";")[:130]
        f.write(f'"{name}",{r["Calls"]},{int(r["TotalDurationNs"])/1e6:.3f},{float(r["AverageNs"])/1e3:.1f},{r["Percentage"]},{int(r["MinNs"])/1e3:.1f},{int(r["MaxNs"])/1e3:.1f}\n')
