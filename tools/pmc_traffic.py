"""Per-launch HBM-side traffic of rr_step_kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units).
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]"""
import csv, json, statistics, sys
def vals(path, counter):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if "rr_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
f = vals(sys.argv[1], "FETCH_SIZE")[1:]     # [0] is the reset (forward-only) launch
w = vals(sys.argv[2], "WRITE_SIZE")[1:]
out = {"kernel": "rr_step_kernel", "launches_sampled": [len(f), len(w)],
       "FETCH_SIZE_KB_median": statistics.median(f), "WRITE_SIZE_KB_median": statistics.median(w),
       "fetch_bytes_raw": statistics.median(f) * 1024, "write_bytes": statistics.median(w) * 1024,
       "note": "gfx950: FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x; this kernel issues 4 B/lane "
               "accesses (uncalibrated), so the raw value is reported; WRITE_SIZE is exact for the store shapes calibrated "
               "in MI355X_MICROARCH.md"}
out["traffic_bytes_per_launch"] = out["fetch_bytes_raw"] + out["write_bytes"]
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
