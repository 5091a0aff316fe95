"""Condense the rocprofv3 --pmc passes of tools/profile_round3.sh: per launch of the MULTI-STEP instance of rr_step_kernel (the one the
bench times; the single-step launches of the warm-up and of the for-the-record pass are other instances and are left out).
usage: pmc_round3.py "<bench command>" <outdir> <fetch dir> <write dir> <sq dir>...      (env RR_PMC_INSTANCE=unroll|pair, RR_PMC_T, RR_PMC_N)"""
import collections, csv, glob, json, statistics, sys

import os
cmd, out = sys.argv[1], sys.argv[2]
WHICH = os.environ.get("RR_PMC_INSTANCE", "unroll")
T, N = int(os.environ.get("RR_PMC_T", "20")), int(os.environ.get("RR_PMC_N", "2048"))       # env steps per launch, envs


def rows(d):
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        yield from csv.DictReader(open(p))


def unroll(name):      # template arguments ..., NEWTON, UNROLL, ACTOR, PAIR, DYN: the multi-step instance has UNROLL = true, the two-wave one PAIR
    if "rr_step_kernel" not in name:
        return False
    args = name.replace(" ", "").split(">(")[0].split(",")
    if len(args) < 11:
        return False
    tail = args[-5:]            # NEWTON, UNROLL, ACTOR, PAIR, DYN
    return tail[1] == "true" if WHICH == "unroll" else tail[3] == "true"


acc = collections.defaultdict(list)
names = set()
for d in sys.argv[3:]:
    for r in rows(d):
        if unroll(r["Kernel_Name"]):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            names.add(r["Kernel_Name"].split("(")[0])
med = {k: statistics.median(v) for k, v in acc.items()}
if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
    t = {"kernel": "rr_step_kernel (%s instance)" % ("multi-step" if WHICH == "unroll" else "two-wave PAIR"), "instances": sorted(names), "command": cmd, "env_steps_per_launch": T, "envs": N,
         "launches_sampled": [len(acc["FETCH_SIZE"]), len(acc["WRITE_SIZE"])],
         "FETCH_SIZE_KB_median": med["FETCH_SIZE"], "WRITE_SIZE_KB_median": med["WRITE_SIZE"],
         "fetch_bytes_raw": med["FETCH_SIZE"] * 1024, "write_bytes": med["WRITE_SIZE"] * 1024,
         "note": "gfx950: FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x; this kernel issues 4 B/lane accesses "
                 "(uncalibrated), so the raw value is reported; WRITE_SIZE is exact for the calibrated store shapes (MI355X_MICROARCH.md)"}
    t["traffic_bytes_per_launch"] = t["fetch_bytes_raw"] + t["write_bytes"]
    t["traffic_bytes_per_env_step_of_all_envs"] = t["traffic_bytes_per_launch"] / T
    json.dump(t, open(out + "/pmc_traffic.json", "w"), indent=1)
if "SQ_INSTS_VALU" in med:
    per = lambda k: med[k] / (N * T) if k in med else None
    s = {"source": f"tools/profile_round3.sh: rocprofv3 --kernel-trace --pmc on `{cmd}`, one counter group per run, median over the "
                   f"launches of the {'multi-step' if WHICH == 'unroll' else 'two-wave PAIR'} instance; raw values per launch ({N} envs x {T} env steps)",
         "env_steps_per_launch": T, "counters": med,
         "instructions_per_env_step": sum(med.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")) / (N * T),
         "valu_instructions_per_env_step": per("SQ_INSTS_VALU"), "lds_instructions_per_env_step": per("SQ_INSTS_LDS")}
    if "SQ_ACTIVE_INST_VALU" in med and "SQ_WAVE_CYCLES" in med:
        s["valu_busy_frac"] = 2 * med["SQ_ACTIVE_INST_VALU"] / med["SQ_WAVE_CYCLES"]       # 2 waves per SIMD in both instances
        s["valu_cycles_per_instruction"] = 4 * med["SQ_ACTIVE_INST_VALU"] / med["SQ_INSTS_VALU"]
    if "SQ_LDS_IDX_ACTIVE" in med and "SQ_BUSY_CYCLES" in med:
        s["lds_bank_conflict_frac"] = med["SQ_LDS_BANK_CONFLICT"] / med["SQ_LDS_IDX_ACTIVE"]
    s["notes"] = "valu_busy_frac = 2 waves per SIMD x VALU-active share of a wave's cycles (both in quad-cycles)"
    json.dump(s, open(out + "/sq.json", "w"), indent=1)
