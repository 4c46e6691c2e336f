#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/prof.sh: per-kernel stats + per-dispatch PMC sums for the mapping kernel."""
import csv
import glob
import os
import sys

out = sys.argv[1]


def find(sub, pattern):
    return sorted(glob.glob(os.path.join(out, sub, "**", pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("kt", "*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        print("  %-70s calls %5s  total %12s ns  avg %12s ns  %6s %%" % (row.get("Name", "")[:70], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
durs = []
for f in find("kt", "*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        if "bgr_align" in row.get("Kernel_Name", ""):
            durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            last = row
if durs:
    print("  mapping kernel dispatches: %d, durations ms: %s" % (len(durs), ", ".join("%.3f" % (d / 1e6) for d in durs)))
    print("  VGPR %s  SGPR %s  LDS %s  grid %s  wg %s" % (last.get("VGPR_Count"), last.get("SGPR_Count"), last.get("LDS_Block_Size"), last.get("Grid_Size_X"), last.get("Workgroup_Size_X")))
    big = [d for d in durs if d > 1_000_000]
    if big:
        print("  full-size dispatches: %d, mean %.4f ms" % (len(big), sum(big) / len(big) / 1e6))

print("== PMC (per dispatch of the mapping kernel; mean over dispatches) ==")
vals = {}
for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in find(sub, "*counter_collection.csv"):
        acc = {}
        for row in csv.DictReader(open(f)):
            if "bgr_align" not in row.get("Kernel_Name", ""):
                continue
            if int(row["End_Timestamp"]) - int(row["Start_Timestamp"]) < 1_000_000:
                continue  # skip the small parity-sample launches: only the timed full-size dispatches
            key = (row["Counter_Name"], row["Dispatch_Id"])
            acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
        per = {}
        for (name, did), v in acc.items():
            per.setdefault(name, []).append(v)
        for name, lst in per.items():
            vals[name] = sum(lst) / len(lst)
for k in sorted(vals):
    print("  %-24s %18.1f" % (k, vals[k]))
if "FETCH_SIZE" in vals:
    # MI355X_MICROARCH.md: FETCH_SIZE is in KiB-like units of 1024 B? rocprofv3 reports FETCH_SIZE in KB; gfx950 under-counts wide streams by 2x.
    print("  FETCH_SIZE*1024 = %.1f MB (x2 gfx950 wide-stream correction = %.1f MB)" % (vals["FETCH_SIZE"] * 1024 / 1e6, vals["FETCH_SIZE"] * 2048 / 1e6))
if "WRITE_SIZE" in vals:
    print("  WRITE_SIZE*1024 = %.1f MB" % (vals["WRITE_SIZE"] * 1024 / 1e6))
if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
    print("  L2 hit rate = %.4f" % (vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])))
