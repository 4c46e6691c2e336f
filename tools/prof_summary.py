#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/prof.sh: per-kernel stats + per-dispatch PMC sums for the mapping kernel."""
import csv
import glob
import os
import sys

import re

out = sys.argv[1]


def short(name):
    m = re.search(r"(bgr_[a-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def find(sub, pattern):
    return sorted(glob.glob(os.path.join(out, sub, "**", pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("kt", "*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        print("  %-70s calls %5s  total %12s ns  avg %12s ns  %6s %%" % (row.get("Name", "")[:70], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
bydur = {}
for f in find("kt", "*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        if "bgr_" in row.get("Kernel_Name", ""):
            nm = row["Kernel_Name"]
            nm = nm[nm.find("bgr_"):][:60]
            bydur.setdefault(nm, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row))
kt_ms = {}
for nm, lst in bydur.items():
    durs = [d for d, _ in lst]
    big = [d for d in durs if d * 2 >= max(durs)]
    kt_ms[short(nm)] = sum(big) / len(big) / 1e6
    last = lst[-1][1]
    print("  %-60s dispatches %3d  full-size %3d  mean %.4f ms  (VGPR %s SGPR %s LDS %s grid %s wg %s)" % (
        nm, len(durs), len(big), sum(big) / len(big) / 1e6, last.get("VGPR_Count"), last.get("SGPR_Count"), last.get("LDS_Block_Size"),
        last.get("Grid_Size_X"), last.get("Workgroup_Size_X")))

print("== PMC (mean per full-size dispatch, per kernel) ==")

# full-size dispatches of a kernel = those within 2x of its longest one (the parity-sample launches are much shorter)
vals = {}   # kernel -> counter -> mean
counts = {}
for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in find(sub, "*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "bgr_" in r.get("Kernel_Name", "")]
        longest = {}
        for r in rows:
            k = short(r["Kernel_Name"])
            longest[k] = max(longest.get(k, 0), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        acc = {}
        for r in rows:
            k = short(r["Kernel_Name"])
            if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 2 < longest[k]:
                continue
            key = (k, r["Counter_Name"], r["Dispatch_Id"])
            acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
        per = {}
        for (k, name, did), v in acc.items():
            per.setdefault((k, name), []).append(v)
        for (k, name), lst in per.items():
            vals.setdefault(k, {})[name] = sum(lst) / len(lst)
            counts[k] = len(lst)
for k in sorted(vals):
    v = vals[k]
    print("  -- %s (%d dispatches)" % (k, counts.get(k, 0)))
    for name in sorted(v):
        print("     %-24s %18.1f" % (name, v[name]))
    if "FETCH_SIZE" in v:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; MI355X_MICROARCH.md: gfx950 tallies 128-B read requests of wide streams at 64 B (x2 upper bound)
        print("     FETCH_SIZE*1024 = %.1f MB (x2 gfx950 wide-stream correction = %.1f MB)" % (v["FETCH_SIZE"] * 1024 / 1e6, v["FETCH_SIZE"] * 2048 / 1e6))
    if "WRITE_SIZE" in v:
        print("     WRITE_SIZE*1024 = %.1f MB" % (v["WRITE_SIZE"] * 1024 / 1e6))
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_ACTIVE_INST_VALU2" in v:
        # busy vector issue slots = instructions minus those issued as the second of a pair (profiles/r03_valu2_pmc_calibration.txt)
        busy = v["SQ_ACTIVE_INST_VALU"] - v["SQ_ACTIVE_INST_VALU2"]
        print("     VALU busy issue slots = ACTIVE_INST_VALU - ACTIVE_INST_VALU2 = %.4e  (x4 = %.4e SIMD-cycles; %.3f cycles per instruction)" % (busy, 4 * busy, 4 * busy / max(1.0, v["SQ_ACTIVE_INST_VALU"])))
        if v.get("SQ_BUSY_CU_CYCLES"):
            print("     VALU busy fraction of the CU-busy cycles = slots / SQ_BUSY_CU_CYCLES = %.4f" % (busy / v["SQ_BUSY_CU_CYCLES"]))
        if k in kt_ms:
            print("     VALU busy fraction at 1024 SIMDs x 2.4 GHz over the kernel's mean duration (%.4f ms, kernel trace) = %.4f" % (kt_ms[k], 4 * busy / (1024 * 2.4e9 * kt_ms[k] / 1e3)))
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
        print("     L2 hit rate = %.4f" % (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
