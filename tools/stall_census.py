#!/usr/bin/env python3
"""Summarise tools/stall_census.sh: per mapping kernel, counters per launch and the ratios that say what the waves wait for."""
import csv
import glob
import os
import re
import sys

out = sys.argv[1]
tot, calls = {}, {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    seen = {}
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name", "")
        if "bgr_" not in kn:
            continue
        m = re.search(r"(bgr_[a-z0-9_]+)(<[^>]*>)?", kn)
        nm = m.group(1) + (m.group(2) or "")
        c = row["Counter_Name"]
        tot.setdefault(nm, {})
        tot[nm][c] = tot[nm].get(c, 0.0) + float(row["Counter_Value"])
        seen.setdefault((nm, c), set()).add(row.get("Dispatch_Id"))
    for (nm, c), ids in seen.items():
        calls.setdefault(nm, {})[c] = len(ids)
for nm, cs in sorted(tot.items(), key=lambda x: -x[1].get("SQ_WAVE_CYCLES", 0)):
    g = lambda c: cs.get(c, 0.0) / max(1, calls[nm].get(c, 1))   # per launch
    wc = g("SQ_WAVE_CYCLES")
    if wc <= 0:
        continue
    print("== %s  (%d launches profiled; per launch) ==" % (nm, calls[nm].get("SQ_WAVE_CYCLES", 0)))
    for c in sorted(cs):
        print("   %-28s %16.0f" % (c, g(c)))
    r = lambda a, b: (g(a) / g(b)) if g(b) else float("nan")
    print("   -- ratios --")
    print("   waves resident on average (SQ_LEVEL_WAVES / SQ_BUSY_CYCLES ... per SE)      %8.2f" % r("SQ_LEVEL_WAVES", "SQ_BUSY_CYCLES"))
    print("   of a wave's resident cycles: waiting for anything (SQ_WAIT_ANY)              %8.3f" % r("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"))
    print("   ... waiting in s_waitcnt (SQ_WAIT_INST_ANY)                                  %8.3f" % r("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"))
    print("   ... of which for LDS (SQ_WAIT_INST_LDS)                                      %8.3f" % r("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"))
    print("   ... issuing any instruction (SQ_ACTIVE_INST_ANY)                             %8.3f" % r("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"))
    for c in ("VALU", "SCA", "LDS", "VMEM", "FLAT", "MISC"):
        print("   ... issuing %-5s (SQ_ACTIVE_INST_%s)                                      %8.3f" % (c, c, r("SQ_ACTIVE_INST_" + c, "SQ_WAVE_CYCLES")))
    print("   vector memory reads in flight per resident wave (LEVEL_VMEM / WAVE_CYCLES)   %8.3f" % r("SQ_INST_LEVEL_VMEM", "SQ_WAVE_CYCLES"))
    print("   mean latency of a vector memory read, cycles (LEVEL_VMEM x 4 / INSTS_VMEM_RD)%8.0f" % (4 * r("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM_RD")))
    print("   mean latency of an LDS instruction, cycles (LEVEL_LDS x 4 / INSTS_LDS)       %8.0f" % (4 * r("SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS")))
    print("   LDS bank conflict cycles / LDS active cycles                                 %8.3f" % r("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"))
    print("   instruction fetches per 100 instructions issued                              %8.2f" % (100 * g("SQ_IFETCH") / max(1.0, g("SQ_INSTS_VALU") + g("SQ_INSTS_SALU"))))
    print("   mean instruction fetch latency, cycles (IFETCH_LEVEL x 4 / IFETCH)           %8.0f" % (4 * r("SQ_IFETCH_LEVEL", "SQ_IFETCH")))
    print("   branches per 100 instructions                                                %8.2f" % (100 * g("SQ_INSTS_BRANCH") / max(1.0, g("SQ_INSTS_VALU") + g("SQ_INSTS_SALU"))))
    print("   lanes switched on per vector instruction (THREAD_CYCLES_VALU / ACTIVE_INST_VALU / 64 ... if both count quad-cycles) %8.3f" % (r("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU") / 64.0))
    print()
