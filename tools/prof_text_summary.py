#!/usr/bin/env python3
"""Summarise tools/prof_text_route.sh: per kernel of the text route (calls, mean duration, share) and per copy direction (bytes, busy time,
GB/s while a copy is running, against the link's ~50 GB/s measured per copy)."""
import csv
import glob
import os
import re
import sys

out, n_reads = sys.argv[1], int(sys.argv[2])


def short(name):
    m = re.search(r"(bgr_[a-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


for mode in ("fasta", "fastq"):
    d = os.path.join(out, mode)
    print("== bin/bgreat, %s, %d reads x 150 bp, -t 16, 1 GPU (rocprofv3 --kernel-trace --memory-copy-trace --stats) ==" % (mode.upper(), n_reads))
    for l in open(os.path.join(out, mode + ".err")):
        if l.startswith("bgreat:"):
            print("   " + l.rstrip())
    k = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            nm = short(row["Kernel_Name"])
            dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            k.setdefault(nm, []).append(dur)
    tot = sum(sum(v) for v in k.values()) or 1
    print("   %-52s %7s %12s %12s %7s" % ("kernel", "calls", "mean us", "total ms", "share"))
    for nm, v in sorted(k.items(), key=lambda x: -sum(x[1])):
        print("   %-52s %7d %12.1f %12.2f %6.1f%%" % (nm[:52], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, 100.0 * sum(v) / tot))
    print("   all kernels: %.1f ms of device time = %.0f Mreads/s if nothing else ran" % (tot / 1e6, n_reads / (tot / 1e9) / 1e6))
    c = {}
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            kind = row.get("Direction") or row.get("Name") or "?"
            dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            nbytes = int(row.get("Bytes") or row.get("Size") or 0) if (row.get("Bytes") or row.get("Size")) else 0
            c.setdefault(kind, []).append((dur, nbytes))
    # (this rocprofv3 writes no byte counts into the copy trace: the bytes are the run's own -- the input file up, the two output files down)
    sizes = {}
    for l in open(os.path.join(out, mode + ".files")):
        f = l.split()
        if len(f) >= 9:
            sizes[os.path.basename(f[-1])] = int(f[4])
    up = sizes.get("r.fa", 0)   # (FASTQ pieces travel without their '+' and quality lines: the bytes of the header and read lines = the FASTA file's)
    down = sizes.get("paths", 0) + sizes.get("notAligned.fa", 0)
    for kind, v in sorted(c.items()):
        td = sum(d_ for d_, _ in v)
        nb = up if "HOST_TO_DEVICE" in kind else down if "DEVICE_TO_HOST" in kind else 0
        print("   copies %-28s %7d  %10.1f MB  busy %9.2f ms  = %5.1f GB/s while a copy runs, %5.1f bytes per read" % (kind[:28], len(v), nb / 1e6, td / 1e6, nb / max(1, td), nb / n_reads))
    print()
