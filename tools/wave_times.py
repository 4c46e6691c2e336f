#!/usr/bin/env python3
"""When do the waves of bgr_align_greedy_multi_kernel start, get their table, finish their share of the batch and finish their queue?
Needs the diagnostic build (make -C bgreat_amd BUILD=build_phase LIBDIR=lib_phase EXTRA=-DBGR_PHASE_TIMING lib_phase/libbgreat_gpu.so) loaded
through BGR_LIB_PATH.  usage (GPU box): BGR_LIB_PATH=... python tools/wave_times.py [--workload ecoli|small|chr1] [--reads N]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bgreat_amd as B  # noqa: E402
from tools.synth import Synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="ecoli")
ap.add_argument("--reads", type=int, default=0)
args = ap.parse_args()
G, d, L, R, AL, M, MODE = {"ecoli": (4_600_000, 140, 150, 5_000_000, 2, 2, 0), "small": (250_000, 75, 100, 1_000_000, 2, 2, 0), "chr1": (230_000_000, 175, 150, 5_000_000, 2, 2, 0),
                          "branchy": (50_000_000, 36, 250, 2_000_000, 4, 5, 1)}[args.workload]
R = args.reads or R
s = Synth(G, d, AL, 31, 20261003)
seqs, offs = s.unitigs()
g = B.Graph.build(31, seqs, offs)
al = B.Aligner(g, 0)
reads, _ = s.reads(0, R, L, M, 77, threads=16)
db = B.DeviceBuffer(0, reads)
do = B.DeviceBuffer(0, np.arange(R + 1, dtype=np.uint64) * np.uint64(L))
for _ in range(3):
    al.align_device(db.data_ptr(), do.data_ptr(), R, R * L, L, m=M, effort=2, mode=MODE)
al.sync()
lib = B.lib()
lib.bgr_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
n = C.c_uint64()
buf = np.zeros(4 * 65536, dtype=np.uint64)
B._check(lib.bgr_debug_wave_times(al.h, buf.ctypes.data, 65536, C.byref(n)))
t = buf[: 4 * n.value].reshape(-1, 4).astype(np.float64) / 100.0   # microseconds (100 MHz)
t0 = t[:, 0].min()
t -= t0
end = t[:, 3].max()
print("workload %s, %d reads, %d waves (launch %s); all times in microseconds from the first wave's start; kernel ends at %.1f" % (args.workload, R, n.value, al.launch_info(), end))
names = ["wave starts", "table staged (first barrier passed)", "share of the batch done", "queue done = wave ends"]
for j in range(4):
    c = t[:, j]
    print("   %-38s min %8.1f  p10 %8.1f  median %8.1f  p90 %8.1f  max %8.1f" % (names[j], c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
life = t[:, 3] - t[:, 0]
print("   a wave lives %.1f us on average = %.3f of the kernel's duration; busy wave-slots over time:" % (life.mean(), life.mean() / end))
edges = np.linspace(0, end, 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    alive = ((t[:, 0] <= mid) & (t[:, 3] > mid)).sum()
    staged = ((t[:, 1] <= mid) & (t[:, 3] > mid)).sum()
    print("      %7.0f us  resident %5d  past the staging barrier %5d" % (mid, alive, staged))
# per workgroup: spread of its waves' ends (a workgroup holds its LDS until its last wave ends)
wpb = al.launch_info()["threads"] // 64
wg_end = t[:, 3].reshape(-1, wpb)
print("   workgroups: last wave ends %.1f us (median) after the workgroup's first one; slowest workgroup ends at %.1f, median workgroup at %.1f" % (
    np.median(wg_end.max(axis=1) - wg_end.min(axis=1)), wg_end.max(), np.median(wg_end.max(axis=1))))
