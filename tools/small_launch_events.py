"""What the HIP events around a mapping launch's kernels cost a small launch (diagnostic): python tools/small_launch_events.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bgreat_amd as B
from tools.synth import Synth
s = Synth(4_600_000, 140, 2, 31, 20261003)
seqs, offs = s.unitigs()
g = B.Graph.build(31, seqs, offs)
for R in (131072, 262144, 1048576):
    reads, _ = s.reads(0, R, 150, 2, 77, threads=16)
    db = B.DeviceBuffer(0, reads)
    do = B.DeviceBuffer(0, np.arange(R + 1, dtype=np.uint64) * np.uint64(150))
    for ev in (1, 0, 1, 0):
        al = B.Aligner(g, 0)
        al.set_knob(B.KNOB_KERNEL_EVENTS, ev)
        for _ in range(5):
            al.align_device(db.data_ptr(), do.data_ptr(), R, R * 150, 150, m=2, effort=2, mode=0)
        al.sync()
        t0 = time.perf_counter()
        K = 60
        for _ in range(K):
            al.align_device(db.data_ptr(), do.data_ptr(), R, R * 150, 150, m=2, effort=2, mode=0)
        al.sync()
        dt = (time.perf_counter() - t0) / K
        print("reads %8d  events %d  %.1f us per launch  %.0f Mreads/s" % (R, ev, dt * 1e6, R / dt / 1e6), flush=True)
        al.close()
    db.free(); do.free()
