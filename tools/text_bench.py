#!/usr/bin/env python3
"""Time bgr_align_fasta_text on one piece of synthetic FASTA (diagnostic): python tools/text_bench.py [reads] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import bgreat_amd as B  # noqa: E402
from tools.synth import Synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
s = Synth(4_600_000, 140, 2, 31, 20261003)
seqs, offs = s.unitigs()
g = B.Graph.build(31, seqs, offs)
al = B.Aligner(g, 0)
f = "/tmp/text_bench.fa"
s.write_reads(f, 0, n, 150, 2, 77, threads=16)
text = np.fromfile(f, dtype=np.uint8)
lib = B.lib()


def pinned(nbytes):
    p = C.c_void_p()
    B._check(lib.bgr_host_alloc(nbytes, C.byref(p)))
    return p, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))


hp, tin = pinned(len(text) + 64)
tin[: len(text)] = text
ho, pout = pinned(len(text))
hn, nout = pinned(len(text))
b = B.TextBatch(C.sizeof(B.TextBatch), tin.ctypes.data, len(text), 1, 0, pout.ctypes.data, len(text), nout.ctypes.data, len(text), 0, 0, 0, 0, None)
p = B.Params(B.MODE_GREEDY, 2, 2, 0)
for i in range(reps):
    t0 = time.perf_counter()
    B._check(lib.bgr_align_fasta_text(al.h, C.byref(p), C.byref(b)))
    dt = time.perf_counter() - t0
    print("rep %d: %.2f ms  %.1f Mreads/s  (%d records, %d accepted, %d + %d bytes out, irregular %d)" % (i, dt * 1e3, n / dt / 1e6, b.n_records, b.n_accepted, b.paths_bytes, b.notaligned_bytes, b.irregular), flush=True)
print(al.kernel_times())

if hasattr(lib, "bgr_x_times_read") or True:
    try:
        t = (C.c_ulonglong * 64)()
        lib.bgr_x_times_read.restype = C.c_int
        if lib.bgr_x_times_read(t) == 0:
            for slot, name in ((0, "parse tile 0"), (1, "parse tile mid"), (2, "format tile 0"), (3, "format tile mid")):
                v = [t[slot * 16 + i] for i in range(16)]
                print(name, " ".join("%d:%.1f" % (i, (v[i] - v[0]) / 100.0) for i in range(1, 16) if v[i]), "(us, 100 MHz clock assumed)")
    except AttributeError:
        pass
