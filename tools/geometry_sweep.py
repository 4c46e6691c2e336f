"""Is the launch planner's choice (bgreat_amd/csrc/launch_plan.h: constants measured on the five bench graphs) any good on graphs of OTHER shapes?
For each of a handful of graphs with different unitig lengths, allele counts and key-table sizes, and per mode, the device-resident rate of the
planner's own geometry next to forced alternatives (bgr_aligner_configure: key table staged / not staged, waves per workgroup, workgroups per
CU); prints each alternative as a fraction of the best and flags a planner choice below 0.90 of the best.  GPU box; ~3 minutes.
(Test infrastructure / measurement; results kept as profiles/rNN_geometry_sweep.txt.)

  python tools/geometry_sweep.py [reads per launch, default 1000000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bgreat_amd as B
from tools.synth import Synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
GRAPHS = [  # name, genome, spacing, alleles, read length, m
    ("long unitigs (a site every ~600 bp)", 6_000_000, 600, 2, 150, 2),
    ("short unitigs, 3 alleles every ~45 bp", 3_000_000, 45, 3, 150, 3),
    ("mid-size table (12 Mb genome: staged once per CU)", 12_000_000, 140, 2, 150, 2),
    ("tiny graph, 100 bp reads", 120_000, 90, 2, 100, 2),
    ("250 bp reads, 2 alleles every ~100 bp", 8_000_000, 100, 2, 250, 4),
]
ALTS = [("planner", (0, 0, 0)), ("unstaged", (0, 0, 1)), ("staged", (0, 0, 2)), ("staged 16x1", (16, 1, 2)), ("staged 12x2", (12, 2, 2)), ("staged 8x3", (8, 3, 2)),
        ("unstaged 16x2", (16, 2, 1)), ("unstaged 8x4", (8, 4, 1)), ("unstaged 4x6", (4, 6, 1))]


def rate(g, reads_d, offs_d, n, L, m, mode, cfg, steps=8):
    al = B.Aligner(g, 0)
    try:
        al.configure(*cfg)
        for _ in range(2):
            al.align_device(reads_d.data_ptr(), offs_d.data_ptr(), n, n * L, L, m=m, effort=2, mode=mode)
        al.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            al.align_device(reads_d.data_ptr(), offs_d.data_ptr(), n, n * L, L, m=m, effort=2, mode=mode)
        al.sync()
        dt = time.perf_counter() - t0
        info = al.launch_info()
        return steps * n / dt / 1e6, info
    finally:
        al.close()


worst = 1.0
for name, G, d, a, L, m in GRAPHS:
    s = Synth(G, d, a, 31, 4242)
    seqs, offs = s.unitigs()
    g = B.Graph.build(31, seqs, offs)
    gi = g.info()
    reads, _ = s.reads(0, N, L, m, 99, threads=16)
    reads_d = B.DeviceBuffer(0, reads)
    offs_d = B.DeviceBuffer(0, np.arange(N + 1, dtype=np.uint64) * np.uint64(L))
    print("== %s: %d unitigs, key table %d KB, blob %.1f MB, %d x %d bp" % (name, gi["n_unitigs"], gi["mphf_bytes"] // 1024, gi["blob_bytes"] / 1e6, N, L), flush=True)
    for mode, mname in ((B.MODE_GREEDY, "greedy"), (B.MODE_EXHAUSTIVE, "exhaustive")):
        res = []
        for an, cfg in ALTS:
            try:
                r, info = rate(g, reads_d, offs_d, N, L, m, mode, cfg)
                res.append((an, r, info))
            except B.BgrError as ex:
                res.append((an, 0.0, {"error": str(ex)[:60]}))
        best = max(r for _, r, _ in res)
        pl = res[0][1]
        worst = min(worst, pl / best)
        print("  %-10s planner %7.1f Mreads/s = %.2f of the best (%s: %d x %d threads, table in LDS %s)%s" % (mname, pl, pl / best, max(res, key=lambda x: x[1])[0], res[0][2].get("blocks", 0), res[0][2].get("threads", 0),
              res[0][2].get("mphf_in_lds"), "   <-- BELOW 0.90" if pl < 0.9 * best else ""))
        print("             " + "  ".join("%s %.2f" % (an, r / best) for an, r, _ in res[1:]), flush=True)
    reads_d.free(); offs_d.free(); g.close()
print("planner's worst case: %.2f of the best alternative" % worst)
