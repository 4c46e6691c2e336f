"""Randomised parity campaign against the oracle (written for the exhaustive level search, exh_dp; also greedy / anchors): 60 random (k, site spacing, alleles,
read length, m, -i, N rate, level cap) configurations per seed, N in unitigs now and then, small caps to push reads through
the depth-first and HBM-stack passes.  Run on a GPU box: python tools/fuzz_parity.py [seed] [exhaustive|greedy|anchors] [by-level|depth-first|auto].
(Test infrastructure.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bgreat_amd as B, oracle_py
B.set_options_from_string(os.environ.get("BGR_FUZZ_OPTIONS"))   # (library options of this campaign: the library itself reads no environment)
from synth import Synth
SEARCH = {"by-level": 2, "depth-first": 1, "auto": 0}[sys.argv[3] if len(sys.argv) > 3 else "by-level"]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
MODE = sys.argv[2] if len(sys.argv) > 2 else "exhaustive"   # exhaustive | greedy | anchors
bad = 0
fracs = []
t0 = time.time()
for it in range(60):
    k = int(rng.choice([5, 8, 12, 15, 21, 25, 31, 32]))
    d = int(rng.integers(k + 2, 4 * k))
    alleles = int(rng.integers(2, 5))
    L = int(rng.integers(k + 2, 320))
    m = int(rng.integers(0, 7))
    partial = bool(rng.integers(0, 2))
    nfrac = float(rng.choice([0, 0, 0.002, 0.01]))
    s = Synth(int(rng.integers(20000, 150000)), d, alleles, k, 1000 + it)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, 2500, L, m + 1, 5000 + it)
    if nfrac:
        reads = reads.copy(); idx = rng.choice(len(reads), size=max(1, int(len(reads) * nfrac)), replace=False); reads[idx] = ord("N")
    if it % 7 == 3:   # unitigs with N as well
        seqs = seqs.copy(); seqs[rng.choice(len(seqs), size=20, replace=False)] = ord("N")
    anc = MODE == "anchors"
    gamma = float(rng.choice([0.0, 0.0, 1.07, 1.8, 3.0]))   # key table fill: dense (LDS staging) ... sparse (L2 probing)
    g = B.Graph.build(k, seqs, offs, gamma, anchors=anc); al = B.Aligner(g, 0); o = oracle_py.Oracle(k, seqs, offs, anchors=anc)
    if it % 3 == 1: al.configure(0, 0, 1)   # key table probed in L2 instead of LDS (second bucket read only when the first is full)
    effort = int(rng.choice([0, 1, 2, 2, 3, 8]))
    cap = int(rng.choice([3, 6, 16, 24]))
    al.set_knob(B.KNOB_EXH_FRAME_CAP, cap)
    al.set_knob(B.KNOB_EXH_SEARCH, SEARCH)
    if MODE == "exhaustive":
        p1, po1, st1 = al.align(reads, roffs, m=m, mode=B.MODE_EXHAUSTIVE, partial=partial)
        p2, po2, st2 = o.align(reads, roffs, m=m, mode=1, partial=partial)
    else:
        gm, om = (B.MODE_ANCHORS, 2) if anc else (B.MODE_GREEDY, 0)
        p1, po1, st1 = al.align(reads, roffs, m=m, effort=effort, mode=gm)
        p2, po2, st2 = o.align(reads, roffs, m=m, effort=effort, mode=om)
    c1, c2 = al.counters(), o.counters()
    if MODE != "exhaustive":
        c1["overlaps"] = c2["overlaps"] = 0
    ok = np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2) and c1 == c2
    fracs.append(float(((st1 & 3) == 2).mean()))
    if not ok:
        bad += 1
        print("MISMATCH", it, dict(k=k, d=d, alleles=alleles, L=L, m=m, partial=partial, nfrac=nfrac, cap=cap), flush=True)
print("configs 60 bad", bad, "aligned fractions: mean %.2f, >0.5 in %d configs, >0.1 in %d" % (np.mean(fracs), sum(f > 0.5 for f in fracs), sum(f > 0.1 for f in fracs)), "%.1fs" % (time.time() - t0))
