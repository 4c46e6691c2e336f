"""Randomised parity campaign at batch sizes that exercise the run-time work distribution of the several-reads-per-wave kernels
(claim_task: workgroup stock, guided refills from the launch's counter; the greedy kernel's per-wave follow-up ring): tools/fuzz_parity.py
maps 2 500 reads per configuration -- fewer tasks than the launch has waves -- so here a configuration is 40 000 .. 400 000 reads (several
tasks per wave, several refills per workgroup), random k / site spacing / alleles / read length / budget / effort / N rate / table regime,
every row and the counters against the oracle.  Run on a GPU box: python tools/fuzz_big_batches.py [seed] [greedy|exhaustive|anchors] [configs].
(Test infrastructure.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bgreat_amd as B, oracle_py
B.set_options_from_string(os.environ.get("BGR_FUZZ_OPTIONS"))   # (library options of this campaign: the library itself reads no environment)
from synth import Synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
MODE = sys.argv[2] if len(sys.argv) > 2 else "greedy"
NCFG = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(seed)
bad = 0
t0 = time.time()
for it in range(NCFG):
    k = int(rng.choice([12, 15, 21, 25, 31, 31, 32]))
    d = int(rng.integers(k + 2, 5 * k))
    alleles = int(rng.integers(2, 5))
    L = int(rng.choice([k + 3, 60, 100, 150, 150, 250, 320, 460]))
    if L <= k:
        L = k + 3
    m = int(rng.integers(0, 6))
    nfrac = float(rng.choice([0, 0, 0.0005, 0.005]))
    effort = int(rng.choice([0, 1, 2, 2, 3, 8]))
    n = int(rng.choice([40_000, 70_001, 131_072, 200_003, 400_000, 600_001, 1_100_000]))   # (from 512 k reads on bgr_align_batch runs in pieces on four streams)
    if n > 400_000 and (L > 150 or MODE != "greedy"):
        n = 400_000
    if MODE == "exhaustive":
        n = min(n, 131_072); m = min(m, 3)       # (the oracle's recursion on branchy graphs: keep a configuration in seconds)
    if L >= 250:
        n = min(n, 200_003)
    s = Synth(int(rng.integers(100_000, 3_000_000)), d, alleles, k, 3000 + 17 * seed + it)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, n, L, m + 1, 7000 + it)
    if nfrac:
        reads = reads.copy(); idx = rng.choice(len(reads), size=max(1, int(len(reads) * nfrac)), replace=False); reads[idx] = ord("N")
    ragged = bool(rng.random() < 0.3)
    if ragged:   # a batch of mixed lengths: every read cut to a random length k-1 .. L (no anchor position, one, a few ... and some of full length)
        lens = rng.integers(k - 1, L + 1, size=n).astype(np.uint64)   # (shorter than k-1: the reference's substr throws, and so does the oracle)
        lens[rng.random(n) < 0.05] = L
        keep = (np.arange(L, dtype=np.uint64)[None, :] < lens[:, None]).reshape(-1)
        reads = reads[keep]
        roffs = np.concatenate([np.zeros(1, np.uint64), np.cumsum(lens, dtype=np.uint64)])
    anc = MODE == "anchors"
    gamma = float(rng.choice([0.0, 0.0, 1.07, 1.8]))
    g = B.Graph.build(k, seqs, offs, gamma, anchors=anc); al = B.Aligner(g, 0); o = oracle_py.Oracle(k, seqs, offs, anchors=anc)
    regime = int(rng.choice([0, 0, 1, 2]))
    if regime:
        al.configure(0, 0, regime)   # 1: key table probed in L2, 2: LDS staging forced
    t1 = time.time()
    if MODE == "exhaustive":
        p1, po1, st1 = al.align(reads, roffs, m=m, mode=B.MODE_EXHAUSTIVE)
        p2, po2, st2 = o.align(reads, roffs, m=m, mode=1)
    else:
        gm, om = (B.MODE_ANCHORS, 2) if anc else (B.MODE_GREEDY, 0)
        p1, po1, st1 = al.align(reads, roffs, m=m, effort=effort, mode=gm)
        p2, po2, st2 = o.align(reads, roffs, m=m, effort=effort, mode=om)
    c1, c2 = al.counters(), o.counters()
    if MODE != "exhaustive":
        c1["overlaps"] = c2["overlaps"] = 0
    ok = np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2) and c1 == c2
    cfg = dict(k=k, d=d, alleles=alleles, L=L, ragged=ragged, m=m, effort=effort, nfrac=nfrac, n=n, gamma=gamma, regime=regime, unitigs=len(offs) - 1)
    print("%s %s aligned %.2f passes %s  %.1fs" % ("ok      " if ok else "MISMATCH", cfg, float(((st1 & 3) == 2).mean()), al.pass_counts(), time.time() - t1), flush=True)
    bad += 0 if ok else 1
    al.close()
print("configs %d bad %d  reads per configuration 40 000 .. 400 000  %.1fs" % (NCFG, bad, time.time() - t0))
sys.exit(1 if bad else 0)
