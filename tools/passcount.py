"""How many reads each pass of a greedy mapping launch hands on (bgr_aligner_pass_counts), effort 2 and 1, on the default
E. coli-scale workload: python tools/passcount.py (GPU box).  Diagnostic."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, bgreat_amd as B
from tools.synth import Synth
s = Synth(4600000, 140, 2, 31, 20261003)
seqs, offs = s.unitigs()
g = B.Graph.build(31, seqs, offs); al = B.Aligner(g, 0)
reads, roffs = s.reads(0, 1000000, 150, 2, 77)
for e in (2, 1):
    p, po, st = al.align(reads, roffs, m=2, effort=e)
    print("effort", e, "listed", al.pass_counts(), "status hist", np.bincount(st & 3, minlength=3), "rc", int(((st & 4) != 0).sum()))
