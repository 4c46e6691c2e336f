"""One long-lived bgr_aligner, a random sequence of calls: every entry point of the batch C-ABI (bgr_align_batch, its packed form, the ticketed
asynchronous form, the device-resident form + bgr_aligner_fetch, the text form) in every mode (greedy, exhaustive, anchors), batches of 1 ..
600 000 reads of 20 .. 600 bases with and without N, key table staged or probed, one call after the other on the SAME aligner -- whose device
buffers, twins, stages and lists are grown, recycled and left dirty by the calls before.  Every row and the counters against the oracle (the
text form: against the batch form's rows formatted as the reference writes them).  Run on a GPU box:
python tools/fuzz_reuse.py [seed] [calls].  (Test infrastructure; written after tools/fuzz_text_route.py found a kernel that read recycled
buffers.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bgreat_amd as B, oracle_py
B.set_options_from_string(os.environ.get("BGR_FUZZ_OPTIONS"))   # (library options of this campaign: the library itself reads no environment)
from synth import Synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NCALL = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)
k = int(rng.choice([15, 21, 31, 31, 32]))
s = Synth(int(rng.integers(150_000, 2_000_000)), int(rng.integers(k + 5, 4 * k)), int(rng.integers(2, 4)), k, 5000 + seed)
seqs, offs = s.unitigs()
g = B.Graph.build(k, seqs, offs, float(rng.choice([0.0, 0.0, 1.8])), anchors=True)
o = oracle_py.Oracle(k, seqs, offs, anchors=True)
al = B.Aligner(g, 0)
print("graph", g.info()["n_unitigs"], "unitigs, k", k, flush=True)


def records(headers, reads, roffs, paths, poffs):
    """the two streams as the reference writes them (alignerGreedy.cpp:406-427)"""
    p, n = [], []
    for i, h in enumerate(headers):
        a, b = int(poffs[i]), int(poffs[i + 1])
        if b > a:
            p.append(h + b"\n" + b"".join(b"%d." % v for v in paths[a:b]) + b"\n")
        else:
            n.append(h + b"\n" + reads[int(roffs[i]): int(roffs[i + 1])].tobytes() + b"\n")
    return b"".join(p), b"".join(n)


bad = 0
t0 = time.time()
for it in range(NCALL):
    mode_name = str(rng.choice(["greedy", "greedy", "exhaustive", "anchors"]))
    gm, om = {"greedy": (B.MODE_GREEDY, 0), "exhaustive": (B.MODE_EXHAUSTIVE, 1), "anchors": (B.MODE_ANCHORS, 2)}[mode_name]
    form = str(rng.choice(["batch", "batch", "packed", "ticket", "device", "text"]))
    L = int(rng.choice([max(20, k + 1), 60, 100, 150, 150, 250, 600]))
    n = int(rng.choice([1, 7, 63, 1000, 20_000, 131_073, 300_000, 600_000]))
    if L >= 250 or mode_name != "greedy":
        n = min(n, 131_073)
    if mode_name == "exhaustive" and L >= 250:
        n = min(n, 20_000)
    m = int(rng.integers(0, 5 if mode_name != "exhaustive" else 3)); effort = int(rng.choice([0, 1, 2, 2, 4]))
    reads, roffs = s.reads(int(rng.integers(0, 1 << 30)), n, L, m + 1, int(rng.integers(1, 1 << 30)))
    nfrac = float(rng.choice([0, 0, 0.001]))
    if nfrac and form != "text":
        reads = reads.copy(); reads[rng.choice(len(reads), size=max(1, int(len(reads) * nfrac)), replace=False)] = ord("N")
    al.configure(0, 0, int(rng.choice([0, 0, 1, 2])))
    t1 = time.time()
    p2, po2, st2 = o.align(reads, roffs, m=m, effort=effort, mode=om)
    ok = True
    if form == "batch":
        p1, po1, st1 = al.align(reads, roffs, m=m, effort=effort, mode=gm)
    elif form == "packed":
        p1, po1, st1 = al.align_packed(B.pack_reads(reads, roffs), m=m, effort=effort, mode=gm)
    elif form == "ticket":
        if n > 2_000_000:
            continue
        t = al.align_begin(reads, roffs, m=m, effort=effort, mode=gm)
        while not al.align_test(t):
            time.sleep(0.0005)
        p1, po1, st1 = al.align_wait(t)
    elif form == "device":
        dr, do = B.DeviceBuffer(0, reads), B.DeviceBuffer(0, roffs)
        al.align_device(dr.data_ptr(), do.data_ptr(), n, int(roffs[n]), L, m=m, effort=effort, mode=gm)
        p1, po1, st1 = al.fetch(n, int(roffs[n]) + 8 * n + 8)
        dr.free(); do.free()
    else:   # the text form: FASTA bytes in, record bytes out -- against the oracle's rows in the reference's record format (reads of at most k bases are dropped)
        headers = [b">r%d" % i for i in range(n)]
        rr = reads.reshape(n, L)
        text = b"".join(h + b"\n" + rr[i].tobytes() + b"\n" for i, h in enumerate(headers))
        pt, nt, info = al.align_fasta_text(text, m=m, effort=effort, mode=gm, staged=bool(rng.integers(0, 2)))
        if info["irregular"] and len(text) < 34 * n:
            ok = pt == b"" and nt == b""   # (records of fewer than 32 bytes on average: more record starts per 32 KB than the parse launch lists -- the host parser's piece)
        elif L > k:
            ep, en = records(headers, reads, roffs, p2, po2)
            ok = (not info["irregular"]) and pt == ep and nt == en
        else:
            ok = (not info["irregular"]) and pt == b"" and nt == b""
        p1, po1, st1 = p2, po2, st2
    if form != "text":
        ok = np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)
    print("%s call %d %s" % ("ok      " if ok else "MISMATCH", it, dict(mode=mode_name, form=form, n=n, L=L, m=m, effort=effort, nfrac=nfrac)), "%.1fs" % (time.time() - t1), flush=True)
    bad += 0 if ok else 1
print("calls %d bad %d  %.1fs" % (NCALL, bad, time.time() - t0))
sys.exit(1 if bad else 0)
