"""Randomised campaign over the text route (bgr_align_fasta_text: the device finds the records of a FASTA / FASTQ piece, applies getReads'
accept rules, maps and formats) against the host route (the exact iostream state machine of fastx.cpp + host formatter) and -- where the
compiled reference finishes in seconds and its mode writes files -- against oracle/_ref/bgreat at -t 1: random graphs, record counts
of 5 000 .. 600 000, fixed and mixed read lengths (down to 8 bases: more records than the device's table holds), header styles, batch / chunk sizes, thread counts, modes (greedy, -c, -G, -b with
and without --write-exhaustive, --no-overlap), and irregular records injected at a random rate (lower case, N, CR, empty sequence lines, multi-line sequences,
blank lines, '>' inside headers and at the start of sequence lines, reads of at most k bases, a last record without its newline,
text in front of the first header; FASTQ: '@' / '+' look-alikes in quality lines, truncated tails).  Bytes of both output files and
the counters must agree.  Run on a GPU box: python tools/fuzz_text_route.py [seed] [configs].  (Test infrastructure.)"""
import os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bgreat_amd as B
B.set_options_from_string(os.environ.get("BGR_FUZZ_OPTIONS"))
from synth import Synth
from irregular_files import make_file

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NCFG = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed)
REF = os.path.join(ROOT, "oracle", "_ref", "bgreat")


def same(a, b):
    ea, eb = os.path.exists(a), os.path.exists(b)
    if not ea or not eb:
        return ea == eb     # (-b without --write-exhaustive writes nothing)
    return os.path.getsize(a) == os.path.getsize(b) and open(a, "rb").read() == open(b, "rb").read()


bad = 0
t0 = time.time()
for it in range(NCFG):
    k = int(rng.choice([15, 21, 25, 31, 31, 32]))
    s = Synth(int(rng.integers(60_000, 1_500_000)), int(rng.integers(k + 2, 5 * k)), int(rng.integers(2, 5)), k, 900 + 31 * seed + it)
    mode = str(rng.choice(["greedy", "greedy", "greedy", "correct", "anchors", "exhaustive"]))
    fastq = bool(rng.random() < 0.35)
    n = int(rng.choice([5_000, 20_000, 29_999, 60_000, 250_000, 600_000]))
    irr = float(rng.choice([0.0, 0.0, 1e-4, 2e-3, 0.05]))
    mixed = bool(rng.random() < 0.4)
    hdr = int(rng.integers(0, 3))
    m = int(rng.integers(0, 5)); effort = int(rng.choice([1, 2, 2, 3]))
    batch = int(rng.choice([0, 0, 5_000, 33_333, 100_000])); chunk = int(rng.choice([0, 0, 1 << 16, 1 << 20])); threads = int(rng.choice([1, 4, 8]))
    nfiles = int(rng.choice([1, 1, 2]))
    wex = bool(rng.random() < 0.7)            # -b: with --write-exhaustive, or counts only (the reference's behaviour: record info from the device, progress blocks)
    novl = bool(mode == "greedy" and rng.random() < 0.15)   # --no-overlap FILE (host route by design: both runs take it)
    d = tempfile.mkdtemp(prefix="bgr_fzt_")
    cfg = dict(k=k, mode=mode, fastq=fastq, n=n, irr=irr, mixed=mixed, hdr=hdr, m=m, effort=effort, batch=batch, chunk=chunk, threads=threads, files=nfiles, wex=wex, novl=novl)
    try:
        files = []
        for j in range(nfiles):
            f = os.path.join(d, "r%d.%s" % (j, "fq" if fastq else "fa"))
            L = make_file(rng, f, s, k, n if j == 0 else max(1, n // 3), fastq, irr, mixed, hdr)
            files.append(f)
        seqs, offs = s.unitigs()
        g = B.Graph.build(k, seqs, offs, 0.0, anchors=(mode == "anchors"))
        kw = dict(m=m, effort=effort, threads=threads, batch_reads=batch, chunk_bytes=chunk, fastq=fastq, correction=(mode == "correct"),
                  mode={"greedy": B.MODE_GREEDY, "correct": B.MODE_GREEDY, "anchors": B.MODE_ANCHORS, "exhaustive": B.MODE_EXHAUSTIVE}[mode],
                  write_exhaustive=(mode == "exhaustive" and wex), no_overlap_file=(os.path.join(d, "o%d") if novl else None))
        res = {}
        for route in (0, 1):
            try:
                kwr = dict(kw)
                if kwr.get("no_overlap_file"):
                    kwr["no_overlap_file"] = kwr["no_overlap_file"] % route
                c, _ = B.align_all(g, ",".join(files), os.path.join(d, "p%d" % route), os.path.join(d, "n%d" % route), route=route, **kwr)
                res[route] = ("ok", c)
            except B.BgrError as ex:    # (-c: the reference's "bug compaction" exit is an error of the run on both routes)
                res[route] = ("err", str(ex)[:80])
        ok = res[0][0] == res[1][0] and (res[0][0] == "err" or (res[0][1] == res[1][1] and same(os.path.join(d, "p0"), os.path.join(d, "p1")) and same(os.path.join(d, "n0"), os.path.join(d, "n1"))
                                                                   and (not novl or same(os.path.join(d, "o0"), os.path.join(d, "o1")))))
        # a split run (one pipeline per lane over contiguous shares of the input, N output pairs; option test.lanes_on_one_device: every lane on this
        # box's one device): the pairs concatenated in lane order must be the single pipeline's bytes
        split = "-"
        if ok and res[0][0] == "ok" and not fastq and mode in ("greedy", "anchors") and not novl and rng.random() < 0.5:
            lanes = int(rng.choice([2, 3, 5]))
            B.set_option("test.lanes_on_one_device", 1)
            try:
                kws = dict(kw); kws["threads"] = max(threads, lanes)
                cs, _ = B.align_all(g, ",".join(files), os.path.join(d, "ps"), os.path.join(d, "ns"), route=int(rng.integers(0, 2)), n_gpus=lanes, split_output=True, **kws)
                cat = lambda stem: b"".join(open(os.path.join(d, "%s.%d" % (stem, i)), "rb").read() for i in range(lanes))
                sok = cs == res[0][1] and cat("ps") == open(os.path.join(d, "p0"), "rb").read() and cat("ns") == open(os.path.join(d, "n0"), "rb").read()
                split = "%d lanes %s" % (lanes, "same" if sok else "DIFFERENT")
                ok = ok and sok
            except B.BgrError as ex:
                split = "error %s" % str(ex)[:80]
                ok = False
            finally:
                B.set_option("test.lanes_on_one_device", 0)
        ref = "-"
        if ok and res[0][0] == "ok" and os.path.exists(REF) and mode != "exhaustive" and not novl and n <= 60_000 and not (fastq and (mixed or irr or L < k)):
            # the compiled reference at -t 1 (FASTQ reads shorter than k-1 make it throw: regular FASTQ only)
            s.write_unitigs(os.path.join(d, "u.fa"))
            cmd = [REF, "-r", ",".join(files), "-k", str(k), "-g", os.path.join(d, "u.fa"), "-m", str(m), "-e", str(effort), "-t", "1", "-f", os.path.join(d, "pr"), "-a", os.path.join(d, "nr")]
            cmd += ["-q"] if fastq else []
            cmd += ["-c"] if mode == "correct" else []
            cmd += ["-G"] if mode == "anchors" else []
            try:
                p = subprocess.run(cmd, cwd=d, capture_output=True, timeout=120)
                if p.returncode == 0:
                    rs = same(os.path.join(d, "p0"), os.path.join(d, "pr")) and same(os.path.join(d, "n0"), os.path.join(d, "nr"))
                    ref = "same" if rs else "DIFFERENT"
                    ok = ok and rs
                else:
                    ref = "ref rc %d" % p.returncode
            except subprocess.TimeoutExpired:
                ref = "ref timeout"
        print("%s %s routes %s/%s reads %s reference %s split %s" % ("ok      " if ok else "MISMATCH", cfg, res[0][0], res[1][0], res[0][1]["reads"] if res[0][0] == "ok" else res[0][1], ref, split), flush=True)
        if not ok:
            bad += 1
            keep = os.path.join(ROOT, "gpurun_out", "fuzztext_bad_%d_%d" % (seed, it))
            os.makedirs(keep, exist_ok=True)
            for f in files[:1]:
                if os.path.getsize(f) < (8 << 20):
                    shutil.copy(f, keep)
    finally:
        shutil.rmtree(d, ignore_errors=True)
print("configs %d bad %d  %.1fs" % (NCFG, bad, time.time() - t0))
sys.exit(1 if bad else 0)
