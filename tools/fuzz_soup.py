"""Randomised campaign on DEGENERATE graphs (the synthetic workloads of the other fuzzers are genomes with variant sites: tidy bubbles): a
soup of unitigs with everything a compacted de Bruijn graph should not have and the reference nevertheless takes -- more than four unitigs
on one (k-1)-overlap (the slot-4 overwrite of aligner.cpp:466-533), palindromic overlaps, hairpins and self-loops (a unitig whose end
overlaps its own start, on either strand), duplicated unitigs, unitigs stored reverse complemented, unitigs of length k-1+1 .. , homopolymers,
N and lower case inside unitigs, chains cut out of one random sequence (long walks) next to unrelated random strings; k from 4 (every overlap
shared by dozens of unitigs) to 32.  Reads: walks through the chains with substitutions, random strings, N.

  python tools/fuzz_soup.py cpu [seed] [configs]   oracle CLI (oracle/bgreat_oracle) against the compiled reference (oracle/_ref/bgreat, -t 1): bytes
                                                   of paths / notAligned.fa and the counters -- runs anywhere /root/reference was compiled (no GPU)
                                                   exhaustive soups also through the oracle's remembered-calls form (ORACLE_EXH_MEMO): same bytes as the literal one
  python tools/fuzz_soup.py gpu [seed] [configs]   the C-ABI on a GPU box against the oracle, row for row, greedy / exhaustive / anchors; exhaustive mode WITH the
                                                   ingredients that duplicate k-mers (the last pass is polynomial since round 5), checked by the oracle's
                                                   remembered-calls form (the literal recursion is exponential there -- `cpu` pins the one to the other)
  python tools/fuzz_soup.py gpuref [seed] [configs] exhaustive soups through bin/bgreat -b --write-exhaustive on a GPU box against the compiled reference
                                                   (oracle/_ref/bgreat_exh, -t 1, 120 s): bytes and counters wherever the reference finishes
(Test infrastructure.)"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

what = sys.argv[1] if len(sys.argv) > 1 else "cpu"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
NCFG = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(seed)
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def rs(n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))


def rc(s):
    return "".join(COMP.get(c, "A") for c in reversed(s))


def soup(k, no_dup_kmers=False):
    """no_dup_kmers: leave out what duplicates k-mers between unitigs on a grand scale (a homopolymer on both strands, copied unitigs).  On such a
    graph -- no compacted de Bruijn graph has it -- the reference's exhaustive recursion is exponential: every base of a poly-A read is a
    choice between slots that name the same move, 150 s for ONE 79-base read in the compiled reference and in the oracle's literal form.
    (Round 4 left these out of the device's exhaustive soups: its last pass was that recursion.  Since round 5 it remembers its calls and
    every campaign runs with them.)"""
    K1 = k - 1
    us = []
    chains = []
    for _ in range(int(rng.integers(1, 6))):          # chains: consecutive pieces of one sequence, overlapping by k-1
        G = rs(int(rng.integers(3 * k, 40 * k)))
        chains.append(G)
        p = 0
        while p + k <= len(G):
            ln = int(rng.integers(k, k + 1 + int(rng.choice([2, k, 4 * k]))))
            u = G[p: p + ln]
            if len(u) < k:
                break
            us.append(rc(u) if rng.random() < 0.4 else u)
            p += len(u) - K1
    for _ in range(int(rng.integers(0, 60))):          # unrelated strings
        us.append(rs(int(rng.integers(k, 3 * k + 2))))
    for _ in range(int(rng.integers(0, 30))):          # fans: unitigs that share an end overlap with an existing one (more than four per overlap now and then)
        u = us[int(rng.integers(0, len(us)))]
        ov = u[-K1:] if rng.random() < 0.5 else u[:K1]
        for _ in range(int(rng.integers(1, 8))):
            v = ov + rs(int(rng.integers(1, 2 * k))) if rng.random() < 0.5 else rs(int(rng.integers(1, 2 * k))) + ov
            us.append(rc(v) if rng.random() < 0.3 else v)
    if K1 % 2 == 0 and rng.random() < 0.7:              # palindromic overlaps
        for _ in range(3):
            h = rs(K1 // 2)
            pal = h + rc(h)
            us.append(pal + rs(int(rng.integers(1, k))))
            us.append(rs(int(rng.integers(1, k))) + pal)
    for _ in range(int(rng.integers(0, 4))):           # self-loops and hairpins
        ov = rs(K1)
        us.append(ov + rs(int(rng.integers(1, k))) + ov)            # the end overlaps the start
        us.append(ov + rs(int(rng.integers(1, k))) + rc(ov))        # the end overlaps the start of its own reverse complement
    us += ["A" * int(rng.integers(k, 2 * k + 2)), "AC" * k] + ([] if no_dup_kmers else ["T" * k])
    for _ in range(0 if no_dup_kmers else int(rng.integers(0, 5))):           # duplicates, on either strand
        u = us[int(rng.integers(0, len(us)))]
        us.append(u if rng.random() < 0.5 else rc(u))
    if rng.random() < 0.3:                              # characters outside ACGT
        for _ in range(int(rng.integers(1, 6))):
            i = int(rng.integers(0, len(us)))
            u = list(us[i]); u[int(rng.integers(0, len(u)))] = str(rng.choice(list("NNacgtRY"))); us[i] = "".join(u)
    order = rng.permutation(len(us))
    us = [us[i] for i in order]
    reads = []
    nr = int(rng.integers(200, 3000))
    for _ in range(nr):
        t = rng.random()
        if t < 0.6 and chains:
            G = chains[int(rng.integers(0, len(chains)))]
            L = int(rng.integers(k + 1, min(len(G), 8 * k) + 1))
            p = int(rng.integers(0, len(G) - L + 1))
            r = list(G[p: p + L])
            for _ in range(int(rng.integers(0, 4))):
                r[int(rng.integers(0, L))] = "ACGT"[int(rng.integers(0, 4))]
            r = "".join(r)
            if rng.random() < 0.5:
                r = rc(r)
        elif t < 0.8:
            a, b = us[int(rng.integers(0, len(us)))], us[int(rng.integers(0, len(us)))]
            r = (a + b[K1:])[: int(rng.integers(k + 1, 6 * k))]
            r = "".join(c if c in "ACGT" else "A" for c in r.upper())
            if len(r) <= k:
                r = rs(k + 3)
        else:
            r = rs(int(rng.integers(k + 1, 5 * k)))
        if rng.random() < 0.05:
            r = list(r); r[int(rng.integers(0, len(r)))] = "N"; r = "".join(r)
        reads.append(r)
    return us, reads


def write_inputs(d, us, reads):
    with open(os.path.join(d, "u.fa"), "w") as f:
        f.write("".join(">%d\n%s\n" % (i + 1, u) for i, u in enumerate(us)))
    with open(os.path.join(d, "r.fa"), "w") as f:
        f.write("".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(reads)))


bad = 0
t0 = time.time()
if what == "cpu":
    from util import parse_counters, run_cli
    REF, REFX, ORC = (os.path.join(ROOT, "oracle", "_ref", "bgreat"), os.path.join(ROOT, "oracle", "_ref", "bgreat_exh"), os.path.join(ROOT, "oracle", "bgreat_oracle"))
    for it in range(NCFG):
        k = int(rng.choice([4, 5, 6, 7, 8, 10, 12, 15, 21, 31, 32]))
        us, reads = soup(k)
        mode = str(rng.choice(os.environ["FUZZ_MODES"].split(",") if os.environ.get("FUZZ_MODES") else ["greedy", "greedy", "correct", "anchors", "exhaustive", "exhaustive_i"]))
        m = int(rng.integers(0, 6)); e = int(rng.choice([0, 1, 2, 2, 3, 8]))
        if mode.startswith("exhaustive"):
            reads = reads[:400]   # (the literal recursion is exponential on the duplicated k-mers: keep the checkers in seconds)
        with tempfile.TemporaryDirectory() as d:
            write_inputs(d, us, reads)
            args = ["-r", os.path.join(d, "r.fa"), "-k", str(k), "-g", os.path.join(d, "u.fa"), "-m", str(m), "-e", str(e), "-t", "1"]
            args += {"greedy": [], "correct": ["-c"], "anchors": ["-G"], "exhaustive": ["-b"], "exhaustive_i": ["-b", "-i"]}[mode]
            exh = mode.startswith("exhaustive")
            try:
                o1, p1, n1 = run_cli(REFX if exh else REF, args, timeout=120)
            except Exception as ex:   # the reference itself gives up on some soups (exit, exception, endless search): nothing to compare with
                print("skipped  %s reference: %s" % (dict(k=k, mode=mode, m=m, e=e, unitigs=len(us), reads=len(reads)), str(ex)[:90].replace("\n", " ")), flush=True)
                continue
            try:
                o2, p2, n2 = run_cli(ORC, args, env={"ORACLE_EXH_WRITES": "1"} if exh else None, timeout=300)
                ok = parse_counters(o1) == parse_counters(o2) and p1 == p2 and n1 == n2
                if exh:   # the remembered-calls form of the oracle: what checks the device where the literal recursion does not come back
                    o3, p3, n3 = run_cli(ORC, args, env={"ORACLE_EXH_WRITES": "1", "ORACLE_EXH_MEMO": "1"}, timeout=300)
                    ok = ok and parse_counters(o1) == parse_counters(o3) and p1 == p3 and n1 == n3
            except Exception as ex:
                ok = False
                print("oracle failed:", str(ex)[:200])
            print("%s %s aligned %s" % ("ok      " if ok else "MISMATCH", dict(k=k, mode=mode, m=m, e=e, unitigs=len(us), reads=len(reads)), parse_counters(o1).get("aligned")), flush=True)
            if not ok:
                bad += 1
                keep = os.path.join(ROOT, "gpurun_out", "soup_bad_%d_%d" % (seed, it))
                os.makedirs(keep, exist_ok=True)
                write_inputs(keep, us, reads)
                open(os.path.join(keep, "args.txt"), "w").write(" ".join(args))
elif what == "gpuref":
    import bgreat_amd as B
    from util import parse_counters, run_cli
    REFX = os.path.join(ROOT, "oracle", "_ref", "bgreat_exh")
    skipped = 0
    for it in range(NCFG):
        k = int(rng.choice([4, 5, 6, 7, 8, 10, 12, 15, 21, 31, 32]))
        us, reads = soup(k)
        reads = reads[:400]
        m = int(rng.integers(0, 6)); mode = str(rng.choice(["exhaustive", "exhaustive_i"]))
        with tempfile.TemporaryDirectory() as d:
            write_inputs(d, us, reads)
            args = ["-r", os.path.join(d, "r.fa"), "-k", str(k), "-g", os.path.join(d, "u.fa"), "-m", str(m), "-t", "1", "-b"] + (["-i"] if mode == "exhaustive_i" else [])
            tg = time.time()
            o2, p2, n2 = run_cli(B.CLI_PATH, args + ["--write-exhaustive"], timeout=600)
            tg = time.time() - tg
            try:
                tr = time.time()
                o1, p1, n1 = run_cli(REFX, args, timeout=120)
                tr = time.time() - tr
            except Exception as ex:
                skipped += 1
                print("skipped  %s reference: %s (device %.1fs)" % (dict(k=k, mode=mode, m=m, unitigs=len(us), reads=len(reads)), str(ex)[:60].replace("\n", " "), tg), flush=True)
                continue
            ok = parse_counters(o1) == parse_counters(o2) and p1 == p2 and n1 == n2
            print("%s %s aligned %s device %.1fs reference %.1fs" % ("ok      " if ok else "MISMATCH", dict(k=k, mode=mode, m=m, unitigs=len(us), reads=len(reads)), parse_counters(o1).get("aligned"), tg, tr), flush=True)
            if not ok:
                bad += 1
                keep = os.path.join(ROOT, "gpurun_out", "soup_bad_%d_%d" % (seed, it))
                os.makedirs(keep, exist_ok=True)
                write_inputs(keep, us, reads)
                open(os.path.join(keep, "args.txt"), "w").write(" ".join(args))
    print("reference did not finish within 120 s:", skipped)
else:
    import bgreat_amd as B, oracle_py
    for it in range(NCFG):
        k = int(rng.choice([4, 5, 6, 7, 8, 10, 12, 15, 21, 31, 32]))
        mode = str(rng.choice(os.environ["FUZZ_MODES"].split(",") if os.environ.get("FUZZ_MODES") else ["greedy", "greedy", "anchors", "exhaustive", "exhaustive_i"]))
        us, reads = soup(k)
        m = int(rng.integers(0, 6)); e = int(rng.choice([0, 1, 2, 2, 3, 8]))
        seqs = np.frombuffer("".join(us).encode(), dtype=np.uint8)
        offs = np.concatenate([[0], np.cumsum([len(u) for u in us])]).astype(np.uint64)
        rb = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
        roffs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
        anc = mode == "anchors"
        g = B.Graph.build(k, seqs, offs, float(rng.choice([0.0, 1.07, 1.8])), anchors=anc); al = B.Aligner(g, 0); o = oracle_py.Oracle(k, seqs, offs, anchors=anc)
        al.configure(0, 0, int(rng.choice([0, 1, 2])))
        if rng.random() < 0.3:
            al.set_knob(B.KNOB_EXH_FRAME_CAP, int(rng.choice([3, 6, 16])))
        gm, om = {"greedy": (B.MODE_GREEDY, 0), "anchors": (B.MODE_ANCHORS, 2), "exhaustive": (B.MODE_EXHAUSTIVE, 3), "exhaustive_i": (B.MODE_EXHAUSTIVE, 3)}[mode]   # (3: remembered calls)
        if gm == B.MODE_EXHAUSTIVE and rng.random() < 0.3:
            al.set_knob(B.KNOB_EXH_MEMO_CAP, int(rng.choice([8, 64])))   # (the last pass's table: reads handed back and run again)
        partial = mode == "exhaustive_i"
        if gm == B.MODE_EXHAUSTIVE and len(reads) > 1000:
            roffs = roffs[:1001]; rb = rb[: int(roffs[1000])]
        tg = time.time()
        p1, po1, st1 = al.align(rb, roffs, m=m, effort=e, mode=gm, partial=partial)
        tg = time.time() - tg
        p2, po2, st2 = o.align(rb, roffs, m=m, effort=e, mode=om, partial=partial)
        ok = np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)
        print("%s %s aligned %.2f gpu %.2fs" % ("ok      " if ok else "MISMATCH", dict(k=k, mode=mode, m=m, e=e, unitigs=len(us), reads=len(roffs) - 1), float(((st1 & 3) == 2).mean()), tg), flush=True)
        bad += 0 if ok else 1
        al.close()
print("configs %d bad %d  %.1fs" % (NCFG, bad, time.time() - t0))
sys.exit(1 if bad else 0)
