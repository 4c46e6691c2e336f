#!/usr/bin/env python3
"""Generate tests/golden/: small inputs + the outputs of the COMPILED REFERENCE on them.

Run in the build container only (needs oracle/_ref/bgreat and oracle/_ref/bgreat_exh, which
oracle/Makefile compiles from /root/reference).  The fixtures are data: inputs and the reference's
output bytes (or their sha256 where the bytes would bloat the repo), plus the stdout counters.
`bgreat_exh` is the reference with alignerExhaustive.cpp:285/292/299 redirected to the open FILE*s --
"what the reference computes but does not write" (SURVEY.md fact 0.5); the unmodified binary supplies
the exhaustive counters.

    python tests/make_golden.py        # rewrites tests/golden/
"""
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.synth import Synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(ROOT, "oracle", "_ref", "bgreat")
REF_EXH = os.path.join(ROOT, "oracle", "_ref", "bgreat_exh")
KEEP_BYTES_LIMIT = 6000  # expected outputs up to this size are stored verbatim, larger ones as sha256 only


def sha(b):
    return hashlib.sha256(b).hexdigest()


def parse_counters(out):
    c = {}
    for line in out.splitlines():
        for key, name in (("Reads : ", "reads"), ("No overlap : ", "no_overlap"), ("Got overlap : ", "got_overlap"),
                          ("Overlap and aligned : ", "aligned"), ("Overlap but not aligned : ", "not_aligned")):
            if line.startswith(key):
                c[name] = int(line[len(key):].split()[0])
    return c


def run_ref(binary, args):
    d = tempfile.mkdtemp()
    try:
        out = subprocess.run([binary] + args, cwd=d, capture_output=True, text=True, check=True).stdout
        paths = open(os.path.join(d, "paths"), "rb").read()
        na = open(os.path.join(d, "notAligned.fa"), "rb").read()
        return out, paths, na
    finally:
        shutil.rmtree(d)


def rs(rng, n, al="ACGT"):
    return "".join(rng.choice(al) for _ in range(n))


def write(name, text, mode="w"):
    with open(os.path.join(GOLD, name), mode) as f:
        f.write(text)


def make_inputs():
    rng = random.Random(20261003)
    # (i) bundled toy (data files of the reference: 2 reads, 6 unitigs, k=4)
    write("toy_reads.fa", ">nomiss\nAAATCGATTCA\n>nomiss\nAAATCGTTTCA\n")
    write("toy_unitig.fa", ">1\nAAATC\n>2\nATCCT\n>3\nATCG\n>4\nTCGATTC\n>5\nTCGTTTC\n>6\nTGAA\n")
    # (ii) synthetic k=31 graphs: short unitigs (~75 bp) and long unitigs (longer than the reads)
    s = Synth(36000, 75, 2, 31, 101)
    s.write_unitigs(os.path.join(GOLD, "syn_unitig.fa"))
    s.write_reads(os.path.join(GOLD, "syn_r100.fa"), 0, 300, 100, 5, 201)
    s.write_reads(os.path.join(GOLD, "syn_r150.fa"), 0, 400, 150, 3, 202)
    s.write_reads(os.path.join(GOLD, "syn_r250.fa"), 0, 150, 250, 5, 203)
    s2 = Synth(60000, 420, 3, 31, 102)
    s2.write_unitigs(os.path.join(GOLD, "long_unitig.fa"))
    s2.write_reads(os.path.join(GOLD, "long_r150.fa"), 0, 300, 150, 3, 204)
    s2.write_reads(os.path.join(GOLD, "long_r150.fq"), 1000, 120, 150, 2, 205, fastq=True)
    # (iii) edge-case FASTA on the syn graph: N, lowercase, short, multi-line, blank line, CR, header with
    # spaces, no-anchor read, size k / k+1, last record without trailing newline.
    base = open(os.path.join(GOLD, "syn_r150.fa")).read().split("\n")
    good = [base[2 * i + 1] for i in range(40)]
    recs = []
    recs.append(">plain\n" + good[0] + "\n")
    r = list(good[1]); r[5] = "N"; recs.append(">withN first window\n" + "".join(r) + "\n")
    r = list(good[2]); r[70] = "N"; recs.append(">withN rolled\n" + "".join(r) + "\n")
    r = list(good[3]); r[40] = "N"; r[90] = "N"; r[140] = "N"; recs.append(">withN x3\n" + "".join(r) + "\n")
    recs.append(">lower\n" + good[4].lower() + "\n")
    recs.append(">mixedcase\n" + good[5][:50] + "a" + good[5][51:] + "\n")
    recs.append(">multi line\n" + good[6][:60] + "\n" + good[6][60:100] + "\n" + good[6][100:] + "\n")
    recs.append(">blank line inside\n" + good[7][:80] + "\n\n" + good[7][80:] + "\n")
    recs.append(">crlf\r\n" + good[8] + "\r\n")
    recs.append(">len k\n" + good[9][:31] + "\n")
    recs.append(">len k+1\n" + good[9][:32] + "\n")
    recs.append(">len 2\nAC\n")
    recs.append(">len 3\nACG\n")
    recs.append(">random no anchor\n" + rs(rng, 150) + "\n")
    recs.append(">allN\n" + "N" * 60 + "\n")
    recs.append(">polyA\n" + "A" * 80 + "\n")
    recs.append(">  spaces  in header  \n" + good[10] + "\n")
    recs.append(">other char\n" + good[11][:30] + "R" + good[11][31:] + "\n")
    recs.append(">\n" + good[12] + "\n")
    for i in range(13, 40):
        r = list(good[i])
        for _ in range(rng.randint(1, 4)):
            r[rng.randrange(len(r))] = "N"
        recs.append(">nmix%d\n%s\n" % (i, "".join(r)))
    recs.append(">last no newline\n" + good[0][::-1][:120])
    write("edge_reads.fa", "".join(recs))
    # (iv) FASTQ: phantom record at EOF (file ends with newline), short / lowercase / N reads
    fq = []
    for i in range(6):
        fq.append("@q%d desc\n%s\n+\n%s\n" % (i, good[20 + i], "I" * 150))
    fq.append("@short\nACGTA\n+\nIIIII\n")
    fq.append("@two\nAC\n+\nII\n")
    fq.append("@lower\n%s\n+\n%s\n" % (good[30].lower(), "I" * 150))
    r = list(good[31]); r[77] = "N"
    fq.append("@withN\n%s\n+\n%s\n" % ("".join(r), "I" * 150))
    write("edge_reads.fq", "".join(fq))
    write("edge_reads_nonl.fq", "".join(fq)[:-1])  # no trailing newline: no phantom record
    # (v) degenerate graph, k=5: palindromic overlaps, hairpins, >4 unitigs per overlap, unitig of length k
    us = [rs(rng, rng.randint(5, 14)) for _ in range(300)]
    us += ["ACGTA", "TACGT", "AAAAA", "TTTTT", "ACGTACGT", "GATCGATC", "AATTAATT"]
    write("deg_unitig.fa", "".join(">%d\n%s\n" % (i + 1, u) for i, u in enumerate(us)))
    us2 = list(us)
    for i in range(0, len(us2), 6):
        u = list(us2[i]); u[rng.randrange(len(u))] = rng.choice("NNNacgtRY"); us2[i] = "".join(u)
    write("deg_unitig_exc.fa", "".join(">%d\n%s\n" % (i + 1, u) for i, u in enumerate(us2)))
    reads = []
    for i in range(1500):
        L = rng.randint(6, 40)
        r = list(rs(rng, L))
        if rng.random() < 0.12:
            for _ in range(rng.randint(1, 3)):
                r[rng.randrange(L)] = "N"
        reads.append("".join(r))
    write("deg_reads.fa", "".join(">d%d\n%s\n" % (i, r) for i, r in enumerate(reads)))
    # truncated unitig file: loading stops at the first sequence shorter than k
    write("short_stop_unitig.fa", "".join(">%d\n%s\n" % (i + 1, u) for i, u in enumerate(us[:40] + ["ACG"] + us[40:80])))


def cases():
    c = []
    c.append(("toy", ["-r", "toy_reads.fa", "-k", "4", "-g", "toy_unitig.fa"]))
    for rd in ("syn_r100.fa", "syn_r150.fa", "syn_r250.fa"):
        for m in (0, 2, 5):
            for e in (1, 2, 4):
                c.append(("syn", ["-r", rd, "-k", "31", "-g", "syn_unitig.fa", "-m", str(m), "-e", str(e)]))
    for m in (0, 2, 5):
        c.append(("long", ["-r", "long_r150.fa", "-k", "31", "-g", "long_unitig.fa", "-m", str(m)]))
    c.append(("long_fq", ["-r", "long_r150.fq", "-k", "31", "-g", "long_unitig.fa", "-m", "2", "-q"]))
    c.append(("multi", ["-r", "syn_r100.fa,syn_r150.fa", "-k", "31", "-g", "syn_unitig.fa", "-m", "2"]))
    for m in (0, 2, 5):
        c.append(("edge", ["-r", "edge_reads.fa", "-k", "31", "-g", "syn_unitig.fa", "-m", str(m)]))
    c.append(("edge_k25", ["-r", "edge_reads.fa", "-k", "25", "-g", "syn_unitig.fa", "-m", "3", "-e", "3"]))
    c.append(("edge_fq", ["-r", "edge_reads.fq", "-k", "31", "-g", "syn_unitig.fa", "-m", "2", "-q"]))
    c.append(("edge_fq_nonl", ["-r", "edge_reads_nonl.fq", "-k", "31", "-g", "syn_unitig.fa", "-m", "2", "-q"]))
    for g in ("deg_unitig.fa", "deg_unitig_exc.fa", "short_stop_unitig.fa"):
        for m in (0, 2, 5):
            for e in (2, 4):
                c.append(("deg", ["-r", "deg_reads.fa", "-k", "5", "-g", g, "-m", str(m), "-e", str(e)]))
    c.append(("deg_k6", ["-r", "deg_reads.fa", "-k", "6", "-g", "deg_unitig.fa", "-m", "3"]))
    # correction mode (-c): header + the read as spelled by its path (recoverPath, aligner.cpp:270-290)
    for rd, m in (("syn_r150.fa", 0), ("syn_r150.fa", 2), ("syn_r250.fa", 5), ("syn_r100.fa", 5)):
        c.append(("corr", ["-r", rd, "-k", "31", "-g", "syn_unitig.fa", "-m", str(m), "-c"]))
    c.append(("corr", ["-r", "edge_reads.fa", "-k", "31", "-g", "syn_unitig.fa", "-m", "5", "-c"]))
    c.append(("corr", ["-r", "long_r150.fa", "-k", "31", "-g", "long_unitig.fa", "-m", "2", "-c"]))
    c.append(("corr", ["-r", "long_r150.fq", "-k", "31", "-g", "long_unitig.fa", "-m", "2", "-c", "-q"]))
    for g in ("deg_unitig.fa", "deg_unitig_exc.fa"):
        c.append(("corr", ["-r", "deg_reads.fa", "-k", "5", "-g", g, "-m", "2", "-e", "4", "-c"]))
    # exhaustive (-b): counters from the unmodified reference, bytes from bgreat_exh
    for m in (0, 2, 5):
        c.append(("exh", ["-r", "syn_r150.fa", "-k", "31", "-g", "syn_unitig.fa", "-m", str(m), "-b"]))
        c.append(("exh", ["-r", "deg_reads.fa", "-k", "5", "-g", "deg_unitig.fa", "-m", str(m), "-b"]))
        c.append(("exh", ["-r", "deg_reads.fa", "-k", "5", "-g", "deg_unitig_exc.fa", "-m", str(m), "-b", "-i"]))
    c.append(("exh", ["-r", "syn_r250.fa", "-k", "31", "-g", "syn_unitig.fa", "-m", "5", "-b"]))
    c.append(("exh", ["-r", "long_r150.fa", "-k", "31", "-g", "long_unitig.fa", "-m", "5", "-b", "-i"]))
    c.append(("exh", ["-r", "edge_reads.fa", "-k", "31", "-g", "syn_unitig.fa", "-m", "5", "-b"]))
    # anchors mode (-G): k-mer anchors looked up WITHOUT a key check (aligner.cpp:387-389), so these bytes also pin the
    # index values of the BooPHF cascade (false-positive ranks included)
    c.append(("dog", ["-r", "toy_reads.fa", "-k", "4", "-g", "toy_unitig.fa", "-G"]))
    for rd, m, e in (("syn_r100.fa", 1, 2), ("syn_r150.fa", 0, 2), ("syn_r150.fa", 2, 2), ("syn_r150.fa", 5, 4), ("syn_r250.fa", 5, 3)):
        c.append(("dog", ["-r", rd, "-k", "31", "-g", "syn_unitig.fa", "-m", str(m), "-e", str(e), "-G"]))
    c.append(("dog", ["-r", "syn_r250.fa", "-k", "25", "-g", "syn_unitig.fa", "-m", "5", "-G"]))
    c.append(("dog", ["-r", "long_r150.fa", "-k", "31", "-g", "long_unitig.fa", "-m", "3", "-G"]))
    c.append(("dog", ["-r", "long_r150.fq", "-k", "31", "-g", "long_unitig.fa", "-m", "3", "-e", "5", "-q", "-G"]))
    c.append(("dog", ["-r", "edge_reads.fa", "-k", "31", "-g", "syn_unitig.fa", "-m", "2", "-G"]))
    c.append(("dog", ["-r", "edge_reads.fq", "-k", "31", "-g", "syn_unitig.fa", "-m", "2", "-q", "-G"]))
    c.append(("dog", ["-r", "deg_reads.fa", "-k", "5", "-g", "deg_unitig.fa", "-m", "1", "-G"]))
    c.append(("dog", ["-r", "deg_reads.fa", "-k", "5", "-g", "deg_unitig_exc.fa", "-m", "2", "-e", "4", "-G"]))
    c.append(("dog", ["-r", "syn_r150.fa", "-k", "31", "-g", "syn_unitig.fa", "-m", "2", "-c", "-G"]))
    c.append(("dog", ["-r", "long_r150.fa", "-k", "31", "-g", "long_unitig.fa", "-m", "5", "-b", "-G"]))   # -b wins: -G has no effect
    return c


def main():
    if not (os.path.exists(REF) and os.path.exists(REF_EXH)):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    if os.path.isdir(GOLD):
        shutil.rmtree(GOLD)
    os.makedirs(GOLD)
    make_inputs()
    expected = []
    for idx, (group, args) in enumerate(cases()):
        full = [os.path.join(GOLD, a) if a.endswith((".fa", ".fq")) and "," not in a else a for a in args]
        full = [",".join(os.path.join(GOLD, x) for x in a.split(",")) if "," in a else a for a in full]
        out, paths, na = run_ref(REF, full)
        entry = {"id": idx, "group": group, "args": args, "counters": parse_counters(out)}
        if "-b" in args:
            # unmodified reference writes nothing in exhaustive mode
            assert paths == b"" and na == b"", "reference -b unexpectedly wrote output"
            out2, paths, na = run_ref(REF_EXH, full)
            assert parse_counters(out2) == entry["counters"]
            entry["exhaustive_bytes_from"] = "bgreat_exh"
        entry["paths_sha256"] = sha(paths)
        entry["paths_len"] = len(paths)
        entry["notaligned_sha256"] = sha(na)
        entry["notaligned_len"] = len(na)
        if len(paths) <= KEEP_BYTES_LIMIT:
            entry["paths"] = paths.decode("latin-1")
        if len(na) <= KEEP_BYTES_LIMIT:
            entry["notaligned"] = na.decode("latin-1")
        expected.append(entry)
    with open(os.path.join(GOLD, "expected.json"), "w") as f:
        json.dump(expected, f, indent=1)
    write("README.md", """# tests/golden

Inputs and the COMPILED REFERENCE's outputs on them (generated by `tests/make_golden.py` in the build
container from `oracle/_ref/bgreat`, i.e. /root/reference compiled as-is).  `expected.json` holds, per
case, the command-line arguments, the stdout counters, sha256 + length of `paths` and `notAligned.fa`,
and the bytes themselves when small.

Exhaustive (`-b`) cases: the unmodified reference writes NOTHING in this mode (it builds the path string
and discards it, alignerExhaustive.cpp:285; failures go to never-opened streams).  For those cases the
counters come from the unmodified binary and the bytes from `oracle/_ref/bgreat_exh` -- the same sources
with those three statements redirected to the open files -- i.e. what the reference computes but does
not write.

`toy_reads.fa` / `toy_unitig.fa` are the reference's bundled data files (reads.fa / unitig.fa).
""")
    tot = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
    print("wrote %d cases, %d bytes in %s" % (len(expected), tot, GOLD))


if __name__ == "__main__":
    main()
