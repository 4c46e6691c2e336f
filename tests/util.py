"""Shared helpers for the parity tests (test infrastructure)."""
import hashlib
import json
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def sha(b):
    return hashlib.sha256(b).hexdigest()


def golden_cases():
    with open(os.path.join(GOLD, "expected.json")) as f:
        return json.load(f)


def resolve_args(args):
    """Golden args name files relative to tests/golden; make them absolute (also inside comma lists)."""
    out = []
    for a in args:
        if a.endswith((".fa", ".fq")):
            out.append(",".join(os.path.join(GOLD, x) for x in a.split(",")))
        else:
            out.append(a)
    return out


def parse_counters(out):
    c = {}
    for line in out.splitlines():
        for key, name in (("Reads : ", "reads"), ("No overlap : ", "no_overlap"), ("Got overlap : ", "got_overlap"),
                          ("Overlap and aligned : ", "aligned"), ("Overlap but not aligned : ", "not_aligned")):
            if line.startswith(key):
                c[name] = int(line[len(key):].split()[0])
    return c


def run_cli(binary, args, env=None, timeout=600):
    """Run a bgreat-style CLI in a scratch dir -> (stdout, paths bytes, notAligned bytes)."""
    d = tempfile.mkdtemp()
    try:
        e = dict(os.environ)
        if env:
            e.update(env)
        p = subprocess.run([binary] + list(args), cwd=d, capture_output=True, text=True, env=e, timeout=timeout)
        if p.returncode != 0:
            raise RuntimeError("%s %s failed (%d): %s" % (binary, args, p.returncode, p.stderr[-2000:]))
        paths = open(os.path.join(d, "paths"), "rb").read() if os.path.exists(os.path.join(d, "paths")) else b""
        na = open(os.path.join(d, "notAligned.fa"), "rb").read() if os.path.exists(os.path.join(d, "notAligned.fa")) else b""
        return p.stdout, paths, na
    finally:
        shutil.rmtree(d)


def check_against_golden(case, out, paths, na):
    assert parse_counters(out) == case["counters"], (case["args"], parse_counters(out), case["counters"])
    if "paths" in case:
        assert paths.decode("latin-1") == case["paths"], case["args"]
    if "notaligned" in case:
        assert na.decode("latin-1") == case["notaligned"], case["args"]
    assert len(paths) == case["paths_len"] and sha(paths) == case["paths_sha256"], case["args"]
    assert len(na) == case["notaligned_len"] and sha(na) == case["notaligned_sha256"], case["args"]


import numpy as np


def homopolymer_soup(k=15, seed=5):
    """Unitigs that duplicate each other's k-mers wholesale -- poly-A of six lengths (every walk step is a choice between four unitigs that advance
    1, 2, 3 or 4 bases: levels of the search spread over many (overlap, position) nodes), poly-T, random strings -- and reads on which the reference's
    recursion is exponential: poly-A with more mismatches than the budget at the FAR end (every walk is explored to its end before it fails: x 2 per
    base, 2 s at 36 bases in the oracle's literal form), with a few near the anchor, clean ones, walks from random sequence into the homopolymer."""
    rng = np.random.default_rng(seed)
    rs = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    us = ["A" * (k + j) for j in (0, 1, 2, 3, 5, 8)] + ["T" * (k + 2)] + [rs(40) for _ in range(20)]
    tail = rs(30)
    us.append(tail + "A" * k)
    short, long_ = [], []
    for L in (24, 26, 28, 30, 32):
        short += ["A" * (L - 5) + "CACAC", "A" * L, "A" * (L - 9) + "C" + "A" * 8, "C" + "A" * (L - 1)]
    for L in (40, 60, 79, 120, 200, 250):
        long_ += ["A" * (L - 5) + "CACAC", "A" * L, "CAC" + "A" * (L - 8) + "CACAC", "A" * (L // 2) + "C" + "A" * (L - L // 2 - 1), tail[-20:] + "A" * (L - 20),
                  "T" * (L - 4) + "GTGT", "A" * (L - 30) + rs(30)]
    pack = lambda strs: (np.frombuffer("".join(strs).encode(), dtype=np.uint8), np.concatenate([[0], np.cumsum([len(x) for x in strs])]).astype(np.uint64))
    return k, pack(us), pack(short), pack(long_)
