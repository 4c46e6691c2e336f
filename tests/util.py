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
