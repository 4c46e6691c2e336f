"""Full-config parity against the COMPILED REFERENCE (oracle/_ref, built from /root/reference in the build container; test infrastructure):
the whole read sets of BASELINE configs[1] (10 M x 100 bp, the ten 1 M-read launches bench.py times) and configs[2] (50 M x 150 bp), and
2 M x 250 bp of configs[4] (exhaustive, m = 5, 4-allele graph), each through bin/bgreat (this repository's CLI, one GPU) and through
oracle/_ref/bgreat -t N (configs[4]: bgreat_exh -b, the reference with its three fwrite redirects).  With -t N the reference writes
its records in whatever order its workers take the mutex: compared are the SORTED multisets of 2-line records -- sha256 of
`paste - - | LC_ALL=C sort` of both sides' `paths` and `notAligned.fa` -- the record counts, and the counters both programs print.  The
product's own files are additionally the -t 1 order by construction (tests compare that byte for byte on smaller inputs).

  python tools/full_parity.py [--out gpurun_out/r05/full_parity.txt] [--configs 1,2,4] [--threads 16] [--scale 1.0]

Runs on the GPU box (the reference binaries travel as oracle/_ref/*; /root/reference itself is not needed).  ~4-6 minutes, most of it the
reference's own mapping time (0.7 Mreads/s at -t 16).  bench.py --full-parity calls run() and carries the verdict in its line."""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools.synth import Synth  # noqa: E402

SEED_GRAPH, SEED_READS = 20261003, 77   # bench.py's
CONFIGS = {   # the graphs and read shapes of bench.py's presets
    1: dict(name="configs[1]: 10 M x 100 bp, ~10 k-unitig graph, greedy, m=2", genome=250_000, spacing=75, alleles=2, L=100, m=2, n=10_000_000, exhaustive=False),
    2: dict(name="configs[2]: 50 M x 150 bp, E. coli-scale graph, greedy, m=2", genome=4_600_000, spacing=140, alleles=2, L=150, m=2, n=50_000_000, exhaustive=False),
    4: dict(name="configs[4] graph: 2 M x 250 bp, 4 alleles every ~36 bp, exhaustive (-b), m=5", genome=50_000_000, spacing=36, alleles=4, L=250, m=5, n=2_000_000, exhaustive=True),
}


def sorted_records_sha256(path, threads, tmp):
    """sha256 and count of the file's 2-line records in byte order (`paste - - | LC_ALL=C sort`)."""
    env = dict(os.environ, LC_ALL="C")
    p1 = subprocess.Popen(["paste", "-", "-"], stdin=open(path, "rb"), stdout=subprocess.PIPE, env=env)
    p2 = subprocess.Popen(["sort", "--parallel=%d" % threads, "-S", "4G", "-T", tmp], stdin=p1.stdout, stdout=subprocess.PIPE, env=env)
    p1.stdout.close()
    h, n = hashlib.sha256(), 0
    while True:
        blk = p2.stdout.read(1 << 24)
        if not blk:
            break
        h.update(blk)
        n += blk.count(b"\n")
    if p2.wait() != 0 or p1.wait() != 0:
        raise RuntimeError("sort failed on %s" % path)
    return h.hexdigest(), n


def counters(stdout):
    out = {}
    for line in stdout.splitlines():
        for key, tag in (("reads", "Reads :"), ("no_overlap", "No overlap :"), ("aligned", "Overlap and aligned :"), ("not_aligned", "Overlap but not aligned :")):
            if line.startswith(tag):
                out[key] = int(line[len(tag):].split()[0])
    return out


def run_config(c, cfg, threads, scale, log):
    n = max(1000, int(cfg["n"] * scale))
    d = tempfile.mkdtemp(prefix="bgr_fullparity_%d_" % c)
    rec = {"config": cfg["name"], "reads": n}
    try:
        syn = Synth(cfg["genome"], cfg["spacing"], cfg["alleles"], 31, SEED_GRAPH)
        syn.write_unitigs(os.path.join(d, "u.fa"))
        t0 = time.time()
        syn.write_reads(os.path.join(d, "r.fa"), 0, n, cfg["L"], cfg["m"], SEED_READS, threads=threads)
        rec["input_bytes"] = os.path.getsize(os.path.join(d, "r.fa"))
        log("  input written: %.1f GB in %.1f s" % (rec["input_bytes"] / 1e9, time.time() - t0))
        base = ["-r", os.path.join(d, "r.fa"), "-k", "31", "-g", os.path.join(d, "u.fa"), "-m", str(cfg["m"]), "-t", str(threads)]
        sides = {}
        for side, exe, extra in (("gpu", os.path.join(ROOT, "bgreat_amd", "bin", "bgreat"), ["-b", "--write-exhaustive"] if cfg["exhaustive"] else []),
                                 ("ref", os.path.join(ROOT, "oracle", "_ref", "bgreat_exh" if cfg["exhaustive"] else "bgreat"), ["-b"] if cfg["exhaustive"] else [])):
            if not os.path.exists(exe):
                raise RuntimeError("%s is missing (make -C oracle ref / make -C bgreat_amd)" % exe)
            wd = os.path.join(d, side)
            os.makedirs(wd)
            t0 = time.time()
            p = subprocess.run([exe] + base + extra, cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if p.returncode != 0:
                raise RuntimeError("%s failed (rc %d): %s" % (exe, p.returncode, p.stderr[-400:]))
            wall = time.time() - t0
            cnt = counters(p.stdout)
            s = {"wall_s": round(wall, 1), "counters": cnt}
            for fn in ("paths", "notAligned.fa"):
                sha, nrec = sorted_records_sha256(os.path.join(wd, fn), threads, d)
                s[fn] = {"bytes": os.path.getsize(os.path.join(wd, fn)), "records": nrec, "sha256_sorted_records": sha}
                os.unlink(os.path.join(wd, fn))
            sides[side] = s
            log("  %s: %.1f s, %s" % (side, wall, cnt))
        rec.update(sides)
        rec["equal"] = bool(sides["gpu"]["counters"] == sides["ref"]["counters"] and sides["gpu"]["counters"].get("reads") == n and
                            all(sides["gpu"][fn] == sides["ref"][fn] for fn in ("paths", "notAligned.fa")))
    except Exception as ex:
        rec["error"] = "%s: %s" % (type(ex).__name__, ex)
        rec["equal"] = False
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return rec


def run(configs=(1, 2, 4), threads=16, scale=1.0, out=None, log=lambda s: print(s, file=sys.stderr, flush=True)):
    """-> {"equal": bool, "configs": {...}}; writes the report to `out` when given."""
    res = {"equal": True, "threads": threads, "scale": scale, "configs": {}}
    t0 = time.time()
    for c in configs:
        log("full parity, %s" % CONFIGS[c]["name"])
        r = run_config(c, CONFIGS[c], threads, scale, log)
        res["configs"][str(c)] = r
        res["equal"] = res["equal"] and r["equal"]
    res["seconds"] = round(time.time() - t0, 1)
    if out:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        with open(out, "w") as f:
            f.write("# tools/full_parity.py: bin/bgreat (1 x MI355X) against the compiled reference (oracle/_ref, -t %d) on whole read sets; sorted multisets of 2-line records\n" % threads)
            f.write("# verdict: %s   (%.0f s)\n" % ("EQUAL on every config" if res["equal"] else "DIFFERENT", res["seconds"]))
            for c, r in res["configs"].items():
                f.write("\n## %s -- %d reads -- %s\n" % (r["config"], r["reads"], "equal" if r["equal"] else "DIFFERENT" + (" (%s)" % r["error"] if "error" in r else "")))
                for side in ("gpu", "ref"):
                    if side in r:
                        s = r[side]
                        f.write("%s  wall %.1f s  counters %s\n" % ({"gpu": "bin/bgreat        ", "ref": "compiled reference"}[side], s["wall_s"], json.dumps(s["counters"], sort_keys=True)))
                        for fn in ("paths", "notAligned.fa"):
                            f.write("    %-14s %12d bytes %10d records  sha256(sorted records) %s\n" % (fn, s[fn]["bytes"], s[fn]["records"], s[fn]["sha256_sorted_records"]))
            f.write("\n" + json.dumps(res) + "\n")
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "full_parity.txt"))
    ap.add_argument("--configs", default="1,2,4")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of each config's read count (tests: 0.001)")
    a = ap.parse_args()
    r = run(tuple(int(x) for x in a.configs.split(",")), a.threads, a.scale, a.out)
    print(json.dumps({"equal": r["equal"], "seconds": r["seconds"], "configs": {c: {"reads": v["reads"], "equal": v["equal"], **({"error": v["error"]} if "error" in v else {})} for c, v in r["configs"].items()}}))
    sys.exit(0 if r["equal"] else 1)
