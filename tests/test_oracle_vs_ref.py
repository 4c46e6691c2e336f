"""Live cross-check oracle <-> compiled reference on fresh seeded inputs (only where oracle/_ref exists,
i.e. wherever /root/reference was available to oracle/Makefile; the binary travels to the GPU box)."""
import os
import tempfile

import pytest

from tools.synth import Synth
from util import parse_counters, run_cli


@pytest.mark.parametrize("seed,k,L,m,e,threads", [(1, 31, 150, 2, 2, 1), (2, 31, 100, 2, 2, 1), (3, 21, 120, 4, 3, 1),
                                                  (4, 31, 250, 5, 4, 1), (5, 32, 150, 2, 2, 1), (6, 31, 150, 2, 2, 4)])
def test_greedy_random(oracle_bins, seed, k, L, m, e, threads):
    if not oracle_bins["ref"]:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    with tempfile.TemporaryDirectory() as d:
        s = Synth(50000, 80, 2, k, 1000 + seed)
        s.write_unitigs(os.path.join(d, "u.fa"))
        s.write_reads(os.path.join(d, "r.fa"), 0, 2000, L, m + 1, 2000 + seed)
        args = ["-r", os.path.join(d, "r.fa"), "-k", str(k), "-g", os.path.join(d, "u.fa"), "-m", str(m), "-e", str(e), "-t", str(threads)]
        o1, p1, n1 = run_cli(oracle_bins["ref"], args)
        o2, p2, n2 = run_cli(oracle_bins["cli"], args)
        assert parse_counters(o1) == parse_counters(o2)
        if threads == 1:
            assert p1 == p2 and n1 == n2
        else:  # record order is only deterministic with -t 1: compare 2-line record multisets
            def recs(b):
                ls = b.split(b"\n")
                return sorted(zip(ls[0::2], ls[1::2]))
            assert recs(p1) == recs(p2) and recs(n1) == recs(n2)


@pytest.mark.parametrize("seed,m,partial", [(1, 2, False), (2, 5, False), (3, 5, True)])
def test_exhaustive_random(oracle_bins, seed, m, partial):
    if not oracle_bins["ref_exh"]:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    with tempfile.TemporaryDirectory() as d:
        s = Synth(30000, 45, 4, 31, 3000 + seed)
        s.write_unitigs(os.path.join(d, "u.fa"))
        s.write_reads(os.path.join(d, "r.fa"), 0, 1000, 200, m, 4000 + seed)
        args = ["-r", os.path.join(d, "r.fa"), "-k", "31", "-g", os.path.join(d, "u.fa"), "-m", str(m), "-b"] + (["-i"] if partial else [])
        o1, p1, n1 = run_cli(oracle_bins["ref_exh"], args)
        o2, p2, n2 = run_cli(oracle_bins["cli"], args, env={"ORACLE_EXH_WRITES": "1"})
        assert parse_counters(o1) == parse_counters(o2)
        assert p1 == p2 and n1 == n2


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_degenerate_graphs(oracle_bins, seed):
    """tools/fuzz_soup.py cpu: unitig soups with fans of more than four unitigs per overlap, palindromic overlaps, self-loops, hairpins,
    duplicates, homopolymers, N, k from 4 to 32, all modes -- the oracle's bytes and counters against the compiled reference's (the
    campaign of profiles/r04_fuzz_campaign.txt ran 1 800 of them; here 20 per seed)."""
    if not oracle_bins["ref"] or not oracle_bins["ref_exh"]:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    import subprocess
    import sys
    from util import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_soup.py"), "cpu", str(seed), "20"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "bad 0" in p.stdout and p.stdout.count("\nok ") + p.stdout.startswith("ok ") >= 15, p.stdout[-2000:] + p.stderr[-2000:]
