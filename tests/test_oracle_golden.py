"""The oracle (oracle/bgreat_oracle.cpp, this repo's CPU restatement) against the committed outputs of the
compiled reference (tests/golden/expected.json).  This is what pins the oracle on machines where
/root/reference does not exist."""
import pytest

from util import check_against_golden, golden_cases, resolve_args, run_cli

CASES = golden_cases()


@pytest.mark.parametrize("case", CASES, ids=["%02d-%s" % (c["id"], c["group"]) for c in CASES])
def test_oracle_matches_reference_golden(case, oracle_bins):
    env = {"ORACLE_EXH_WRITES": "1"} if "-b" in case["args"] else None
    out, paths, na = run_cli(oracle_bins["cli"], resolve_args(case["args"]), env=env)
    check_against_golden(case, out, paths, na)


def test_oracle_exhaustive_writes_nothing_by_default(oracle_bins):
    """the reference's -b mode leaves paths / notAligned.fa empty."""
    case = next(c for c in CASES if "-b" in c["args"])
    out, paths, na = run_cli(oracle_bins["cli"], resolve_args(case["args"]))
    assert paths == b"" and na == b""


# ---- the remembered-calls form of the oracle's exhaustive mode (Oracle::exh_memo; what checks the device on unitig sets where the literal
# recursion -- the reference's -- is exponential) must return what the literal form returns ----
@pytest.mark.parametrize("case", [c for c in CASES if "-b" in c["args"]], ids=lambda c: "%02d-%s" % (c["id"], c["group"]))
def test_oracle_with_remembered_calls_matches_reference_golden(case, oracle_bins):
    out, paths, na = run_cli(oracle_bins["cli"], resolve_args(case["args"]), env={"ORACLE_EXH_WRITES": "1", "ORACLE_EXH_MEMO": "1"})
    check_against_golden(case, out, paths, na)


def test_oracle_with_remembered_calls_on_the_soup_the_reference_needs_150_s_for(oracle_bins):
    """tests/golden/soup_polyA_*: expected bytes = the compiled reference's own (tests/make_golden_soup.py; one 79-base poly-A read costs it 150 s)."""
    import os
    from util import GOLD, parse_counters
    args = ["-r", os.path.join(GOLD, "soup_polyA_reads.fa"), "-k", "31", "-g", os.path.join(GOLD, "soup_polyA_unitig.fa"), "-m", "1", "-e", "2", "-t", "1", "-b"]
    out, paths, na = run_cli(oracle_bins["cli"], args, env={"ORACLE_EXH_WRITES": "1", "ORACLE_EXH_MEMO": "1"}, timeout=60)
    assert paths == open(os.path.join(GOLD, "soup_polyA_expected_paths"), "rb").read()
    assert na == open(os.path.join(GOLD, "soup_polyA_expected_notAligned.fa"), "rb").read()
    assert parse_counters(out)["aligned"] == 141 and parse_counters(out)["reads"] == 400


@pytest.mark.parametrize("m", [0, 1, 2, 5])
def test_oracle_literal_and_remembered_forms_agree_where_the_literal_one_is_exponential(m):
    """Poly-A unitigs of six lengths; reads with their mismatches at the far end: the literal recursion doubles per base (0.1 s at 32 bases)."""
    import numpy as np
    import oracle_py
    from util import homopolymer_soup
    k, (seqs, offs), (sr, so), _ = homopolymer_soup()
    o = oracle_py.Oracle(k, seqs, offs)
    for partial in (False, True):
        a, b = o.align(sr, so, m=m, mode=1, partial=partial), o.align(sr, so, m=m, mode=3, partial=partial)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
        assert int(((a[2] & 3) == 2).sum()) > 0
