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
