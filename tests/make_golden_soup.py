"""Expected outputs of tests/golden/soup_polyA_*: a degenerate unitig set (tools/fuzz_soup.py gpu, seed 2, configuration 20 as the tool stood when it found the
case: chains, fans, palindromes, self-loops, copies, and a homopolymer on BOTH strands -- A x 55 and T x 31) and the first 400 of its reads, exhaustive mode,
k = 31, m = 1.  One read (r207: 55 A's, then unrelated bases) costs the reference 150 s: every base of the run is a choice between slots that name the same move
and its recursion enumerates them all (alignerExhaustive.cpp:61-259).  The expected bytes are the reference's own: oracle/_ref/bgreat_exh (the reference with its
three fwrite sites of exhaustive mode redirected to the open files, oracle/Makefile) at -t 1.  Run from the repo root where /root/reference was compiled:
    python tests/make_golden_soup.py        (~3 minutes)
(Test infrastructure: generates fixtures, data only.)"""
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(ROOT, "oracle", "_ref", "bgreat_exh")

if __name__ == "__main__":
    d = tempfile.mkdtemp()
    try:
        subprocess.check_call([REF, "-r", os.path.join(GOLD, "soup_polyA_reads.fa"), "-k", "31", "-g", os.path.join(GOLD, "soup_polyA_unitig.fa"), "-m", "1", "-e", "2", "-t", "1", "-b",
                               "-f", os.path.join(d, "p"), "-a", os.path.join(d, "n")], cwd=d, stdout=subprocess.DEVNULL)
        shutil.copy(os.path.join(d, "p"), os.path.join(GOLD, "soup_polyA_expected_paths"))
        shutil.copy(os.path.join(d, "n"), os.path.join(GOLD, "soup_polyA_expected_notAligned.fa"))
    finally:
        shutil.rmtree(d)
