"""Parity tests proper: the HIP path (through the C-ABI of include/bgreat_gpu.h) against
  (1) the committed outputs of the compiled reference (tests/golden), byte for byte, and
  (2) the oracle (oracle/liboracle.so) on fresh seeded inputs: identical path ints, offsets and status bytes.
Integer work: the bar is bit-exact everywhere."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

import bgreat_amd as B
import oracle_py
from tools.synth import Synth
from util import GOLD, ROOT, check_against_golden, golden_cases, resolve_args, run_cli

pytestmark = pytest.mark.gpu

CASES = golden_cases()
GREEDY = [c for c in CASES if "-b" not in c["args"] and "-c" not in c["args"]]
CORR = [c for c in CASES if "-c" in c["args"]]
EXH = [c for c in CASES if "-b" in c["args"]]


def _argval(args, flag, default):
    return args[args.index(flag) + 1] if flag in args else default


def _format(reads, roffs, heads, hoffs, paths, poffs):
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    with tempfile.TemporaryDirectory() as d:
        pf = libc.fopen(os.path.join(d, "p").encode(), b"wb")
        nf = libc.fopen(os.path.join(d, "n").encode(), b"wb")
        rc = B.lib().bgr_write_records(pf, nf, len(roffs) - 1, heads.ctypes.data, hoffs.ctypes.data, reads.ctypes.data, roffs.ctypes.data,
                                       paths.ctypes.data if len(paths) else None, poffs.ctypes.data)
        libc.fclose(pf)
        libc.fclose(nf)
        assert rc == 0
        return open(os.path.join(d, "p"), "rb").read(), open(os.path.join(d, "n"), "rb").read()


@pytest.fixture(scope="module")
def graphs():
    cache = {}

    def get(path, k, anchors=False):
        key = (path, k, anchors)
        if key not in cache:
            g = B.Graph.from_fasta(path, k, anchors=anchors)
            cache[key] = (g, B.Aligner(g, 0))
        return cache[key]
    yield get
    cache.clear()


@pytest.mark.parametrize("case", GREEDY + EXH, ids=["%02d-%s" % (c["id"], c["group"]) for c in GREEDY + EXH])
def test_gpu_matches_reference_golden(case, graphs):
    """Greedy cases: bytes of the unmodified reference.  Exhaustive (-b) cases: counters of the unmodified reference,
    bytes of what it computes but does not write (tests/golden/README.md)."""
    args = case["args"]
    k = int(_argval(args, "-k", "30"))
    m = int(_argval(args, "-m", "2"))
    e = int(_argval(args, "-e", "2"))
    fastq = "-q" in args
    mode = B.MODE_EXHAUSTIVE if "-b" in args else (B.MODE_ANCHORS if "-G" in args else B.MODE_GREEDY)   # -b wins over -G
    g, al = graphs(os.path.join(GOLD, _argval(args, "-g", None)), k, mode == B.MODE_ANCHORS)
    al.reset_counters()
    # alternate between the two formulations of the exhaustive search over the -b cases
    al.set_knob(B.KNOB_EXH_SEARCH, (B.SEARCH_BY_LEVEL if case["id"] & 1 else B.SEARCH_DEPTH_FIRST) if "-b" in args else B.SEARCH_AUTO)
    pbytes, nbytes = b"", b""
    for f in _argval(args, "-r", None).split(","):
        reads, roffs, heads, hoffs = B.load_reads(os.path.join(GOLD, f), k, fastq)
        paths, poffs, status = al.align(reads, roffs, m=m, effort=e, mode=mode, partial="-i" in args)
        p, n = _format(reads, roffs, heads, hoffs, paths, poffs)
        pbytes += p
        nbytes += n
    cnt = al.counters()
    out = "Reads : %d\nNo overlap : %d x\nGot overlap : %d x\nOverlap and aligned : %d x\nOverlap but not aligned : %d x\n" % (
        cnt["reads"], cnt["no_overlap"], cnt["aligned"] + cnt["not_aligned"], cnt["aligned"], cnt["not_aligned"])
    check_against_golden(case, out, pbytes, nbytes)


@pytest.mark.parametrize("case", [c for c in GREEDY if c["group"] in ("toy", "edge", "edge_fq", "multi", "long_fq", "deg")][:12] +
                         [c for c in GREEDY if c["group"] == "dog"],
                         ids=lambda c: "%02d-%s" % (c["id"], c["group"]))
def test_cli_matches_reference_golden(case):
    out, paths, na = run_cli(B.CLI_PATH, resolve_args(case["args"]))
    check_against_golden(case, out, paths, na)


@pytest.mark.parametrize("case", [c for c in GREEDY if c["group"] in ("edge", "multi", "syn", "deg")][::5],
                         ids=lambda c: "%02d-%s" % (c["id"], c["group"]))
def test_cli_pipeline_threads_and_tiny_chunks_keep_the_t1_stream(case):
    """Any thread count, batch size and parser chunk size must give the reference's -t 1 bytes."""
    out, paths, na = run_cli(B.CLI_PATH, resolve_args(case["args"]) + ["-t", "5", "--batch", "37", "--chunk-bytes", "600"])
    check_against_golden(case, out, paths, na)


@pytest.mark.parametrize("k,L,m,e,seed", [(31, 150, 2, 2, 1), (21, 100, 3, 1, 2), (32, 250, 5, 4, 3), (12, 60, 1, 3, 4), (31, 33, 0, 2, 5),
                                          (25, 400, 4, 0, 6), (8, 40, 2, 5, 7)])
def test_anchors_mode_random_vs_oracle(k, L, m, e, seed):
    """-G on fresh graphs and reads (both strands, substitutions, some N): identical ints, offsets and status bytes."""
    s = Synth(60000, 3 * k, 3, k, 100 + seed)
    seqs, offs = s.unitigs()
    g = B.Graph.build(k, seqs, offs, anchors=True)
    al = B.Aligner(g, 0)
    o = oracle_py.Oracle(k, seqs, offs, anchors=True)
    n = 3000
    reads, roffs = s.reads(0, n, L, m + 1, 200 + seed)
    reads = reads.copy()
    rng = np.random.default_rng(seed)
    idx = rng.choice(len(reads), size=n // 10, replace=False)
    reads[idx] = ord("N")
    p1, po1, st1 = al.align(reads, roffs, m=m, effort=e, mode=B.MODE_ANCHORS)
    p2, po2, st2 = o.align(reads, roffs, m=m, effort=e, mode=2)
    assert np.array_equal(st1, st2), np.nonzero(st1 != st2)[0][:10]
    assert np.array_equal(po1, po2) and np.array_equal(p1, p2)
    assert ((st1 & 3) == B.ST_ALIGNED).sum() > n // 20
    assert al.counters() == o.counters()


@pytest.mark.parametrize("seed,k,L,m,e,nfrac", [(1, 31, 150, 2, 2, 0.0), (2, 21, 100, 3, 1, 0.002), (3, 32, 250, 5, 4, 0.0), (4, 12, 60, 1, 3, 0.0), (5, 31, 40, 0, 0, 0.0),
                                                  (6, 9, 90, 2, 8, 0.0)])
def test_anchors_four_reads_per_wave_pass_equals_general_kernel_and_oracle(seed, k, L, m, e, nfrac):
    """-G maps with bgr_align_anchors4_kernel (four reads per wave) and leaves N reads and very long paths to the one-read-per-wave
    kernel.  With and without the first pass, and the oracle, must agree row for row, counters included."""
    s = Synth(120000, max(k + 5, 60), 2, k, 6100 + seed)
    seqs, offs = s.unitigs()
    n = 9003 - seed
    reads, roffs = s.reads(0, n, L, m + 1, 6200 + seed)
    if nfrac:
        reads = _inject_n(reads, np.random.default_rng(seed), nfrac)
    g = B.Graph.build(k, seqs, offs, anchors=True)
    al = B.Aligner(g, 0)
    o = oracle_py.Oracle(k, seqs, offs, anchors=True)
    p2, po2, st2 = o.align(reads, roffs, m=m, effort=e, mode=2)
    p1, po1, st1 = al.align(reads, roffs, m=m, effort=e, mode=B.MODE_ANCHORS)
    assert al.launch_info()["four_reads_per_wave"]
    assert np.array_equal(st1, st2), np.nonzero(st1 != st2)[0][:10]
    assert np.array_equal(po1, po2) and np.array_equal(p1, p2)
    c1, c2 = al.counters(), o.counters()
    c1["overlaps"] = c2["overlaps"] = 0
    assert c1 == c2
    al.set_knob(B.KNOB_ANCHORS_FAST, 1)
    p3, po3, st3 = al.align(reads, roffs, m=m, effort=e, mode=B.MODE_ANCHORS)
    assert not al.launch_info()["four_reads_per_wave"]
    assert np.array_equal(st3, st2) and np.array_equal(po3, po2) and np.array_equal(p3, p2)


def test_anchors_mode_needs_the_anchors_index():
    s = Synth(30000, 90, 2, 31, 3)
    seqs, offs = s.unitigs()
    al = B.Aligner(B.Graph.build(31, seqs, offs), 0)
    reads, roffs = s.reads(0, 10, 100, 1, 4)
    with pytest.raises(B.BgrError, match="BGR_BUILD_ANCHORS"):
        al.align(reads, roffs, mode=B.MODE_ANCHORS)


def test_cli_end_to_end_against_reference_binary(oracle_bins):
    """200k reads through the whole host pipeline (mmap, chunk-parallel parse, pinned batches, 2 streams, parallel
    formatting, ordered writer) == the compiled reference at -t 1, byte for byte."""
    ref = oracle_bins["ref"] or oracle_bins["cli"]
    with tempfile.TemporaryDirectory() as d:
        s = Synth(500000, 120, 2, 31, 4242)
        s.write_unitigs(os.path.join(d, "u.fa"))
        s.write_reads(os.path.join(d, "r.fa"), 0, 200000, 150, 3, 4243)
        args = ["-r", os.path.join(d, "r.fa"), "-k", "31", "-g", os.path.join(d, "u.fa"), "-m", "2"]
        o1, p1, n1 = run_cli(ref, args + ["-t", "1"])
        from util import parse_counters
        def timeless(o):
            return [ln for ln in o.splitlines() if "seconds" not in ln]
        # both routes: the device takes the file as text (parse, map, format on the GPU), and the host pipeline (--host-route)
        for extra in ([], ["--host-route"], ["--batch", "7001"]):
            o2, p2, n2 = run_cli(B.CLI_PATH, args + ["-t", "8", "--batch", "30000", "--chunk-bytes", "1000000"] + extra)
            assert parse_counters(o1) == parse_counters(o2), extra
            assert p1 == p2 and n1 == n2, extra
            # the whole stdout, line for line, but for the three wall-clock lines (aligner.cpp:546,559,588-596)
            assert timeless(o1) == timeless(o2), extra


@pytest.mark.parametrize("fastq", [False, True])
def test_cli_on_messy_files_against_reference_binary(oracle_bins, fastq, tmp_path):
    """Files whose lines are a random mix of mappable reads, junk, blank lines, CR, lower case, stray '>' / '@' /
    '+', with or without a final newline: the whole CLI (chunk-parallel parse, GPU, ordered writer) == the compiled
    reference (or, without it, the oracle CLI) at -t 1."""
    ref = oracle_bins["ref"] or oracle_bins["cli"]
    s = Synth(200000, 100, 2, 31, 555)
    s.write_unitigs(str(tmp_path / "u.fa"))
    good, goffs = s.reads(0, 4000, 120, 2, 556)
    rng = np.random.default_rng(99 + fastq)
    for it in range(3):
        lines = []
        for i in range(6000):
            r = rng.integers(0, 24)
            if r < 12:
                j = int(rng.integers(0, 4000))
                rd = good[int(goffs[j]):int(goffs[j + 1])].tobytes()
                if rng.integers(0, 6) == 0:
                    # (FASTQ: the reference dies with std::out_of_range on an accepted sequence shorter than k-1, so
                    # every all-ACGTN line of the FASTQ variant is kept at 31+ characters)
                    cut = int(rng.integers(31, len(rd) - 31)) if fastq else int(rng.integers(1, len(rd)))
                    rd = rd[:cut] + b"\n" + rd[cut:]          # multi-line record
                lines.append(((b"@" if fastq else b">") + b"r%d" % i))
                lines.append(rd)
                if fastq:
                    lines.append(b"+")
                    lines.append(b"I" * 120)
            elif r < 14:
                lines.append(b"")
            elif r < 16:
                lines.append((b"@" if fastq else b">") + b"junk header %d" % i)
            elif r < 18:
                lines.append(bytes(rng.choice(list(b"ACGTNacgtRX>@+ "), size=int(rng.integers(1, 90))).astype(np.uint8)) + (b"x" if fastq else b""))
            elif r < 20:
                lines.append(good[:int(rng.integers(1, 60))].tobytes() + b"\r")
            elif r < 22:
                lines.append(b"+")
            else:
                lines.append(b"N" * int(rng.integers(31 if fastq else 1, 80)))
        text = b"\n".join(lines) + (b"\n" if it != 1 else b"")
        f = str(tmp_path / ("m%d.fx" % it))
        open(f, "wb").write(text)
        args = ["-r", f, "-k", "31", "-g", str(tmp_path / "u.fa"), "-m", "2"] + (["-q"] if fastq else [])
        o1, p1, n1 = run_cli(ref, args + ["-t", "1"])
        from util import parse_counters
        # (FASTA: by default every piece goes to the device as text first and comes back as irregular -- the fall-back per piece,
        # then per file; --host-route: the host parser from the start)
        for extra in ([], ["--host-route"]):
            o2, p2, n2 = run_cli(B.CLI_PATH, args + ["-t", "5", "--batch", "700", "--chunk-bytes", "3000"] + extra)
            assert parse_counters(o1) == parse_counters(o2), (it, extra)
            assert p1 == p2 and n1 == n2, (it, extra)
        assert parse_counters(o1)["aligned"] > 200


@pytest.mark.parametrize("case", CORR, ids=lambda c: "%02d-%s" % (c["id"], c["group"]))
def test_cli_correction_mode_matches_reference_golden(case):
    """-c: mapped reads are written as header + the read spelled by its path (aligner.cpp:270-290, alignerGreedy.cpp:394-404)."""
    out, paths, na = run_cli(B.CLI_PATH, resolve_args(case["args"]) + ["-t", "3", "--batch", "101"])
    check_against_golden(case, out, paths, na)


def test_cli_no_overlap_split_is_opt_in():
    """--no-overlap FILE moves exactly the reads without any anchor out of notAligned.fa; everything else is unchanged."""
    case = next(c for c in GREEDY if c["group"] == "edge" and c["args"][7] == "2")
    with tempfile.TemporaryDirectory() as d:
        nov = os.path.join(d, "noOverlap.fa")
        out, paths, na = run_cli(B.CLI_PATH, resolve_args(case["args"]) + ["--no-overlap", nov])
        novb = open(nov, "rb").read()
    assert paths.decode("latin-1") == case["paths"]
    recs = lambda b: list(zip(b.split(b"\n")[0::2], b.split(b"\n")[1::2]))
    want = recs(case["notaligned"].encode("latin-1"))
    got = recs(na) + recs(novb)
    assert sorted(want) == sorted(got)
    assert len(recs(novb)) == case["counters"]["no_overlap"]


def test_cli_exhaustive_writes_nothing_unless_asked():
    case = next(c for c in EXH if c["args"][1] == "syn_r150.fa" and c["args"][7] == "2")
    out, paths, na = run_cli(B.CLI_PATH, resolve_args(case["args"]))
    assert paths == b"" and na == b""
    from util import parse_counters
    assert parse_counters(out) == case["counters"]
    out, paths, na = run_cli(B.CLI_PATH, resolve_args(case["args"]) + ["--write-exhaustive"])
    check_against_golden(case, out, paths, na)


@pytest.mark.parametrize("fastq", [False, True])
def test_cli_exhaustive_stdout_against_reference_binary(oracle_bins, fastq, tmp_path):
    """-b: stdout is all the reference shows, including the block its worker prints after every tenth
    getReads() call (alignerExhaustive.cpp:306-316; `iter` starts at 1, aligner.h:103, and runs on over the files).  At
    -t 1 every line of it but Reads/seconds is deterministic: the CLI prints the same lines in the same places, whatever
    its threads, batch and chunk sizes.  A call is 10 000 record ATTEMPTS, so the files carry dropped records too."""
    ref = oracle_bins["ref"]
    if not ref:
        pytest.skip("needs the compiled reference (oracle/_ref/bgreat)")
    s = Synth(200000, 100, 3, 31, 808)
    s.write_unitigs(str(tmp_path / "u.fa"))
    rng = np.random.default_rng(5 + fastq)
    files = []
    for fi, n in enumerate((127003, 40000, 36000)):   # calls 13 + 4 + 4: blocks after calls 10 and 20 (file 3), none at a file end
        good, goffs = s.reads(fi * 200000, n, 64, 3, 809 + fi)
        out = []
        for i in range(n):
            rd = good[int(goffs[i]):int(goffs[i + 1])].tobytes()
            r = int(rng.integers(0, 50))
            if r == 0:
                rd = rd[:20].lower() + rd[20:]                 # dropped (characters), still an iteration
            elif r == 1 and not fastq:
                rd = rd[:25]                                   # FASTA: dropped (size <= k)
            elif r == 2 and not fastq:
                rd = rd[:30] + b"\n" + rd[30:]                 # multi-line record: one iteration
            out.append((b"@" if fastq else b">") + b"f%d_%d\n" % (fi, i) + rd + (b"\n+\n" + b"I" * len(rd) if fastq else b"") + b"\n")
        f = str(tmp_path / ("p%d.%s" % (fi, "fq" if fastq else "fa")))
        open(f, "wb").write(b"".join(out))
        files.append(f)
    args = ["-r", ",".join(files), "-k", "31", "-g", str(tmp_path / "u.fa"), "-m", "3", "-b"] + (["-q"] if fastq else [])
    o1, p1, n1 = run_cli(ref, args + ["-t", "1"])
    def timeless(o):
        return [ln for ln in o.splitlines() if "seconds" not in ln]
    assert sum(ln.startswith("Read : ") for ln in o1.splitlines()) == 2
    for extra in (["-t", "6", "--batch", "7001", "--chunk-bytes", "40000"], ["-t", "2"]):
        o2, p2, n2 = run_cli(B.CLI_PATH, args + extra)
        assert timeless(o1) == timeless(o2), extra
        assert p2 == b"" and n2 == b""


def test_cli_reads_and_unitigs_through_fifos(tmp_path):
    """Inputs that are not regular files (FIFOs, process substitution: st_size is 0 and mmap is impossible) are read to
    the end like the reference's ifstream does; same bytes as from the files."""
    import threading
    s = Synth(150000, 90, 2, 31, 31)
    ufa, rfa = str(tmp_path / "u.fa"), str(tmp_path / "r.fa")
    s.write_unitigs(ufa)
    s.write_reads(rfa, 0, 30000, 120, 2, 32)
    args = ["-k", "31", "-m", "2", "-t", "3"]
    o1, p1, n1 = run_cli(B.CLI_PATH, ["-r", rfa, "-g", ufa] + args)
    ufifo, rfifo = str(tmp_path / "u.fifo"), str(tmp_path / "r.fifo")
    os.mkfifo(ufifo)
    os.mkfifo(rfifo)
    def feed(src, dst):
        with open(dst, "wb") as o, open(src, "rb") as i:
            o.write(i.read())
    ts = [threading.Thread(target=feed, args=(ufa, ufifo)), threading.Thread(target=feed, args=(rfa, rfifo))]
    for t in ts:
        t.start()
    o2, p2, n2 = run_cli(B.CLI_PATH, ["-r", rfifo, "-g", ufifo] + args)
    for t in ts:
        t.join()
    from util import parse_counters
    assert parse_counters(o1) == parse_counters(o2) and parse_counters(o1)["aligned"] > 20000
    assert p1 == p2 and n1 == n2


@pytest.mark.parametrize("seed,k,L,m,partial,nfrac,d,alleles", [
    (1, 31, 150, 2, False, 0.0, 75, 2), (2, 31, 250, 5, False, 0.0, 40, 4), (3, 31, 200, 5, True, 0.002, 45, 4),
    (4, 21, 120, 3, False, 0.004, 50, 3), (5, 8, 60, 4, False, 0.0, 20, 4), (6, 31, 150, 0, True, 0.0, 140, 2), (7, 32, 100, 6, False, 0.01, 60, 4)])
@pytest.mark.parametrize("search", ["depth-first", "by-level"])
def test_gpu_exhaustive_matches_oracle_random(seed, k, L, m, partial, nfrac, d, alleles, search):
    """Both formulations of the exhaustive search (exh_search, exh_dp; the library picks one per graph and budget)."""
    s = Synth(100000, d, alleles, k, 9000 + seed)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, 8000, L, m + 1, 9500 + seed)
    if nfrac:
        reads = _inject_n(reads, np.random.default_rng(seed), nfrac)
    g = B.Graph.build(k, seqs, offs)
    al = B.Aligner(g, 0)
    al.set_knob(B.KNOB_EXH_SEARCH, B.SEARCH_BY_LEVEL if search == "by-level" else B.SEARCH_DEPTH_FIRST)
    o = oracle_py.Oracle(k, seqs, offs)
    p1, po1, st1 = al.align(reads, roffs, m=m, mode=B.MODE_EXHAUSTIVE, partial=partial)
    p2, po2, st2 = o.align(reads, roffs, m=m, mode=1, partial=partial)
    assert np.array_equal(st1, st2), np.nonzero(st1 != st2)[0][:10]
    assert np.array_equal(po1, po2)
    assert np.array_equal(p1, p2)
    assert al.counters() == o.counters()


@pytest.mark.parametrize("seed,k,L,m,d,alleles,nfrac", [(1, 31, 250, 5, 36, 4, 0.0), (2, 31, 150, 2, 140, 2, 0.001), (3, 15, 120, 4, 25, 3, 0.0),
                                                      (4, 31, 440, 6, 60, 4, 0.0), (5, 9, 80, 3, 12, 4, 0.0), (6, 32, 100, 0, 50, 2, 0.0)])
def test_exhaustive_four_reads_per_wave_pass_equals_level_search_and_oracle(seed, k, L, m, d, alleles, nfrac):
    """-b maps with bgr_align_exhaustive4_kernel first (four reads per wave; one node per level, first anchor) and leaves the
    rest -- levels with two nodes, failing first anchors, N, long walks (small k) -- to the level / depth-first passes.  With and
    without that first pass, and the oracle, must agree row for row, counters included."""
    s = Synth(150000, d, alleles, k, 8100 + seed)
    seqs, offs = s.unitigs()
    n = 12003 - seed
    reads, roffs = s.reads(0, n, L, m + 1, 8200 + seed)
    if nfrac:
        reads = _inject_n(reads, np.random.default_rng(seed), nfrac)
    g = B.Graph.build(k, seqs, offs)
    al = B.Aligner(g, 0)
    o = oracle_py.Oracle(k, seqs, offs)
    p2, po2, st2 = o.align(reads, roffs, m=m, mode=1)
    p1, po1, st1 = al.align(reads, roffs, m=m, mode=B.MODE_EXHAUSTIVE)
    assert al.launch_info()["four_reads_per_wave"]
    left = al.pass_counts()[2]
    assert np.array_equal(st1, st2), np.nonzero(st1 != st2)[0][:10]
    assert np.array_equal(po1, po2) and np.array_equal(p1, p2)
    assert al.counters() == o.counters()
    if k == 31 and not nfrac:
        assert left < 0.2 * n, (left, n)   # the fast pass settles most reads on graphs of isolated bubbles
    al.reset_counters()
    al.set_knob(B.KNOB_EXH_FAST, 1)
    p3, po3, st3 = al.align(reads, roffs, m=m, mode=B.MODE_EXHAUSTIVE)
    assert not al.launch_info()["four_reads_per_wave"]
    assert np.array_equal(st3, st2) and np.array_equal(po3, po2) and np.array_equal(p3, p2)
    assert al.counters() == o.counters()


@pytest.mark.parametrize("cap", ["2", "3", "5"])
@pytest.mark.parametrize("search", ["depth-first", "by-level"])
def test_exhaustive_deep_stack_second_pass(cap, search):
    """Pass 1 of the exhaustive kernel has a shallow stack (depth-first search) or few levels (level search); reads that
    need more go through pass 2.  With a tiny cap nearly every mapped read takes the second pass; results must not change."""
    s = Synth(60000, 30, 4, 12, 606)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, 6000, 120, 4, 607)
    g = B.Graph.build(12, seqs, offs)
    al = B.Aligner(g, 0)
    al.set_knob(B.KNOB_EXH_FRAME_CAP, int(cap))
    al.set_knob(B.KNOB_EXH_SEARCH, B.SEARCH_BY_LEVEL if search == "by-level" else B.SEARCH_DEPTH_FIRST)
    o = oracle_py.Oracle(12, seqs, offs)
    p1, po1, st1 = al.align(reads, roffs, m=3, mode=B.MODE_EXHAUSTIVE)
    p2, po2, st2 = o.align(reads, roffs, m=3, mode=1)
    assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)
    assert al.counters() == o.counters()
    # golden degenerate graph (unitigs of 5..14 bases, walks of many steps)
    case = next(c for c in EXH if c["args"][1] == "deg_reads.fa" and c["args"][7] == "5" and "-i" not in c["args"])
    g2 = B.Graph.from_fasta(os.path.join(GOLD, "deg_unitig.fa"), 5)
    a2 = B.Aligner(g2, 0)
    reads, roffs, heads, hoffs = B.load_reads(os.path.join(GOLD, "deg_reads.fa"), 5)
    paths, poffs, status = a2.align(reads, roffs, m=5, mode=B.MODE_EXHAUSTIVE)
    pb, nb = _format(reads, roffs, heads, hoffs, paths, poffs)
    from util import sha
    assert sha(pb) == case["paths_sha256"] and sha(nb) == case["notaligned_sha256"]


def _inject_n(reads, rng, frac):
    reads = reads.copy()
    idx = rng.random(reads.size) < frac
    reads[idx] = ord("N")
    return reads


@pytest.mark.parametrize("seed,k,L,m,e,nfrac,d,alleles", [
    (1, 31, 150, 2, 2, 0.0, 75, 2), (2, 31, 100, 2, 2, 0.0, 75, 2), (3, 31, 250, 5, 4, 0.0, 60, 3), (4, 21, 120, 3, 3, 0.002, 50, 2),
    (5, 32, 150, 2, 1, 0.0, 140, 2), (6, 31, 150, 0, 2, 0.001, 75, 2), (7, 31, 150, 2, 2, 0.01, 400, 2), (8, 15, 80, 4, 8, 0.005, 40, 4),
    (9, 31, 33, 2, 2, 0.0, 75, 2), (10, 31, 1000, 8, 2, 0.0005, 75, 2), (11, 8, 60, 3, 1000, 0.0, 20, 4),
    (12, 31, 150, 2, 0, 0.002, 75, 2), (13, 5, 40, 2, 2, 0.01, 12, 3), (14, 4, 30, 1, 3, 0.0, 9, 2), (15, 31, 150, 200, 2, 0.001, 75, 2),
    (16, 3, 20, 1, 2, 0.0, 8, 2)])
def test_gpu_matches_oracle_random(seed, k, L, m, e, nfrac, d, alleles):
    s = Synth(120000, d, alleles, k, 7000 + seed)
    seqs, offs = s.unitigs()
    n = 20000 if L <= 250 else 3000
    reads, roffs = s.reads(0, n, L, m + 1, 8000 + seed)
    if nfrac:
        reads = _inject_n(reads, np.random.default_rng(seed), nfrac)
    g = B.Graph.build(k, seqs, offs)
    al = B.Aligner(g, 0)
    o = oracle_py.Oracle(k, seqs, offs)
    p1, po1, st1 = al.align(reads, roffs, m=m, effort=e)
    p2, po2, st2 = o.align(reads, roffs, m=m, effort=e)
    assert np.array_equal(st1, st2), np.nonzero(st1 != st2)[0][:10]
    assert np.array_equal(po1, po2)
    assert np.array_equal(p1, p2)
    assert al.counters() == {**o.counters(), "overlaps": 0}


@pytest.mark.parametrize("seed,k,L,m,e,nfrac,d,alleles", [
    (1, 31, 150, 2, 2, 0.0, 140, 2), (2, 31, 100, 2, 2, 0.003, 75, 2), (3, 21, 250, 5, 1, 0.0, 40, 4), (4, 8, 90, 3, 3, 0.0, 12, 3),
    (5, 31, 150, 0, 0, 0.0, 100, 2), (6, 32, 440, 4, 2, 0.001, 60, 4), (7, 12, 64, 1, 8, 0.0, 20, 2), (8, 31, 33, 2, 2, 0.0, 90, 2)])
def test_four_reads_per_wave_pass_equals_general_kernel_and_oracle(seed, k, L, m, e, nfrac, d, alleles):
    """Greedy mode maps with bgr_align_greedy_multi_kernel (eight reads per wave, 8 lanes each; three launches up the reference's
    retry ladder) and hands what that does not settle -- N reads, long paths (small k: many short unitigs) -- to the general
    kernel.  Both routes and the oracle must agree row for row, counters included; the batch sizes leave 0..7 reads in the
    last wave."""
    s = Synth(150000, d, alleles, k, 7100 + seed)
    seqs, offs = s.unitigs()
    n = 20003 - seed
    reads, roffs = s.reads(0, n, L, m + 1, 7200 + seed)
    if nfrac:
        reads = _inject_n(reads, np.random.default_rng(seed), nfrac)
    g = B.Graph.build(k, seqs, offs)
    al = B.Aligner(g, 0)
    o = oracle_py.Oracle(k, seqs, offs)
    p2, po2, st2 = o.align(reads, roffs, m=m, effort=e)
    p1, po1, st1 = al.align(reads, roffs, m=m, effort=e)
    assert al.launch_info()["four_reads_per_wave"]
    assert np.array_equal(st1, st2), np.nonzero(st1 != st2)[0][:10]
    assert np.array_equal(po1, po2) and np.array_equal(p1, p2)
    assert al.counters() == o.counters()
    al.reset_counters()
    al.set_knob(B.KNOB_GREEDY_FAST, 1)   # general kernel only
    p3, po3, st3 = al.align(reads, roffs, m=m, effort=e)
    assert not al.launch_info()["four_reads_per_wave"]
    assert np.array_equal(st3, st2) and np.array_equal(po3, po2) and np.array_equal(p3, p2)
    assert al.counters() == o.counters()


@pytest.mark.parametrize("seed,k,L,m,e,nfrac", [(1, 31, 150, 2, 2, 0.0), (2, 21, 100, 3, 1, 0.002), (3, 32, 250, 5, 4, 0.001), (4, 25, 60, 1, 3, 0.0), (5, 31, 40, 0, 0, 0.0),
                                                (6, 27, 400, 4, 6, 0.0), (7, 31, 31, 2, 2, 0.0), (8, 22, 150, 2, 2, 0.0)])
def test_minimizer_filter_in_front_of_the_key_table(seed, k, L, m, e, nfrac, monkeypatch):
    """The filter of large graphs (graph_layout.h bgr_mmx_*: block chosen by the (k-1)-mer's minimizer, worked out across the lanes of a scan
    with whole-wave DPP shifts, 65 - (k-16) positions per scan step) forced onto a small graph whose table is probed in memory: eight-reads-
    per-wave kernel and general kernel (N reads: per-key minimizer) against the oracle, and against the same graph without filter."""
    B.set_option("build_filter", 2)   # (put back by conftest.py's fixture)
    B.set_option("exh_filter", 1)     # (the default; spelled out: exhaustive mode goes through the filter as well)
    s = Synth(150000, 3 * k, 3, k, 9100 + seed)
    seqs, offs = s.unitigs()
    n = 20000
    reads, roffs = s.reads(0, n, L, m + 1, 9200 + seed)
    if nfrac:
        reads = _inject_n(reads, np.random.default_rng(seed), nfrac)
    g = B.Graph.build(k, seqs, offs)
    hdr = np.array(g.blob())[:4096].view(np.uint64)
    assert int(hdr[18]) & 0xFFFFFFFF == 2 and int(hdr[25]) >= 1024
    al = B.Aligner(g, 0)
    al.configure(lds_mphf=1)   # table not staged in LDS: the filter is in use
    o = oracle_py.Oracle(k, seqs, offs)
    p2, po2, st2 = o.align(reads, roffs, m=m, effort=e)
    p1, po1, st1 = al.align(reads, roffs, m=m, effort=e)
    assert not al.launch_info()["mphf_in_lds"] and al.launch_info()["four_reads_per_wave"]
    assert np.array_equal(st1, st2), np.nonzero(st1 != st2)[0][:10]
    assert np.array_equal(po1, po2) and np.array_equal(p1, p2)
    assert al.counters() == o.counters()
    assert ((st1 & 3) == B.ST_ALIGNED).sum() > n // 100
    al.reset_counters()
    al.set_knob(B.KNOB_GREEDY_FAST, 1)   # general kernel only
    p3, po3, st3 = al.align(reads, roffs, m=m, effort=e)
    assert np.array_equal(st3, st2) and np.array_equal(po3, po2) and np.array_equal(p3, p2)
    assert al.counters() == o.counters()
    # exhaustive mode: the eight-reads-per-wave pass, then the depth-first and the level search alone
    al.set_knob(B.KNOB_GREEDY_FAST, 0)
    sub_r, sub_o = reads[: int(roffs[3000])], roffs[:3001]
    p5, po5, st5 = o.align(sub_r, sub_o, m=m, effort=e, mode=1)
    for fast_off, search in ((0, B.SEARCH_AUTO), (1, B.SEARCH_DEPTH_FIRST), (1, B.SEARCH_BY_LEVEL)):
        al.set_knob(B.KNOB_EXH_FAST, fast_off)
        al.set_knob(B.KNOB_EXH_SEARCH, search)
        p4, po4, st4 = al.align(sub_r, sub_o, m=m, effort=e, mode=B.MODE_EXHAUSTIVE)
        assert np.array_equal(st4, st5) and np.array_equal(po4, po5) and np.array_equal(p4, p5), (fast_off, search)


@pytest.mark.parametrize("mode", [B.MODE_GREEDY, B.MODE_EXHAUSTIVE])
def test_large_batch_in_overlapped_pieces_equals_one_launch(mode):
    """bgr_align_batch maps a batch of >= 512 k reads in four pieces on two streams (copies of one piece under the kernels of
    the other): rows, offsets and counters must be those of one launch over the whole batch (BGR_KNOB_BATCH_OVERLAP = 1), ragged
    read lengths, and a paths buffer that is too small must be reported as such."""
    k = 31
    s = Synth(300000, 90, 2, k, 5151)
    seqs, offs = s.unitigs()
    n = 600001
    reads, roffs = s.reads(0, n, 100, 2, 5252)
    keep = np.ones(len(reads), dtype=bool)   # ragged: cut a few bases off every third read
    lens = np.full(n, 100, dtype=np.int64)
    lens[::3] -= np.arange(len(lens[::3])) % 37
    idx = np.repeat(np.arange(n, dtype=np.int64) * 100, 100) + np.tile(np.arange(100, dtype=np.int64), n)
    keep = (np.tile(np.arange(100, dtype=np.int64), n) < np.repeat(lens, 100))
    reads = reads[keep]
    roffs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    g = B.Graph.build(k, seqs, offs)
    al = B.Aligner(g, 0)
    p1, po1, st1 = al.align(reads, roffs, m=2, mode=mode)
    c1 = al.counters()
    al.reset_counters()
    al.set_knob(B.KNOB_BATCH_OVERLAP, 1)
    p2, po2, st2 = al.align(reads, roffs, m=2, mode=mode)
    assert al.counters() == c1
    assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)
    o = oracle_py.Oracle(k, seqs, offs)
    sub = slice(299990, 300020)            # rows around a piece boundary against the oracle
    rs = reads[int(roffs[sub.start]):int(roffs[sub.stop])]
    ro = (roffs[sub.start:sub.stop + 1] - roffs[sub.start]).astype(np.uint64)
    p3, po3, st3 = o.align(rs, ro, m=2, mode=1 if mode == B.MODE_EXHAUSTIVE else 0)
    assert np.array_equal(st1[sub], st3)
    assert np.array_equal(p1[int(po1[sub.start]):int(po1[sub.stop])], p3)
    if mode == B.MODE_GREEDY:   # a paths buffer that cannot hold the result: reported, not overrun
        import ctypes as C
        al.set_knob(B.KNOB_BATCH_OVERLAP, 0)
        small = np.empty(1000, dtype=np.int32)
        poffs = np.empty(n + 1, dtype=np.uint64)
        status = np.empty(n, dtype=np.uint8)
        prm = B.Params(mode, 2, 2, 0)
        rc = B.lib().bgr_align_batch(al.h, C.byref(prm), reads.ctypes.data, roffs.ctypes.data, n, small.ctypes.data, len(small), poffs.ctypes.data, status.ctypes.data)
        assert rc != 0 and b"too small" in B.lib().bgr_last_error()


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 9, 15, 17, 63, 65])
def test_tiny_batches_through_the_many_reads_per_wave_kernels(n):
    """A single, partly filled wave (fewer reads than a wave takes, a last group alone) in all three modes."""
    k = 31
    s = Synth(60000, 70, 2, k, 4242)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, n, 150, 3, 4300 + n)
    g = B.Graph.build(k, seqs, offs, anchors=True)
    al = B.Aligner(g, 0)
    o = oracle_py.Oracle(k, seqs, offs, anchors=True)
    for mode in (B.MODE_GREEDY, B.MODE_EXHAUSTIVE, B.MODE_ANCHORS):
        p1, po1, st1 = al.align(reads, roffs, m=2, mode=mode)
        p2, po2, st2 = o.align(reads, roffs, m=2, mode={B.MODE_GREEDY: 0, B.MODE_EXHAUSTIVE: 1, B.MODE_ANCHORS: 2}[mode])
        assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2), mode


@pytest.mark.parametrize("mode,m", [(0, 2), (1, 3), (2, 2)])
def test_host_packed_batch_equals_ascii_batch(mode, m):
    """bgr_align_batch_packed (reads packed to 2 bits on the host, N-mask words as a sparse list scattered on the device) gives
    the rows of bgr_align_batch (ASCII in, packed by the pre-pass kernel) in every mode; ragged lengths, N in 1 % of the reads."""
    k = 31
    s = Synth(200000, 70, 3, k, 4100 + mode)
    seqs, offs = s.unitigs()
    g = B.Graph.build(k, seqs, offs, anchors=(mode == 2))
    al = B.Aligner(g, 0)
    rng = np.random.default_rng(5 + mode)
    full, foffs = s.reads(0, 30001, 180, m, 4200 + mode)
    lens = rng.integers(33, 181, size=30001)     # ragged: cut every read
    keep = np.concatenate([full[int(foffs[i]):int(foffs[i]) + int(lens[i])] for i in range(len(lens))])
    roffs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    reads = _inject_n(keep, rng, 0.0005)
    p1, po1, st1 = al.align(reads, roffs, m=m, effort=2, mode=mode)
    c1 = al.counters()
    al.reset_counters()
    pk = B.pack_reads(reads, roffs)
    assert len(pk["nm_index"]) > 0
    p2, po2, st2 = al.align_packed(pk, m=m, effort=2, mode=mode)
    assert np.array_equal(st1, st2), np.nonzero(st1 != st2)[0][:10]
    assert np.array_equal(po1, po2) and np.array_equal(p1, p2) and al.counters() == c1


def test_devices_init_on_one_gpu():
    """bgr_devices_init (graph resident on the GPUs of one process: one upload, then RCCL broadcast / xGMI peer copies).  A
    one-GPU box can only check the single-device paths: nothing to distribute, the forced RCCL path through a one-rank
    communicator (run-time lookup of librccl, ncclCommInitAll, in-place ncclBroadcast), argument errors."""
    s = Synth(120000, 90, 2, 31, 17)
    seqs, offs = s.unitigs()
    g = B.Graph.build(31, seqs, offs)
    assert g.devices_init(0, 1) == 0 and g.device_blob(0)
    assert g.devices_init(0, 1, how=2) == 0
    assert g.devices_init(0, 1, how=1) == 1       # BGR_FANOUT_RCCL
    with pytest.raises(B.BgrError):
        g.devices_init(0, B.device_count() + 1)
    with pytest.raises(B.BgrError):
        g.devices_init(0, 1, how=7)
    al = B.Aligner(g, 0)
    reads, roffs = s.reads(0, 3000, 150, 2, 18)
    o = oracle_py.Oracle(31, seqs, offs)
    p1, po1, st1 = al.align(reads, roffs)
    p2, po2, st2 = o.align(reads, roffs)
    assert np.array_equal(p1, p2) and np.array_equal(po1, po2) and np.array_equal(st1, st2)   # the blob survived the in-place broadcast


@pytest.mark.parametrize("mode,m", [(0, 2), (1, 3), (2, 2)])
def test_batch_of_mixed_lengths_keeps_the_fast_pass(mode, m):
    """A few long reads in a batch of short ones: the four-reads-per-wave kernels still take the short reads (one lane per
    word: < 480 bases) and list the long ones for the one-read-per-wave kernels.  Rows equal the oracle's."""
    k = 31
    s = Synth(300000, 80, 2, k, 5100 + mode)
    seqs, offs = s.unitigs()
    g = B.Graph.build(k, seqs, offs, anchors=(mode == 2))
    al = B.Aligner(g, 0)
    o = oracle_py.Oracle(k, seqs, offs, anchors=(mode == 2))
    short, soffs = s.reads(0, 6000, 150, m, 5200 + mode)
    long_, loffs = s.reads(6000, 40, 2500, m, 5300 + mode)
    # interleave: a long read after every 150 short ones
    parts, lens = [], []
    li = 0
    for i in range(6000):
        parts.append(short[i * 150:(i + 1) * 150]); lens.append(150)
        if i % 150 == 149 and li < 40:
            parts.append(long_[li * 2500:(li + 1) * 2500]); lens.append(2500); li += 1
    reads = np.concatenate(parts)
    roffs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    p1, po1, st1 = al.align(reads, roffs, m=m, effort=2, mode=mode)
    assert al.launch_info()["four_reads_per_wave"]
    listed = al.pass_counts()
    p2, po2, st2 = o.align(reads, roffs, m=m, effort=2, mode=mode)
    assert np.array_equal(st1, st2), np.nonzero(st1 != st2)[0][:10]
    assert np.array_equal(po1, po2) and np.array_equal(p1, p2)
    if mode:
        assert listed[2] >= li   # the long reads went on the list of the first pass


def test_asynchronous_tickets_double_buffer_on_one_thread():
    """bgr_align_batch_begin / _wait: one host thread keeps two aligners busy (begin A, begin B, wait A, begin A, wait B ...); every
    batch's rows, offsets and status equal the blocking call's, counters add up, stale tickets and a second begin are refused."""
    s = Synth(200000, 100, 2, 31, 911)
    seqs, offs = s.unitigs()
    g = B.Graph.build(31, seqs, offs)
    a1, a2, ref = B.Aligner(g, 0), B.Aligner(g, 0), B.Aligner(g, 0)
    batches = [s.reads(i * 30000, 30000 - 7 * i, 150, 3, 912) for i in range(6)]
    want = [ref.align(r, o, m=2, effort=2) for r, o in batches]
    als = [a1, a2]
    tickets = [None, None]
    got = [None] * len(batches)
    for i, (r, o) in enumerate(batches):
        j = i & 1
        if tickets[j] is not None:
            got[tickets[j][1]] = als[j].align_wait(tickets[j][0])
        tickets[j] = (als[j].align_begin(r, o, m=2, effort=2), i)
        if i == 2:
            with pytest.raises(B.BgrError):
                als[j].align_begin(r, o)      # one batch in flight per aligner
    for j in range(2):
        als[j].align_test(tickets[j][0])
        got[tickets[j][1]] = als[j].align_wait(tickets[j][0])
        with pytest.raises(B.BgrError):
            als[j].align_wait(tickets[j][0])  # a ticket is waited for once
    for (p1, po1, st1), (p2, po2, st2) in zip(got, want):
        assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)
    c1, c2, cr = a1.counters(), a2.counters(), ref.counters()
    assert {k: c1[k] + c2[k] for k in cr} == cr
    # an empty batch
    t = a1.align_begin(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    p, po, st = a1.align_wait(t)
    assert len(p) == 0 and list(po) == [0]


def test_ragged_and_empty_batches():
    s = Synth(60000, 75, 2, 31, 77)
    seqs, offs = s.unitigs()
    g = B.Graph.build(31, seqs, offs)
    al = B.Aligner(g, 0)
    o = oracle_py.Oracle(31, seqs, offs)
    # empty batch
    p, po, st = al.align(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert len(p) == 0 and len(st) == 0 and list(po) == [0]
    # ragged lengths 32..400 in one batch (the FASTA parser never hands over reads of length <= k)
    rng = np.random.default_rng(3)
    chunks, lens = [], []
    for i in range(3000):
        L = int(rng.integers(32, 400))
        r, _ = s.reads(i, 1, L, 2, 99)
        chunks.append(r)
        lens.append(L)
    reads = np.concatenate(chunks)
    roffs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    p1, po1, st1 = al.align(reads, roffs)
    p2, po2, st2 = o.align(reads, roffs)
    assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)


@pytest.mark.parametrize("stage,gamma", [(1, 0.0), (2, 0.0), (1, 1.8), (2, 1.8)])
def test_mphf_in_lds_and_in_hbm_agree(stage, gamma):
    """Key table staged in LDS (both buckets read at once) or probed in L2 (second bucket only when the first is full),
    at the tight fill of small graphs and the sparse one of large graphs."""
    s = Synth(200000, 75, 2, 31, 55)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, 30000, 150, 3, 56)
    g = B.Graph.build(31, seqs, offs, gamma)
    al = B.Aligner(g, 0)
    al.configure(lds_mphf=stage)
    o = oracle_py.Oracle(31, seqs, offs)
    p1, po1, st1 = al.align(reads, roffs)
    assert al.launch_info()["mphf_in_lds"] == (stage == 2)
    p2, po2, st2 = o.align(reads, roffs)
    assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)


@pytest.mark.parametrize("mode", [B.MODE_GREEDY, B.MODE_EXHAUSTIVE])
@pytest.mark.parametrize("stage", [1, 2, 3])
def test_keys_in_the_fallback_list_are_found(mode, stage, monkeypatch):
    """BUILD_NO_EVICTIONS (test hook) leaves the keys whose two buckets were full in the sorted fallback list: the kernels'
    bisection path, which ordinary graphs never take.  stage 3 = table in memory behind the (forced) minimizer filter, where the
    lanes the filter lets through compare their bucket's four keys directly."""
    if stage == 3:
        B.set_option("build_filter", 2)   # (put back by conftest.py's fixture)
        B.set_option("exh_filter", 1)
        stage = 1
    s = Synth(120000, 60, 3, 31, 91)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, 12000, 120, 3, 92)
    g = B.Graph.build(31, seqs, offs, 1.07, no_evictions=True)
    assert g.info()["n_fallback"] > 100
    al = B.Aligner(g, 0)
    al.configure(lds_mphf=stage)
    o = oracle_py.Oracle(31, seqs, offs)
    m = 2 if mode == B.MODE_GREEDY else 4
    p1, po1, st1 = al.align(reads, roffs, m=m, mode=mode)
    p2, po2, st2 = o.align(reads, roffs, m=m, mode=1 if mode == B.MODE_EXHAUSTIVE else 0)
    assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)


def test_blob_adopted_from_device_memory_maps_identically():
    """The multi-GPU path: a rank that received the blob bytes in its HBM (RCCL broadcast) adopts them in place."""
    import torch
    s = Synth(80000, 75, 2, 31, 21)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, 5000, 150, 2, 22)
    g = B.Graph.build(31, seqs, offs)
    blob = torch.from_numpy(np.array(g.blob())).to("cuda:0")
    torch.cuda.synchronize()
    g2 = B.Graph.adopt_device_blob(0, blob.data_ptr(), blob.numel())
    a1, a2 = B.Aligner(g, 0), B.Aligner(g2, 0)
    r1, r2 = a1.align(reads, roffs), a2.align(reads, roffs)
    for x, y in zip(r1, r2):
        assert np.array_equal(x, y)


def test_full_size_batch_properties():
    """At the benchmark's batch size (the oracle would need minutes): size-independent properties.
    (1) determinism: the same batch twice gives identical rows; (2) sharding invariance: mapping the two halves
    separately and concatenating equals mapping the whole (what the multi-GPU split relies on); (3) counters add up;
    (4) a seeded sample of rows equals the oracle's."""
    s = Synth(4_600_000, 140, 2, 31, 20261003)
    seqs, offs = s.unitigs()
    n, L = 2_000_000, 150
    reads, roffs = s.reads(0, n, L, 2, 77, threads=16)
    g = B.Graph.build(31, seqs, offs)
    al = B.Aligner(g, 0)
    al.reset_counters()
    p1, po1, st1 = al.align(reads, roffs)
    c1 = al.counters()
    p2, po2, st2 = al.align(reads, roffs)
    assert np.array_equal(p1, p2) and np.array_equal(po1, po2) and np.array_equal(st1, st2)
    h = n // 2 + 12345
    pa, poa, sta = al.align(reads[: h * L], roffs[: h + 1])
    pb, pob, stb = al.align(reads[h * L:], roffs[h:] - roffs[h])
    assert np.array_equal(np.concatenate([pa, pb]), p1)
    assert np.array_equal(np.concatenate([sta, stb]), st1)
    assert np.array_equal(np.concatenate([poa, pob[1:] + poa[-1]]), po1)
    assert c1["reads"] == n and c1["reads"] == c1["no_overlap"] + c1["aligned"] + c1["not_aligned"]
    assert c1["aligned"] == int(((st1 & 3) == 2).sum()) and c1["no_overlap"] == int(((st1 & 3) == 0).sum())
    rng = np.random.default_rng(5)
    idx = np.sort(rng.choice(n, 5000, replace=False))
    sub = np.concatenate([reads[i * L:(i + 1) * L] for i in idx])
    o = oracle_py.Oracle(31, seqs, offs)
    ps, pos, sts = o.align(sub, np.arange(len(idx) + 1, dtype=np.uint64) * L)
    for j, i in enumerate(idx):
        assert st1[i] == sts[j]
        assert np.array_equal(p1[int(po1[i]): int(po1[i + 1])], ps[int(pos[j]): int(pos[j + 1])])


def test_configs1_at_its_stated_size_equals_the_oracle():
    """BASELINE configs[1] as stated -- 1 M x 100 bp reads, k=31, m=2, ~10 k-unitig graph (250 kb genome, a 2-allele site every ~75 bp:
    SURVEY 8d config 2), greedy -- one launch, EVERY row against the oracle (sixteen oracle instances on host threads), plus the counters."""
    from concurrent.futures import ThreadPoolExecutor
    s = Synth(250_000, 75, 2, 31, 20261003)
    seqs, offs = s.unitigs()
    assert 9_000 < len(offs) - 1 < 12_000
    n, L = 1_000_000, 100
    reads, roffs = s.reads(0, n, L, 2, 77, threads=16)
    g = B.Graph.build(31, seqs, offs)
    al = B.Aligner(g, 0)
    al.reset_counters()
    p1, po1, st1 = al.align(reads, roffs, m=2, effort=2)
    c1 = al.counters()
    parts = 16
    cuts = [n * i // parts for i in range(parts + 1)]

    def part(i):
        o = oracle_py.Oracle(31, seqs, offs)
        lo, hi = cuts[i], cuts[i + 1]
        return o.align(reads[lo * L: hi * L], np.arange(hi - lo + 1, dtype=np.uint64) * L, m=2, effort=2), o.counters()

    with ThreadPoolExecutor(parts) as ex:
        res = list(ex.map(part, range(parts)))
    tot = {k: 0 for k in res[0][1]}
    for i, ((p2, po2, st2), c2) in enumerate(res):
        lo, hi = cuts[i], cuts[i + 1]
        assert np.array_equal(st1[lo:hi], st2)
        assert np.array_equal(po1[lo: hi + 1] - po1[lo], po2)
        assert np.array_equal(p1[int(po1[lo]): int(po1[hi])], p2)
        for k in tot:
            tot[k] += c2[k]
    for k in ("reads", "no_overlap", "aligned", "not_aligned"):
        assert c1[k] == tot[k], (k, c1, tot)
    assert c1["reads"] == n and c1["aligned"] > 0.8 * n


def test_oversized_batch_is_mapped_in_pieces():
    """bgr_align_batch splits a batch whose path arena would not fit 32-bit addressing; with the limit lowered the split
    path runs on a small batch and must give the rows of the unsplit call."""
    s = Synth(300000, 90, 2, 31, 4)
    seqs, offs = s.unitigs()
    al = B.Aligner(B.Graph.build(31, seqs, offs), 0)
    reads, roffs = s.reads(0, 50000, 150, 2, 5)
    p1, po1, st1 = al.align(reads, roffs)
    al.reset_counters()
    al.set_knob(B.KNOB_BATCH_SPLIT_LIMIT, 3_000_000)   # pieces of ~4 600 reads
    p2, po2, st2 = al.align(reads, roffs)
    assert np.array_equal(p1, p2) and np.array_equal(po1, po2) and np.array_equal(st1, st2)
    assert al.counters()["reads"] == 50000
    al.set_knob(B.KNOB_BATCH_SPLIT_LIMIT, 4096)        # one read per piece
    p3, po3, st3 = al.align(reads[: 300 * 150], roffs[:301])
    assert np.array_equal(p3, p1[: int(po1[300])]) and np.array_equal(po3, po1[:301]) and np.array_equal(st3, st1[:300])


def test_tiny_and_very_long_reads_through_the_api():
    """Lengths the parsers never hand over (shorter than k-1) and reads of several kb: no crash, oracle-identical."""
    s = Synth(300000, 90, 2, 31, 909)
    seqs, offs = s.unitigs()
    g = B.Graph.build(31, seqs, offs)
    al = B.Aligner(g, 0)
    o = oracle_py.Oracle(31, seqs, offs)
    chunks, lens = [], []
    for i, L in enumerate([30, 31, 32, 40, 3000, 64, 8000, 33, 257, 512, 513, 1024, 20000]):
        r, _ = s.reads(i, 1, L, 3, 910)
        chunks.append(r)
        lens.append(L)
    reads = np.concatenate(chunks)
    roffs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    p1, po1, st1 = al.align(reads, roffs, m=6)
    p2, po2, st2 = o.align(reads, roffs, m=6)
    assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)
    # exhaustive: pass 1 keeps a shallow search stack in LDS, pass 2 the worst-case state in HBM; a batch whose longest
    # read does not even fit pass 1's LDS layout (the 20 kb one) goes through the pass-2 kernel only.  m stays small:
    # the search is exponential in m on long reads (m=6 on 20 kb does not finish on the CPU reference either)
    for limit in (1024, 8000, 20000):
        keep = [i for i, L in enumerate(lens) if L <= limit]
        reads_e = np.concatenate([chunks[i] for i in keep])
        roffs_e = np.concatenate([[0], np.cumsum([lens[i] for i in keep])]).astype(np.uint64)
        p1, po1, st1 = al.align(reads_e, roffs_e, m=3, mode=B.MODE_EXHAUSTIVE)
        p2, po2, st2 = o.align(reads_e, roffs_e, m=3, mode=1)
        assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2), limit
        assert all((st1[j] & 3) == B.ST_ALIGNED for j, i in enumerate(keep) if lens[i] >= 257)   # not vacuous
    # shorter than k-1: the reference reads out of range here; the library reports "not mapped" instead of crashing
    short = np.frombuffer(b"ACGTACGTACGTAC" + b"ACG" + b"A", dtype=np.uint8)
    so = np.array([0, 14, 17, 18, 18], dtype=np.uint64)
    p, po, st = al.align(short, so)
    assert len(p) == 0 and list(st & 3) == [0, 0, 0, 0]
    p, po, st = al.align(short, so, mode=B.MODE_EXHAUSTIVE)
    assert len(p) == 0 and list(st & 3) == [1, 1, 1, 1]


@pytest.mark.parametrize("nbytes,env", [(8 << 20, None), (3 << 20, None), (4096, None), (8 << 20, "1")])
def test_page_locked_host_buffers(nbytes, env, monkeypatch):
    """bgr_host_alloc: buffers of 2 MB and more come from huge-page mappings registered with the runtime, smaller ones (and all of them
    with the option huge_pinned = 0) from hipHostMalloc; either kind is writable, feeds a batch call and is freed by bgr_host_free."""
    import ctypes
    if env:
        B.set_option("huge_pinned", 0)   # (put back by conftest.py's fixture)
    L = B.lib()
    p = ctypes.c_void_p()
    B._check(L.bgr_host_alloc(nbytes, ctypes.byref(p)))
    assert p.value
    buf = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,))
    s = Synth(60000, 75, 2, 31, 5)
    seqs, offs = s.unitigs()
    n = min(2000, nbytes // 200)
    reads, roffs = s.reads(0, n, 150, 2, 6)
    buf[: len(reads)] = reads                      # the reads live in the page-locked buffer
    al = B.Aligner(B.Graph.build(31, seqs, offs), 0)
    p1 = al.align(buf[: len(reads)], roffs)
    p2 = al.align(reads, roffs)
    for x, y in zip(p1, p2):
        assert np.array_equal(x, y)
    del buf
    B._check(L.bgr_host_free(p))


def test_bench_line_contract():
    """bench.py (what the driver runs at round end) prints ONE JSON line carrying the contract's fields, the roofline and
    cpu_baseline objects, and a parity sample that matches the oracle -- here on a small workload, without the counter passes."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--reads-per-step", "200000", "--e2e-reads", "200000",
           "--pcie-steps", "1", "--cpu-sample", "20000", "--cpu-sample-all", "30000", "--alg-sample", "4000", "--no-pmc", "--no-sub", "--genome", "400000",
           "--full-parity", "--full-parity-scale", "0.002"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "Mreads/s" and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    # without the counter passes the line carries the HBM view only (own compulsory bytes of the dominant kernel / its duration):
    # a FRACTION of the peak, <= 1; the SURVEY 8d figure of the reference's control flow sits beside it under its own keys
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] is None
    assert 0 < r["frac"] <= 1 and 0 < r["hbm"]["frac"] <= 1 and 0 < r["hbm"]["whole_launch"]["frac"] <= 1
    assert r["reference_alg_bytes_per_read"] > 1000 and "dominant_kernel_ms" in r and r["dominant_kernel"].startswith("bgr_")
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "reference" and c["value"] > 0 and c["cores"] >= 1 and c["gpu_matches_cpu_records"] is True
    assert c["t1"]["value"] > 0 and c["t1"]["cores"] == 1 and c["cpu_model"]
    assert d["parity_sample"]["gpu_equals_oracle"] is True
    assert d["pcie_inclusive"]["value"] > 0 and d["e2e"]["value"] > 0
    # SURVEY 8d's two metrics as first-class values: (i) PCIe inclusive, one blocking caller; (ii) end to end = the MEDIAN of the runs
    assert d["value_pcie_inclusive"] == d["pcie_inclusive"]["value"] and d["value_e2e"] == d["e2e"]["value"]
    runs = d["e2e"]["runs_mreads_per_s"]
    assert len(runs) == 3 and min(runs) - 0.1 <= d["e2e"]["value"] <= max(runs) + 0.1 and d["e2e"]["worst"] <= d["e2e"]["value"] <= d["e2e"]["best"]
    if c.get("all_cores"):   # (only on hosts with more visible CPUs than --cpu-threads)
        assert c["all_cores"]["value"] > 0 and c["all_cores"]["cores"] > c["cores"]
    # round 5: the scalar copies the driver's record keeps (it drops nested objects), and --full-parity (tools/full_parity.py at 1/500 of the configs' sizes:
    # bin/bgreat against the compiled reference, sorted record multisets + counters)
    assert d["config"]["value_e2e"] == d["value_e2e"] and d["config"]["value_pcie_inclusive"] == d["value_pcie_inclusive"] and d["config"]["parity_sample_ok"] is True
    assert d["cpu_baseline"]["t1_value"] == c["t1"]["value"] and "hbm_compulsory_frac" in r
    fp = d["full_parity"]
    assert fp["equal"] is True and fp["c1_equal"] and fp["c2_equal"] and fp["c4_equal"] and fp["c2_reads"] == 100000 and d["config"]["full_parity_equal"] is True


def test_bench_sub_record_of_the_exhaustive_config():
    """The default bench line carries configs[1], [3] and [4] as sub-records (children `bench.py --workload W --sub-record`); here the
    exhaustive one on a small graph: one JSON line, the device-resident leg only, a parity sample, and the reference's -b counters on
    a bounded sample next to the GPU's."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "branchy", "--sub-record", "--steps", "2", "--warmup", "1", "--reads-per-step", "100000",
           "--genome", "300000", "--cpu-sample-exh", "5000", "--alg-sample", "2000", "--no-pmc"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["e2e"] is None and d["pcie_inclusive"] is None and d["other_configs"] is None
    assert "exhaustive" in d["config"]["workload"] and d["parity_sample"]["gpu_equals_oracle"] is True
    c = d["cpu_baseline"]
    assert "error" not in c, c
    assert c["kind"] == "reference" and c["value"] > 0 and c["reference_reads"] == 5000 and c["gpu_matches_cpu_counters"] is True


def test_bench_two_rank_rehearsal_on_one_gpu():
    """The N > 1 code path of bench.py as the driver launches it (torch.distributed.run, one rank per GPU), rehearsed with two ranks that
    share this box's one GPU and the gloo backend (BGR_BENCH_BACKEND=gloo; never a reported number): one JSON line from rank 0 with the
    whole-job value, the ranks that took part, the graph broadcast's bytes and time, the counters summed over the ranks, the end-to-end leg
    per rank plus the CLI's own forms in one process (ordered pair and split run: bytes identical), and the C-ABI's one-process form
    (bgr_devices_init + an aligner and a host thread per device)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, BGR_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads-per-step", "300000", "--e2e-reads", "400000",
           "--alg-sample", "3000", "--genome", "400000", "--no-pmc"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and d["parity_sample"]["gpu_equals_oracle"] is True
    assert d["counters"]["reads"] == 2 * 2 * 300000
    m = d["multi_gpu"]
    assert m["ranks_seen"] == 2 and m["graph_broadcast_bytes"] > 1000 and m["graph_broadcast_ms"] > 0
    o = d["one_process_all_gpus"]
    assert "error" not in o, o
    assert o["value"] > 0 and o["n_gpus"] >= 1 and sum(o["reads_mapped_per_device"]) == o["n_gpus"] * (o["steps"] + 1) * 300000
    e = d["e2e"]
    assert e["value"] > 0 and e["n_gpus"] == 2 and e["host_route"]["identical_bytes_to_the_text_route"] is True
    one = e["one_process_all_gpus"]
    assert "error" not in one, one
    assert one["identical_bytes_to_one_gpu"] is True and one["split_output"]["identical_bytes_concatenated"] is True
    assert d["cpu_baseline"] is None and d["pcie_inclusive"] is None and d["other_configs"] is None   # N = 1 only


def test_bench_line_comes_out_when_an_optional_leg_hangs():
    """bench.py's LineGuard: the N > 1 legs behind the timed region drive collectives between real devices that have only ever been
    rehearsed on one GPU; a leg that never returns (test hook: the one-process form sleeps for ever) must cost the line its optional
    fields, not the line -- rank 0 prints what it holds when the deadline passes (marked `incomplete`), every rank leaves with exit
    code 0 and torch.distributed.run reports success."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, BGR_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", BGR_BENCH_TEST_HANG="one_process_all_gpus", BGR_BENCH_DEADLINE="25")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29519",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads-per-step", "200000", "--e2e-reads", "0",
           "--alg-sample", "2000", "--genome", "400000", "--no-pmc"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["parity_sample"]["gpu_equals_oracle"] is True and d["counters"]["reads"] == 2 * 2 * 200000
    assert "one_process_all_gpus" in d["incomplete"] and d["one_process_all_gpus"] is None and d["roofline"]["dominant_kernel"].startswith("bgr_")


@pytest.mark.parametrize("mode", ["anchors", "large_table"])
def test_forced_staging_that_cannot_be_had_runs_without(mode):
    """bgr_aligner_configure(lds_mphf = 2) asks for the key table in LDS.  Anchors mode has no key table to stage (it probes its own
    index), and a table beyond a CU's LDS cannot be staged: such launches run unstaged (launch_info says so) instead of failing with
    "read too long" (found by tools/fuzz_big_batches.py, anchors seed 13)."""
    if mode == "anchors":
        s = Synth(139510, 47, 3, 21, 3222)
        seqs, offs = s.unitigs()
        reads, roffs = s.reads(0, 20000, 100, 5, 7001)
        g = B.Graph.build(21, seqs, offs, 1.07, anchors=True)
        o = oracle_py.Oracle(21, seqs, offs, anchors=True)
        gm, om, m = B.MODE_ANCHORS, 2, 4
    else:
        s = Synth(14_000_000, 75, 2, 31, 77)      # ~200 k overlap keys: a table of more than 160 KB
        seqs, offs = s.unitigs()
        reads, roffs = s.reads(0, 20000, 150, 3, 78)
        g = B.Graph.build(31, seqs, offs, 1.07)
        assert g.info()["mphf_bytes"] > 170_000
        o = oracle_py.Oracle(31, seqs, offs)
        gm, om, m = B.MODE_GREEDY, 0, 2
    al = B.Aligner(g, 0)
    al.configure(lds_mphf=2)
    p1, po1, st1 = al.align(reads, roffs, m=m, effort=2, mode=gm)
    assert al.launch_info()["mphf_in_lds"] is False
    p2, po2, st2 = o.align(reads, roffs, m=m, effort=2, mode=om)
    assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)


def test_exhaustive_on_duplicated_kmers_comes_back_and_equals_the_reference(tmp_path):
    """A unitig set that duplicates its own k-mers (a homopolymer on both strands next to chains, fans, self-loops: tests/make_golden_soup.py) makes the
    reference's exhaustive recursion enumerate exponentially many identical walks -- 150 s for one 79-base read.  The device's depth-first passes bound
    their work per search and hand such a read to the last pass, whose level search over tables in HBM is polynomial: the run is back in seconds (it did
    not come back at all before), with the reference's bytes (found by tools/fuzz_soup.py)."""
    import time
    from util import run_cli
    args = ["-r", os.path.join(GOLD, "soup_polyA_reads.fa"), "-k", "31", "-g", os.path.join(GOLD, "soup_polyA_unitig.fa"), "-m", "1", "-e", "2", "-t", "4", "-b", "--write-exhaustive"]
    t0 = time.time()
    out, paths, na = run_cli(B.CLI_PATH, args, timeout=120)
    assert time.time() - t0 < 60
    assert paths == open(os.path.join(GOLD, "soup_polyA_expected_paths"), "rb").read()
    assert na == open(os.path.join(GOLD, "soup_polyA_expected_notAligned.fa"), "rb").read()
    from util import parse_counters
    assert parse_counters(out)["aligned"] == 141 and parse_counters(out)["reads"] == 400
    # the batch form, tiny search caps: the same rows through every pass
    us = [l.strip() for l in open(os.path.join(GOLD, "soup_polyA_unitig.fa")) if not l.startswith(">")]
    rd = [l.strip() for l in open(os.path.join(GOLD, "soup_polyA_reads.fa")) if not l.startswith(">")]
    seqs = np.frombuffer("".join(us).encode(), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(u) for u in us])]).astype(np.uint64)
    rb = np.frombuffer("".join(rd).encode(), dtype=np.uint8)
    roffs = np.concatenate([[0], np.cumsum([len(r) for r in rd])]).astype(np.uint64)
    g = B.Graph.build(31, seqs, offs, 1.8)
    rows = None
    for knobs in ({}, {B.KNOB_EXH_FRAME_CAP: 3}, {B.KNOB_EXH_FAST: 1}, {B.KNOB_EXH_FAST: 1, B.KNOB_EXH_SEARCH: 1}, {B.KNOB_EXH_FAST: 1, B.KNOB_EXH_SEARCH: 2, B.KNOB_EXH_FRAME_CAP: 3}):
        al = B.Aligner(g, 0)
        for kk, v in knobs.items():
            al.set_knob(kk, v)
        p, po, st = al.align(rb, roffs, m=1, effort=2, mode=B.MODE_EXHAUSTIVE)
        assert int(((st & 3) == 2).sum()) == 141, knobs
        if rows is None:
            rows = (p, po, st)
        else:
            assert np.array_equal(p, rows[0]) and np.array_equal(po, rows[1]) and np.array_equal(st, rows[2]), knobs
        al.close()


def test_exhaustive_last_pass_runs_again_with_a_larger_table():
    """The last exhaustive pass is the reference's recursion memoised on (overlap, position) (exh_memo): a read whose table of remembered calls fills up is
    handed back and run again by the host with a table sixteen times as large, until it fits -- never an error, never a row that differs.  Tiny search caps
    push every read of a 4-allele graph into that pass, a table of 8 entries per wave makes its first run hand most of them back."""
    s = Synth(60000, 40, 4, 31, 88)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, 3000, 250, 4, 89)
    g = B.Graph.build(31, seqs, offs)
    o = oracle_py.Oracle(31, seqs, offs)
    p2, po2, st2 = o.align(reads, roffs, m=4, mode=1)

    def run(memo_cap, partial=False):
        al = B.Aligner(g, 0)
        al.set_knob(B.KNOB_EXH_FAST, 1)
        al.set_knob(B.KNOB_EXH_FRAME_CAP, 3)
        al.set_knob(B.KNOB_EXH_MEMO_CAP, memo_cap)
        try:
            return al.align(reads, roffs, m=4, mode=B.MODE_EXHAUSTIVE, partial=partial), al.pass_counts(), al.last_pass_runs(), al.counters()
        finally:
            al.close()
    (p1, po1, st1), passes, (runs, cap), c = run(0)     # the table as shipped: every read through the last pass, once
    assert passes[1] > 1000 or passes[0] > 1000, passes
    assert runs == 1 and cap >= 1024
    assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)
    assert c["reads"] == 3000 and c["aligned"] == int(((st2 & 3) == 2).sum())
    (p1, po1, st1), _, (runs, cap), c = run(8)         # 8 entries: most reads come back for a second (128 entries) and third run
    assert runs >= 2 and cap >= 128, (runs, cap)
    assert np.array_equal(st1, st2) and np.array_equal(po1, po2) and np.array_equal(p1, p2)
    assert c["reads"] == 3000 and c["aligned"] == int(((st2 & 3) == 2).sum())   # (a read handed back is counted once, by the run that maps it)
    p3, po3, st3 = o.align(reads, roffs, m=4, mode=1, partial=True)             # -i
    (p1, po1, st1), _, (runs, cap), _ = run(8, partial=True)
    assert runs >= 2
    assert np.array_equal(st1, st3) and np.array_equal(po1, po3) and np.array_equal(p1, p3)


@pytest.mark.parametrize("m", [0, 2, 5])
def test_exhaustive_where_the_reference_is_exponential(m):
    """The hole round 4 left: on such unitig sets the device's last pass ran the reference's recursion and gave up with an error after 2^26 steps.  It now
    remembers its calls (exh_memo): every read comes back in milliseconds with the rows of the recursion -- checked against the oracle's literal form on
    the reads that one finishes (<= 32 bases) and against its remembered-calls form (pinned to the literal one and to the compiled reference by
    tools/fuzz_soup.py cpu and tests/test_oracle_golden.py) on reads of up to 250 bases, through every route into the last pass and with tables that
    start at 8 entries."""
    import time
    from util import homopolymer_soup
    k, (seqs, offs), (sr, so), (lr, lo) = homopolymer_soup()
    g = B.Graph.build(k, seqs, offs)
    o = oracle_py.Oracle(k, seqs, offs)
    exp_short = o.align(sr, so, m=m, mode=1)
    assert all(np.array_equal(a, b) for a, b in zip(exp_short, o.align(sr, so, m=m, mode=3)))
    exp_long = o.align(lr, lo, m=m, mode=3)
    for knobs in ({}, {B.KNOB_EXH_MEMO_CAP: 8}, {B.KNOB_EXH_FAST: 1, B.KNOB_EXH_SEARCH: 1}, {B.KNOB_EXH_FAST: 1, B.KNOB_EXH_SEARCH: 2, B.KNOB_EXH_FRAME_CAP: 3, B.KNOB_EXH_MEMO_CAP: 64}):
        al = B.Aligner(g, 0)
        for kk, v in knobs.items():
            al.set_knob(kk, v)
        t0 = time.time()
        got_s = al.align(sr, so, m=m, mode=B.MODE_EXHAUSTIVE)
        got_l = al.align(lr, lo, m=m, mode=B.MODE_EXHAUSTIVE)
        assert time.time() - t0 < 30, knobs
        assert all(np.array_equal(a, b) for a, b in zip(got_s, exp_short)), knobs
        assert all(np.array_equal(a, b) for a, b in zip(got_l, exp_long)), knobs
        al.close()


@pytest.mark.parametrize("mode", ["greedy", "exhaustive", "anchors"])
def test_staging_from_the_characters_equals_the_pre_pass_and_the_oracle(mode):
    """Round 5: a launch handed ASCII reads stages them inside the mapping kernels (ascii_word: v_perm table lookup, no planes); KNOB_GREEDY_PREPASS = 1
    keeps the pre-pass + planes of rounds 2-4.  Both against the oracle, in every mode, on reads with N (one-read-per-wave kernels through load_packed's
    ASCII form), mixed lengths (partial last words, a read too long for the several-reads-per-wave kernels) and the batch's last read (no byte behind it)."""
    k = 31
    s = Synth(200000, 70, 3, k, 515)
    seqs, offs = s.unitigs()
    rng = np.random.default_rng(5)
    base, _ = s.reads(0, 6000, 300, 3, 616)
    lens = rng.choice([32, 33, 63, 64, 65, 100, 128, 150, 151, 250, 300], size=6000)
    lens[-1] = 150
    lens[17] = 300
    reads = np.concatenate([base[i * 300: i * 300 + int(l)] for i, l in enumerate(lens)])
    roffs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    reads = _inject_n(reads, rng, 0.0008)
    anchors = mode == "anchors"
    g = B.Graph.build(k, seqs, offs, anchors=anchors)
    o = oracle_py.Oracle(k, seqs, offs, anchors=anchors)
    gm, om = {"greedy": (B.MODE_GREEDY, 0), "exhaustive": (B.MODE_EXHAUSTIVE, 1), "anchors": (B.MODE_ANCHORS, 2)}[mode]
    exp = o.align(reads, roffs, m=3, effort=2, mode=om)
    for prepass in (0, 1):
        al = B.Aligner(g, 0)
        al.set_knob(B.KNOB_GREEDY_PREPASS, prepass)
        got = al.align(reads, roffs, m=3, effort=2, mode=gm)
        names = [n for n, _ in al.kernel_times()[1]]
        assert any(n.startswith("bgr_pack_reads_kernel") for n in names) == bool(prepass or mode == "anchors"), names   # (anchors mode keeps its pre-pass: measured faster)
        assert all(np.array_equal(a, b) for a, b in zip(got, exp)), (mode, prepass)
        al.close()


def test_launches_without_a_several_reads_per_wave_pass_keep_the_pre_pass():
    """Staging from the characters pays behind the several-reads-per-wave passes; the one-read-per-wave kernels ALONE (knobs; -i; budgets beyond 254) are faster
    from planes, so such launches keep bgr_pack_reads_kernel -- same rows either way."""
    s = Synth(150000, 75, 2, 31, 717)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, 4000, 150, 2, 718)
    g = B.Graph.build(31, seqs, offs)
    o = oracle_py.Oracle(31, seqs, offs)
    for gm, om, knob, kw in ((B.MODE_GREEDY, 0, B.KNOB_GREEDY_FAST, {}), (B.MODE_EXHAUSTIVE, 1, B.KNOB_EXH_FAST, {}), (B.MODE_EXHAUSTIVE, 1, None, {"partial": True})):
        exp = o.align(reads, roffs, m=2, effort=2, mode=om, **kw)
        al = B.Aligner(g, 0)
        if knob is not None:
            al.set_knob(knob, 1)
        got = al.align(reads, roffs, m=2, effort=2, mode=gm, **kw)
        assert any(n.startswith("bgr_pack_reads_kernel") for n, _ in al.kernel_times()[1])
        assert all(np.array_equal(a, b) for a, b in zip(got, exp))
        al.close()


@pytest.mark.parametrize("k,L", [(31, 150), (32, 250), (31, 100), (27, 150)])
def test_minimizer_filter_scan_steps_of_64_positions(k, L):
    """k = 31 / 32: a scan step behind the minimizer filter covers 64 positions -- lanes 50.. take the 16-mers of positions 64.. out of the low halves of the
    windows of lanes 48.. (scan_mblock_wide); k = 27 keeps 65 - (k - 16) positions per step.  Long unitigs, so that a read's first overlap lies anywhere
    (also in the lanes that lack their right neighbours, and in the second and third step): the oracle's rows, and the rows of the same graph without filter."""
    B.set_option("build_filter", 2)   # (put back by conftest.py's fixture)
    s = Synth(600000, 230, 2, k, 9900 + k + L)
    seqs, offs = s.unitigs()
    n = 30000
    reads, roffs = s.reads(0, n, L, 2, 9950 + k)
    g = B.Graph.build(k, seqs, offs)
    al = B.Aligner(g, 0)
    al.configure(lds_mphf=1)   # table not staged in LDS: the filter is in use
    o = oracle_py.Oracle(k, seqs, offs)
    exp = o.align(reads, roffs, m=2, effort=3)
    got = al.align(reads, roffs, m=2, effort=3)
    assert not al.launch_info()["mphf_in_lds"] and al.launch_info()["four_reads_per_wave"]
    assert all(np.array_equal(a, b) for a, b in zip(got, exp))
    assert ((exp[2] & 3) == B.ST_ALIGNED).sum() > n // 10
    al.close()


@pytest.mark.parametrize("extra", [[], ["--host-route"], ["-c"]])
def test_cli_paths_into_a_fifo_and_into_a_file(extra, tmp_path):
    """Regular output files are written at known offsets by several workers (pwrite); a FIFO (or a pipe) gets the same bytes through ONE fwrite per stream,
    in order: `-f FIFO` read by another thread = the file of the plain run; 120 000 reads x 3 host threads, several pieces and slices."""
    import subprocess
    import threading
    s = Synth(150000, 90, 2, 31, 41)
    ufa, rfa = str(tmp_path / "u.fa"), str(tmp_path / "r.fa")
    s.write_unitigs(ufa)
    s.write_reads(rfa, 0, 120000, 120, 2, 42)
    args = ["-r", rfa, "-g", ufa, "-k", "31", "-m", "2", "-t", "3", "--batch", "20000"] + extra
    o1, p1, n1 = run_cli(B.CLI_PATH, args)
    assert len(p1) > 1_000_000
    fifo = str(tmp_path / "paths.fifo")
    os.mkfifo(fifo)
    got = []
    t = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()))
    t.start()
    wd = tmp_path / "wd"
    wd.mkdir()
    p = subprocess.run([B.CLI_PATH] + args + ["-f", fifo], cwd=str(wd), capture_output=True, text=True, timeout=600)
    t.join()
    assert p.returncode == 0, p.stderr[-2000:]
    assert got[0] == p1 and open(wd / "notAligned.fa", "rb").read() == n1
