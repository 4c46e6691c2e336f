"""The text form of the C-ABI (bgr_align_fasta_text, text_kernels.hip): a piece of a FASTA file in, the bytes of `paths` and
`notAligned.fa` out.  The device takes pieces of the regular shape (header line, one sequence line); every other piece must come
back flagged `irregular` with nothing mapped.  Expected bytes = the host route: the exact getReads parser (bgr_readset_load) +
bgr_align_batch (itself pinned to the oracle and the reference by test_gpu_parity.py) + the reference's record format."""
import os

import numpy as np
import pytest

import bgreat_amd as B
from tools.synth import Synth
from util import GOLD

pytestmark = pytest.mark.gpu


def _host_route(al, path_or_bytes, k, tmp_path, m=2, effort=2, mode=B.MODE_GREEDY, fastq=False):
    if isinstance(path_or_bytes, (bytes, bytearray)):
        f = str(tmp_path / "piece.fa")
        open(f, "wb").write(path_or_bytes)
    else:
        f = path_or_bytes
    reads, roffs, heads, hoffs = B.load_reads(f, k, fastq=fastq)
    n = len(roffs) - 1
    if n == 0:
        return b"", b"", 0
    p, po, st = al.align(reads, roffs, m=m, effort=effort, mode=mode)
    pb, nb = [], []
    hb, rb = heads.tobytes(), reads.tobytes()
    for i in range(n):
        h = hb[int(hoffs[i]): int(hoffs[i + 1])]
        if po[i + 1] > po[i]:   # alignerGreedy.cpp:406-411 + printPath aligner.cpp:600-609
            pb.append(h + b"\n" + b"".join(b"%d." % v for v in p[int(po[i]): int(po[i + 1])]) + b"\n")
        else:                   # alignerGreedy.cpp:421-427
            nb.append(h + b"\n" + rb[int(roffs[i]): int(roffs[i + 1])] + b"\n")
    return b"".join(pb), b"".join(nb), n


@pytest.mark.parametrize("reads_file,unitigs,k,m", [("syn_r100.fa", "syn_unitig.fa", 31, 2), ("syn_r150.fa", "syn_unitig.fa", 31, 2), ("syn_r250.fa", "syn_unitig.fa", 31, 5),
                                                     ("long_r150.fa", "long_unitig.fa", 31, 2), ("toy_reads.fa", "toy_unitig.fa", 4, 2)])
def test_text_route_equals_host_route_on_golden_files(reads_file, unitigs, k, m, tmp_path):
    g = B.Graph.from_fasta(os.path.join(GOLD, unitigs), k)
    al = B.Aligner(g, 0)
    text = open(os.path.join(GOLD, reads_file), "rb").read()
    want_p, want_n, n = _host_route(al, os.path.join(GOLD, reads_file), k, tmp_path, m=m)
    c0 = al.counters()
    al.reset_counters()
    got_p, got_n, info = al.align_fasta_text(text, m=m)
    if info["irregular"]:
        # a golden file of another shape (multi-line records ...): nothing may have been mapped
        assert got_p == b"" and got_n == b"" and al.counters()["reads"] == 0
        pytest.skip("piece not of the regular shape")
    assert info["n_accepted"] == n
    assert got_p == want_p
    assert got_n == want_n
    assert al.counters() == c0


def _mixed_piece(seed, n, L, k):
    s = Synth(120000, 90, 2, k, 4200 + seed)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, n, L, 3, 4300 + seed)
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        r = reads[i * L:(i + 1) * L].tobytes()
        x = rng.random()
        if x < 0.02:
            r = r.lower()                      # not ACGTN: dropped, not counted (aligner.cpp:79-84)
        elif x < 0.04:
            r = r[: int(rng.integers(0, k + 1))]  # size <= k (or <= 2, or empty): dropped
        elif x < 0.07:
            b = bytearray(r); b[int(rng.integers(0, L))] = ord("N"); r = bytes(b)   # N is admitted
        elif x < 0.08:
            b = bytearray(r); b[int(rng.integers(0, L))] = 0; r = bytes(b)           # a NUL byte: dropped
        elif x < 0.09:
            r = r + b"\r"                      # CR before the newline: dropped
        h = b">r%d some text > with a '>' inside %d" % (i, i) if i % 7 == 0 else b">r%d" % i
        if seed == 6 and i % 13 == 0:
            h = b">long header " + b"y" * int(rng.integers(200, 900)) + b" %d" % i   # longer than one pass of the record scan
        recs.append(h + b"\n" + r + b"\n")
    return s, seqs, offs, b"".join(recs)


@pytest.mark.parametrize("seed,n,L,k,mode", [(1, 40000, 150, 31, B.MODE_GREEDY), (2, 20011, 100, 21, B.MODE_GREEDY), (3, 9000, 250, 31, B.MODE_EXHAUSTIVE), (4, 5000, 40, 12, B.MODE_GREEDY),
                                             (5, 12000, 150, 31, B.MODE_ANCHORS), (6, 6000, 600, 31, B.MODE_GREEDY)])
def test_text_route_with_dropped_records_equals_host_route(seed, n, L, k, mode, tmp_path):
    """Regular pieces whose records the parser drops (lower case, NUL, CR, size <= k, empty sequence line) or admits (N), headers with
    '>' inside: same bytes and counters as the host parser + bgr_align_batch."""
    s, seqs, offs, text = _mixed_piece(seed, n, L, k)
    g = B.Graph.build(k, seqs, offs, anchors=(mode == B.MODE_ANCHORS))
    al = B.Aligner(g, 0)
    want_p, want_n, n_acc = _host_route(al, text, k, tmp_path, m=2, mode=mode)
    c0 = al.counters()
    al.reset_counters()
    got_p, got_n, info = al.align_fasta_text(text, m=2, mode=mode)
    assert not info["irregular"] and info["n_records"] == n and info["n_accepted"] == n_acc and n_acc < n
    assert got_p == want_p and got_n == want_n
    assert al.counters() == c0
    # too small a paths buffer: BGR_E_CAPACITY, then the same bytes through bgr_aligner_fetch_text
    al.reset_counters()
    got_p2, got_n2, _ = al.align_fasta_text(text, m=2, mode=mode, paths_cap=1000)
    assert got_p2 == want_p and got_n2 == want_n and al.counters() == c0
    # the piece sent ahead through a stage (its own copy stream)
    al.reset_counters()
    got_p4, got_n4, _ = al.align_fasta_text(text, m=2, mode=mode, staged=True)
    assert got_p4 == want_p and got_n4 == want_n and al.counters() == c0
    # counters only
    al.reset_counters()
    p3, n3, info3 = al.align_fasta_text(text, m=2, mode=mode, want_output=False)
    assert p3 == b"" and n3 == b"" and info3["n_accepted"] == n_acc and al.counters() == c0


@pytest.mark.parametrize("name,text", [
    ("multi-line sequence", b">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\nACGTACGTACGT\n>b\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n"),
    ("blank line", b">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n\n>b\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n"),
    ("no final newline", b">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n>b\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT"),
    ("dangling header", b">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n>b\n"),
    ("header only, no newline", b">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n>b"),
    ("does not start with a header", b"ACGT\n>a\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n"),
    ("no record at all", b"ACGTACGTACGT\nACGT\n"),
    ("sequence line that starts with '>'", b">a\n>ACGTACGTACGTACGTACGTACGTACGTACGTACGT\n>b\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n"),
    ("trailing blank lines", b">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n\n\n"),
])
def test_pieces_of_another_shape_are_left_to_the_host_parser(name, text):
    s = Synth(60000, 75, 2, 31, 77)
    seqs, offs = s.unitigs()
    al = B.Aligner(B.Graph.build(31, seqs, offs), 0)
    # such a record in the middle of many regular ones, and alone
    many = b"".join(b">x%d\n%s\n" % (i, b"ACGT" * 40) for i in range(3000))
    for piece in (text, many + text) if not text.startswith(b"ACGT") else (text,):
        p, n, info = al.align_fasta_text(piece)
        assert info["irregular"], name
        assert p == b"" and n == b"" and al.counters()["reads"] == 0


def test_empty_and_tiny_pieces():
    s = Synth(60000, 75, 2, 31, 78)
    seqs, offs = s.unitigs()
    al = B.Aligner(B.Graph.build(31, seqs, offs), 0)
    assert al.align_fasta_text(b"") == (b"", b"", {"irregular": False, "n_records": 0, "n_accepted": 0})
    p, n, info = al.align_fasta_text(b">only\nACG\n")   # size <= 2: dropped, nothing to map
    assert (p, n) == (b"", b"") and not info["irregular"] and info["n_records"] == 1 and info["n_accepted"] == 0
    p, n, info = al.align_fasta_text(b">one\n" + b"ACGT" * 20 + b"\n")
    assert not info["irregular"] and info["n_accepted"] == 1 and (p + n).startswith(b">one\n")


def _fastq_records(seed, n, L, k):
    s = Synth(120000, 90, 2, k, 5200 + seed)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, n, L, 3, 5300 + seed)
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        r = reads[i * L:(i + 1) * L].tobytes()
        x = rng.random()
        if x < 0.02:
            r = r.lower()                                  # not ACGTN: dropped (aligner.cpp:56-65)
        elif x < 0.04:
            r = r[: int(rng.integers(0, 3))]               # size <= 2 (or empty): dropped
        elif x < 0.06:
            r = r[: int(rng.integers(3, k + 1))]           # 2 < size <= k: KEPT in FASTQ (no size > k test there), never anchored
        elif x < 0.09:
            b = bytearray(r); b[int(rng.integers(0, L))] = ord("N"); r = bytes(b)
        elif x < 0.10:
            r = r + b"\r"
        h = [b"@q%d" % i, b"@q%d 1:N:0 @ > + text" % i, b"", b">not an at sign %d" % i][0 if i % 11 else (i // 11) % 4]
        if seed == 4 and i % 17 == 0:
            h = b"@long header " + b"x" * int(rng.integers(200, 700)) + b" %d" % i   # a header line longer than one pass of the record scan
        plus = b"+" if i % 5 else b"+q%d" % i
        qual = bytes(rng.integers(33, 74, len(r)).astype(np.uint8)) if i % 13 else b"@" * len(r)   # '@', '+', '>' are quality characters too
        recs.append(h + b"\n" + r + b"\n" + plus + b"\n" + qual + b"\n")
    return s, seqs, offs, recs


@pytest.mark.parametrize("seed,n,L,k,mode", [(1, 40000, 150, 31, B.MODE_GREEDY), (2, 20000, 100, 21, B.MODE_GREEDY), (3, 10000, 250, 31, B.MODE_EXHAUSTIVE),
                                             (4, 10000, 600, 31, B.MODE_GREEDY)])
def test_fastq_pieces_equal_the_host_route(seed, n, L, k, mode, tmp_path):
    """-q: a piece of whole four-line records (n a multiple of the reference's 10000-record getReads() call, so that the host parser sees
    no phantom record at the end of the file): header = line 0 whatever it holds, read = line 1, kept iff size > 2 and ACGTN."""
    s, seqs, offs, recs = _fastq_records(seed, n, L, k)
    text = b"".join(recs)
    g = B.Graph.build(k, seqs, offs)
    al = B.Aligner(g, 0)
    want_p, want_n, n_acc = _host_route(al, text, k, tmp_path, m=2, mode=mode, fastq=True)
    c0 = al.counters()
    al.reset_counters()
    got_p, got_n, info = al.align_fasta_text(text, m=2, mode=mode, fastq=True)
    assert not info["irregular"] and info["n_records"] == n and info["n_accepted"] == n_acc and n_acc < n
    assert got_p == want_p and got_n == want_n
    assert al.counters() == c0
    al.reset_counters()
    got_p2, got_n2, _ = al.align_fasta_text(text, m=2, mode=mode, fastq=True, staged=True, paths_cap=500)
    assert got_p2 == want_p and got_n2 == want_n and al.counters() == c0
    # a piece that is not whole records is refused (the pipeline cuts at record starts)
    with pytest.raises(B.BgrError):
        al.align_fasta_text(text[:-1], fastq=True)
    # the same records without their '+' and quality lines (fastq = 2: what bgr_align_all sends -- half the bytes over PCIe): same streams,
    # as one host range, staged, and staged in host ranges cut at arbitrary bytes (bgr_text_stage_upload_parts)
    two = b"".join(b"\n".join(r.split(b"\n")[:2]) + b"\n" for r in recs)
    assert len(two) < 0.62 * len(text)
    for kw in ({}, {"staged": True}, {"staged": True, "parts": [1, len(two) // 3, len(two) // 3 + 1, len(two) - 7]}):
        al.reset_counters()
        got_p3, got_n3, info3 = al.align_fasta_text(two, m=2, mode=mode, fastq=2, **kw)
        assert not info3["irregular"] and info3["n_records"] == n and info3["n_accepted"] == n_acc
        assert got_p3 == want_p and got_n3 == want_n and al.counters() == c0


@pytest.mark.parametrize("n,batch,extra", [(25003, 0, []), (25003, 7000, ["-c"]), (30000, 4096, []), (10001, 0, ["-G"]), (9999, 0, []),
                                           (45003, 3000, ["--chunk-bytes", "20000"])])   # (the last: pieces cut while later chunks are still being counted)
def test_cli_fastq_text_route_equals_host_route(n, batch, extra, tmp_path):
    """-q through the text route (device pieces up to the file's last getReads() boundary, the tail with its phantom record on the
    host) == -q with --host-route, two files in a row."""
    k, L = 31, 120
    s, seqs, offs, recs = _fastq_records(7, n, L, k)
    s.write_unitigs(str(tmp_path / "u.fa"))
    open(tmp_path / "a.fq", "wb").write(b"".join(recs))
    open(tmp_path / "b.fq", "wb").write(b"".join(recs[: n // 3]) + b"@last\nACGTACGTAC")   # truncated tail
    args = ["-r", "%s,%s" % (tmp_path / "a.fq", tmp_path / "b.fq"), "-q", "-k", str(k), "-g", str(tmp_path / "u.fa"), "-m", "2", "-t", "4"] + extra
    if batch:
        args += ["--batch", str(batch)]
    (oa, pa, na), (ob, pb, nb) = _cli_pair(args, [], ["--host-route"])
    assert pa == pb and na == nb
    assert [l for l in oa.splitlines() if "seconds" not in l] == [l for l in ob.splitlines() if "seconds" not in l]
    assert pa.count(b"\n") > n // 4
    # (default: only the header and read lines of a piece cross PCIe; --set fastq_gather=0: the four-line records as they are)
    from util import run_cli
    oc, pc, nc = run_cli(B.CLI_PATH, args + ["--set", "fastq_gather=0"])
    assert pc == pa and nc == na


def _cli_pair(args, extra_a, extra_b):
    from util import run_cli
    oa, pa, na = run_cli(B.CLI_PATH, args + extra_a)
    ob, pb, nb = run_cli(B.CLI_PATH, args + extra_b)
    return (oa, pa, na), (ob, pb, nb)


@pytest.mark.parametrize("seed,L,k,m", [(1, 150, 31, 2), (2, 100, 21, 3), (3, 250, 31, 5), (4, 60, 12, 1)])
def test_cli_correction_mode_on_the_device_equals_the_host_formatter(seed, L, k, m, tmp_path):
    """-c through the text route (the device spells every mapped read from its path and the 2-bit unitig store, reverse complemented
    when the path was found on the other strand) == -c with --host-route (recover_path on the host, itself pinned to the reference's
    goldens): same bytes, same stdout."""
    s = Synth(150000, 3 * k, 3, k, 8100 + seed)
    s.write_unitigs(str(tmp_path / "u.fa"))
    s.write_reads(str(tmp_path / "r.fa"), 0, 30000, L, m + 1, 8200 + seed)
    args = ["-r", str(tmp_path / "r.fa"), "-k", str(k), "-g", str(tmp_path / "u.fa"), "-m", str(m), "-c", "-t", "4"]
    (oa, pa, na), (ob, pb, nb) = _cli_pair(args, [], ["--host-route"])
    assert pa == pb and na == nb
    assert [l for l in oa.splitlines() if "seconds" not in l] == [l for l in ob.splitlines() if "seconds" not in l]
    assert pa.count(b"\n") > 20000   # most reads map: header + corrected read
    # a corrected read is as long as its read (the walk covers it) and made of ACGT
    lines = pa.split(b"\n")
    assert all(len(x) == L and set(x) <= set(b"ACGT") for x in lines[1:2000:2])


@pytest.mark.parametrize("lanes,extra", [(2, []), (3, ["--host-route"]), (4, ["--batch", "3000"])])
def test_cli_split_output_pairs_concatenate_to_the_single_run(lanes, extra, tmp_path):
    """--gpus N --split-output: one pipeline per device over contiguous shares of the input, N output pairs; `cat` in device order must be
    the single pipeline's files.  (--set test.lanes_on_one_device=1 puts every lane on this box's one GPU: the split run's own code --
    shares cut at record starts over two input files, concurrent pipelines, the text route's fall-back per piece on the messy file -- on
    real device calls.)"""
    import subprocess
    k, L, n = 31, 110, 30000
    s = Synth(150000, 90, 2, k, 6100)
    s.write_unitigs(str(tmp_path / "u.fa"))
    s.write_reads(str(tmp_path / "a.fa"), 0, n, L, 2, 6101)
    reads, _ = s.reads(n, 4000, L, 2, 6101)
    with open(tmp_path / "b.fa", "wb") as f:   # a second file of another shape: multi-line sequences, a lower-case read, a short one
        for i in range(4000):
            r = reads[i * L:(i + 1) * L].tobytes()
            if i % 7 == 0:
                r = r[:50] + b"\n" + r[50:]
            if i % 97 == 0:
                r = r.lower()
            if i % 131 == 0:
                r = r[:20]
            f.write(b">b%d\n" % i + r + b"\n")
    args = ["-r", "%s,%s" % (tmp_path / "a.fa", tmp_path / "b.fa"), "-k", str(k), "-g", str(tmp_path / "u.fa"), "-m", "2", "-t", "6"] + extra
    from util import run_cli
    o1, p1, n1 = run_cli(B.CLI_PATH, args)
    d = tmp_path / "split"
    d.mkdir()
    pr = subprocess.run([B.CLI_PATH] + args + ["--gpus", str(lanes), "--split-output", "--set", "test.lanes_on_one_device=1"], cwd=d, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-2000:]
    ps = b"".join(open(d / ("paths.%d" % i), "rb").read() for i in range(lanes))
    ns = b"".join(open(d / ("notAligned.fa.%d" % i), "rb").read() for i in range(lanes))
    assert ps == p1 and ns == n1 and len(p1) > 100000
    assert not (d / "paths").exists()
    keep = lambda o: [l for l in o.splitlines() if "seconds" not in l]
    assert keep(pr.stdout) == keep(o1)   # file names, then the reference's closing block with the same counters


@pytest.mark.parametrize("mode", [B.MODE_GREEDY, B.MODE_EXHAUSTIVE])
def test_record_info_says_what_became_of_every_record(mode, tmp_path):
    """bgr_text_batch.record_info_out (what bgr_align_all's writer turns into the reference's -b progress blocks): one word per record of
    the piece, kept << 31 | mapped << 30 | read length -- against the host parser (which records getReads keeps) and bgr_align_batch
    (which of those are mapped), with and without the formatted streams."""
    k = 31
    s, seqs, offs, text = _mixed_piece(5, 6000, 150, k)
    al = B.Aligner(B.Graph.build(k, seqs, offs), 0)
    f = tmp_path / "piece.fa"
    f.write_bytes(text)
    reads, roffs, heads, hoffs = B.load_reads(str(f), k)
    paths, poffs, status = al.align(reads, roffs, m=2, mode=mode)
    kept = {bytes(heads[hoffs[i]:hoffs[i + 1]]): (int(roffs[i + 1] - roffs[i]), bool(poffs[i + 1] > poffs[i])) for i in range(len(roffs) - 1)}
    assert len(kept) == len(roffs) - 1   # (headers are unique in this piece)
    headers = [l for l in text.split(b"\n")[0::2] if l]
    for want in (True, False):
        _, _, info = al.align_fasta_text(text, m=2, mode=mode, want_output=want, record_info=True)
        rec = info["records"]
        assert not info["irregular"] and len(rec) == info["n_records"] == len(headers) and int((rec >> 31).sum()) == info["n_accepted"] == len(kept)
        for h, v in zip(headers, rec):
            if h in kept:
                assert (int(v) >> 31) == 1 and (int(v) & 0x3FFFFFFF) == kept[h][0] and bool(int(v) & 0x40000000) == kept[h][1]
            else:
                assert int(v) == 0


@pytest.mark.parametrize("fastq", [0, 1, 2])
def test_pieces_of_tiny_records_beyond_the_record_table_go_to_the_host(fastq, tmp_path):
    """The device keeps room for one record per 24 bytes of a piece; a piece with more record starts (reads of a dozen bases: a k-mer list,
    adapter fragments) must come back flagged, nothing mapped.  The kernels behind the record count are enqueued before the host has read
    it: they must run on zeroes, not on what the (here: poisoned) buffers held -- found by tools/fuzz_text_route.py as a GPU memory fault
    in a long-lived process (FASTQ, k = 15, reads of 11 .. 18 bases).  The aligner goes on mapping regular pieces afterwards, and the CLI
    writes the same bytes on both routes."""
    s = Synth(120000, 60, 2, 15, 41)
    seqs, offs = s.unitigs()
    g = B.Graph.build(15, seqs, offs)
    # device memory full of ones, handed back to the allocator: what the aligner's fresh buffers are likely to be carved from
    poison = [B.DeviceBuffer(0, np.full(8 << 20, 0xFF, dtype=np.uint8)) for _ in range(6)]
    for pb in poison:
        pb.free()
    al = B.Aligner(g, 0)
    reads, _ = s.reads(0, 60000, 12, 1, 42)
    reads = reads.reshape(60000, 12)
    if fastq == 0:
        text = b"".join(b">%d\n%s\n" % (i % 10, reads[i].tobytes()) for i in range(60000))               # 16 bytes per record
    elif fastq == 1:
        text = b"".join(b"@\n%s\n+\n%s\n" % (reads[i, :4].tobytes(), b"IIII") for i in range(60000))      # 14 bytes per four-line record
    else:
        text = b"".join(b"@%d\n%s\n" % (i % 10, reads[i].tobytes()) for i in range(60000))               # header and read lines only
    p, n, info = al.align_fasta_text(text, fastq=fastq)
    assert info["irregular"] and info["n_records"] > len(text) // 24 + 1024 and (p, n) == (b"", b"")
    # the same aligner, a regular piece behind it
    r2, o2 = s.reads(0, 3000, 60, 2, 43)
    text2 = b"".join(b">r%d\n%s\n" % (i, r2[60 * i: 60 * i + 60].tobytes()) for i in range(3000))
    p2, n2, info2 = al.align_fasta_text(text2)
    assert not info2["irregular"] and info2["n_accepted"] == 3000 and p2.count(b">") + n2.count(b">") == 3000
    if fastq != 2:   # whole files through the CLI: the text route (pieces handed back to the host parser) == the host route
        f = tmp_path / ("tiny.fq" if fastq else "tiny.fa")
        f.write_bytes(text)
        u = tmp_path / "u.fa"
        s.write_unitigs(str(u))
        a = _cli_pair(["-r", str(f), "-k", "15", "-g", str(u), "-t", "4"] + (["-q"] if fastq else []), [], ["--host-route"])
        assert a[0][1:] == a[1][1:]


@pytest.mark.parametrize("memo_cap", [0, 8])
def test_text_form_in_exhaustive_mode_when_the_last_pass_hands_reads_back(memo_cap, tmp_path):
    """Exhaustive mode on a unitig set where the reference's recursion is exponential (tests/util.py homopolymer_soup), through the TEXT form: the kernels behind
    the mapping launch (record info, sizes, formatting) are enqueued before the host knows whether the last pass handed reads back for a larger table of
    remembered calls (BGR_KNOB_EXH_MEMO_CAP 8: it does); then the launch is settled and they are enqueued again -- same record bytes and the same record info
    as the host route (parser + bgr_align_batch + the reference's record format), with and without the formatted streams."""
    from util import homopolymer_soup
    k, (seqs, offs), (sr, so), (lr, lo) = homopolymer_soup()
    reads = [bytes(sr[int(so[i]):int(so[i + 1])]) for i in range(len(so) - 1)] + [bytes(lr[int(lo[i]):int(lo[i + 1])]) for i in range(len(lo) - 1)]
    text = b"".join(b">q%d\n%s\n" % (i, r) for i, r in enumerate(reads))
    g = B.Graph.build(k, seqs, offs)
    al = B.Aligner(g, 0)
    al.set_knob(B.KNOB_EXH_MEMO_CAP, memo_cap)
    pb, nb, n = _host_route(al, text, k, tmp_path, m=2, mode=B.MODE_EXHAUSTIVE)
    assert n == len(reads) and len(pb) > 0 and len(nb) > 0
    p1, n1, info = al.align_fasta_text(text, m=2, mode=B.MODE_EXHAUSTIVE, want_output=True, record_info=True)
    assert not info["irregular"] and p1 == pb and n1 == nb
    if memo_cap:
        assert al.last_pass_runs()[0] >= 2    # (reads were handed back: the kernels behind the launch ran twice)
    mapped = int(((info["records"] >> 30) & 1).sum())
    assert mapped == pb.count(b"\n") // 2
    _, _, info0 = al.align_fasta_text(text, m=2, mode=B.MODE_EXHAUSTIVE, want_output=False, record_info=True)
    assert np.array_equal(info0["records"], info["records"])


@pytest.mark.parametrize("L,hdr,fastq", [(1500, 0, False), (5000, 0, False), (25000, 3000, False), (700, 2500, False), (150, 400, False), (3000, 0, True), (24000, 1200, True), (150, 300, 2)])
def test_records_that_leave_a_tile_and_its_window(L, hdr, fastq, tmp_path):
    """The parse launch reads a record's shape off bit masks of 32 KB of text + 1 KB behind it; a record that leaves that window (long reads, long
    headers) is scanned from the text by a 16-lane group, and a workgroup's stretch of the paths stream that is larger than its LDS buffer (long
    headers) goes out record by record: same bytes as the host parser + formatter; a dropped record, an N, and the tile borders anywhere."""
    n = max(40, min(4000, 6_000_000 // (L + hdr + 10)))
    s = Synth(300000, 110, 2, 31, 9100 + L)
    seqs, offs = s.unitigs()
    reads, roffs = s.reads(0, n, L, 3, 9200 + L)
    rng = np.random.default_rng(L + hdr)
    recs = []
    for i in range(n):
        r = reads[i * L:(i + 1) * L].tobytes()
        x = rng.random()
        if x < 0.03:
            b = bytearray(r); b[int(rng.integers(0, L))] = ord("x"); r = bytes(b)     # dropped: a character that is not one of ACGTN, anywhere in a long read
        elif x < 0.06:
            b = bytearray(r); b[int(rng.integers(0, L))] = ord("N"); r = bytes(b)
        h = b">r%d" % i + (   # (FASTQ: the header is line 0 whatever it holds -- a '>' here, so that the expected bytes come from the FASTA host parser
                             # on the header and read lines; the reference's FASTQ reader adds a phantom record at the end of a FILE, tests above)
                           b" " + bytes(rng.integers(97, 123, int(rng.integers(hdr // 2, hdr + 1))).astype(np.uint8)) if hdr else b"")
        if fastq == 1:
            recs.append(h + b"\n" + r + b"\n+\n" + bytes(rng.integers(33, 74, len(r)).astype(np.uint8)) + b"\n")
        else:
            recs.append(h + b"\n" + r + b"\n")
    text = b"".join(recs)
    al = B.Aligner(B.Graph.build(31, seqs, offs), 0)
    host_text = text if fastq != 1 else b"".join(b"\n".join(rec.split(b"\n")[:2]) + b"\n" for rec in recs)
    want_p, want_n, n_acc = _host_route(al, host_text, 31, tmp_path, m=3)
    c0 = al.counters()
    for kw in ({}, {"staged": True}):
        al.reset_counters()
        got_p, got_n, info = al.align_fasta_text(text, m=3, fastq=fastq, **kw)
        assert not info["irregular"] and info["n_records"] == n and info["n_accepted"] == n_acc and n_acc < n
        assert got_p == want_p and got_n == want_n and len(want_p) > 0
        assert al.counters() == c0
    # a last record that leaves the window AND lacks its newline / its sequence line: another shape
    if not fastq:
        for cut in (text[:-1], text[: len(text) - L - 1], text + b">tail with no newline"):
            p, nn, info = al.align_fasta_text(cut, m=3)
            assert info["irregular"] and p == b"" and nn == b""


def test_the_chains_of_the_text_kernels_across_the_wrap_of_their_epoch(tmp_path):
    """The parse and format launches pass running totals between workgroups through chain words tagged with a 22-bit epoch, never cleared between launches; when
    the epochs run out the host clears the chains and starts over.  With the test hook the wrap comes after three pieces: same bytes before, at and behind it, on
    pieces of several tiles each (so that stale words of the piece before would be read)."""
    s, seqs, offs, text = _mixed_piece(1, 30000, 150, 31)
    g = B.Graph.build(31, seqs, offs)
    with B.options(**{"test.text_epoch": 0x3FFFFF - 6}):
        al = B.Aligner(g, 0)
        want_p, want_n, n_acc = _host_route(al, text, 31, tmp_path, m=2)
        half = text[: text.index(b">r15000\n")]
        for i in range(8):   # (two epochs per piece: parse, format)
            piece = text if i % 2 == 0 else half
            got_p, got_n, info = al.align_fasta_text(piece, m=2)
            assert not info["irregular"]
            if i % 2 == 0:
                assert info["n_accepted"] == n_acc and got_p == want_p and got_n == want_n, i
            else:
                assert want_p.startswith(got_p) and want_n.startswith(got_n) and 0 < len(got_p) < len(want_p), i
        al.close()
