"""CPU-side checks of the product library: it loads, exports every symbol include/bgreat_gpu.h declares, the host
parts (index build, blob round trip, read parser, record writer) behave like the reference, and mapping fails
loudly without a GPU.  No kernel is launched here."""
import ctypes as C
import os
import re
import tempfile

import numpy as np
import pytest

import bgreat_amd as B
import oracle_py
from tools.synth import Synth
from util import GOLD, ROOT, golden_cases


@pytest.fixture(scope="module", autouse=True)
def built():
    B.build()


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bgreat_gpu.h")).read()
    declared = set(re.findall(r"\b(bgr_[a-z_0-9]+)\s*\(", hdr))
    types = {"bgr_graph", "bgr_aligner", "bgr_readset", "bgr_params", "bgr_graph_info_t"}
    declared -= types
    assert declared == set(B.SYMBOLS), declared ^ set(B.SYMBOLS)
    L = B.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_graph_build_info_and_blob_roundtrip():
    s = Synth(30000, 75, 2, 31, 5)
    seqs, offs = s.unitigs()
    g = B.Graph.build(31, seqs, offs)
    info = g.info()
    assert info["k"] == 31 and info["n_unitigs"] == len(offs) - 1
    assert info["n_keys"] <= info["n_left_keys"] + info["n_right_keys"]
    assert info["total_bases"] == 2 * int(offs[-1])
    blob = np.array(g.blob())
    g2 = B.Graph.from_blob(blob)
    assert g2.info() == info
    assert np.array_equal(np.array(g2.blob()), blob)
    with pytest.raises(B.BgrError):
        B.Graph.from_blob(blob[:1000])
    bad = blob.copy()
    bad[0] ^= 1
    with pytest.raises(B.BgrError):
        B.Graph.from_blob(bad)


def test_blob_validation_refuses_corrupt_headers():
    """A blob from a file or from another rank is only trusted after every section and the key table's size have been
    checked against its size (the kernels index HBM with those numbers): corrupt single header fields one at a time."""
    s = Synth(30000, 75, 2, 31, 6)
    seqs, offs = s.unitigs()
    blob = np.array(B.Graph.build(31, seqs, offs, anchors=True).blob())
    B.Graph.from_blob(blob)   # intact: accepted
    hdr = blob[:4096].view(np.uint64)
    # BgrBlobHeader (graph_layout.h): u64 words 0 magic, 1 version|k, 2 blob_bytes, 3 n_unitigs, 4 n_keys, 5 n_placed, 6 n_fallback,
    # 7 seq_words, 8 total_bases, 9 n_buckets, 10.. off_table, off_keys, off_recs, off_meta, off_seq, off_exc, off_excn, off_fallback
    def corrupt(word, value):
        bad = blob.copy()
        bad[:4096].view(np.uint64)[word] = value
        with pytest.raises(B.BgrError):
            B.Graph.from_blob(bad)
    corrupt(3, int(hdr[3]) + (1 << 40))          # n_unitigs: meta section would leave the blob
    corrupt(4, int(hdr[4]) + 1)                  # n_keys no longer 4 * n_buckets + n_fallback
    corrupt(4, (1 << 61))                        # count * size overflows 64 bits
    corrupt(7, int(hdr[7]) + (1 << 30))          # seq_words beyond the blob
    corrupt(9, 0)                                # no buckets at all
    corrupt(9, int(hdr[9]) + 1)                  # n_buckets: the index space no longer matches the table
    corrupt(10, int(hdr[10]) + 8)                # off_table not 256-byte aligned
    corrupt(12, len(blob))                       # off_recs at the end of the blob
    corrupt(14, 0xFFFFFFFFFFFFFF00)              # off_seq + size wraps around
    with pytest.raises(B.BgrError):
        B.Graph.from_blob(blob[:-256])           # truncated


_HDR_N_SLOTS = 23   # BgrBlobHeader.n_slots as a uint64 index (graph_layout.h: ... max_unitig_len 19, n_left_keys 20, n_right_keys 21, slot_fill 22, n_slots 23)


def _canonical_end_kmers(seqs, offs, k):
    """The reference's overlap key set (aligner.cpp:466-533): canonical first and last (k-1)-mers of every unitig."""
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    K1 = k - 1
    keys = set()
    for i in range(len(offs) - 1):
        u = bytes(seqs[int(offs[i]):int(offs[i + 1])])
        if len(u) < k:
            break
        for w in (u[:K1], u[-K1:]):
            x = 0
            for ch in w:
                x = x << 2 | code.get(ch, 3)
            rc = 0
            for ch in reversed(w):
                rc = rc << 2 | (3 - code.get(ch, 3))
            keys.add(min(x, rc))
    return keys


@pytest.mark.parametrize("gamma,no_evictions", [(0.0, False), (1.03, False), (1.8, False), (4.0, False), (1.07, True)])
def test_overlap_key_table(gamma, no_evictions):
    """Two-choice fingerprint table (graph_layout.h): every overlap key is found in a slot of its own, non-members are
    rejected, and a key sits in its second bucket only when the first is full (what the L2 probing of find_key relies on).
    no_evictions is the test hook that sends keys to the fallback list."""
    k = 21
    s = Synth(60000, 45, 3, k, 17)
    seqs, offs = s.unitigs()
    g = B.Graph.build(k, seqs, offs, gamma, no_evictions=no_evictions)
    info = g.info()
    keys = _canonical_end_kmers(seqs, offs, k)
    assert info["n_keys"] == len(keys)
    assert (info["n_fallback"] > 0) == no_evictions
    slots = [g.key_lookup(x) for x in keys]
    assert None not in slots and len(set(slots)) == len(slots)
    rng = np.random.default_rng(5)
    for x in rng.integers(0, 1 << (2 * (k - 1)), size=20000, dtype=np.uint64):
        assert (g.key_lookup(int(x)) is not None) == (int(x) in keys)
    # blob side: the fingerprint bytes, the key per slot, the placement invariant
    blob = np.array(g.blob())
    hdr = blob[:4096].view(np.uint64)
    n_buckets, off_table, off_keys = int(hdr[9]), int(hdr[10]), int(hdr[11])
    table = blob[off_table:off_table + 4 * n_buckets]
    kent = blob[off_keys:off_keys + 16 * (4 * n_buckets + info["n_fallback"])]   # 16 B per entry: key, handle of the left half, of the right half
    kslot = kent.view(np.uint64)[0::2]
    khand = kent.view(np.uint32).reshape(-1, 4)[:, 2:4]
    assert int((table != 0).sum()) == len(keys) - info["n_fallback"]
    assert np.array_equal(table == 0, kslot[:4 * n_buckets] == np.uint64(0xFFFFFFFFFFFFFFFF))
    full = (table.reshape(-1, 4) != 0).all(axis=1)
    M = (1 << 64) - 1
    for x, slot in zip(keys, slots):
        if slot >= 4 * n_buckets:
            continue                                   # fallback list entry
        y = x ^ (x >> 32)
        m = (y * 0x9E3779B97F4A7C15) & M               # bgr_mix64
        b1 = ((m & 0xFFFFFFFF) * n_buckets) >> 32
        b2 = ((m >> 32) * n_buckets) >> 32
        assert slot // 4 in (b1, b2) and int(kslot[slot]) == x
        assert int(table[slot]) == max(1, (m >> 32) & 0xFF)
        if slot // 4 != b1:
            assert full[b1]
    if not no_evictions and gamma in (0.0, 1.03):
        assert 4 * n_buckets <= 1.08 * len(keys) + 4  # the tight fill small graphs are staged in LDS with
    # the neighbour records, compact (graph_layout.h): the slots of a key's left / right half stand next to each other from its handle
    # on, the last one flagged, ids in the reference's fill order (first free of indice1..3, later ones overwrite indice4:
    # aligner.cpp:466-533), nothing else in the array; every slot's two "where the walk goes on" words name the start of a half
    off_recs, n_slots = int(hdr[12]), int(blob[:4096].view(np.uint64)[_HDR_N_SLOTS])
    sl = blob[off_recs:off_recs + 32 * (n_slots + 4)].view(np.uint32).reshape(-1, 8)
    assert not sl[n_slots:].any()                      # four zero slots behind the last one
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    K1 = k - 1
    left, right = {}, {}

    def attach(tab, key, i):
        lst = tab.setdefault(key, [])
        if len(lst) < 3:
            lst.append(i)
        elif len(lst) == 3:
            lst.append(i)
        else:
            lst[3] = i
    for i in range(len(offs) - 1):
        u = bytes(seqs[int(offs[i]):int(offs[i + 1])])
        if len(u) < k:
            break
        enc = lambda w: (sum(code.get(ch, 3) << (2 * (K1 - 1 - j)) for j, ch in enumerate(w)), sum((3 - code.get(ch, 3)) << (2 * j) for j, ch in enumerate(w)))
        beg, rcbeg = enc(u[:K1])
        end, rcend = enc(u[-K1:])
        attach(left, beg, i + 1) if beg <= rcbeg else attach(right, rcbeg, i + 1)
        attach(right, end, i + 1) if end <= rcend else attach(left, rcend, i + 1)
    HNONE, LAST = 0x0FFFFFFF, 32
    starts, seen = set(), 0
    for x, slot in zip(keys, slots):
        for side, tab in ((0, left), (1, right)):
            want = tab.get(x, [])
            h = int(khand[slot, side])
            assert (h == HNONE) == (not want)
            if not want:
                continue
            starts.add(h)
            got = sl[h:h + len(want)]
            assert [int(v) & 0x3FFFFFFF for v in got[:, 0]] == want
            assert [bool(int(v) & LAST) for v in got[:, 3]] == [False] * (len(want) - 1) + [True]
            seen += len(want)
    assert seen == n_slots
    nxt = np.concatenate([sl[:n_slots, 5], sl[:n_slots, 6]]) & HNONE
    assert all(int(v) == HNONE or int(v) in starts for v in nxt)


def _rc16(x):
    y = 0
    for _ in range(16):
        y = (y << 2) | (3 - (x & 3))
        x >>= 2
    return y


def _mmx_hash(x):
    c = min(x, _rc16(x))
    return ((c * 0x9E3779B1) & 0xFFFFFFFF) | 1        # bgr_mmx_hash


def _mmx_of_key(key, K1):
    return max(_mmx_hash((key >> (2 * (K1 - 16 - j))) & 0xFFFFFFFF) for j in range(K1 - 15))   # bgr_mmx_of_key


def _filter_passes(bl, kind, bits, key, K1):
    M = (1 << 64) - 1
    y = key ^ (key >> 32)
    m = (y * 0x9E3779B97F4A7C15) & M                   # bgr_mix64
    if kind == 1:
        bit = (m >> 20) & (bits - 1)                   # bgr_bloom_bit
        return (int(bl[bit >> 5]) >> (bit & 31)) & 1
    shift = 32 - ((bits // 512).bit_length() - 1)
    blk = ((_mmx_of_key(key, K1) * 0x85EBCA6B) & 0xFFFFFFFF) >> shift          # bgr_mmx_block
    want = (1 << ((m >> 12) & 31)) | (1 << ((m >> 17) & 31))                   # bgr_mmx_bits
    return (int(bl[blk * 16 + ((m >> 8) & 15)]) & want) == want                # bgr_mmx_word


@pytest.mark.parametrize("k,env,kind", [(31, None, 2), (21, None, 2), (32, None, 2), (20, None, 1), (31, "1", 1), (25, "2", 2)])
def test_large_tables_get_a_filter_that_passes_every_key(k, env, kind, monkeypatch):
    """A key table too large for LDS staging is built with a filter in front (graph_layout.h): minimizer-blocked for k-1 >= 20 (24-48
    bits per key in 64-byte blocks chosen by the key's minimizer), one hash (4-8 bits per key) for shorter k; every key passes (a member
    is never turned away), the filter is not saturated, and both strands of a key choose the same block."""
    if env is not None:
        B.set_option("build_filter", int(env))   # (put back by conftest.py's fixture)
    small = env == "2"
    s = Synth(60000 if small else 2_600_000, 60, 2, k, 23)
    seqs, offs = s.unitigs()
    g = B.Graph.build(k, seqs, offs)
    keys = _canonical_end_kmers(seqs, offs, k)
    assert small or len(keys) * 1.07 > 73000
    blob = np.array(g.blob())
    hdr = blob[:4096].view(np.uint64)
    off_bloom, bits = int(hdr[24]), int(hdr[25])
    assert int(hdr[18]) & 0xFFFFFFFF == kind
    lo, hi = (24, 48) if kind == 2 else (4, 8)
    assert bits & (bits - 1) == 0 and (lo * len(keys) <= bits < hi * len(keys) or bits == 1024)
    bl = blob[off_bloom:off_bloom + bits // 8].view(np.uint32)
    K1 = k - 1
    rng = np.random.default_rng(5)
    for x in list(keys)[:20000]:
        assert _filter_passes(bl, kind, bits, x, K1)
    if kind == 2:   # strand symmetry of the minimizer: the reverse complement of a key has the same 16-mer set
        for x in list(keys)[:200]:
            rc = 0
            t = x
            for _ in range(K1):
                rc = (rc << 2) | (3 - (t & 3))
                t >>= 2
            assert _mmx_of_key(x, K1) == _mmx_of_key(rc, K1)
    # random non-keys are mostly turned away
    keyset = set(keys)
    non = [int(v) for v in rng.integers(0, 1 << (2 * K1), 4000, dtype=np.uint64) if int(v) not in keyset]
    passed = sum(1 for v in non if _filter_passes(bl, kind, bits, v, K1))
    assert passed < (0.05 if kind == 2 else 0.3) * len(non)
    ones = int(np.unpackbits(bl.view(np.uint8)).sum())
    assert 0.01 * bits < ones <= 2 * len(keys)


def test_small_tables_get_no_filter():
    s2 = Synth(60000, 45, 3, 31, 17)
    sq, of = s2.unitigs()
    h2 = np.array(B.Graph.build(31, sq, of).blob())[:4096].view(np.uint64)
    assert int(h2[25]) == 0


def test_graph_build_rejects_bad_k():
    seqs = np.frombuffer(b"ACGTACGTAC", dtype=np.uint8)
    offs = np.array([0, 10], dtype=np.uint64)
    for k in (0, 1, 33, 64):
        with pytest.raises(B.BgrError):
            B.Graph.build(k, seqs, offs)


def test_unitig_loading_stops_at_first_short_sequence():
    g = B.Graph.from_fasta(os.path.join(GOLD, "short_stop_unitig.fa"), 5)
    assert g.info()["n_unitigs"] == 40   # aligner.cpp:418-420
    g = B.Graph.from_fasta(os.path.join(GOLD, "toy_unitig.fa"), 4)
    assert g.info()["n_unitigs"] == 6


def test_index_build_does_not_depend_on_thread_count(tmp_path):
    """The parallel build (key sort, cascade with atomic state words, range-partitioned slot fill) gives one blob."""
    s = Synth(1300000, 60, 3, 31, 11)   # > 2^16 end k-mers per side so the parallel sort and cascade really split
    seqs, offs = s.unitigs()
    seqs = seqs.copy()
    seqs[np.arange(5, len(seqs), 40001)] = ord("N")   # exception planes too
    fa = str(tmp_path / "u.fa")
    with open(fa, "wb") as f:
        for i in range(len(offs) - 1):
            f.write(b">%d\n" % i + seqs[int(offs[i]):int(offs[i + 1])].tobytes() + b"\n")
    blobs = []
    try:
        for T in (1, 2, 3, 8):
            B.lib().bgr_set_build_threads(T)
            g = B.Graph.build(31, seqs, offs)
            assert g.info()["n_left_keys"] > (1 << 15) and g.info()["n_unitigs"] > (1 << 16) and g.info()["has_exceptions"] == 1
            blobs.append(np.array(g.blob()))
            blobs.append(np.array(B.Graph.from_fasta(fa, 31).blob()))
    finally:
        B.lib().bgr_set_build_threads(0)
    assert all(np.array_equal(blobs[0], b) for b in blobs[1:])


@pytest.mark.parametrize("text,n", [(b"", 0), (b">a", 0), (b">a\n", 0), (b">a\nACGTA", 1), (b">a\nACGTA\n", 1), (b">a\nACGTA\n>b", 1),
                                    (b">a\nACGTA\n\n\nACGTACG\n", 1), (b">a\nACGTA\n>b\nACG\n>c\nACGTAA\n", 1),
                                    (b"ACGTA\nACGTAC\nxx\nACGTACGT", 2), (b">a\nACGTA\r\n>b\nACGT\r\n", 2)])
def test_unitig_fasta_two_lines_per_record(tmp_path, text, n):
    """aligner.cpp:415-420: header line ignored whatever it holds, a missing line reads as empty, stop at |seq| < k."""
    fa = str(tmp_path / "u.fa")
    open(fa, "wb").write(text)
    assert B.Graph.from_fasta(fa, 5).info()["n_unitigs"] == n


def _canon(x, k):
    rc = 0
    y = x
    for _ in range(k):
        rc = (rc << 2) | (3 - (y & 3))
        y >>= 2
    return min(x, rc)


@pytest.mark.parametrize("fa,k", [("toy_unitig.fa", 4), ("deg_unitig.fa", 5), ("deg_unitig_exc.fa", 5), ("syn_unitig.fa", 31),
                                  ("syn_unitig.fa", 32), ("syn_unitig.fa", 12), ("long_unitig.fa", 31)])
def test_anchors_index_is_boophf_bit_for_bit(fa, k):
    """-G consumes MPHF answers for non-keys (aligner.cpp:387-389): the product's anchors index must return what the
    oracle's BooPHF restatement (pinned by the -G goldens) returns, for every unitig k-mer and for random non-keys."""
    path = os.path.join(GOLD, fa)
    try:
        graphs = []
        for T in (1, 4):
            B.lib().bgr_set_build_threads(T)
            graphs.append(B.Graph.from_fasta(path, k, anchors=True))
    finally:
        B.lib().bgr_set_build_threads(0)
    g = graphs[0]
    assert g.info()["has_anchors"] == 1 and np.array_equal(np.array(g.blob()), np.array(graphs[1].blob()))
    o = oracle_py.Oracle(k, fasta=path, anchors=True)
    code = {"A": 0, "C": 1, "G": 2}
    keys = []
    for line in open(path).read().split("\n")[1::2]:
        if len(line) < k:
            break
        for j in range(0, max(0, len(line) - k)):
            x = 0
            for ch in line[j:j + k]:
                x = (x << 2) | code.get(ch, 3)
            keys.append(_canon(x, k))
    assert keys
    rng = np.random.default_rng(5)
    keys = keys[:4000] + [int(v) & ((1 << (2 * k)) - 1) for v in rng.integers(0, 1 << 62, 3000)] + [0, (1 << (2 * k)) - 1]
    hits = 0
    for x in keys:
        a, b = g.anchor_lookup(x), o.anchor_lookup(x)
        assert a == b, (x, a, b)
        hits += a[0] is not None
    assert hits >= min(4000, len(keys) - 3002)
    g2 = B.Graph.from_blob(np.array(g.blob()))   # the sections survive the blob round trip
    assert g2.anchor_lookup(keys[0]) == g.anchor_lookup(keys[0]) and g2.info() == g.info()
    with pytest.raises(B.BgrError):
        B.Graph.from_fasta(path, k).anchor_lookup(keys[0])


def test_unitig_fasta_through_a_fifo(tmp_path):
    """-g may be a FIFO / process substitution (st_size 0, not mappable): read to the end, same graph as from the file."""
    import threading
    s = Synth(60000, 80, 2, 31, 77)
    ufa, fifo = str(tmp_path / "u.fa"), str(tmp_path / "u.fifo")
    s.write_unitigs(ufa)
    os.mkfifo(fifo)
    def feed():
        with open(fifo, "wb") as o, open(ufa, "rb") as i:
            o.write(i.read())
    t = threading.Thread(target=feed)
    t.start()
    g2 = B.Graph.from_fasta(fifo, 31)
    t.join()
    g1 = B.Graph.from_fasta(ufa, 31)
    assert g1.info()["n_unitigs"] > 500 and np.array_equal(np.array(g1.blob()), np.array(g2.blob()))


def _pack_reference(reads, offs):
    """The plane layout of include/bgreat_gpu.h, written out plainly: codes A0 C1 G2 else 3, 32 per word, first base in the top
    bits, read r at word (offs[r] >> 5) + r; N reads flagged, with a mask word (3 per N) for every word of theirs."""
    code = np.full(256, 3, dtype=np.uint64)
    code[ord("A")], code[ord("C")], code[ord("G")], code[ord("T")] = 0, 1, 2, 3
    n = len(offs) - 1
    words = (int(offs[n]) >> 5) + n + 4
    fw3 = np.zeros(words, dtype=np.uint64)
    hasn = np.zeros((n + 31) // 32 + 1, dtype=np.uint32)
    nm = {}
    for r in range(n):
        seq = reads[int(offs[r]):int(offs[r + 1])]
        w0 = (int(offs[r]) >> 5) + r
        isn = seq == ord("N")
        for j in range((len(seq) + 31) // 32):
            chunk = seq[32 * j:32 * j + 32]
            v = 0
            for i, c in enumerate(chunk):
                v |= int(code[c]) << (62 - 2 * i)
            fw3[w0 + j] = v
            if isn.any():
                m = 0
                for i in np.nonzero(isn[32 * j:32 * j + 32])[0]:
                    m |= 3 << (62 - 2 * int(i))
                nm[w0 + j] = m
        if isn.any():
            hasn[r >> 5] |= np.uint32(1 << (r & 31))
    return fw3, hasn, nm


def test_host_packer_layout():
    """bgr_pack_reads (SSSE3 or scalar) against the layout written out in numpy: ragged lengths 1..200 around the 16- and
    32-base boundaries, N in a few reads, a batch that does not start at offset 0."""
    rng = np.random.default_rng(12)
    lens = np.concatenate([np.arange(1, 70), rng.integers(1, 200, size=300), [15, 16, 17, 31, 32, 33, 63, 64, 65, 128, 150]])
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    reads = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(offs[-1]))
    for r in rng.choice(len(lens), size=25, replace=False):
        a, b = int(offs[r]), int(offs[r + 1])
        reads[rng.integers(a, b, size=min(3, b - a))] = ord("N")
    pk = B.pack_reads(reads, offs)
    fw3, hasn, nm = _pack_reference(reads, offs)
    assert np.array_equal(pk["fw3"][:len(fw3)], fw3)
    assert np.array_equal(pk["hasn"][:len(hasn)], hasn) and pk["max_read_len"] == int(lens.max())
    assert dict(zip(pk["nm_index"].tolist(), pk["nm_value"].tolist())) == nm and len(pk["nm_index"]) == len(nm)
    # the same reads inside a larger buffer: offsets not starting at 0 are rebased
    pad = 77
    reads2 = np.concatenate([np.full(pad, ord("T"), dtype=np.uint8), reads])
    pk2 = B.pack_reads(reads2, offs + np.uint64(pad))
    assert np.array_equal(pk2["fw3"], pk["fw3"]) and np.array_equal(pk2["read_offsets"], offs) and np.array_equal(pk2["nm_index"], pk["nm_index"])


def test_exception_planes_only_when_needed():
    assert B.Graph.from_fasta(os.path.join(GOLD, "deg_unitig.fa"), 5).info()["has_exceptions"] == 0
    assert B.Graph.from_fasta(os.path.join(GOLD, "deg_unitig_exc.fa"), 5).info()["has_exceptions"] == 1


@pytest.mark.parametrize("name,k,fastq", [("edge_reads.fa", 31, False), ("edge_reads.fa", 25, False), ("edge_reads.fq", 31, True),
                                          ("edge_reads_nonl.fq", 31, True), ("syn_r150.fa", 31, False), ("long_r150.fq", 31, True),
                                          ("deg_reads.fa", 5, False), ("toy_reads.fa", 4, False)])
def test_parser_matches_oracle_getreads(name, k, fastq):
    path = os.path.join(GOLD, name)
    r1, ro1, h1, ho1 = B.load_reads(path, k, fastq)
    r2, ro2, h2, ho2 = oracle_py.parse_file(path, k, fastq)
    assert np.array_equal(ro1, ro2) and np.array_equal(ho1, ho2)
    assert np.array_equal(r1, r2) and np.array_equal(h1, h2)


def _write_tmp(text):
    f = tempfile.NamedTemporaryFile("wb", suffix=".fx", delete=False)
    f.write(text)
    f.close()
    return f.name


@pytest.mark.parametrize("text,fastq", [
    (b"", False), (b"", True), (b">h\n", False), (b">h", False), (b">h\nACGTACGT", False), (b"\n>h\nACGTACGT\n", False),
    (b">a\nACGTACGT\n>b", False), (b">a\nACGTACGT\n>b\n", False), (b">a\nACGT\nACGT\n\n\n>b\nAAAAAAAA\n\n", False),
    (b"ACGTACGT\nACGTACGT\n", False), (b">a\r\nACGTACGT\r\n", False),
    (b"@a\nACGTACGT\n+\nIIIIIIII\n", True), (b"@a\nACGTACGT\n+\nIIIIIIII", True), (b"@a\nACGTACGT\n+\n", True),
    (b"@a\nACGTACGT\n", True), (b"@a\nACGTACGT\n+\nIIIIIIII\n@b\nacgtacgt\n+\nIIIIIIII\n", True),
    (b"@a\nACGTACGT\n+\nIIIIIIII\n\n", True),
])
def test_parser_corner_cases_match_oracle(text, fastq):
    p = _write_tmp(text)
    try:
        a = B.load_reads(p, 5, fastq)
        b = oracle_py.parse_file(p, 5, fastq)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), (text, a, b)
    finally:
        os.unlink(p)


@pytest.mark.parametrize("fastq", [False, True])
def test_parser_random_line_soup_matches_oracle(fastq, tmp_path):
    """Random files built from lines of every kind (sequence lines of all lengths around the 16-byte SIMD step, headers,
    '>' / '@' in odd places, blanks, CR, lower case, other letters, with and without a final newline): the product
    parser (sequential and chunk-parallel, fast path and general machine) == the oracle's ifstream state machine."""
    rng = np.random.default_rng(77 + fastq)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)

    def seq(n):
        return alpha[rng.integers(0, 4, n)].tobytes()

    def line():
        r = rng.integers(0, 20)
        n = int(rng.choice([0, 1, 2, 3, 5, 6, 7, 15, 16, 17, 31, 32, 33, 47, 48, 49, 100]))
        if r < 9:
            return seq(n)
        if r < 13:
            return (b"@" if fastq else b">") + b"h%d" % rng.integers(0, 99)
        if r == 13:
            return b""
        if r == 14:
            return seq(n) + b"\r"
        if r == 15:
            x = bytearray(seq(n + 1)); x[int(rng.integers(0, n + 1))] = int(rng.choice(list(b"NNacgtRX> @+")))
            return bytes(x)
        if r == 16:
            return b">" + seq(n)
        if r == 17:
            return b"+"
        if r == 18:
            return b"I" * n
        return seq(n) + b"N"
    for it in range(120):
        nl = int(rng.integers(0, 40))
        text = b"\n".join(line() for _ in range(nl))
        if rng.integers(0, 2):
            text += b"\n"
        p = str(tmp_path / ("s%d.fx" % it))
        open(p, "wb").write(text)
        want = oracle_py.parse_file(p, 5, fastq)
        for kw in ({}, {"threads": 3, "chunk_bytes": int(rng.choice([1, 9, 40, 200]))}):
            got = B.load_reads(p, 5, fastq, **kw)
            for x, y in zip(got, want):
                assert np.array_equal(x, y), (it, kw, text)


@pytest.mark.parametrize("name,k", [("edge_reads.fa", 31), ("syn_r150.fa", 31), ("deg_reads.fa", 5), ("toy_reads.fa", 4), ("long_r150.fa", 31)])
@pytest.mark.parametrize("chunk", [1, 7, 64, 300, 4096])
def test_chunk_parallel_parser_equals_sequential(name, k, chunk):
    """split_fasta may only cut where an independent reader is in the same state as the sequential one."""
    path = os.path.join(GOLD, name)
    a = B.load_reads(path, k, False, threads=4, chunk_bytes=chunk)
    b = oracle_py.parse_file(path, k, False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("text", [
    b">a\n>b\nACGTACGT\n>c\nACGTACGTAA\n", b"ACGTACGT\n>x\nACGTACGT\n>y\nACGTACGTT\n", b">a\nACGTACGT\n\n>b\nACGTACGTA\n>c\n>d\n>e\nACGTAAAA\n",
    b">a\nACGTACGT\n>b\n>ACGT\n>c\nGGGGGGGG\n\n\n>d\nTTTTTTTT", b"\n\n>a\nACGTACGT\n>b\nCCCCCCCC\n", b">a\r\nACGTACGT\r\n>b\r\nACGTACGG\r\n",
    b">a\nACGT\nACGT\n>b\nAC\nGT\nAC\nGT\nAAAA\n>c\nACGTNNNN\n"])
def test_chunk_parallel_parser_torture(text):
    p = _write_tmp(text)
    try:
        want = oracle_py.parse_file(p, 5, False)
        for chunk in (1, 2, 3, 5, 8, 13, 1000):
            got = B.load_reads(p, 5, False, threads=3, chunk_bytes=chunk)
            for x, y in zip(got, want):
                assert np.array_equal(x, y), (text, chunk)
    finally:
        os.unlink(p)


@pytest.mark.parametrize("n,tail", [(0, b""), (1, b""), (3, b""), (9999, b""), (10000, b""), (10001, b""), (20000, b""), (25003, b""),
                                    (20000, b"@x\nACGTACGTAA\n"), (10002, b"@x\nACGTACGTAA\n+\nIIIIIIIIII"), (30001, b"\n"), (12000, b"@x\nACGTACGTAA\n+\n")])
@pytest.mark.parametrize("chunk", [97, 5000, 1 << 20])
def test_fastq_chunk_parallel_parser_equals_sequential(n, tail, chunk):
    rng = np.random.default_rng(n + len(tail))
    recs = []
    for i in range(n):
        L = int(rng.integers(1, 40))
        seq = bytes(rng.choice(list(b"ACGTNacgt"), L, p=[.24, .24, .24, .24, .02, .005, .005, .005, .005]).astype(np.uint8))
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, seq, b"I" * L))
    p = _write_tmp(b"".join(recs) + tail)
    try:
        a = B.load_reads(p, 5, True, threads=5, chunk_bytes=chunk)
        b = oracle_py.parse_file(p, 5, True)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    finally:
        os.unlink(p)


def test_fastq_phantom_depends_on_batch_boundary():
    """aligner.cpp:51-68: the phantom record appears unless the record count is a multiple of the 10000-read batch."""
    rec = b"@r\nACGTACGTAC\n+\nIIIIIIIIII\n"
    for n in (9999, 10000, 10001, 20000):
        p = _write_tmp(rec * n)
        try:
            a = B.load_reads(p, 5, True)
            b = oracle_py.parse_file(p, 5, True)
            assert len(a[1]) == len(b[1]) and np.array_equal(a[0], b[0])
            assert len(a[1]) - 1 == (n if n % 10000 == 0 else n + 1)
        finally:
            os.unlink(p)


def test_write_records_matches_golden_format():
    """printPath + fwrite sites (aligner.cpp:600-609, alignerGreedy.cpp:406-427), fed with the oracle's paths."""
    case = next(c for c in golden_cases() if c["group"] == "edge" and c["args"][7] == "5")
    reads, roffs, heads, hoffs = B.load_reads(os.path.join(GOLD, "edge_reads.fa"), 31)
    o = oracle_py.Oracle(31, fasta=os.path.join(GOLD, "syn_unitig.fa"))
    paths, poffs, status = o.align(reads, roffs, m=5, effort=2)
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    with tempfile.TemporaryDirectory() as d:
        pf = libc.fopen(os.path.join(d, "paths").encode(), b"wb")
        nf = libc.fopen(os.path.join(d, "na").encode(), b"wb")
        rc = B.lib().bgr_write_records(pf, nf, len(roffs) - 1, heads.ctypes.data, hoffs.ctypes.data, reads.ctypes.data, roffs.ctypes.data,
                                       paths.ctypes.data if len(paths) else None, poffs.ctypes.data)
        libc.fclose(pf)
        libc.fclose(nf)
        assert rc == 0
        assert open(os.path.join(d, "paths"), "rb").read().decode("latin-1") == case["paths"]
        assert open(os.path.join(d, "na"), "rb").read().decode("latin-1") == case["notaligned"]


def test_mapping_fails_loudly_without_gpu():
    if B.device_count() > 0:
        pytest.skip("a GPU is present")
    g = B.Graph.from_fasta(os.path.join(GOLD, "toy_unitig.fa"), 4)
    with pytest.raises(B.BgrError, match="(?i)no HIP device|hip"):
        B.Aligner(g, 0)


def test_structs_of_another_header_version_are_refused(tmp_path):
    """bgr_run_options / bgr_text_batch carry their own size: a caller built against another layout of the header gets BGR_E_ARG instead of having
    fields read past the end of its struct (both structs have grown in every round)."""
    import ctypes as C
    L = B.lib()
    o = B.RunOptions(C.sizeof(B.RunOptions) - 8, 1, 1)
    p = B.Params(0, 2, 2, 0)
    cnt = (C.c_uint64 * 5)()
    secs = C.c_double(0)
    rc = L.bgr_align_all(C.c_void_p(1), C.byref(p), C.byref(o), b"x.fa", str(tmp_path / "p").encode(), str(tmp_path / "n").encode(), cnt, C.byref(secs))
    assert rc == -1 and b"struct_size" in L.bgr_last_error()
    o.struct_size = 0
    assert L.bgr_align_all(C.c_void_p(1), C.byref(p), C.byref(o), b"x.fa", str(tmp_path / "p").encode(), str(tmp_path / "n").encode(), cnt, C.byref(secs)) == -1
    b = B.TextBatch(0)
    assert L.bgr_align_fasta_text(C.c_void_p(1), C.byref(p), C.byref(b)) == -1 and b"struct_size" in L.bgr_last_error()


def test_options_are_named_bounded_and_put_back():
    names = dict(B.option_names())
    assert {"timing", "build_filter", "poison_device_buffers", "test.bases_cap", "test.lanes_on_one_device"} <= set(names)
    assert B.get_option("build_filter") == -1
    with B.options(build_filter=2, timing=1):
        assert B.get_option("build_filter") == 2 and B.get_option("timing") == 1
    assert B.get_option("build_filter") == -1 and B.get_option("timing") == 0
    with pytest.raises(B.BgrError, match="unknown option"):
        B.set_option("no_such_option", 1)
    with pytest.raises(B.BgrError, match="out of range"):
        B.set_option("build_filter", 7)


def test_the_library_reads_no_environment_variables():
    """Options come in through bgr_set_option / --set; getenv appears nowhere in the product's sources."""
    import glob
    for f in glob.glob(os.path.join(ROOT, "bgreat_amd", "csrc", "*")):
        assert "getenv" not in open(f, errors="replace").read(), f
