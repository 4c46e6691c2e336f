"""GPU parity at the graph scales of BASELINE.json configs[3] and configs[4] (the read counts there are 8-GPU totals; the
code paths a graph of that size takes are what needs covering): a cascade too large for LDS staging (probed in L2/HBM with
the look-ahead load), a 0.7 GB blob, 32-bit sequence addressing near its range, the level search without staging.
Each test checks the default launch geometry really is the one claimed, determinism, shard invariance, agreement between the
kernel formulations, and sampled rows against the oracle (building the oracle's own BooPHF over the graph takes ~15-30 s)."""
import numpy as np
import pytest

import bgreat_amd as B
import oracle_py
from tools.synth import Synth

pytestmark = pytest.mark.gpu


def _rows(p, po, idx):
    return [p[int(po[i]):int(po[i + 1])].tolist() for i in idx]


def _check_sample_against_oracle(k, seqs, offs, reads, roffs, L, p, po, st, idx, **kw):
    o = oracle_py.Oracle(k, seqs, offs)
    sub = np.concatenate([reads[int(roffs[i]):int(roffs[i + 1])] for i in idx])
    soffs = np.arange(len(idx) + 1, dtype=np.uint64) * np.uint64(L)
    p2, po2, st2 = o.align(sub, soffs, **kw)
    assert np.array_equal(st[idx], st2), np.nonzero(st[idx] != st2)[0][:10]
    assert _rows(p, po, idx) == _rows(p2, po2, range(len(idx)))
    return st2


def test_chr1_scale_graph_greedy():
    """configs[3] graph: 230 Mb genome, a 2-allele site every ~175 bp -> ~4 M unitigs, blob ~0.7 GB; 1 M x 150 bp, m=2."""
    k, L, n = 31, 150, 1_000_000
    s = Synth(230_000_000, 175, 2, k, 20261003)
    seqs, offs = s.unitigs()
    g = B.Graph.build(k, seqs, offs)
    info = g.info()
    assert info["n_unitigs"] > 3_500_000 and info["blob_bytes"] > 500_000_000 and info["mphf_bytes"] > 1_000_000
    al = B.Aligner(g, 0)
    reads, roffs = s.reads(0, n, L, 2, 77)
    p1, po1, st1 = al.align(reads, roffs, m=2, effort=2)
    li = al.launch_info()
    assert li["mphf_in_lds"] is False and li["four_reads_per_wave"] is True, li   # the default geometry for this graph
    c1 = al.counters()
    assert c1["reads"] == n and c1["aligned"] + c1["no_overlap"] + c1["not_aligned"] == n and c1["aligned"] > 0.7 * n
    assert int(((st1 & 3) == 2).sum()) == c1["aligned"] and int(((st1 & 3) == 0).sum()) == c1["no_overlap"]
    # determinism
    al.reset_counters()
    p2, po2, st2 = al.align(reads, roffs, m=2, effort=2)
    assert np.array_equal(p1, p2) and np.array_equal(po1, po2) and np.array_equal(st1, st2) and al.counters() == c1
    # shard invariance: ragged pieces, concatenated in order
    cuts = [0, 1, 333_333, 333_336, 700_001, n]
    ps, sts, lens = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        pa, poa, sta = al.align(reads[a * L:b * L], roffs[a:b + 1] - roffs[a], m=2, effort=2)
        ps.append(pa)
        sts.append(sta)
        lens.append(np.diff(poa.astype(np.int64)))
    assert np.array_equal(np.concatenate(ps), p1) and np.array_equal(np.concatenate(sts), st1)
    assert np.array_equal(np.concatenate(lens), np.diff(po1.astype(np.int64)))
    # the general kernel alone (one read per wave) gives the same rows
    al.set_knob(B.KNOB_GREEDY_FAST, 1)
    p3, po3, st3 = al.align(reads, roffs, m=2, effort=2)
    assert al.launch_info()["four_reads_per_wave"] is False and al.launch_info()["mphf_in_lds"] is False
    assert np.array_equal(p1, p3) and np.array_equal(po1, po3) and np.array_equal(st1, st3)
    # 5 000 sampled rows against the oracle
    idx = np.sort(np.random.default_rng(3).choice(n, size=5000, replace=False))
    st_o = _check_sample_against_oracle(k, seqs, offs, reads, roffs, L, p1, po1, st1, idx, m=2, effort=2)
    assert (st_o & 4).any() and ((st_o & 3) == 0).any()   # the sample covers reverse-complement answers and no-anchor reads


def test_branchy_graph_exhaustive_level_search_without_staging():
    """configs[4] graph: 50 Mb genome, 4 alleles every ~36 bp (~7 M unitigs); 250 bp reads, m=5, -b.  The cascade does not fit
    LDS, so the level search runs as bgr_align_exhaustive_dp_kernel<false>; what it cannot hold goes through the depth-first
    passes.  Both formulations, forced, must agree with each other and with the oracle."""
    k, L, n = 31, 250, 200_000
    s = Synth(50_000_000, 36, 4, k, 20261003)
    seqs, offs = s.unitigs()
    g = B.Graph.build(k, seqs, offs)
    al = B.Aligner(g, 0)
    reads, roffs = s.reads(0, n, L, 5, 77)
    p1, po1, st1 = al.align(reads, roffs, m=5, mode=B.MODE_EXHAUSTIVE)
    li = al.launch_info()
    assert li["level_search"] is True and li["mphf_in_lds"] is False, li   # what the library picks for this graph
    c1 = al.counters()
    assert c1["reads"] == n and c1["aligned"] + c1["not_aligned"] == n and c1["aligned"] > 0.5 * n
    assert c1["overlaps"] == n * (L - (k - 1) + 1)
    al.reset_counters()
    p2, po2, st2 = al.align(reads, roffs, m=5, mode=B.MODE_EXHAUSTIVE)
    assert np.array_equal(p1, p2) and np.array_equal(po1, po2) and np.array_equal(st1, st2) and al.counters() == c1
    # the depth-first formulation on a part of the batch (it is several times slower on this graph)
    h = 60_000
    al.set_knob(B.KNOB_EXH_SEARCH, B.SEARCH_DEPTH_FIRST)
    p3, po3, st3 = al.align(reads[:h * L], roffs[:h + 1], m=5, mode=B.MODE_EXHAUSTIVE)
    assert al.launch_info()["level_search"] is False
    assert np.array_equal(p3, p1[:int(po1[h])]) and np.array_equal(po3, po1[:h + 1]) and np.array_equal(st3, st1[:h])
    idx = np.sort(np.random.default_rng(4).choice(n, size=3000, replace=False))
    _check_sample_against_oracle(k, seqs, offs, reads, roffs, L, p1, po1, st1, idx, m=5, mode=1)


def test_largest_single_launch():
    """One launch addresses its path arena and its planes with 32 bits: 2 x (bases + 16 x reads) < 2^32 - 2^28.  A batch just below that -- 12 M x 150 bp,
    read rows, arena offsets and plane words near the top of their range -- through bgr_align_device + bgr_aligner_fetch: every row equal to what two
    launches of half the batch give, a sample of rows at both ends and in the middle equal to the oracle, counters consistent; one read more than the
    limit admits is refused."""
    k, L, n = 31, 150, 12_000_000
    assert 2 * (n * L + 16 * n) < 2**32 - 2**28 <= 2 * ((n + 800_000) * L + 16 * (n + 800_000))
    s = Synth(4_600_000, 140, 2, k, 20261003)
    seqs, offs = s.unitigs()
    g = B.Graph.build(k, seqs, offs)
    al = B.Aligner(g, 0)
    reads, roffs = s.reads(0, n, L, 2, 4711, threads=16)
    dr, do = B.DeviceBuffer(0, reads), B.DeviceBuffer(0, roffs)
    al.align_device(dr.data_ptr(), do.data_ptr(), n, n * L, L, m=2, effort=2)
    p1, po1, st1 = al.fetch(n, 4 * n + 1024)
    c1 = al.counters()
    assert c1["reads"] == n and c1["aligned"] + c1["no_overlap"] + c1["not_aligned"] == n and c1["aligned"] > 0.8 * n
    assert int(((st1 & 3) == 2).sum()) == c1["aligned"] and po1[n] == len(p1)
    # the same reads in two launches
    h = n // 2
    parts = []
    for a, b in ((0, h), (h, n)):
        sub_offs = B.DeviceBuffer(0, roffs[a:b + 1] - roffs[a])
        al.align_device(dr.data_ptr() + a * L, sub_offs.data_ptr(), b - a, (b - a) * L, L, m=2, effort=2)
        parts.append(al.fetch(b - a, 4 * (b - a) + 1024))
        sub_offs.free()
    assert np.array_equal(np.concatenate([parts[0][0], parts[1][0]]), p1) and np.array_equal(np.concatenate([parts[0][2], parts[1][2]]), st1)
    assert np.array_equal(np.concatenate([np.diff(parts[0][1].astype(np.int64)), np.diff(parts[1][1].astype(np.int64))]), np.diff(po1.astype(np.int64)))
    idx = np.concatenate([np.arange(0, 2000), np.arange(h - 1000, h + 1000), np.arange(n - 2000, n)])
    _check_sample_against_oracle(k, seqs, offs, reads, roffs, L, p1, po1, st1, idx, m=2, effort=2)
    # beyond the limit of one launch: refused (bgr_align_batch would cut such a batch; the device-resident form says so)
    with pytest.raises(B.BgrError):
        al.align_device(dr.data_ptr(), do.data_ptr(), n, 2**31 + 2**28, L, m=2, effort=2)
    dr.free(); do.free()
