"""The launch geometry (bgreat_amd/csrc/launch_plan.h, exported as bgr_plan_launch) is a pure function of numbers -- graph header, device, tuning,
batch -- so it is swept here on the CPU: synthetic headers with mean unitig lengths from 40 to 10^5 bases, slot fills from 1.0 to 3.5, key tables from
a kilobyte to a gigabyte, reads of 32 bases to 100 kb, all three modes.  The plan must never fail on a batch that can be mapped, never put more LDS on
a CU than it has, never stage a table it cannot hold, and the five bench workloads must keep the geometry their numbers were measured with."""
import itertools

import numpy as np
import pytest

import bgreat_amd as B

LDS = 160 * 1024
CUS = 256


def plan(mean_unitig=75, fill=1.5, table_bytes=69000, n_unitigs=99000, k=31, mode=B.MODE_GREEDY, m=2, L=150, n=5_000_000, **kw):
    return B.plan_launch(k=k, slot_fill_x100=int(fill * 100), table_bytes=table_bytes, graph_bases=2 * mean_unitig * n_unitigs, n_unitigs=n_unitigs,
                         max_unitig_len=kw.pop("max_unitig_len", 20 * mean_unitig), mode=mode, max_mismatch=m, max_read_len=L, n_reads=n, total_bases=n * L,
                         anchors=int(mode == B.MODE_ANCHORS), anchor_levels=kw.pop("anchor_levels", 7), **kw)


def check(p, table_bytes, n_items_min=1):
    used = [v for v in p.values() if isinstance(v, dict) and v["used"]]
    assert used
    for q in used:
        assert 1 <= q["waves_per_block"] <= 16 and q["blocks"] >= 1
        assert q["lds_bytes"] <= LDS
        if q["table_staged"]:
            assert q["lds_bytes"] >= table_bytes + 512          # the table is really in there ...
            assert table_bytes + 512 <= LDS - 64                # ... and fits a CU
        per_cu = -(-q["blocks"] // CUS)                          # workgroups a CU holds when the grid is CUs x b
        if q["blocks"] >= CUS:
            assert per_cu * q["lds_bytes"] <= LDS, q
    assert p["memo_cap"] & (p["memo_cap"] - 1) == 0
    assert p["deep_scratch_bytes"] <= (1 << 30) + (64 << 20)


@pytest.mark.parametrize("mode", [B.MODE_GREEDY, B.MODE_EXHAUSTIVE, B.MODE_ANCHORS])
def test_sweep_of_synthetic_headers_never_fails_and_never_overfills_a_cu(mode):
    n_checked = 0
    for mean_unitig, fill, table_bytes, L, n in itertools.product([40, 75, 300, 2000, 100_000], [1.0, 1.5, 2.2, 3.5],
                                                                  [1024, 20_000, 69_000, 71_000, 150_000, 155_000, 400_000, 40_000_000, 1_000_000_000],
                                                                  [32, 100, 150, 250, 479, 481, 1000, 5000, 20_000], [1, 63, 4096, 262_144, 5_000_000]):
        if n * L >= 1 << 30:
            continue   # (beyond one launch's 32-bit path arena: refused, see below)
        for m in (0, 2, 5, 255):
            p = plan(mean_unitig, fill, table_bytes, mode=mode, m=m, L=L, n=n)
            check(p, table_bytes)
            n_checked += 1
    assert n_checked > 10_000


def test_reads_beyond_the_lds_layouts():
    """greedy / anchors: a read must fit the per-wave LDS staging (~30 kb); exhaustive mode sends such batches through its last pass alone (~160 kb)."""
    with pytest.raises(B.BgrError, match="read too long"):
        plan(L=60_000, n=10)
    p = plan(mode=B.MODE_EXHAUSTIVE, L=60_000, n=10)
    assert p["deep_only"] and p["last"]["used"]
    check(p, 69000)
    with pytest.raises(B.BgrError, match="read too long"):
        plan(mode=B.MODE_EXHAUSTIVE, L=400_000, n=10)
    with pytest.raises(B.BgrError, match="batch too large"):
        plan(L=150, n=40_000_000)


def test_explicit_tuning_is_honoured_or_shrunk_to_fit():
    for waves, bpc, stage in itertools.product([0, 1, 4, 7, 16], [0, 1, 2, 6], [0, 1, 2]):
        for table_bytes in (1024, 69_000, 155_000, 10_000_000):
            for mode in (B.MODE_GREEDY, B.MODE_EXHAUSTIVE, B.MODE_ANCHORS):
                p = plan(table_bytes=table_bytes, mode=mode, cfg_waves=waves, cfg_blocks_per_cu=bpc, cfg_lds_mphf=stage)
                check(p, table_bytes)
                if stage == 1 or mode == B.MODE_ANCHORS or table_bytes > LDS:
                    assert not any(v["table_staged"] for v in p.values() if isinstance(v, dict))
                if table_bytes > LDS and waves:   # staging asked for a table no CU can hold: the workgroup keeps the waves asked for (it used to shrink to one wave)
                    q = p["greedy16"] if mode == B.MODE_GREEDY else p["exhaustive8"] if mode == B.MODE_EXHAUSTIVE else p["anchors4"]
                    assert q["waves_per_block"] == min(waves, 16) or q["lds_bytes"] > LDS // 2, (waves, bpc, stage, table_bytes, q)


def test_smaller_devices():
    """A device with 64 KB of LDS per CU and 104 CUs (an MI200-class part): the same rules, other numbers."""
    global LDS, CUS
    old = LDS, CUS
    LDS, CUS = 64 * 1024, 104
    try:
        for table_bytes, L, mode in itertools.product([1024, 30_000, 69_000, 10_000_000], [100, 250, 2000], [B.MODE_GREEDY, B.MODE_EXHAUSTIVE, B.MODE_ANCHORS]):
            p = plan(table_bytes=table_bytes, mode=mode, L=L, n=200_000, num_cus=104, lds_per_cu=64 * 1024)
            check(p, table_bytes)
    finally:
        LDS, CUS = old


def test_the_bench_workloads_keep_their_measured_geometry():
    """What profiles/r04_* and r05_* were measured with (MI355X: 256 CUs, 160 KB LDS per CU)."""
    # default line: E. coli-scale graph (66 k keys at 1.046 slots per key: 69 008 bytes), 5 M x 150 bp: two staged workgroups of 16 waves per CU
    p = plan(table_bytes=69_008, n_unitigs=98_935)
    assert p["greedy16"] == dict(used=True, blocks=512, waves_per_block=16, lds_bytes=p["greedy16"]["lds_bytes"], table_staged=True)
    # configs[1]: 10 k unitigs, 100 bp, 1 M reads per launch
    p = plan(table_bytes=7_200, n_unitigs=10_000, L=100, n=1_000_000)
    assert p["greedy16"]["table_staged"] and p["greedy16"]["blocks"] * p["greedy16"]["waves_per_block"] == 256 * 32
    # configs[3] graph: 2.6 M keys at 1.8 slots per key: probed in L2 behind the minimizer filter, 32 waves per CU
    p = plan(table_bytes=4_700_000 * 4, n_unitigs=3_960_000, mean_unitig=88)
    assert not p["greedy16"]["table_staged"] and p["greedy16"]["blocks"] * p["greedy16"]["waves_per_block"] == 256 * 32
    # configs[4]: 4-allele graph, 250 bp, m = 5: level search behind the eight-reads-per-wave pass, 16 levels per side, 24 waves per CU
    p = plan(table_bytes=30_000_000, n_unitigs=5_900_000, mean_unitig=47, fill=3.4, mode=B.MODE_EXHAUSTIVE, m=5, L=250, n=2_000_000)
    assert p["level_search"] and p["x4_levels"] == 16 and p["exhaustive8"]["used"] and not p["exhaustive8"]["table_staged"]
    assert p["exhaustive8"]["blocks"] * p["exhaustive8"]["waves_per_block"] == 256 * 24
    assert p["last"]["used"] and p["deep_scratch_bytes"] < 64 << 20      # (round 4: ~170 MB per exhaustive aligner)
    # exhaustive on the E. coli-scale graph, m = 2: short walks: 8 levels per side, one staged workgroup of 16 waves
    p = plan(table_bytes=69_008, n_unitigs=98_935, mode=B.MODE_EXHAUSTIVE)
    assert p["x4_levels"] == 8 and p["exhaustive8"]["table_staged"] and not p["level_search"]
