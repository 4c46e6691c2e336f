"""The N>1 plumbing (bgreat_amd/dist.py) on CPU: two processes over gloo.  The mapping itself needs a GPU, so the
per-shard mapper here is the oracle (checker role only); what is under test is sharding + the two collectives:
C1 blob broadcast (bytes arrive intact and re-open as the same graph), C2 counter all-reduce, and the property that
rank-ordered concatenation of per-shard outputs equals the unsharded `-t 1` stream."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bgreat_amd as B
        from bgreat_amd import dist as D
        import oracle_py
        from tools.synth import Synth
        s = Synth(40000, 75, 2, 31, 9)
        seqs, offs = s.unitigs()
        g0 = B.Graph.build(31, seqs, offs) if rank == 0 else None
        g, keep = D.broadcast_graph(g0, dist, device=None)                      # C1
        info = g.info()
        ref = B.Graph.build(31, seqs, offs)                                      # every rank can rebuild to compare
        assert info == ref.info()
        assert np.array_equal(np.array(g.blob()), np.array(ref.blob()))
        n, L = 3000, 150
        reads, roffs = s.reads(0, n, L, 3, 11)
        lo, hi = D.shard_range(n, world, rank)
        o = oracle_py.Oracle(31, seqs, offs)
        p, po, st = o.align(reads[lo * L: hi * L], roffs[lo: hi + 1] - roffs[lo], m=2, effort=2)
        rec = b"".join(b">r%d\n%s\n" % (lo + i, b"".join(b"%d." % v for v in p[int(po[i]): int(po[i + 1])])) for i in range(hi - lo) if po[i + 1] > po[i])
        allrec = D.gather_bytes_in_rank_order(rec, dist)
        cnt = D.reduce_counters(o.counters(), dist)                              # C2
        t = D.max_over_ranks(1.0 + rank, dist)
        if rank == 0:
            o2 = oracle_py.Oracle(31, seqs, offs)
            p2, po2, st2 = o2.align(reads, roffs, m=2, effort=2)
            want = b"".join(b">r%d\n%s\n" % (i, b"".join(b"%d." % v for v in p2[int(po2[i]): int(po2[i + 1])])) for i in range(n) if po2[i + 1] > po2[i])
            q.put(("ok", allrec == want, cnt == o2.counters(), t == float(world), [D.shard_range(10, 3, r) for r in range(3)]))
    except Exception as e:  # noqa: BLE001
        q.put(("err", rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "rank failed"
    res = q.get(timeout=10)
    assert res[0] == "ok", res
    assert res[1] and res[2] and res[3]
    assert res[4] == [(0, 3), (3, 6), (6, 10)]
