"""Multi-GPU plumbing of the mapping path: one process per GPU over torch.distributed (backend "nccl" = RCCL on
ROCm; "gloo" on CPU for the world_size-2 tests).  The path shards by reads and has exactly two collectives:

  C1  broadcast of the immutable graph blob from rank 0, once (bgr_graph_blob -> bytes -> every rank's HBM),
  C2  all-reduce(sum) of the aligner.h:68 counters at the end.

Reads never move between ranks.  Rank r maps the contiguous, input-ordered slice shard_range(n, world, r);
concatenating the ranks' outputs in rank order reproduces the reference's `-t 1` byte stream.
"""
import numpy as np

COUNTER_KEYS = ("reads", "no_overlap", "aligned", "not_aligned", "overlaps")


def shard_range(n, world, rank):
    """[lo, hi) of the n input-ordered units owned by `rank` (sizes differ by at most one)."""
    return n * rank // world, n * (rank + 1) // world


def broadcast_graph(graph, dist, device=None):
    """C1.  `graph` is a bgreat_amd.Graph on rank 0 and None elsewhere.  With a CUDA/HIP `device` index the blob is
    broadcast device-to-device and adopted in place (bgr_graph_adopt_device_blob); with device=None (gloo/CPU) the
    bytes travel as a CPU tensor and are re-wrapped with bgr_graph_from_blob.  Returns (graph, keepalive_tensor)."""
    import torch
    import bgreat_amd as B
    rank = dist.get_rank()
    dev = "cpu" if device is None else "cuda:%d" % device
    nbytes = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == 0:
        blob = np.array(graph.blob())
        nbytes[0] = blob.size
    dist.broadcast(nbytes, 0)
    n = int(nbytes.item())
    if rank == 0:
        t = torch.from_numpy(blob).to(dev)
    else:
        t = torch.empty(n, dtype=torch.uint8, device=dev)
    dist.broadcast(t, 0)
    if device is not None:
        torch.cuda.synchronize(device)
    if rank == 0:
        return graph, t
    if device is None:
        return B.Graph.from_blob(t.numpy()), t
    return B.Graph.adopt_device_blob(device, t.data_ptr(), n), t


def reduce_counters(counters, dist, device=None):
    """C2.  Sum the per-rank counter dicts over all ranks."""
    import torch
    dev = "cpu" if device is None else "cuda:%d" % device
    t = torch.tensor([int(counters.get(k, 0)) for k in COUNTER_KEYS], dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    return dict(zip(COUNTER_KEYS, (int(x) for x in t.tolist())))


def max_over_ranks(value, dist, device=None):
    import torch
    dev = "cpu" if device is None else "cuda:%d" % device
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_bytes_in_rank_order(data, dist):
    """Rank 0 receives every rank's byte string, in rank order (the -t 1 stream when each rank holds the formatted
    records of its shard); other ranks get None."""
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(data, out, dst=0)
    return b"".join(out) if dist.get_rank() == 0 else None
