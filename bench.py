#!/usr/bin/env python3
"""bench.py -- Mreads/s of the BGREAT mapping hot path (greedy, k=31, 150 bp, m=2) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one launch of the mapping kernel over one batch of synthetic reads that is already resident in HBM
(ASCII reads + offsets, as the C-ABI's bgr_align_device takes them).  Workload = BASELINE.json configs[2]
("Synthetic 50M x 150 bp reads, k=31, m=2, E.coli-scale graph (~100k unitigs), greedy"): with the defaults
(10 steps x 5M reads) one run maps exactly that read set on one GPU.  With N>1 every rank maps its own
equally sized shard (weak scaling); the read-only graph blob is built on rank 0 and broadcast once over RCCL.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads-per-step", type=int, default=5_000_000, help="reads per launch per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--mismatch", type=int, default=2)
    ap.add_argument("--effort", type=int, default=2)
    ap.add_argument("--genome", type=int, default=4_600_000, help="synthetic genome length (E. coli scale)")
    ap.add_argument("--site-spacing", type=int, default=140)
    ap.add_argument("--alleles", type=int, default=2)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="reads timed on the host CPU (0 disables)")
    ap.add_argument("--alg-sample", type=int, default=20_000, help="reads used to count ALGORITHMIC bytes/read with the oracle")
    ap.add_argument("--lds-mphf", type=int, default=0, help="0 auto, 1 HBM/L2 only, 2 force LDS staging")
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for input generation and the CPU baseline")
    ap.add_argument("--workload", default="ecoli", choices=["ecoli", "small", "chr1", "branchy"],
                    help="ecoli = BASELINE configs[2] (default, the metric's config); small = configs[1]; chr1 = configs[3] graph scale; "
                         "branchy = configs[4] (exhaustive, m=5, 250 bp)")
    ap.add_argument("--exhaustive", action="store_true")
    ap.add_argument("--anchors", action="store_true", help="-G: greedy mapping from k-mer anchors (diagnostic; not the headline metric)")
    ap.add_argument("--gamma", type=float, default=0.0, help="MPHF positions per key and level (0 = library default)")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    args = ap.parse_args()

    presets = {  # SURVEY.md 8d synthetic inputs; explicit flags given on the command line win over the preset
        "small": dict(genome=250_000, site_spacing=75, alleles=2, read_len=100, reads_per_step=1_000_000),
        "chr1": dict(genome=230_000_000, site_spacing=175, alleles=2),
        "branchy": dict(genome=50_000_000, site_spacing=36, alleles=4, read_len=250, mismatch=5, reads_per_step=2_000_000, exhaustive=True),
    }
    for key, val in presets.get(args.workload, {}).items():
        if getattr(args, key) == ap.get_default(key):
            setattr(args, key, val)
    mode = 1 if args.exhaustive else (2 if args.anchors else 0)

    import torch
    import bgreat_amd as B
    from tools.synth import Synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BGR_BENCH_BACKEND=gloo: rehearsal of the N>1 code path on a box with fewer GPUs than ranks (ranks share
        # the visible devices, collectives run on CPU tensors).  Never used for a reported number.
        backend = os.environ.get("BGR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
            local_rank %= max(1, torch.cuda.device_count())
    rehearsal = world > 1 and os.environ.get("BGR_BENCH_BACKEND", "nccl") != "nccl"
    dev = local_rank
    coll_dev = None if rehearsal else dev
    torch.cuda.set_device(dev)
    if B.device_count() < 1:
        sys.exit("bench.py: no HIP device (the mapping path has no CPU fallback)")

    K, W, R, L = args.steps, args.warmup, args.reads_per_step, args.read_len
    seed_graph, seed_reads = 20261003, 77

    # ---- graph: built on rank 0, blob broadcast once over RCCL/xGMI, adopted in place by the other ranks ----
    t0 = time.time()
    syn = Synth(args.genome, args.site_spacing, args.alleles, args.k, seed_graph)  # every rank needs the genome to draw its reads
    graph_info = None
    g = None
    if rank == 0:
        seqs, offs = syn.unitigs()
        tb = time.time()
        g = B.Graph.build(args.k, seqs, offs, args.gamma, anchors=(mode == 2))
        graph_info = g.info()
        log("index build: %.2fs on the host (%d unitigs)" % (time.time() - tb, graph_info["n_unitigs"]))
    blob_keepalive = None
    if world > 1:
        from bgreat_amd import dist as D
        g, blob_keepalive = D.broadcast_graph(g, dist, device=coll_dev)  # C1: the only data-path collective; reads never move
    al = B.Aligner(g, dev)
    al.configure(args.waves, args.blocks_per_cu, args.lds_mphf)
    if rank == 0:
        log("graph: %s  (%.1fs)" % (graph_info, time.time() - t0))

    # ---- reads: rank r owns global reads [r*K*R, (r+1)*K*R); generated on the host, parked in HBM -----------
    t0 = time.time()
    ncpu = min(len(os.sched_getaffinity(0)), args.cpu_threads)  # the GPU box gives one GPU a 16-core share
    offs_np = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
    offs_t = torch.from_numpy(offs_np.view(np.int64)).to("cuda")
    batches = []
    first_host = None
    for s in range(K):
        arr, _ = syn.reads((rank * K + s) * R, R, L, args.mismatch, seed_reads, threads=max(1, ncpu // max(1, min(world, 8))))
        if s == 0 and rank == 0:
            first_host = arr[: max(args.cpu_sample, args.alg_sample) * L].copy()
        batches.append(torch.from_numpy(arr).to("cuda"))
        del arr
    torch.cuda.synchronize()
    if rank == 0:
        log("reads: %d x %d x %d bp per GPU resident in HBM (%.1fs)" % (K, R, L, time.time() - t0))

    def step(i):
        b = batches[i % K]
        al.align_device(b.data_ptr(), offs_t.data_ptr(), R, R * L, L, m=args.mismatch, effort=args.effort, mode=mode)

    for i in range(W):
        step(i)
    al.sync()
    al.reset_kernel_time()
    al.reset_counters()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for i in range(K):
        step(i)
    al.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        elapsed = D.max_over_ranks(elapsed, dist, device=coll_dev)
    launches, kernel_ms = al.kernel_time()
    counters = al.counters()
    if dist is not None:  # C2: sum the aligner.h:68 counters over ranks
        counters = D.reduce_counters(counters, dist, device=coll_dev)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    total_reads = world * K * R
    value = total_reads / elapsed / 1e6
    avg_kernel_ms = kernel_ms / max(1, launches)

    # ---- ALGORITHMIC bytes per read: SURVEY.md 8d formula, counted by the oracle on a sample of this workload ----
    import oracle_py
    seqs, offs = syn.unitigs()
    orc = oracle_py.Oracle(args.k, seqs, offs, anchors=(mode == 2))
    ns = min(args.alg_sample, R)
    s_reads = first_host[: ns * L]
    s_offs = np.arange(ns + 1, dtype=np.uint64) * np.uint64(L)
    p2, po2, st2 = orc.align(s_reads, s_offs, m=args.mismatch, effort=args.effort, mode=mode)
    alg_bytes_per_read = orc.alg_bytes() / ns
    work = orc.work()
    # parity of the same sample through the GPU path (outside the timed region)
    p1, po1, st1 = al.align(s_reads, s_offs, m=args.mismatch, effort=args.effort, mode=mode)
    parity_ok = bool(np.array_equal(p1, p2) and np.array_equal(po1, po2) and np.array_equal(st1, st2))
    achieved = alg_bytes_per_read * R / (avg_kernel_ms / 1e3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("reads_per_launch") == R and tj.get("read_len") == L and tj.get("workload") == args.workload and mode == 0:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "kernel": ("bgr_align_greedy_kernel", "bgr_align_exhaustive_dp_kernel" if al.launch_info().get("level_search") else "bgr_align_exhaustive_kernel", "bgr_align_anchors_kernel")[mode], "avg_launch_ms": round(avg_kernel_ms, 4), "launches": launches,
                "alg_bytes_per_read": round(alg_bytes_per_read, 1), "reads_per_launch": R}

    # ---- CPU baseline: the compiled reference (oracle/_ref/bgreat -t cores) on a bounded sample, N=1 only -------
    cpu = None
    if world == 1 and args.cpu_sample > 0 and mode == 0:
        nc = min(args.cpu_sample, R)
        ref = os.path.join(ROOT, "oracle", "_ref", "bgreat")
        d = tempfile.mkdtemp(prefix="bgr_cpu_")
        try:
            syn.write_unitigs(os.path.join(d, "u.fa"))
            syn.write_reads(os.path.join(d, "r.fa"), 0, nc, L, args.mismatch, seed_reads)
            cores = min(ncpu, 255)
            if os.path.exists(ref):
                cmd, kind = [ref], "reference"
            else:
                cmd, kind = [os.path.join(ROOT, "oracle", "bgreat_oracle")], "port"
            cmd += ["-r", os.path.join(d, "r.fa"), "-k", str(args.k), "-g", os.path.join(d, "u.fa"), "-m", str(args.mismatch), "-e", str(args.effort), "-t", str(cores)]
            t1 = time.perf_counter()
            subprocess.run(cmd, cwd=d, check=True, stdout=subprocess.DEVNULL)
            wall = time.perf_counter() - t1
            # index-only run (empty read file) to subtract the one-off indexing from the mapping time
            open(os.path.join(d, "empty.fa"), "w").close()
            cmd_i = list(cmd)
            cmd_i[cmd_i.index("-r") + 1] = os.path.join(d, "empty.fa")
            d2 = os.path.join(d, "idx")
            os.makedirs(d2)
            t1 = time.perf_counter()
            subprocess.run(cmd_i, cwd=d2, check=True, stdout=subprocess.DEVNULL)
            wall_idx = time.perf_counter() - t1
            map_s = max(1e-6, wall - wall_idx)
            # parity at scale: GPU records == reference records as a multiset (-t N interleaves records)
            ref_paths = open(os.path.join(d, "paths"), "rb").read().split(b"\n")
            c_reads = first_host[: nc * L]
            c_offs = np.arange(nc + 1, dtype=np.uint64) * np.uint64(L)
            gp, gpo, gst = al.align(c_reads, c_offs, m=args.mismatch, effort=args.effort)
            ref_map = {}
            for h, p in zip(ref_paths[0::2], ref_paths[1::2]):
                ref_map[h] = p
            n_al = int((gpo[1:] > gpo[:-1]).sum())
            ok = n_al == len(ref_map)
            if ok:
                idx = np.nonzero(gpo[1:] > gpo[:-1])[0]
                for i in idx[:: max(1, len(idx) // 200000)]:
                    want = ref_map.get(b">r%d" % i)
                    got = b"".join(b"%d." % v for v in gp[int(gpo[i]): int(gpo[i + 1])])
                    if want != got:
                        ok = False
                        break
            cpu = {"value": round(nc / map_s / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": kind,
                   "sample": "first %d reads of step 0 of this workload, %s -t %d, wall %.2fs minus %.2fs index-only run" % (nc, os.path.basename(cmd[0]), cores, wall, wall_idx),
                   "gpu_matches_cpu_records": bool(ok)}
        finally:
            shutil.rmtree(d, ignore_errors=True)

    out = {
        "metric": "Mreads/s aligned (k=%d, %dbp, m=%d)" % (args.k, L, args.mismatch), "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s: synthetic %.1fM x %d bp reads per GPU (%d steps x %d), k=%d, m=%d, effort=%d, %s, graph of %d unitigs (genome %d bp, %d alleles every ~%d bp)"
                   % ({"ecoli": "BASELINE configs[2]", "small": "BASELINE configs[1]", "chr1": "BASELINE configs[3] graph scale, one GPU's share", "branchy": "BASELINE configs[4] graph, one GPU's share"}[args.workload],
                      K * R / 1e6, L, K, R, args.k, args.mismatch, args.effort, ("greedy", "exhaustive", "greedy from k-mer anchors (-G)")[mode], graph_info["n_unitigs"], args.genome, args.alleles, args.site_spacing),
                   "reads_per_step_per_gpu": R, "read_len": L, "k": args.k, "m": args.mismatch, "effort": args.effort,
                   "parallelism": "reads sharded over %d GPU(s); graph blob broadcast once" % world, "launch": al.launch_info()},
        "roofline": roofline, "cpu_baseline": cpu,
        "counters": counters, "parity_sample": {"reads": ns, "gpu_equals_oracle": parity_ok},
        "oracle_work_per_read": {k: round(v / ns, 2) for k, v in work.items() if k not in ("reads",)},
    }
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
