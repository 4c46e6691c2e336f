#!/usr/bin/env python3
"""bench.py -- Mreads/s of the BGREAT mapping hot path (greedy, k=31, 150 bp, m=2) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one mapping launch (bgr_align_device of the C-ABI) over one batch of synthetic reads that is already resident
in HBM as ASCII (what the reference's getReads hands over): the pre-pass that packs the reads to 2 bits and the passes of
the mapping kernels, enqueued back to back on one stream.  Workload = BASELINE.json configs[2] ("Synthetic 50M x 150 bp
reads, k=31, m=2, E.coli-scale graph (~100k unitigs), greedy"): with the defaults (10 steps x 5M reads) one run maps
exactly that read set on one GPU.  With N>1 every rank maps its own equally sized shard (weak scaling); the read-only
graph blob is built on rank 0 and broadcast once over RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement).  Besides `value` (device-resident) it carries
  roofline       algorithmic bytes (SURVEY 8d formula, oracle counter mode) / launch time, plus what bounds the launch in
                 fact: HBM traffic measured in this run (rocprofv3 PMC passes over a short child run of this script) and
                 the VALU issue fraction
  pcie_inclusive pinned host buffers -> bgr_align_batch (H2D + launch + CSR + D2H), same batch size
  e2e            bgr_align_all: read file -> paths / notAligned.fa (parse, pack, GPU, format, write; index build excluded)
  cpu_baseline   the compiled reference (oracle/_ref/bgreat -t cores) on a bounded sample, rank 0, N=1 only
"""
import argparse
import csv
import ctypes
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLOCK_GHZ = 2.4         # hipDeviceAttributeClockRate on the box; profiles/r02_valu_rates.txt is priced at it
N_SIMD = 1024           # 256 CUs x 4


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
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
    ap.add_argument("--cpu-sample-t1", type=int, default=100_000, help="reads of the `-t 1` leg of the CPU baseline")
    ap.add_argument("--alg-sample", type=int, default=20_000, help="reads used to count ALGORITHMIC bytes/read with the oracle")
    ap.add_argument("--lds-mphf", type=int, default=0, help="0 auto, 1 HBM/L2 only, 2 force LDS staging")
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for input generation, the e2e leg and the CPU baseline")
    ap.add_argument("--workload", default="ecoli", choices=["ecoli", "small", "chr1", "branchy"],
                    help="ecoli = BASELINE configs[2] (default, the metric's config); small = configs[1]; chr1 = configs[3] graph scale; "
                         "branchy = configs[4] (exhaustive, m=5, 250 bp)")
    ap.add_argument("--exhaustive", action="store_true")
    ap.add_argument("--anchors", action="store_true", help="-G: greedy mapping from k-mer anchors (diagnostic; not the headline metric)")
    ap.add_argument("--gamma", type=float, default=0.0, help="overlap key table slots per key (0 = library default)")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--no-kernel-events", action="store_true", help="diagnostic: no HIP events around the kernels of a mapping launch (BGR_KNOB_KERNEL_EVENTS 0): what they cost a small launch; the roofline's per-kernel durations are then missing")
    ap.add_argument("--prepass", action="store_true", help="greedy: 2-bit planes by a pre-pass kernel (rounds 2-4) instead of staging from the characters inside the mapping kernels (diagnostic, A/B)")
    ap.add_argument("--general-kernel-only", action="store_true", help="greedy: skip the eight-reads-per-wave passes (diagnostic)")
    ap.add_argument("--exh-first-pass-off", action="store_true", help="exhaustive: skip the eight-reads-per-wave pass, every read goes through the search kernels (diagnostic)")
    ap.add_argument("--anc-first-pass-off", action="store_true", help="anchors: skip the several-reads-per-wave pass (diagnostic)")
    ap.add_argument("--exh-search", type=int, default=0, help="exhaustive: 0 = choose, 1 = depth-first search, 2 = level search (diagnostic)")
    ap.add_argument("--exh-frame-cap", type=int, default=0, help="exhaustive: search frames / levels per wave (a tiny cap pushes reads through the later passes; diagnostic)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic and the issue fractions become null)")
    ap.add_argument("--pmc-steps", type=int, default=3)
    ap.add_argument("--e2e-reads", type=int, default=100_000_000, help="reads of the end-to-end leg per GPU (0 disables); 100 M x 150 bp = a 16 GB FASTA file: "
                                                                          "at ~170 Mreads/s a smaller file mostly measures the start-up (page-locked staging buffers, device buffers)")
    ap.add_argument("--pcie-steps", type=int, default=3, help="timed bgr_align_batch calls of the PCIe-inclusive leg (0 disables)")
    ap.add_argument("--sorted-reads", action="store_true", help="diagnostic: every batch ordered by the genome position its reads were drawn from (the upper bound of any "
                                                                 "locality ordering of the reads; never the reported configuration)")
    ap.add_argument("--sub-record", action="store_true", help="run as one of the default line's sub-records: the device-resident leg, counters and parity sample only "
                                                               "(exhaustive workloads: + the reference's -b run on a bounded sample)")
    ap.add_argument("--no-sub", action="store_true", help="skip the sub-records of the other BASELINE configs (default run, N=1, workload ecoli: configs[1], [3], [4])")
    ap.add_argument("--sub-timeout", type=int, default=170, help="seconds one sub-record child may take")
    ap.add_argument("--cpu-sample-exh", type=int, default=200_000, help="reads of the exhaustive CPU baseline (reference -b, -t cpu-threads)")
    ap.add_argument("--cpu-sample-all", type=int, default=2_500_000, help="reads of the all-cores leg of the CPU baseline (-t min(255, visible cores)); 0 disables")
    ap.add_argument("--full-parity", action="store_true", help="off by default (+4-6 minutes): tools/full_parity.py -- the WHOLE read sets of configs[1] (10 M) and configs[2] (50 M) "
                                                                "and 2 M reads of configs[4] through bin/bgreat and through the compiled reference (-t cpu-threads); sorted record "
                                                                "multisets + counters; the report goes to gpurun_out/full_parity.txt, the verdict into the line (full_parity, config.full_parity_*)")
    ap.add_argument("--full-parity-scale", type=float, default=1.0, help=argparse.SUPPRESS)
    ap.add_argument("--dry-run", action="store_true", help="no device: the launcher / rendezvous / collective / line-assembly path of an N-rank run on stand-in devices (gloo; the index is "
                                                            "built for real on rank 0 and broadcast, a step sleeps): `python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --dry-run` "
                                                            "walks exactly the argument path the driver uses for --gpus 8 on a box without eight GPUs; the line says dry_run and carries no value")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--debug-stop", type=int, default=0, help="diagnostic builds of the library only (-DBGR_PHASE_TIMING, loaded through BGR_LIB_PATH): "
                                                              "1 = the mapping kernel stops behind the staging of the reads, 2 = behind the anchor scan")
    args = ap.parse_args()
    presets = {  # SURVEY.md 8d synthetic inputs; explicit flags given on the command line win over the preset
        "small": dict(genome=250_000, site_spacing=75, alleles=2, read_len=100, reads_per_step=1_000_000),
        "chr1": dict(genome=230_000_000, site_spacing=175, alleles=2),
        "branchy": dict(genome=50_000_000, site_spacing=36, alleles=4, read_len=250, mismatch=5, reads_per_step=2_000_000, exhaustive=True),
    }
    for key, val in presets.get(args.workload, {}).items():
        if getattr(args, key) == ap.get_default(key):
            setattr(args, key, val)
    if args.sub_record:
        args.e2e_reads, args.pcie_steps = 0, 0
    return args


# ---- HBM traffic and instruction counters of one step, measured in this run ------------------------------------------------
# (FETCH_SIZE takes 3 of the 4 TCC slots of a pass and WRITE_SIZE 2: separate passes, as MI355X_MICROARCH.md prescribes; SQ: 8 slots)
PMC_PASSES = (("fetch", ["FETCH_SIZE", "TCC_REQ_sum", "SQ_BUSY_CU_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"]),
              ("write", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"]),
              ("sq", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"]))
PROFILER_ENV = ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCP_TOOL_LIBRARY")


def under_profiler():
    """True when this process already runs under rocprofv3 (its tool library is preloaded): nested counter passes would
    inherit that environment and the inner rocprofv3 launcher would exec with the GPU initialised (forbidden on this pool)."""
    return any("rocprof" in os.environ.get(v, "").lower() for v in PROFILER_ENV) or bool(os.environ.get("ROCPROF_OUTPUT_PATH"))


def run_pmc_passes(args):
    """rocprofv3 --pmc over short child runs of this script (same workload, same launch geometry, `pmc_steps` launches and
    nothing else on the GPU), one pass per counter group as MI355X_MICROARCH.md prescribes.  Runs BEFORE this process
    touches the GPU.  -> {counter: mean per mapping launch, summed over the kernels of the launch}, or {"error": ...}."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    fwd = ["--pmc-child", "--steps", str(args.pmc_steps), "--warmup", "0", "--workload", args.workload, "--reads-per-step", str(args.reads_per_step),
           "--read-len", str(args.read_len), "--k", str(args.k), "--mismatch", str(args.mismatch), "--effort", str(args.effort),
           "--genome", str(args.genome), "--site-spacing", str(args.site_spacing), "--alleles", str(args.alleles), "--lds-mphf", str(args.lds_mphf),
           "--waves", str(args.waves), "--blocks-per-cu", str(args.blocks_per_cu), "--gamma", str(args.gamma), "--cpu-threads", str(args.cpu_threads),
           "--debug-stop", str(args.debug_stop)]
    for flag in ("exhaustive", "anchors", "general_kernel_only", "exh_first_pass_off", "anc_first_pass_off", "sorted_reads", "prepass"):
        if getattr(args, flag):
            fwd.append("--" + flag.replace("_", "-"))
    fwd += ["--exh-search", str(args.exh_search), "--exh-frame-cap", str(args.exh_frame_cap)]
    totals, per_kernel = {}, {}
    t0 = time.time()
    for tag, counters in PMC_PASSES:
        d = tempfile.mkdtemp(prefix="bgr_pmc_%s_" % tag, dir="/tmp")
        try:
            cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__)] + fwd
            env = {k: v for k, v in os.environ.items() if k not in PROFILER_ENV and not k.startswith("ROCPROF")}
            env["TMPDIR"] = "/tmp"
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return {"error": "rocprofv3 pass '%s' failed (rc %d): %s" % (tag, p.returncode, (p.stderr or "")[-300:])}
            for f in files:
                for row in csv.DictReader(open(f)):
                    kn = row.get("Kernel_Name", "")
                    if "bgr_" not in kn:
                        continue
                    v = float(row["Counter_Value"])
                    totals[row["Counter_Name"]] = totals.get(row["Counter_Name"], 0.0) + v
                    short = kn[kn.find("bgr_"):].split("(")[0]
                    per_kernel.setdefault(short, {})
                    per_kernel[short][row["Counter_Name"]] = per_kernel[short].get(row["Counter_Name"], 0.0) + v
        except Exception as ex:  # the counters are a diagnostic: never fail the bench over them
            return {"error": "%s: %s" % (type(ex).__name__, ex)}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    n = float(args.pmc_steps)
    out = {k: v / n for k, v in totals.items()}
    out["_per_kernel"] = {k: {c: v / n for c, v in d.items()} for k, d in per_kernel.items()}
    out["_seconds"] = round(time.time() - t0, 1)
    return out


def main():
    args = parse_args()
    mode = 1 if args.exhaustive else (2 if args.anchors else 0)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if args.dry_run:
        return dry_run(args, mode, world, rank, local_rank)
    pmc = None
    if not args.pmc_child and not args.no_pmc and world == 1:
        # child processes; this process has not initialised the GPU yet.  Never nested inside another profiler run.
        pmc = {"error": "already running under a profiler: counter passes skipped"} if under_profiler() else run_pmc_passes(args)
        log("pmc passes:", {k: v for k, v in pmc.items() if not k.startswith("_per")} if pmc else None)

    subs = None
    if world == 1 and args.workload == "ecoli" and mode == 0 and not (args.pmc_child or args.sub_record or args.no_sub):
        subs = run_sub_records(args)   # child processes, one after the other; this process has not initialised the GPU yet

    import torch
    import bgreat_amd as B
    from tools.synth import Synth

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
    multi = None
    if world > 1:
        from bgreat_amd import dist as D
        ones = torch.ones(1, dtype=torch.int64, device="cpu" if coll_dev is None else "cuda:%d" % coll_dev)
        dist.all_reduce(ones)                 # (also brings the communicator up before the broadcast is timed)
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        # C1: the only data-path collective; reads never move.  First contact with RCCL between real devices happens HERE: if the broadcast raises, every
        # rank builds the index itself from the same seed (the blob does not depend on the thread count: same bytes) and the line says so.
        bc_form, bc_err = "torch.distributed broadcast over %s, adopted in place" % ("gloo (rehearsal)" if rehearsal else "RCCL"), None
        try:
            g, _blob_keepalive = D.broadcast_graph(g, dist, device=coll_dev)
            torch.cuda.synchronize()
        except Exception as ex:
            bc_err = "%s: %s" % (type(ex).__name__, str(ex)[:300])
            log("bench.py rank %d: graph broadcast failed (%s): building the index on this rank" % (rank, bc_err))
            if rank != 0:
                seqs, offs = syn.unitigs()
                g = B.Graph.build(args.k, seqs, offs, args.gamma, anchors=(mode == 2))
            bc_form = "FAILED (%s): every rank built the index itself from the same unitigs" % bc_err
        bc_ms = (time.perf_counter() - tb0) * 1e3
        # who took part: device, bus id and host of every rank; the RCCL the process group runs on
        who = None
        try:
            props = torch.cuda.get_device_properties(dev)
            mine = {"rank": rank, "local_rank": local_rank, "device": dev, "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None), "host": os.uname().nodename,
                    "blob_bytes": int(g.info()["blob_bytes"])}
            who = [None] * world
            dist.all_gather_object(who, mine)
        except Exception as ex:
            who = "all_gather_object failed: %s" % type(ex).__name__
        try:
            rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version()) if not rehearsal else None
        except Exception:
            rccl_version = None
        multi = {"backend": "gloo (rehearsal)" if rehearsal else "nccl (RCCL)", "rccl_version": rccl_version, "ranks_seen": int(ones.item()),
                 "graph_broadcast_form": bc_form, "graph_broadcast_bytes": int(g.info()["blob_bytes"]),
                 "graph_broadcast_ms": round(bc_ms, 2), "graph_broadcast_GB_per_s": round(g.info()["blob_bytes"] / max(bc_ms, 1e-6) / 1e6, 2),
                 "ranks": who,
                 "blob_bytes_equal_on_all_ranks": (len({w["blob_bytes"] for w in who}) == 1) if isinstance(who, list) and all(isinstance(w, dict) for w in who) else None,
                 "what": "one process per GPU (torch.distributed): all-reduce of ones = ranks that took part; C1 = rank 0's blob to every rank's HBM, adopted in place"}
    al = B.Aligner(g, dev)
    al.configure(args.waves, args.blocks_per_cu, args.lds_mphf)
    if args.general_kernel_only:
        al.set_knob(B.KNOB_GREEDY_FAST, 1)
    if args.prepass:
        al.set_knob(B.KNOB_GREEDY_PREPASS, 1)
    if args.no_kernel_events:
        al.set_knob(B.KNOB_KERNEL_EVENTS, 0)
    if args.debug_stop:
        al.set_knob(B.KNOB_DEBUG_STOP, args.debug_stop)
    if args.exh_first_pass_off:
        al.set_knob(B.KNOB_EXH_FAST, 1)
    if args.anc_first_pass_off:
        al.set_knob(B.KNOB_ANCHORS_FAST, 1)
    if args.exh_search:
        al.set_knob(B.KNOB_EXH_SEARCH, args.exh_search)
    if args.exh_frame_cap:
        al.set_knob(B.KNOB_EXH_FRAME_CAP, args.exh_frame_cap)
    if rank == 0:
        log("graph: %s  (%.1fs)" % (graph_info, time.time() - t0))

    # ---- reads: rank r owns global reads [r*K*R, (r+1)*K*R); generated on the host, parked in HBM -----------
    t0 = time.time()
    ncpu_all = len(os.sched_getaffinity(0))
    ncpu = max(1, min(ncpu_all // max(1, world) if world > 1 else ncpu_all, args.cpu_threads))  # the GPU box gives one GPU a 16-core share
    offs_np = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
    offs_t = B.DeviceBuffer(dev, offs_np)   # parked in HBM through the C-ABI (bgr_device_alloc / bgr_device_upload): torch only synchronises and reduces
    batches = []
    first_host = None
    for s in range(K):
        arr, _ = syn.reads((rank * K + s) * R, R, L, args.mismatch, seed_reads, threads=ncpu)
        if args.sorted_reads:
            order = np.argsort(syn.read_starts((rank * K + s) * R, R, L, seed_reads), kind="stable")
            arr = arr.reshape(R, L)[order].reshape(-1)
        if s == 0 and rank == 0 and not args.pmc_child:
            first_host = arr[: max(args.cpu_sample, args.alg_sample) * L].copy()
        batches.append(B.DeviceBuffer(dev, arr))
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
    if args.pmc_child:
        return
    if dist is not None:
        elapsed = D.max_over_ranks(elapsed, dist, device=coll_dev)
    launches, kernel_ms = al.kernel_time()
    _, slots = al.kernel_times()
    counters = al.counters()
    pass_counts = al.pass_counts()
    if dist is not None:  # C2: sum the aligner.h:68 counters over ranks
        counters = D.reduce_counters(counters, dist, device=coll_dev)

    # ---- rank 0: everything the line needs from the device-resident leg (value, roofline, parity sample) BEFORE the optional legs below, so
    # that a leg that never returns (a collective between real devices that this container has never run, a wedged file system) costs
    # the line its optional fields, not the line: LineGuard prints what is known when the deadline passes
    guard = LineGuard(rank, float(os.environ.get("BGR_BENCH_DEADLINE", "1500" if args.full_parity else "480")))
    line = None
    if rank == 0:
        total_reads = world * K * R
        value = total_reads / elapsed / 1e6
        avg_launch_ms = kernel_ms / max(1, launches)

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
        kernels_ms = [{"kernel": nm, "avg_ms": round(ms / max(1, launches), 4)} for nm, ms in slots]
        dominant, dom_total_ms = max(slots, key=lambda x: x[1]) if slots else (None, 0.0)
        dom_ms = dom_total_ms / max(1, launches)            # the dominant kernel's average duration (HIP events on the aligner's stream)
        dom_short = dominant.split(" ")[0] if dominant else None
        # ---- this implementation's OWN compulsory HBM bytes per read: what one launch must move through HBM if every
        # re-used structure (key table, records, unitig bases: the graph blob) is read from HBM once per launch -- counted from the
        # launch's own numbers: read lengths, path ints written (paths of this run), follow-up items queued (pass_counts)
        blob_bytes = float(graph_info["blob_bytes"])
        path_ints = float(len(p1)) / ns                     # per read, from the parity sample's paths
        items = float(pass_counts[0]) / R if mode == 0 else 0.0
        words = (L + 31) // 32
        has_prepass = any(nm.startswith("bgr_pack_reads_kernel") for nm, _ in slots)   # (greedy mode since round 5: the mapping kernels stage from the characters themselves)
        own = {
            "bgr_pack_reads_kernel": {"ascii_in": L, "offsets_in": 8, "planes_out": 8 * words, "hasn_out": 0.125} if has_prepass else {},
            "mapping": ({"planes_in": 8 * words * (1.0 + items), "hasn_in": 0.125} if has_prepass else {"ascii_in": L * (1.0 + items)}),
        }
        own["mapping"].update({"offsets_in": 8 * (1.0 + items), "results_out": 8, "path_ints_out": 4 * path_ints, "retry_queue_rw": 16 * items, "graph_blob_once_per_launch": blob_bytes / R})
        own_pack = sum(own["bgr_pack_reads_kernel"].values())
        own_map = sum(own["mapping"].values())
        dom_own = own_pack if (dom_short or "").startswith("bgr_pack") else own_map
        hbm = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
               "compulsory_bytes_per_read": {"bgr_pack_reads_kernel": round(own_pack, 1), "mapping_kernels": round(own_map, 1), "split_mapping": {k: round(v, 2) for k, v in own["mapping"].items()}},
               "achieved": round(dom_own * R / (dom_ms / 1e3) / 1e9, 2), "frac": round(dom_own * R / (dom_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
               "whole_launch": {"bytes_per_read": round(own_pack + own_map, 1), "achieved": round((own_pack + own_map) * R / (avg_launch_ms / 1e3) / 1e9, 2),
                                "frac": round((own_pack + own_map) * R / (avg_launch_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5)},
               "definition": "this implementation's compulsory HBM bytes per read of the dominant kernel (planes, offsets, results, path ints, retry queue, the graph blob once per launch) "
                             "x reads per launch / that kernel's mean duration; `traffic` next to it is what the memory side saw (rocprofv3 PMC)"}
        roofline = {"bound": "hbm", "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac"], "traffic": None,
                    "dominant_kernel": dominant, "dominant_kernel_ms": round(dom_ms, 4), "hbm": hbm,
                    "launch": "the kernels of one batch (mapping kernels; a pre-pass in anchors mode), enqueued back to back on one stream (HIP events around every kernel, on the aligner's stream)",
                    "avg_launch_ms": round(avg_launch_ms, 4), "launches": launches, "kernels_ms": kernels_ms, "reads_per_launch": R,
                    # SURVEY 8d's figure describes the REFERENCE's control flow (gamma-10 BooPHF probes, rank words, 24-B records): kept for the
                    # record under its own keys, never as a fraction of this implementation's roofline
                    "reference_alg_bytes_per_read": round(alg_bytes_per_read, 1),
                    "reference_alg_gbps": round(alg_bytes_per_read * R / (avg_launch_ms / 1e3) / 1e9, 1),
                    "reference_alg_note": "SURVEY 8d formula counted by the oracle on %d reads of this workload: bytes the REFERENCE's algorithm would move uncached; this "
                                          "implementation does not perform those probes (its key table is two LDS dwords per position), so this is NOT a fraction of any peak" % ns}
        if pmc and "error" not in pmc:
            pk = pmc.get("_per_kernel", {})
            dom_key = next((k for k in pk if dom_short and k.split("<")[0] == dom_short), None)  # (PMC rows carry the template arguments)
            dom_pmc = pk.get(dom_key, {}) if dom_key else {}
            fetch_kb, write_kb = pmc.get("FETCH_SIZE"), pmc.get("WRITE_SIZE")
            if fetch_kb is not None and write_kb is not None:
                raw = (fetch_kb + write_kb) * 1024.0
                corrected = (2.0 * fetch_kb + write_kb) * 1024.0  # gfx950: FETCH_SIZE tallies 128-B requests of wide reads at 64 B (upper bound for small gathers)
                roofline["traffic"] = round(corrected, 1)
                roofline["traffic_raw"] = round(raw, 1)
                roofline["traffic_note"] = ("HBM-side bytes per launch (all kernels) = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, rocprofv3 --pmc in separate passes over a %d-launch child run of "
                                            "this script in this run; raw = without the gfx950 x2 on FETCH_SIZE (the x2 is exact for wide streaming reads, an upper bound for gathers)" % args.pmc_steps)
                roofline["traffic_frac"] = round(corrected / (avg_launch_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
                roofline["traffic_bytes_per_read"] = round(corrected / R, 1)
                roofline["traffic_over_compulsory"] = round(corrected / ((own_pack + own_map) * R), 3)
                if dom_pmc.get("FETCH_SIZE") is not None and dom_pmc.get("WRITE_SIZE") is not None:
                    dt = (2.0 * dom_pmc["FETCH_SIZE"] + dom_pmc["WRITE_SIZE"]) * 1024.0
                    hbm["dominant_kernel_traffic"] = round(dt, 1)
                    hbm["dominant_kernel_traffic_frac"] = round(dt / (dom_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
            if pmc.get("TCC_HIT_sum") is not None and pmc.get("TCC_MISS_sum") is not None:
                roofline["l2_hit_rate"] = round(pmc["TCC_HIT_sum"] / max(1.0, pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]), 4)
            if pmc.get("TCC_REQ_sum") is not None:
                roofline["l2_requests_per_read"] = round(pmc["TCC_REQ_sum"] / R, 2)
            if dom_pmc.get("SQ_ACTIVE_INST_VALU") is not None and dom_pmc.get("SQ_ACTIVE_INST_VALU2") is not None:
                # MEASURED vector-issue occupancy of the dominant kernel.  gfx950 issues one vector instruction per SIMD and 4-cycle slot, or TWO
                # (from two waves) when both are of the plain VOP1/VOP2 class (add/sub/and/or/xor/mov/not/right shifts: the "2-cycle"
                # class of profiles/r02_valu_rates.txt); SQ_ACTIVE_INST_VALU counts every instruction, SQ_ACTIVE_INST_VALU2 the ones that
                # went out as the second of a pair, so (A - A2) is the number of BUSY issue slots -- calibrated on 50 single-opcode
                # streams, profiles/r03_valu2_pmc_calibration.txt: 4 x (1 - A2/A) reproduces every stream's measured cycles per instruction.
                busy_slots = dom_pmc["SQ_ACTIVE_INST_VALU"] - dom_pmc["SQ_ACTIVE_INST_VALU2"]        # per launch, summed over all SIMDs
                busy_cycles_per_s = 4.0 * busy_slots / (dom_ms / 1e3)
                peak_cycles_per_s = N_SIMD * CLOCK_GHZ * 1e9
                valu = {"bound": "valu_issue", "unit": "G SIMD-cycles/s", "achieved": round(busy_cycles_per_s / 1e9, 2), "peak": round(peak_cycles_per_s / 1e9, 2),
                        "frac": round(busy_cycles_per_s / peak_cycles_per_s, 4),
                        "valu_insts_per_read": round(dom_pmc.get("SQ_INSTS_VALU", 0.0) / R, 1), "paired_share": round(dom_pmc["SQ_ACTIVE_INST_VALU2"] / max(1.0, dom_pmc["SQ_ACTIVE_INST_VALU"]), 4),
                        "cycles_per_valu_inst": round(4.0 * busy_slots / max(1.0, dom_pmc["SQ_ACTIVE_INST_VALU"]), 3),
                        "salu_insts_per_read": round(dom_pmc.get("SQ_INSTS_SALU", 0.0) / R, 1),
                        "salu_issue_frac": round(dom_pmc.get("SQ_INSTS_SALU", 0.0) / (N_SIMD / 4 * (dom_ms / 1e3) * CLOCK_GHZ * 1e9), 4),
                        "definition": "busy vector issue slots of the dominant kernel = SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2 per launch (rocprofv3 PMC, child run of this "
                                      "script), x 4 cycles / (the kernel's mean duration in the timed region) against %d SIMDs x %.1f GHz" % (N_SIMD, CLOCK_GHZ)}
                if dom_pmc.get("SQ_BUSY_CU_CYCLES"):
                    cu_cyc = dom_pmc["SQ_BUSY_CU_CYCLES"]
                    valu["frac_of_cu_busy_cycles"] = round(busy_slots / cu_cyc, 4)   # (4 x slots) / (4 SIMDs x CU-busy cycles): needs no clock and no timer
                roofline["valu_issue"] = valu
                # what bounds the dominant kernel: the larger of the measured vector-issue fraction and the HBM-side fraction
                hf = max(hbm["frac"], hbm.get("dominant_kernel_traffic_frac") or 0.0)
                if valu["frac"] >= hf:
                    roofline.update({"bound": "valu_issue", "achieved": valu["achieved"], "peak": valu["peak"], "unit": valu["unit"], "frac": valu["frac"]})
                roofline["bound_note"] = ("`bound` names the larger of the dominant kernel's measured vector-issue fraction (roofline.valu_issue) and its HBM fraction "
                                          "(roofline.hbm: compulsory bytes, and measured traffic); integer hash + byte compare: no MFMA on this path")
            roofline["pmc_seconds"] = pmc.get("_seconds")
            roofline["pmc_per_kernel"] = {k: {c: round(v, 1) for c, v in d.items()} for k, d in pk.items()}
        elif pmc:
            roofline["traffic_note"] = "not measured in this run: " + pmc["error"]

        line = {
            "metric": "Mreads/s aligned (k=%d, %dbp, m=%d)" % (args.k, L, args.mismatch), "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: synthetic %.1fM x %d bp reads per GPU (%d steps x %d), k=%d, m=%d, effort=%d, %s, graph of %d unitigs (genome %d bp, %d alleles every ~%d bp); "
                                   "reads resident in HBM as ASCII before the timed region, results left in HBM"
                       % ({"ecoli": "BASELINE configs[2]", "small": "BASELINE configs[1]", "chr1": "BASELINE configs[3] graph scale, one GPU's share", "branchy": "BASELINE configs[4] graph, one GPU's share"}[args.workload],
                          K * R / 1e6, L, K, R, args.k, args.mismatch, args.effort, ("greedy", "exhaustive", "greedy from k-mer anchors (-G)")[mode], graph_info["n_unitigs"], args.genome, args.alleles, args.site_spacing),
                       "reads_per_step_per_gpu": R, "read_len": L, "k": args.k, "m": args.mismatch, "effort": args.effort,
                       "parallelism": "reads sharded over %d GPU(s); graph blob broadcast once" % world, "launch": al.launch_info(),
                       "pass_counts_last_launch": pass_counts},
            # SURVEY 8d's two metrics beside `value` (the device-resident kernel/roofline anchor): (i) H2D + launch + CSR + D2H from page-locked
            # host buffers, one blocking caller; (ii) file in -> paths / notAligned.fa out, median of the runs (spread in e2e.runs_mreads_per_s)
            "value_pcie_inclusive": None, "value_e2e": None,
            "roofline": roofline, "pcie_inclusive": None, "e2e": None, "cpu_baseline": None, "multi_gpu": multi, "one_process_all_gpus": None,
            "other_configs": subs,
            "counters": counters, "parity_sample": {"reads": ns, "gpu_equals_oracle": parity_ok},
            "oracle_work_per_read": {k: round(v / ns, 2) for k, v in work.items() if k not in ("reads",)},
        }
    guard.arm(line)

    # ---- end to end: file in -> paths / notAligned.fa out (every rank its own shard file and GPU) ---------------
    e2e = None
    if args.e2e_reads > 0 and mode == 0:
        guard.stage("e2e")
        e2e = run_e2e(args, B, syn, g, rank, world, dev, ncpu, dist, D if dist is not None else None, coll_dev, seed_reads)

    # ---- N > 1: the C-ABI's own multi-GPU form next to the one-process-per-GPU form above -- ONE process, bgr_devices_init (one upload,
    # then device to device over xGMI: RCCL broadcast or peer copies) and one aligner + one host thread per device; rank 0 runs it
    # while the other ranks wait (their batches stay parked: the devices are otherwise idle)
    one_proc = None
    if world > 1:
        guard.stage("one_process_all_gpus")
        one_proc = rank0_alone(dist, rank, "one_process_threads", lambda: run_one_process_all_gpus(args, B, g, syn, world, mode, seed_reads, ncpu, rehearsal))

    if rank != 0:
        guard.leave(dist)
        return

    # ---- PCIe inclusive: pinned host buffers through bgr_align_batch (H2D + launch + CSR + D2H), N=1 only ---------
    pcie = None
    if world == 1 and args.pcie_steps > 0 and mode == 0:
        pcie = run_pcie(args, B, g, al, syn, seed_reads, ncpu, dev)

    # ---- CPU baseline: the compiled reference (oracle/_ref/bgreat -t cores) on a bounded sample, N=1 only -------
    cpu = None
    if world == 1 and args.cpu_sample > 0 and mode == 0 and not args.sub_record:
        cpu = run_cpu_baseline(args, al, syn, first_host, ncpu, seed_reads)
    if world == 1 and args.cpu_sample_exh > 0 and mode == 1:
        cpu = run_cpu_baseline_exhaustive(args, al, syn, first_host, ncpu, seed_reads)

    if world == 1 and args.full_parity and not args.sub_record:
        guard.stage("full_parity")
        try:
            from tools import full_parity
            fp = full_parity.run(threads=ncpu, scale=args.full_parity_scale, out=os.path.join(ROOT, "gpurun_out", "full_parity.txt"), log=log)
            line["full_parity"] = {"equal": fp["equal"], "seconds": fp["seconds"], "against": "oracle/_ref (the reference compiled from its own sources), -t %d" % ncpu,
                                   **{"c%s_%s" % (c, k): v for c, r in fp["configs"].items() for k, v in (("reads", r["reads"]), ("equal", r["equal"]))},
                                   "report": "gpurun_out/full_parity.txt"}
        except Exception as ex:
            line["full_parity"] = {"equal": False, "error": "%s: %s" % (type(ex).__name__, ex)}
    line.update({"value_pcie_inclusive": (pcie or {}).get("value"), "value_e2e": (e2e or {}).get("value"), "pcie_inclusive": pcie, "e2e": e2e, "cpu_baseline": cpu,
                 "one_process_all_gpus": one_proc})
    guard.finish(line, dist)


def dry_run(args, mode, world, rank, local_rank):
    """bench.py --dry-run: everything around the device work of an N-rank run, on stand-in devices -- environment of torch.distributed.run, process group
    (gloo), the all-reduce of ones, C1 (the real blob of the real index, through bgreat_amd.dist.broadcast_graph's CPU form), barriers around a timed region
    whose steps sleep, max over ranks, C2, the LineGuard, one JSON line on rank 0 (value null: nothing was mapped)."""
    import torch
    import torch.distributed as dist
    import bgreat_amd as B
    from bgreat_amd import dist as D
    from tools.synth import Synth
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    K, W, R, L = args.steps, args.warmup, args.reads_per_step, args.read_len
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    syn = Synth(min(args.genome, 300_000), args.site_spacing, args.alleles, args.k, 20261003)
    g = None
    if rank == 0:
        seqs, offs = syn.unitigs()
        g = B.Graph.build(args.k, seqs, offs, args.gamma)
    multi = None
    if world > 1:
        ones = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(ones)
        tb0 = time.perf_counter()
        g, _keep = D.broadcast_graph(g, dist, device=None)
        bc_ms = (time.perf_counter() - tb0) * 1e3
        who = [None] * world
        dist.all_gather_object(who, {"rank": rank, "local_rank": local_rank, "host": os.uname().nodename, "blob_bytes": int(g.info()["blob_bytes"])})
        multi = {"backend": "gloo (dry run, stand-in devices)", "ranks_seen": int(ones.item()), "graph_broadcast_form": "torch.distributed broadcast over gloo (CPU tensors), re-wrapped with bgr_graph_from_blob",
                 "graph_broadcast_bytes": int(g.info()["blob_bytes"]), "graph_broadcast_ms": round(bc_ms, 2), "ranks": who, "blob_bytes_equal_on_all_ranks": len({w["blob_bytes"] for w in who}) == 1}
    guard = LineGuard(rank, float(os.environ.get("BGR_BENCH_DEADLINE", "120")))
    for _ in range(W):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t_start = time.perf_counter()
    for _ in range(K):
        time.sleep(0.001 * (1 + rank % 2))   # (ranks differ: the line must carry the MAX over ranks)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        elapsed = D.max_over_ranks(elapsed, dist, device=None)
        counters = D.reduce_counters({"reads": K * R}, dist, device=None)
    else:
        counters = {"reads": K * R}
    line = None
    if rank == 0:
        line = {"metric": "Mreads/s aligned (k=%d, %dbp, m=%d)" % (args.k, L, args.mismatch), "value": None, "unit": "Mreads/s", "n_gpus": world, "steps": K, "warmup": W,
                "ms_per_step": round(elapsed / max(1, K) * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "dry_run": "no device was used: launcher, rendezvous, collectives and line assembly of a %d-rank run on stand-in devices; steps slept" % world,
                "config": {"workload": "dry run", "reads_per_step_per_gpu": R, "read_len": L, "parallelism": "reads sharded over %d rank(s); graph blob broadcast once" % world},
                "roofline": None, "cpu_baseline": None, "multi_gpu": multi, "counters": counters}
    guard.arm(line)
    if rank != 0:
        guard.leave(dist if world > 1 else None)
        return
    guard.finish(line, dist if world > 1 else None)


TEMP_DIRS = set()   # scratch directories of the legs in flight (tens of GB at default sizes): removed on every way out, the deadline's os._exit included


def scalar_copies(line):
    """The driver's record keeps the scalar members of `config`, `roofline` and `cpu_baseline` (nested objects and other top-level keys survive as
    key names only): every number the line carries elsewhere gets a scalar copy there, so that BENCH_rNN.json proves all five configs, not one."""
    cfg, roof, cpu = line.get("config"), line.get("roofline"), line.get("cpu_baseline")
    if isinstance(cfg, dict):
        cfg["value_e2e"] = line.get("value_e2e")
        cfg["value_pcie_inclusive"] = line.get("value_pcie_inclusive")
        e2e = line.get("e2e") or {}
        if isinstance(e2e.get("host_route"), dict):
            cfg["e2e_host_route_mreads"] = e2e["host_route"].get("value")
        cfg["e2e_routes_identical_bytes"] = (e2e.get("host_route") or {}).get("identical_bytes_to_the_text_route") if isinstance(e2e.get("host_route"), dict) else None
        cfg["parity_sample_ok"] = (line.get("parity_sample") or {}).get("gpu_equals_oracle")
        for w, sub in (line.get("other_configs") or {}).items():
            if not isinstance(sub, dict):
                continue
            cfg["%s_mreads" % w] = sub.get("value")
            cfg["%s_ms_per_step" % w] = sub.get("ms_per_step")
            cfg["%s_parity_ok" % w] = (sub.get("parity_sample") or {}).get("gpu_equals_oracle")
            if "error" in sub:
                cfg["%s_error" % w] = str(sub["error"])[:200]
        fp = line.get("full_parity") or {}
        for k, v in fp.items():
            if isinstance(v, (bool, int, float, str)) or v is None:
                cfg["full_parity_%s" % k] = v
        one = line.get("one_process_all_gpus") or {}
        if isinstance(one, dict) and one.get("value") is not None:
            cfg["one_process_all_gpus_mreads"] = one.get("value")
            cfg["one_process_fanout_method"] = one.get("fanout_method")
            cfg["one_process_fanout_ms"] = one.get("fanout_ms")
            for nm, rec in (one.get("fanout_both_ways") or {}).items():
                cfg["one_process_fanout_%s_ms" % nm] = rec.get("ms") if isinstance(rec, dict) else None
        mg = line.get("multi_gpu") or {}
        if isinstance(mg, dict):
            for k in ("backend", "ranks_seen", "rccl_version", "graph_broadcast_form", "graph_broadcast_bytes", "graph_broadcast_ms", "graph_broadcast_GB_per_s", "blob_bytes_equal_on_all_ranks"):
                if mg.get(k) is not None and isinstance(mg.get(k), (bool, int, float, str)):
                    cfg["multi_gpu_%s" % k] = mg[k]
    if isinstance(roof, dict):
        v = roof.get("valu_issue") or {}
        roof["valu_insts_per_read"] = v.get("valu_insts_per_read")
        roof["salu_insts_per_read"] = v.get("salu_insts_per_read")
        roof["valu_issue_frac"] = v.get("frac")
        roof["hbm_compulsory_frac"] = (roof.get("hbm") or {}).get("frac")
        for w, sub in (line.get("other_configs") or {}).items():
            if not isinstance(sub, dict):
                continue
            r = sub.get("roofline") or {}
            roof["%s_frac" % w] = r.get("frac")
            roof["%s_bound" % w] = r.get("bound")
            roof["%s_dominant_kernel_ms" % w] = sub.get("dominant_kernel_ms")
            roof["%s_valu_insts_per_read" % w] = sub.get("valu_insts_per_read")
            h = sub.get("hbm") or {}
            roof["%s_traffic_over_compulsory" % w] = h.get("traffic_over_compulsory")
            roof["%s_traffic_frac" % w] = h.get("traffic_frac")
            roof["%s_l2_hit_rate" % w] = sub.get("l2_hit_rate")
    if isinstance(cpu, dict):
        if isinstance(cpu.get("all_cores"), dict):
            cpu["all_cores_value"] = cpu["all_cores"].get("value")
            cpu["all_cores_threads"] = cpu["all_cores"].get("cores")
        if isinstance(cpu.get("t1"), dict):
            cpu["t1_value"] = cpu["t1"].get("value")
        x = ((line.get("other_configs") or {}).get("branchy") or {}).get("cpu_baseline")
        if isinstance(x, dict):
            cpu["exhaustive_value"] = x.get("value")
            cpu["exhaustive_cores"] = x.get("cores")
            cpu["exhaustive_sample"] = x.get("sample")
    return line


def drop_temp_dirs():
    for d in list(TEMP_DIRS):
        shutil.rmtree(d, ignore_errors=True)
        TEMP_DIRS.discard(d)


class LineGuard:
    """The ONE JSON line of rank 0 must come out even when an optional leg behind the timed region never returns (the N > 1 legs drive
    collectives between real devices, which this repository has only ever rehearsed on one GPU).  arm(line): rank 0 hands over the line as
    far as it stands behind the device-resident leg; a timer thread prints it -- marked `incomplete`, naming the leg that was running --
    when the deadline passes, and ends the process with exit code 0.  The other ranks arm without a line: they just leave at the
    deadline (a little later than rank 0, so that its line is out first).  finish()/leave(): the normal way out -- rank 0 tells the other
    ranks through the process group's store whether the closing barrier is still safe to enter."""

    def __init__(self, rank, seconds):
        self.rank, self.seconds = rank, seconds
        self.lock = threading.Lock()
        self.printed = False
        self.line = None
        self.what = "?"
        self.timer = None

    def stage(self, what):
        self.what = what

    def arm(self, line):
        self.line = line
        if self.seconds <= 0:
            return
        self.timer = threading.Timer(self.seconds + (0.0 if self.rank == 0 else 20.0), self._expired)
        self.timer.daemon = True
        self.timer.start()

    def _expired(self):
        with self.lock:
            if self.printed:
                return
            self.printed = True
            if self.rank == 0 and self.line is not None:
                d = dict(self.line)
                d["incomplete"] = "the optional legs behind the timed region did not finish within %.0f s (running: %s); value, roofline and parity sample are complete" % (self.seconds, self.what)
                print(json.dumps(scalar_copies(d)), flush=True)
            drop_temp_dirs()
            log("bench.py rank %d: deadline of %.0f s passed in leg '%s': leaving" % (self.rank, self.seconds, self.what))
            sys.stderr.flush()
            os._exit(0)

    @staticmethod
    def _store():
        try:
            from torch.distributed import distributed_c10d as c10d
            return c10d._get_default_store()
        except Exception:
            return None

    def finish(self, line, dist):
        with self.lock:
            if self.printed:
                return
            self.printed = True
            print(json.dumps(scalar_copies(line)), flush=True)
        if self.timer is not None:
            self.timer.cancel()
        if dist is not None:
            st = self._store()
            if st is not None:
                st.set("bgr_line_out", "1")
            dist.barrier()
            dist.destroy_process_group()

    def leave(self, dist):
        """ranks other than 0: wait (on the host) until rank 0 has printed its line, then the closing barrier"""
        if dist is not None:
            st = self._store()
            if st is not None:
                try:
                    st.wait(["bgr_line_out"])
                except Exception as ex:
                    bail_out(self.rank, "waiting for rank 0's line: %s" % type(ex).__name__)
            with self.lock:
                self.printed = True
            if self.timer is not None:
                self.timer.cancel()
            dist.barrier()
            dist.destroy_process_group()


def run_e2e(args, B, syn, g, rank, world, dev, ncpu, dist, D, coll_dev, seed_reads):
    """bgr_align_all on a FASTA file written just before (so it is read from the page cache): mmap + chunk-parallel parse +
    gather into pinned batches + H2D + launch + CSR + D2H + format + write, `ncpu` host threads per GPU.  Index build excluded."""
    n, L = args.e2e_reads, args.read_len
    if world > 1:
        # N ranks write and sync N input files at once on one node: keep the job's total near one rank's default (every os.sync() below
        # waits for ALL dirty pages of the node), but no rank below 25 M reads (a shorter run measures the pipeline's ramp)
        n = max(n // world, min(n, 25_000_000))
    # every rank writes its own input (L + 14 bytes per read) and keeps up to three output pairs (~55 bytes per read each) in the node's
    # temporary directory: with N ranks that is N x ~33 GB at the default size.  Cut the per-rank read count to what half of the free
    # space admits -- the same number on every rank (minimum over the ranks), so that all ranks pass the same barriers
    free = shutil.disk_usage(tempfile.gettempdir()).free
    cap = int(0.5 * free / (world * (L + 14 + 3 * 60)))
    if dist is not None:
        cap = int(-D.max_over_ranks(-float(cap), dist, device=coll_dev))
    if cap < n:
        n = cap
    if n < 100_000:
        return {"error": "not enough free space under %s for the end-to-end leg (%d bytes free, %d ranks)" % (tempfile.gettempdir(), free, world)}
    d = tempfile.mkdtemp(prefix="bgr_e2e_r%d_" % rank)
    TEMP_DIRS.add(d)
    try:
        f = os.path.join(d, "reads.fa")
        syn.write_reads(f, (world + rank) * 1_000_000_000, n, L, args.mismatch, seed_reads, threads=ncpu)
        fsize = os.path.getsize(f)
        best = None
        runs, walls = [], []
        host_route = None
        for rep in range(4):  # later runs have the page-locked staging buffers warm; the box's host cores are shared: runs vary
            route = 1 if rep == 3 else 0   # the last run: the same file through the host parser + host formatter (the round-2 pipeline)
            for fn in ("paths%d" % (rep - 1), "notAligned%d.fa" % (rep - 1)):
                if rep > 1 and os.path.exists(os.path.join(d, fn)):
                    os.unlink(os.path.join(d, fn))      # (keep run 0's outputs for the size / identity check, drop the others: disk space)
            os.sync()  # not timed: dirty pages of the input file / the previous run's outputs would throttle this run's writes
            if dist is not None:
                dist.barrier()
            t1 = time.perf_counter()
            cnt_r, secs = B.align_all(g, f, os.path.join(d, "paths%d" % rep), os.path.join(d, "notAligned%d.fa" % rep), m=args.mismatch, effort=args.effort,
                                      threads=ncpu, n_gpus=1, first_device=dev, route=route)
            if dist is not None:
                dist.barrier()
            wall = time.perf_counter() - t1
            if dist is not None:
                wall = D.max_over_ranks(wall, dist, device=coll_dev)
            if route == 1:
                same = all(_same_file(os.path.join(d, a), os.path.join(d, b)) for a, b in (("paths0", "paths3"), ("notAligned0.fa", "notAligned3.fa")))
                host_route = {"value": round(world * n / wall / 1e6, 3), "unit": "Mreads/s", "identical_bytes_to_the_text_route": bool(same),
                              "what": "one run of the same file with route = 1: host parser, host packer, host formatter (bgr_align_batch_packed)"}
                continue
            cnt = cnt_r
            walls.append(wall)
            runs.append(round(world * n / wall / 1e6, 1))
            if best is None or wall < best:
                best = wall
        out_bytes = os.path.getsize(os.path.join(d, "paths0")) + os.path.getsize(os.path.join(d, "notAligned0.fa"))
        # N > 1: the CLI's own form as well -- ONE process feeding all N devices (bgr_devices_init: one upload, then device to device over
        # xGMI; bgr_align_all with n_gpus = N: one producer, per-device queues and workers, one ordered writer), rank 0's file, the other
        # ranks waiting.  What a future scaling run reads as "does the CLI scale", next to the per-rank figure above.
        one_process = None
        if world > 1:
            def cli_forms():
                try:
                    import torch
                    n_dev = min(world, B.device_count(), max(1, torch.cuda.device_count()))
                    how = g.devices_init(0, n_dev, 0)
                    t1 = time.perf_counter()
                    cnt1, _ = B.align_all(g, f, os.path.join(d, "paths_1p"), os.path.join(d, "notAligned_1p.fa"), m=args.mismatch, effort=args.effort,
                                          threads=min(len(os.sched_getaffinity(0)), ncpu * n_dev), n_gpus=n_dev, first_device=0)
                    w1 = time.perf_counter() - t1
                    t1 = time.perf_counter()   # ... and as a split run: a pipeline per device, N output pairs (bgr_run_options.split_output)
                    cnt2, _ = B.align_all(g, f, os.path.join(d, "paths_sp"), os.path.join(d, "notAligned_sp.fa"), m=args.mismatch, effort=args.effort,
                                          threads=min(len(os.sched_getaffinity(0)), ncpu * n_dev), n_gpus=n_dev, first_device=0, split_output=True)
                    w2 = time.perf_counter() - t1
                    parts_p = [os.path.join(d, "paths_sp.%d" % i) for i in range(n_dev)] if n_dev > 1 else [os.path.join(d, "paths_sp")]
                    parts_n = [os.path.join(d, "notAligned_sp.fa.%d" % i) for i in range(n_dev)] if n_dev > 1 else [os.path.join(d, "notAligned_sp.fa")]
                    split = {"value": round(n / w2 / 1e6, 3), "unit": "Mreads/s", "output_pairs": n_dev,
                             "identical_bytes_concatenated": bool(_same_concat(os.path.join(d, "paths0"), parts_p) and _same_concat(os.path.join(d, "notAligned0.fa"), parts_n))}
                    return {"value": round(n / w1 / 1e6, 3), "unit": "Mreads/s", "n_gpus": n_dev, "reads": n, "host_threads": min(len(os.sched_getaffinity(0)), ncpu * n_dev),
                            "split_output": split,
                            "fanout_method": {0: "none", 1: "rccl broadcast", 2: "peer copies"}.get(how, str(how)),
                            "identical_bytes_to_one_gpu": bool(_same_file(os.path.join(d, "paths0"), os.path.join(d, "paths_1p")) and
                                                               _same_file(os.path.join(d, "notAligned0.fa"), os.path.join(d, "notAligned_1p.fa"))),
                            "what": "bgr_align_all(n_gpus = N) in ONE process on rank 0's file while the other ranks wait (on the host: no collective spins on their GPUs)"}
                except Exception as ex:
                    return {"error": "%s: %s" % (type(ex).__name__, ex)}
            one_process = rank0_alone(dist, rank, "one_process_cli", cli_forms)
        med = float(np.median(walls))
        return {"value": round(world * n / med / 1e6, 3), "unit": "Mreads/s", "reads_per_gpu": n, "n_gpus": world, "host_threads_per_gpu": ncpu, "seconds": round(med, 4),
                "runs_mreads_per_s": runs, "best": round(world * n / best / 1e6, 3), "worst": round(world * n / max(walls) / 1e6, 3),
                "input": "FASTA, %d bytes per GPU, written just before the run: page cache" % fsize, "input_GB_per_s": round(world * fsize / med / 1e9, 2),
                "output_bytes_per_gpu": out_bytes, "aligned": cnt["aligned"], "host_route": host_route, "one_process_all_gpus": one_process,
                "what": "bgr_align_all (the CLI's mapping phase): file -> paths + notAligned.fa, the device taking the FASTA text and returning the record bytes "
                        "(bgr_align_fasta_text); value = MEDIAN of 3 runs in this process (fresh output files, os.sync() before each, not timed; the box's host "
                        "cores are shared, the runs vary); index build excluded"}
    except Exception as ex:
        return {"error": "%s: %s" % (type(ex).__name__, ex)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def bail_out(rank, why):
    """a rank other than 0 that can no longer reach rank 0 (which prints the line) leaves quietly: exit code 0, no collective"""
    log("bench.py rank %d: leaving (%s)" % (rank, why))
    drop_temp_dirs()
    sys.stderr.flush()
    os._exit(0)


def rank0_alone(dist, rank, tag, fn):
    """All ranks meet, then rank 0 runs fn() while the others wait ON THE HOST (a key of the process group's store) -- a dist.barrier() there
    would keep a collective kernel spinning on every other GPU while rank 0 drives those very devices; falls back to a barrier when the
    store cannot be had.  -> fn()'s result on rank 0, None elsewhere."""
    dist.barrier()
    store = None
    try:
        from torch.distributed import distributed_c10d as c10d
        store = c10d._get_default_store()
    except Exception:
        store = None
    out = None
    if rank == 0:
        try:
            out = fn()
        finally:
            if store is not None:
                store.set("bgr_" + tag, "1")
    elif store is not None:
        try:
            store.wait(["bgr_" + tag])
        except Exception as ex:   # the store went away or the wait timed out: rank 0 has left (its LineGuard deadline) or is about to be reaped
            bail_out(rank, "waiting for rank 0's leg '%s': %s" % (tag, type(ex).__name__))
    if store is None:
        dist.barrier()
    return out


def _same_concat(a, parts, chunk=1 << 24):
    """file a == the files `parts` concatenated"""
    if os.path.getsize(a) != sum(os.path.getsize(x) for x in parts):
        return False
    with open(a, "rb") as fa:
        for x in parts:
            with open(x, "rb") as fx:
                while True:
                    y = fx.read(chunk)
                    if not y:
                        break
                    if fa.read(len(y)) != y:
                        return False
    return True


def _same_file(a, b, chunk=1 << 24):
    if os.path.getsize(a) != os.path.getsize(b):
        return False
    with open(a, "rb") as fa, open(b, "rb") as fb:
        while True:
            x, y = fa.read(chunk), fb.read(chunk)
            if x != y:
                return False
            if not x:
                return True


def run_pcie(args, B, g, al, syn, seed_reads, ncpu, dev):
    """Page-locked host buffers -> bgr_align_batch: H2D of the ASCII reads + offsets, the mapping launch, CSR on the device,
    D2H of paths / offsets / status.  One blocking stream, then two aligners (streams) on two host threads, as the CLI's
    pipeline drives them."""
    try:
        R, L = args.reads_per_step, args.read_len
        lib = B.lib()

        def pinned(nbytes, dtype):
            p = ctypes.c_void_p()
            B._check(lib.bgr_host_alloc(nbytes, ctypes.byref(p)))
            return p, np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)).view(dtype)

        def buffers(n):
            cap = 8 * n + 4096
            hr, reads = pinned(n * L, np.uint8)
            ho, offs = pinned((n + 1) * 8, np.uint64)
            hp, paths = pinned(cap * 4, np.int32)
            hq, poffs = pinned((n + 1) * 8, np.uint64)
            hs, status = pinned(n + 8, np.uint8)
            offs[:] = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
            return dict(h=(hr, ho, hp, hq, hs), reads=reads, offs=offs, paths=paths, poffs=poffs, status=status, cap=cap, n=n)

        def call(a, b):
            p = B.Params(B.MODE_GREEDY, args.mismatch, args.effort, 0)
            B._check(lib.bgr_align_batch(a.h, ctypes.byref(p), b["reads"].ctypes.data, b["offs"].ctypes.data, b["n"], b["paths"].ctypes.data, b["cap"],
                                         b["poffs"].ctypes.data, b["status"].ctypes.data))

        t0 = time.time()
        arr, _ = syn.reads(3_000_000_000, R, L, args.mismatch, seed_reads, threads=ncpu)
        one = buffers(R)
        one["reads"][:] = arr
        call(al, one)  # warm-up: device buffers of this size
        t1 = time.perf_counter()
        for _ in range(args.pcie_steps):
            call(al, one)
        t_one = (time.perf_counter() - t1) / args.pcie_steps
        h2d = R * L + (R + 1) * 8
        d2h = int(one["poffs"][R]) * 4 + (R + 1) * 8 + R
        # two streams: two aligners, each with its own half-size pinned batch, driven from two host threads
        half = R // 2
        al2 = B.Aligner(g, dev)
        hb = [buffers(half), buffers(half)]
        hb[0]["reads"][:] = arr[: half * L]
        hb[1]["reads"][:] = arr[half * L: 2 * half * L]
        als = [al, al2]
        for i in range(2):
            call(als[i], hb[i])

        def worker(i):
            for _ in range(args.pcie_steps * 2):
                call(als[i], hb[i])

        t1 = time.perf_counter()
        ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        t_two = time.perf_counter() - t1
        # the same with the reads packed to 2 bits on the host beforehand (what the CLI's gather stage does on its host threads)
        pk = B.pack_reads(arr, np.arange(R + 1, dtype=np.uint64) * np.uint64(L))
        hp = {}
        keep = []
        for key in ("read_offsets", "fw3", "hasn"):
            h, view = pinned(pk[key].nbytes, pk[key].dtype)
            view[:] = pk[key]
            hp[key] = view
            keep.append(h)
        spk = B.PackedReads(hp["read_offsets"].ctypes.data, hp["fw3"].ctypes.data, hp["hasn"].ctypes.data, None, None, 0, L)

        def call_packed(a, b):
            p = B.Params(B.MODE_GREEDY, args.mismatch, args.effort, 0)
            B._check(lib.bgr_align_batch_packed(a.h, ctypes.byref(p), ctypes.byref(spk), R, b["paths"].ctypes.data, b["cap"], b["poffs"].ctypes.data, b["status"].ctypes.data))

        call_packed(al, one)
        t1 = time.perf_counter()
        for _ in range(args.pcie_steps):
            call_packed(al, one)
        t_pk = (time.perf_counter() - t1) / args.pcie_steps
        h2d_pk = pk["fw3"].nbytes + (R + 1) * 8 + pk["hasn"].nbytes
        out = {"value": round(R / t_one / 1e6, 3), "unit": "Mreads/s", "reads_per_call": R, "ms_per_call": round(t_one * 1e3, 3),
               "two_streams": {"value": round(2 * half * args.pcie_steps * 2 / t_two / 1e6, 3), "unit": "Mreads/s", "reads_per_call": half},
               "host_packed": {"value": round(R / t_pk / 1e6, 3), "unit": "Mreads/s", "ms_per_call": round(t_pk * 1e3, 3), "h2d_bytes_per_read": round(h2d_pk / R, 1),
                               "what": "bgr_align_batch_packed: the reads cross PCIe as 2-bit planes packed on the host beforehand (packing not timed: the CLI does it in its "
                                       "gather stage instead of a memcpy); one blocking stream"},
               "h2d_bytes_per_read": round(h2d / R, 1), "d2h_bytes_per_read": round(d2h / R, 1),
               "what": "bgr_align_batch on page-locked host buffers: H2D (ASCII reads + offsets) + pre-pass + mapping passes + CSR on the device + D2H (paths, offsets, status); "
                       "mean of %d blocking calls on one stream; two_streams = two aligners on two host threads (how the CLI's pipeline overlaps copies and kernels)" % args.pcie_steps}
        for h in keep:
            lib.bgr_host_free(h)
        al2.close()
        for b in [one] + hb:
            for h in b["h"]:
                lib.bgr_host_free(h)
        log("pcie leg: %.1fs" % (time.time() - t0))
        return out
    except Exception as ex:
        return {"error": "%s: %s" % (type(ex).__name__, ex)}


def run_sub_records(args):
    """The other BASELINE configs inside the one driver-timed line: configs[1] (`--workload small`), the configs[3] graph (`chr1`, one
    GPU's share of the reads) and configs[4] (`branchy`, exhaustive) each run as a child `bench.py --workload W --sub-record` -- the same
    code path as a builder run of that workload (device-resident leg, its own rocprofv3 counter passes, parity sample against the
    oracle; exhaustive: the reference's -b run on a bounded sample) -- and their lines are cut down to the fields below."""
    out = {}
    for w in ("small", "chr1", "branchy"):
        t0 = time.time()
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", w, "--sub-record", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--cpu-threads", str(args.cpu_threads), "--pmc-steps", str(args.pmc_steps)]
        if args.no_pmc:
            cmd.append("--no-pmc")
        try:
            env = {k: v for k, v in os.environ.items() if k not in PROFILER_ENV and not k.startswith("ROCPROF")} if not under_profiler() else dict(os.environ)
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.sub_timeout)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                out[w] = {"error": "rc %d: %s" % (p.returncode, (p.stderr or "")[-300:])}
                continue
            d = json.loads(lines[-1])
            r = d.get("roofline") or {}
            rec = {"config": d["config"]["workload"], "metric": d["metric"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "ms_per_step": d["ms_per_step"],
                   "reads_per_step": d["config"]["reads_per_step_per_gpu"], "dominant_kernel": r.get("dominant_kernel"), "dominant_kernel_ms": r.get("dominant_kernel_ms"),
                   "roofline": {"bound": r.get("bound"), "frac": r.get("frac"), "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit")},
                   "hbm": {"compulsory_frac": (r.get("hbm") or {}).get("frac"), "traffic_frac": r.get("traffic_frac"), "traffic_over_compulsory": r.get("traffic_over_compulsory"),
                           "traffic_bytes_per_read": r.get("traffic_bytes_per_read")},
                   "l2_hit_rate": r.get("l2_hit_rate"), "valu_insts_per_read": (r.get("valu_issue") or {}).get("valu_insts_per_read"),
                   "parity_sample": d.get("parity_sample"), "counters": d.get("counters"), "launch": d["config"].get("launch"), "seconds": round(time.time() - t0, 1)}
            if d.get("cpu_baseline"):
                rec["cpu_baseline"] = d["cpu_baseline"]
            out[w] = rec
            log("sub-record %s: %s Mreads/s (%.1fs)" % (w, d["value"], time.time() - t0))
        except Exception as ex:  # a sub-record never fails the line
            out[w] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    return out


def run_one_process_all_gpus(args, B, g, syn, world, mode, seed_reads, ncpu, rehearsal):
    """ONE process, N devices: bgr_devices_init fans the blob out (one upload, then device to device), one aligner and one host thread
    per device map device-resident batches through bgr_align_device -- the multi-GPU form a C/C++/cgo host of the C-ABI uses (the CLI's
    `--gpus N`), measured next to the one-process-per-GPU form of this line."""
    try:
        import torch
        if os.environ.get("BGR_BENCH_TEST_HANG") == "one_process_all_gpus":   # test hook (tests/test_gpu_parity.py): this leg never returns
            time.sleep(1e6)
        n_dev = max(1, min(world, B.device_count(), torch.cuda.device_count()))
        K, W, R, L = min(args.steps, 4), 1, args.reads_per_step, args.read_len
        # C1 of the one-process form, BOTH ways on graphs of their own (the same blob bytes): one RCCL broadcast through ncclCommInitAll
        # communicators, and xGMI peer copies in a doubling schedule; then the measured run on whatever BGR_FANOUT_AUTO picks (RCCL, else peer copies)
        fan_both = {}
        if n_dev > 1:
            for name, how_id in (("rccl_broadcast", 1), ("peer_copies", 2)):
                try:
                    g2 = B.Graph.from_blob(np.array(g.blob())) if g.blob() is not None and len(g.blob()) else None
                    if g2 is None:
                        fan_both[name] = {"error": "this rank's graph has no host blob (adopted from a device blob)"}
                        continue
                    tq = time.perf_counter()
                    g2.devices_init(0, n_dev, how_id)
                    fan_both[name] = {"ms": round((time.perf_counter() - tq) * 1e3, 2), "includes": "host -> first device upload, allocation on every device, the copies"}
                    g2.close()
                except Exception as ex:
                    fan_both[name] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
        t0 = time.perf_counter()
        how = g.devices_init(0, n_dev, 0)
        fan_ms = (time.perf_counter() - t0) * 1e3
        offs_np = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
        als, bufs, offs = [], [], []
        host_b = [syn.reads(2 * world * 1_000_000_000 + s * R, R, L, args.mismatch, seed_reads, threads=ncpu)[0] for s in range(2)]   # two batches, the same on every device
        for d in range(n_dev):
            al = B.Aligner(g, d)
            al.configure(args.waves, args.blocks_per_cu, args.lds_mphf)
            als.append(al)
            offs.append(B.DeviceBuffer(d, offs_np))
            bufs.append([B.DeviceBuffer(d, arr) for arr in host_b])   # taken in turn
        del host_b
        start = threading.Barrier(n_dev + 1)
        done = threading.Barrier(n_dev + 1)
        errs = []

        def work(d):
            try:
                for i in range(W):
                    als[d].align_device(bufs[d][i % 2].data_ptr(), offs[d].data_ptr(), R, R * L, L, m=args.mismatch, effort=args.effort, mode=mode)
                als[d].sync()
                start.wait()
                for i in range(K):
                    als[d].align_device(bufs[d][i % 2].data_ptr(), offs[d].data_ptr(), R, R * L, L, m=args.mismatch, effort=args.effort, mode=mode)
                als[d].sync()
            except Exception as ex:
                errs.append("%s: %s" % (type(ex).__name__, ex))
                try:
                    start.abort()
                except Exception:
                    pass
            finally:
                try:
                    done.wait()
                except threading.BrokenBarrierError:
                    pass

        ts = [threading.Thread(target=work, args=(d,)) for d in range(n_dev)]
        for t in ts:
            t.start()
        try:
            start.wait()
            t1 = time.perf_counter()
            done.wait()
            wall = time.perf_counter() - t1
        except threading.BrokenBarrierError:
            wall = None
        for t in ts:
            t.join()
        counters = [als[d].counters() for d in range(n_dev)] if not errs else None
        for al in als:
            al.close()
        for x in offs + [b for db in bufs for b in db]:
            x.free()
        if errs or wall is None:
            return {"error": "; ".join(errs) or "barrier broken"}
        return {"value": round(n_dev * K * R / wall / 1e6, 3), "unit": "Mreads/s", "n_gpus": n_dev, "steps": K, "reads_per_step_per_gpu": R, "ms_per_step": round(wall / K * 1e3, 4),
                "fanout_method": {0: "none (one device)", 1: "rccl broadcast (ncclCommInitAll)", 2: "peer copies, doubling schedule"}.get(how, str(how)),
                "fanout_ms": round(fan_ms, 2), "fanout_both_ways": fan_both, "graph_blob_bytes": int(g.info()["blob_bytes"]), "rehearsal_devices_shared": bool(rehearsal),
                "reads_mapped_per_device": [c["reads"] for c in counters],
                "what": "bgr_devices_init + one bgr_aligner and one host thread per device in ONE process, batches resident in each device's HBM (bgr_device_alloc), "
                        "wall time from a common start barrier to the last device's sync"}
    except Exception as ex:
        return {"error": "%s: %s" % (type(ex).__name__, ex)}


def cpu_quota():
    """CPUs this process may use at once by the cgroup's quota (cpu.max), or None"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(float(q) / float(per), 2)
    except Exception:
        return None


REFERENCE_RERUNS = [0]


def run_reference(cmd, wd):
    """The compiled reference's stdout.  Its progress block divides by the reads counted so far (alignerExhaustive.cpp:314 `overlaps/(alignedRead+notAligned)`,
    unsigned integers shared by the workers): a worker that comes through with an empty batch before another one has counted its first read dies of SIGFPE -- a race of
    the reference itself at -t N on small inputs.  Such a run is started again (at most three times; counted in cpu_baseline.reference_reruns); any other failure raises."""
    for attempt in range(4):
        p = subprocess.run(cmd, cwd=wd, stdout=subprocess.PIPE, text=True)
        if p.returncode == 0:
            return p.stdout
        if p.returncode != -8 or attempt == 3:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        REFERENCE_RERUNS[0] += 1
        for fn in os.listdir(wd):
            os.unlink(os.path.join(wd, fn))
    raise AssertionError("unreachable")


def run_cpu_baseline_exhaustive(args, al, syn, first_host, ncpu, seed_reads):
    """Exhaustive mode on the host: the compiled reference with -b (alignerExhaustive.cpp:262-318) on a bounded sample.  It writes
    nothing in that mode (alignerExhaustive.cpp:285: the fwrite is commented out), so what is compared with the GPU is the count it
    prints at the end (aligner.cpp:588-596 `Overlap and aligned`)."""
    R, L = args.reads_per_step, args.read_len
    nc = min(args.cpu_sample_exh, R)
    ref = os.path.join(ROOT, "oracle", "_ref", "bgreat")
    exe, kind = (ref, "reference") if os.path.exists(ref) else (os.path.join(ROOT, "oracle", "bgreat_oracle"), "port")
    d = tempfile.mkdtemp(prefix="bgr_cpux_")
    TEMP_DIRS.add(d)
    try:
        syn.write_unitigs(os.path.join(d, "u.fa"))
        syn.write_reads(os.path.join(d, "r.fa"), 0, nc, L, args.mismatch, seed_reads)
        open(os.path.join(d, "empty.fa"), "w").close()
        cores = min(ncpu, 255)

        def timed(reads_file, sub):
            wd = os.path.join(d, sub)
            os.makedirs(wd)
            cmd = [exe, "-r", os.path.join(d, reads_file), "-k", str(args.k), "-g", os.path.join(d, "u.fa"), "-m", str(args.mismatch), "-e", str(args.effort), "-t", str(cores), "-b"]
            t1 = time.perf_counter()
            out = run_reference(cmd, wd)
            return time.perf_counter() - t1, out

        wall, out = timed("r.fa", "tN")
        wall_idx, _ = timed("empty.fa", "iN")
        map_s = max(1e-6, wall - wall_idx)
        ref_aligned = ref_reads = None
        tail = out[out.rfind("The End"):] if "The End" in out else out
        for line in tail.splitlines():
            if line.startswith("Overlap and aligned"):
                ref_aligned = int(line.split(":")[1].split()[0])
            elif line.startswith("Reads :"):
                ref_reads = int(line.split(":")[1].split()[0])
        c_reads = first_host[: nc * L]
        c_offs = np.arange(nc + 1, dtype=np.uint64) * np.uint64(L)
        gp, gpo, gst = al.align(c_reads, c_offs, m=args.mismatch, effort=args.effort, mode=1)
        gpu_aligned = int((gpo[1:] > gpo[:-1]).sum())
        return {"value": round(nc / map_s / 1e6, 5), "unit": "Mreads/s", "cores": cores, "kind": kind, "cpu_model": cpu_model(), "host_cores_visible": len(os.sched_getaffinity(0)),
                "cpu_quota": cpu_quota(), "reference_reruns": REFERENCE_RERUNS[0],
                "sample": "first %d reads of step 0 of this workload, %s -b -t %d, wall %.2fs minus %.2fs index-only run" % (nc, os.path.basename(exe), cores, wall, wall_idx),
                "reference_reads": ref_reads, "reference_aligned": ref_aligned, "gpu_aligned": gpu_aligned,
                "gpu_matches_cpu_counters": bool(ref_aligned == gpu_aligned and ref_reads == nc)}
    except Exception as ex:
        log("cpu_baseline (exhaustive) failed: %s: %s" % (type(ex).__name__, ex))
        return {"error": "%s: %s" % (type(ex).__name__, ex)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_cpu_baseline(args, al, syn, first_host, ncpu, seed_reads):
    """The compiled reference (oracle/_ref/bgreat) on a bounded sample of the workload, on this box's host cores: `-t cores` on
    `cpu_sample` reads and `-t 1` on a tenth of them; the mapping time is the program's wall time minus an index-only run (same
    thread count, empty read file), and the reference's own `Reads/seconds` stdout line (aligner.cpp:595) is kept beside it."""
    R, L = args.reads_per_step, args.read_len
    nc = min(args.cpu_sample, R)
    n1 = max(1, min(nc, args.cpu_sample_t1))
    ref = os.path.join(ROOT, "oracle", "_ref", "bgreat")
    d = tempfile.mkdtemp(prefix="bgr_cpu_")
    TEMP_DIRS.add(d)
    try:
        syn.write_unitigs(os.path.join(d, "u.fa"))
        syn.write_reads(os.path.join(d, "r.fa"), 0, nc, L, args.mismatch, seed_reads)
        with open(os.path.join(d, "r.fa"), "rb") as f, open(os.path.join(d, "r1.fa"), "wb") as o:   # the first n1 records (2 lines each)
            for _ in range(2 * n1):
                o.write(f.readline())
        open(os.path.join(d, "empty.fa"), "w").close()
        cores = min(ncpu, 255)
        if os.path.exists(ref):
            exe, kind = ref, "reference"
        else:
            exe, kind = os.path.join(ROOT, "oracle", "bgreat_oracle"), "port"

        def timed(reads_file, threads, sub):
            wd = os.path.join(d, sub)
            os.makedirs(wd)
            cmd = [exe, "-r", os.path.join(d, reads_file), "-k", str(args.k), "-g", os.path.join(d, "u.fa"), "-m", str(args.mismatch), "-e", str(args.effort), "-t", str(threads)]
            t1 = time.perf_counter()
            out = run_reference(cmd, wd)
            wall = time.perf_counter() - t1
            own = None
            for line in out.splitlines():  # "Reads/seconds : N" = reads / (whole mapping seconds + 1), integer arithmetic (aligner.cpp:595)
                if line.startswith("Reads/seconds"):
                    own = line.split(":")[-1].strip()
            return wall, own, wd

        wall, own_line, wd = timed("r.fa", cores, "tN")
        wall_idx, _, _ = timed("empty.fa", cores, "iN")
        wall1, own_line1, _ = timed("r1.fa", 1, "t1")
        wall_idx1, _, _ = timed("empty.fa", 1, "i1")
        map_s = max(1e-6, wall - wall_idx)
        map_s1 = max(1e-6, wall1 - wall_idx1)
        # ... and with a worker per visible CPU (`-t N`, N = all host cores; coreNumber is an unsigned char, bgreat.cpp:75-77: at
        # most 255).  A GPU box's container sees all CPUs of the host but may run only its cgroup quota of them at once (cpu_quota).
        all_cores = None
        n_all = min(255, len(os.sched_getaffinity(0)))
        if args.cpu_sample_all > 0 and n_all > cores:
            na = args.cpu_sample_all
            syn.write_reads(os.path.join(d, "ra.fa"), 0, na, L, args.mismatch, seed_reads, threads=ncpu)
            wall_a, own_a, _ = timed("ra.fa", n_all, "tA")
            wall_ia, _, _ = timed("empty.fa", n_all, "iA")
            all_cores = {"value": round(na / max(1e-6, wall_a - wall_ia) / 1e6, 4), "unit": "Mreads/s", "cores": n_all, "cpu_quota": cpu_quota(),
                         "sample": "first %d reads, %s -t %d, wall %.2fs minus %.2fs index-only run" % (na, os.path.basename(exe), n_all, wall_a, wall_ia),
                         "reference_stdout_reads_per_second": own_a}
        # parity at scale: GPU records == reference records as a multiset (-t N interleaves records)
        ref_paths = open(os.path.join(wd, "paths"), "rb").read().split(b"\n")
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
        return {"value": round(nc / map_s / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": kind, "cpu_model": cpu_model(), "host_cores_visible": len(os.sched_getaffinity(0)),
                "sample": "first %d reads of step 0 of this workload, %s -t %d, wall %.2fs minus %.2fs index-only run" % (nc, os.path.basename(exe), cores, wall, wall_idx),
                "reference_stdout_reads_per_second": own_line, "cpu_quota": cpu_quota(), "reference_reruns": REFERENCE_RERUNS[0], "all_cores": all_cores,
                "t1": {"value": round(n1 / map_s1 / 1e6, 4), "unit": "Mreads/s", "cores": 1,
                       "sample": "first %d reads, %s -t 1, wall %.2fs minus %.2fs index-only run" % (n1, os.path.basename(exe), wall1, wall_idx1),
                       "reference_stdout_reads_per_second": own_line1},
                "gpu_matches_cpu_records": bool(ok)}
    except Exception as ex:
        return {"error": "%s: %s" % (type(ex).__name__, ex)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
