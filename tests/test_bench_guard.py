"""bench.py's LineGuard on the CPU: the one JSON line of rank 0 comes out when an optional leg behind the timed region never returns
(the GPU suite runs the same through two gloo ranks: test_bench_line_comes_out_when_an_optional_leg_hangs)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, time
sys.path.insert(0, %r)
import bench
g = bench.LineGuard(int(sys.argv[1]), 0.5)
g.stage("e2e")
g.arm({"metric": "m", "value": 1.0, "e2e": None} if int(sys.argv[1]) == 0 else None)
if sys.argv[2] == "hang":
    time.sleep(30)
    print("NOT REACHED")
else:
    g.finish({"metric": "m", "value": 1.0, "e2e": {"value": 2.0}}, None)
    time.sleep(1.0)   # (the timer must not fire behind a finished line)
"""


def _run(rank, how):
    return subprocess.run([sys.executable, "-c", SCRIPT % ROOT, str(rank), how], capture_output=True, text=True, timeout=60)


def test_the_line_is_printed_when_a_leg_hangs_and_the_exit_code_is_zero():
    p = _run(0, "hang")
    assert p.returncode == 0 and "NOT REACHED" not in p.stdout
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and d["e2e"] is None and "e2e" in d["incomplete"]


def test_other_ranks_leave_quietly():
    p = _run(1, "hang")
    assert p.returncode == 0 and p.stdout.strip() == ""


def test_a_finished_line_is_printed_once():
    p = _run(0, "finish")
    assert p.returncode == 0
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["e2e"] == {"value": 2.0} and "incomplete" not in lines[0]


def test_dry_run_of_the_eight_rank_command_line_the_driver_uses():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W`
    with --dry-run: no device, stand-in ranks over gloo -- rendezvous, the all-reduce of ones, C1 with the real blob (every rank ends up with the
    same bytes), barriers, max over ranks, C2, the guard, ONE line on rank 0."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--dry-run"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] is None and "dry_run" in d
    m = d["multi_gpu"]
    assert m["ranks_seen"] == 8 and m["blob_bytes_equal_on_all_ranks"] is True and sorted(r["rank"] for r in m["ranks"]) == list(range(8))
    assert d["counters"]["reads"] == 8 * 4 * d["config"]["reads_per_step_per_gpu"]
    assert d["config"]["multi_gpu_ranks_seen"] == 8      # (the scalar copies the driver's record keeps)


def test_scalar_copies_put_every_number_where_the_driver_keeps_it():
    """The driver's record keeps scalar members of config / roofline / cpu_baseline only: bench.scalar_copies mirrors the nested numbers there."""
    sys.path.insert(0, ROOT)
    import bench
    line = {"value": 2800.0, "value_e2e": 210.0, "value_pcie_inclusive": 290.0, "config": {"workload": "w"}, "parity_sample": {"gpu_equals_oracle": True},
            "e2e": {"value": 210.0, "host_route": {"value": 100.0, "identical_bytes_to_the_text_route": True}},
            "roofline": {"frac": 0.85, "hbm": {"frac": 0.07}, "valu_issue": {"valu_insts_per_read": 235.0, "salu_insts_per_read": 98.0, "frac": 0.85}},
            "cpu_baseline": {"value": 0.6, "all_cores": {"value": 0.1, "cores": 255}, "t1": {"value": 0.08}},
            "other_configs": {"small": {"value": 2600.0, "ms_per_step": 0.38, "parity_sample": {"gpu_equals_oracle": True}, "roofline": {"frac": 0.75, "bound": "valu_issue"},
                                        "hbm": {"traffic_over_compulsory": 1.7, "traffic_frac": 0.09}, "l2_hit_rate": 0.88, "valu_insts_per_read": 215.0, "dominant_kernel_ms": 0.37},
                              "chr1": {"error": "timeout"},
                              "branchy": {"value": 1000.0, "ms_per_step": 2.0, "parity_sample": {"gpu_equals_oracle": True}, "roofline": {"frac": 0.75, "bound": "valu_issue"}, "hbm": {},
                                          "cpu_baseline": {"value": 0.08, "cores": 16, "sample": "s"}}},
            "full_parity": {"equal": True, "c2_reads": 50000000, "c2_equal": True},
            "multi_gpu": {"backend": "nccl (RCCL)", "ranks_seen": 8, "rccl_version": "2.22.3", "graph_broadcast_ms": 3.1, "ranks": [{"rank": 0}]},
            "one_process_all_gpus": {"value": 20000.0, "fanout_method": "rccl broadcast (ncclCommInitAll)", "fanout_ms": 12.0, "fanout_both_ways": {"rccl_broadcast": {"ms": 10.0}, "peer_copies": {"error": "x"}}}}
    d = bench.scalar_copies(line)
    c, r, b = d["config"], d["roofline"], d["cpu_baseline"]
    assert c["value_e2e"] == 210.0 and c["value_pcie_inclusive"] == 290.0 and c["e2e_host_route_mreads"] == 100.0 and c["e2e_routes_identical_bytes"] is True
    assert c["small_mreads"] == 2600.0 and c["branchy_mreads"] == 1000.0 and c["chr1_error"] == "timeout" and c["small_parity_ok"] is True and c["parity_sample_ok"] is True
    assert c["full_parity_equal"] is True and c["full_parity_c2_reads"] == 50000000
    assert c["multi_gpu_ranks_seen"] == 8 and c["multi_gpu_rccl_version"] == "2.22.3" and c["one_process_all_gpus_mreads"] == 20000.0
    assert c["one_process_fanout_rccl_broadcast_ms"] == 10.0 and c["one_process_fanout_peer_copies_ms"] is None
    assert r["valu_insts_per_read"] == 235.0 and r["hbm_compulsory_frac"] == 0.07 and r["small_frac"] == 0.75 and r["small_traffic_over_compulsory"] == 1.7 and r["branchy_bound"] == "valu_issue"
    assert b["all_cores_value"] == 0.1 and b["t1_value"] == 0.08 and b["exhaustive_value"] == 0.08 and b["exhaustive_cores"] == 16
    for obj in (c, r, b):   # (what the copies add are scalars)
        for k, v in obj.items():
            if k not in ("hbm", "valu_issue", "all_cores", "t1"):
                assert isinstance(v, (bool, int, float, str)) or v is None, (k, v)
    json.dumps(d)
