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
