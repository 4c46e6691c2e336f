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
