"""Multi-GPU readiness without the hardware: the bookkeeping of bgr_devices_init (bgreat_amd/csrc/fanout.h) driven with stand-in
devices for N in {1, 2, 3, 8} -- collective available / missing / failing half way, peer doubling, devices already resident,
allocation and copy failures (nothing registered, nothing leaked).  Built with ASan+UBSan.  CPU only."""
import os
import subprocess

from util import ROOT


def test_devices_init_bookkeeping_with_stand_in_devices(tmp_path):
    exe = str(tmp_path / "fanout_mock")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "bgreat_amd", "csrc"),
                           os.path.join(ROOT, "tests", "fanout_mock.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "ALL OK" in p.stdout and "FAIL" not in p.stdout and p.stdout.count(" ok ") == 31
    assert "ERROR: " not in p.stderr, p.stderr[-3000:]
