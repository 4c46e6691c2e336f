import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_bins():
    """Build (if needed) and return paths of the parity checkers: oracle CLI, oracle .so, reference binaries."""
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-s", "-C", odir, "liboracle.so", "bgreat_oracle"])
    ref = os.path.join(odir, "_ref", "bgreat")
    ref_exh = os.path.join(odir, "_ref", "bgreat_exh")
    if os.path.isdir("/root/reference") and not (os.path.exists(ref) and os.path.exists(ref_exh)):
        subprocess.check_call(["make", "-s", "-C", odir, "ref"])
    return {
        "cli": os.path.join(odir, "bgreat_oracle"),
        "so": os.path.join(odir, "liboracle.so"),
        "ref": ref if os.path.exists(ref) else None,
        "ref_exh": ref_exh if os.path.exists(ref_exh) else None,
    }
