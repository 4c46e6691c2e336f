import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--bgr-options", default="", help="library options for the whole run, name=value[,name=value] (bgr_set_option), e.g. poison_device_buffers=1")


@pytest.fixture(scope="session", autouse=True)
def library_options_of_the_run(request):
    spec = request.config.getoption("--bgr-options")
    if spec:
        import bgreat_amd as B
        B.set_options_from_string(spec)
    yield


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


@pytest.fixture(autouse=True)
def library_options_put_back():
    """Process-wide library options (bgr_set_option) a test changed are put back behind it: the tests of a run share one process."""
    try:
        import bgreat_amd as B
        before = {n: B.get_option(n) for n, _ in B.option_names()}
    except Exception:   # (library not built: the tests that need it say so themselves)
        yield
        return
    yield
    for n, v in before.items():
        if B.get_option(n) != v:
            B.set_option(n, v)
