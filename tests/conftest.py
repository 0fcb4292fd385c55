import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: imports /root/reference (authoring container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/models/dino")
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    for item in items:
        if "needs_reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def oracle_clib():
    """The plain-C oracle (oracle/msda_ref.c), built on demand with gcc."""
    import ctypes
    import subprocess
    so = os.path.join(ROOT, "oracle", "libmsda_ref.so")
    src = os.path.join(ROOT, "oracle", "msda_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(so)
