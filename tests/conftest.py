import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build what is missing (oracle, host test shim, libpgmove.so). Everything is prebuilt by
    __graft_entry__.build(); this only covers a fresh checkout."""
    need = [os.path.join(ROOT, "oracle", "libgmove_oracle.so"), os.path.join(ROOT, "oracle", "gmove_oracle"),
            os.path.join(ROOT, "poregen_amd", "_pg_hosttest.so"), os.path.join(ROOT, "poregen_amd", "libpgmove.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "-s"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
