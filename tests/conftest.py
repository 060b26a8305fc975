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
def oracle_bin():
    """The C oracle (test infrastructure).  Built on demand with gcc."""
    exe = os.path.join(ROOT, "oracle", "straincall_oracle")
    lib = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(exe) and os.path.exists(lib)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
    return exe
