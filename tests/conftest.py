import os
import sys

import pytest

# torch ships its own HIP runtime: when both torch and libnimble_hip.so live in one process, torch has to be
# loaded first (as bench.py does), otherwise torch finds no device.  Importing it here makes the order the
# same whichever test module runs first.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the CPU-only tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
