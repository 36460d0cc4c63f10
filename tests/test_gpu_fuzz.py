"""A short run of the randomised differential test (tools/fuzz_parity.py): random libraries, settings and reads, the
product's table and per-read records against the CPU oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_parity(seed):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "120", str(seed)],
                        capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert cp.returncode == 0 and "FUZZ OK" in cp.stdout, cp.stdout[-2000:] + cp.stderr[-2000:]
