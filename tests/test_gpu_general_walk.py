"""The general walk (walk(), the launch of indexes with wide classes) on indexes that would take the fast walk: NIMBLE_FAST_ALIGN=0
is read once per process, so the parity suite runs once more in ONE child process with the knob set."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_parity_suite_with_the_general_walk_forced():
    env = dict(os.environ, NIMBLE_FAST_ALIGN="0")
    cp = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                         "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert cp.returncode == 0, cp.stdout[-3000:] + cp.stderr[-1000:]
    assert " passed" in cp.stdout
