"""The other ways through k_align that the README's knobs select -- the general walk on indexes that would take the fast one
(NIMBLE_FAST_ALIGN=0), no "several left flanks" set (NIMBLE_LOCAL_RESEED=0), no first level in front of the presence filter
(NIMBLE_FILTER_L1=0), class masks that are not relative to the component (NIMBLE_UNIFORM_WINDOWS=0).  The knobs are read once
per process, so the parity suite runs once more in ONE child process per setting."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("knob", ["NIMBLE_FAST_ALIGN", "NIMBLE_LOCAL_RESEED", "NIMBLE_FILTER_L1", "NIMBLE_UNIFORM_WINDOWS"])
def test_parity_suite_with_a_knob_off(knob):
    env = dict(os.environ)
    env[knob] = "0"
    cp = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                         "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert cp.returncode == 0, cp.stdout[-3000:] + cp.stderr[-1000:]
    assert " passed" in cp.stdout
