"""CPU-only: bench.py's own launcher (python bench.py --gpus N without torch.distributed.run)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="this node could run the two ranks")
def test_more_ranks_than_devices_is_one_error_line_not_a_launch():
    """The round-3 review: `python3 bench.py --gpus N` with no WORLD_SIZE must start by itself, and a failure must be a JSON error
    line with rc != 0, never a hang.  With fewer devices than ranks nothing is launched at all."""
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                        text=True, timeout=300, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert cp.returncode == 2, (cp.returncode, cp.stderr[-500:])
    line = json.loads(cp.stdout.strip().splitlines()[-1])
    assert "error" in line and line["n_gpus"] == 2
