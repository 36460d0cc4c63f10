"""bench.py itself, at toy sizes: every workload runs, prints one JSON line with the contract's keys, and its built-in
parity check (GPU table == CPU oracle table on the sample) passes -- the paired workload once went through with its
mates in the offsets argument and nobody ran it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload,extra", [
    ("configs2", ["--reads", "60000", "--features", "64", "--e2e-reads", "20000", "--packed-input", "1"]),
    ("configs3", ["--reads", "30000", "--features", "64", "--e2e-reads", "0"]),
    ("families100", ["--reads", "20000", "--e2e-reads", "0"]),
])
def test_bench_workloads_run_and_check_parity(workload, extra):
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "3", "--warmup", "1",
                         "--cpu-sample", "20000"] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert cp.returncode == 0, cp.stderr[-800:]
    assert "HSA_STATUS_ERROR" not in cp.stderr
    d = json.loads(cp.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 3
    assert d["cpu_baseline"]["parity_on_sample"].startswith("bit-exact")
    assert d["config"]["paired"] == (workload == "configs3")
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    if workload == "configs2":
        assert d["packed_input"]["reads_per_s"] > 0
        assert d["e2e_fastq_reads_per_s"] > 0 and d["e2e_fastq_gz_reads_per_s"] > 0
