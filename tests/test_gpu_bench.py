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
    ("configs2", ["--reads", "60000", "--features", "64", "--e2e-reads", "20000", "--e2e-bam-pairs", "20000", "--packed-input", "1"]),
    ("configs3", ["--reads", "30000", "--features", "64", "--e2e-reads", "0"]),
    ("families100", ["--reads", "20000", "--e2e-reads", "0"]),
    ("families500", ["--reads", "20000", "--e2e-reads", "0"]),
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
        assert d["e2e_bam_reads_per_s"] > 0, d.get("e2e_bam_error")


@pytest.mark.parametrize("extra", [["--force-sharded"], ["--native-virtual", "3"]])
def test_bench_native_form_of_the_multi_gpu_step(extra):
    """`--form native`: the multi-GPU step through the C ABI alone (one process, one native thread per rank) -- over real RCCL
    with one rank, and with three ranks sharing this box's GPU; the line reports the native figures and its own parity
    check (the job's table == one call over the union of the ranks' reads)."""
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--form", "native", "--steps", "4", "--warmup", "1",
                         "--reads", "40000", "--features", "64", "--cpu-sample", "0", "--e2e-reads", "0", "--packed-input", "0"]
                        + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert cp.returncode == 0, cp.stderr[-800:]
    d = json.loads(cp.stdout.strip().splitlines()[-1])
    nat = d["native"]
    assert "error" not in nat, nat
    assert nat["parity_on_union"].startswith("bit-exact")
    assert nat["ranks"] == (3 if "--native-virtual" in extra else 1) and nat["rccl"] == ("--force-sharded" in extra)
    assert d["value"] == nat["reads_per_s"] and d["ms_per_step"] == nat["ms_per_step"]


def test_bench_native_child_prints_the_native_object_alone():
    """`--native-child N`: what rank 0 of a torch.distributed.run launch starts (as a process of its own, with a time limit) for
    the native form -- here with three ranks sharing this box's GPU: one JSON object, the native leg's, parity-checked."""
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--native-child", "3", "--native-virtual", "3", "--gpus", "3",
                         "--form", "native", "--steps", "4", "--warmup", "1", "--reads", "40000", "--features", "64"],
                        capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert cp.returncode == 0, cp.stderr[-800:]
    nat = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
    assert "error" not in nat and nat["ranks"] == 3 and nat["parity_on_union"].startswith("bit-exact")
    assert nat["reads_per_s"] > 0 and "metric" not in nat
