#!/bin/bash
# (GPU box) FASTQ pipeline: the call opened while the readers parse their first batches (default) against the late open
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3open}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x > $OUT/pytest.log 2>&1 || { tail -20 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
python3 - <<'PY' | tee $OUT/out.txt
import importlib, os, sys, subprocess, tempfile
sys.path.insert(0, os.getcwd())
synth = importlib.import_module("nimble-aligner_amd.synth")
d = tempfile.mkdtemp(prefix="nimble_e2e_", dir="/tmp")
names, seqs = synth.make_library(1000)
synth.write_library(d + "/lib.json", names, seqs)
N = 10_000_000
reads = synth.make_reads(seqs, N)
synth.write_fastq_fast(d + "/r.fastq", reads, qual="binned")
del reads
exe = "nimble-aligner_amd/lib/nimble"
def run(tag, env, reps=4):
    e = dict(os.environ, NIMBLE_HOST_TIMING="1", **env)
    ts = []
    for rep in range(reps):
        cp = subprocess.run([exe, "-r", d + "/lib.json", "-o", d + "/o.tsv", "-i", d + "/r.fastq"], capture_output=True, text=True, env=e)
        assert cp.returncode == 0, cp.stderr[-500:]
        s = [l for l in cp.stderr.splitlines() if "fastq pipeline" in l][-1]
        ts.append(float(s.split(")")[1].split("s,")[0]))
        cons = [l for l in cp.stderr.splitlines() if "consumer:" in l]
        os.remove(d + "/o.tsv")
    print("%-28s pipeline s %s  best %.1f M reads/s" % (tag, " ".join("%.3f" % t for t in ts), N / 1e6 / min(ts)), flush=True)
    if cons: print("      " + cons[-1], flush=True)
run("early open (default)", {})
run("late open", {"NIMBLE_FASTQ_LATE_OPEN": "1"})
subprocess.run(["rm", "-rf", d])
PY
