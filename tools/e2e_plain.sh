#!/bin/bash
# plain-FASTQ end-to-end rate (development aid): lib/nimble on E2E_READS (default 32 M) reads of 150 bp, parser variants
python3 - <<'PY'
import importlib, os, sys, subprocess, time, tempfile
sys.path.insert(0, os.getcwd())
synth = importlib.import_module("nimble-aligner_amd.synth")
d = tempfile.mkdtemp(prefix="nimble_e2e_", dir="/tmp")
names, seqs = synth.make_library(1000)
synth.write_library(d + "/lib.json", names, seqs)
N = int(os.environ.get('E2E_READS', '32000000'))
reads = synth.make_reads(seqs, N)
synth.write_fastq_fast(d + "/r.fastq", reads, qual="binned")
del reads
exe = "nimble-aligner_amd/lib/nimble"
def run(tag, env, reps=3):
    e = dict(os.environ, NIMBLE_HOST_TIMING="1", **env)
    ts = []
    for rep in range(reps):
        cp = subprocess.run([exe, "-r", d + "/lib.json", "-o", d + "/o.tsv", "-i", d + "/r.fastq"], capture_output=True, text=True, env=e)
        assert cp.returncode == 0, cp.stderr[-500:]
        s = [l for l in cp.stderr.splitlines() if "fastq pipeline" in l][-1]
        ts.append(float(s.split(")")[1].split("s,")[0]))
        cons = [l for l in cp.stderr.splitlines() if "consumer:" in l]
        os.remove(d + "/o.tsv")
    print("%-40s pipeline s %s  best %.1f M reads/s" % (tag, " ".join("%.3f" % t for t in ts), N / 1e6 / min(ts)), flush=True)
    if cons: print("      " + cons[-1], flush=True)
run("defaults", {})
run("buffers not page-locked", {"NIMBLE_FASTQ_NO_PIN": "1"})
run("ASCII batches (no host packing)", {"NIMBLE_FASTQ_PACK": "0"})
subprocess.run(["rm", "-rf", d])
PY
