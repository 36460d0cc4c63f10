#!/bin/bash
# plain-FASTQ end-to-end rate against the number of parser threads (development aid)
python3 - <<'PY'
import importlib, os, sys, subprocess, time, tempfile
sys.path.insert(0, os.getcwd())
synth = importlib.import_module("nimble-aligner_amd.synth")
d = tempfile.mkdtemp(prefix="nimble_e2e_", dir="/tmp")
names, seqs = synth.make_library(1000)
synth.write_library(d + "/lib.json", names, seqs)
N = int(os.environ.get('E2E_READS', '8000000'))
reads = synth.make_reads(seqs, N)
synth.write_fastq_fast(d + "/r.fastq", reads)
exe = "nimble-aligner_amd/lib/nimble"
def run(tag, env):
    e = dict(os.environ, NIMBLE_HOST_TIMING="1", **env)
    best = None
    for rep in range(2):
        cp = subprocess.run([exe, "-r", d + "/lib.json", "-o", d + "/o.tsv", "-i", d + "/r.fastq"], capture_output=True, text=True, env=e)
        assert cp.returncode == 0, cp.stderr[-500:]
        s = [l for l in cp.stderr.splitlines() if "fastq pipeline" in l][-1]
        t = float(s.split(")")[1].split("s,")[0])
        best = t if best is None else min(best, t)
        os.remove(d + "/o.tsv")
    print("%-28s pipeline %.3f s  %.1f M reads/s" % (tag, best, N / 1e6 / best), flush=True)
pass
for t, c in ((24, 8), (24, 16), (32, 8), (32, 16), (48, 16)):
    run("parallel, %2d threads, %2d MiB" % (t, c), {"NIMBLE_FASTQ_THREADS": str(t), "NIMBLE_FASTQ_CHUNK": str(c << 20)})
run("defaults", {})
subprocess.run(["rm", "-rf", d])
PY
