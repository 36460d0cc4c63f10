#!/usr/bin/env python3
"""End-to-end rate of the FASTQ pipeline (SURVEY 8(d) figure B): lib/nimble on a synthetic FASTQ file, parse
included.  usage: tools/e2e_fastq.py [reads=4000000] [features=1000] [gz_reads=1000000]"""
import importlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("nimble-aligner_amd.synth")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
NGZ = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
exe = os.path.join(ROOT, "nimble-aligner_amd", "lib", "nimble")
d = tempfile.mkdtemp(prefix="nimble_e2e_", dir="/tmp")
names, seqs = synth.make_library(T)
libp = os.path.join(d, "lib.json")
synth.write_library(libp, names, seqs)
t0 = time.time()
reads = synth.make_reads(seqs, N)
fq = os.path.join(d, "reads.fastq")
synth.write_fastq_fast(fq, reads)
print("generated %d reads, %.2f GB FASTQ in %.1fs" % (N, os.path.getsize(fq) / 1e9, time.time() - t0), flush=True)
fqz = os.path.join(d, "reads_gz.fastq.gz")
synth.write_fastq_fast(os.path.join(d, "reads_gz.fastq"), reads[:NGZ])
subprocess.run(["gzip", "-1", "-f", os.path.join(d, "reads_gz.fastq")], check=True)
res = {}
def run(tag, inp, n, env_extra):
    out = os.path.join(d, tag + ".tsv")
    env = dict(os.environ, NIMBLE_HOST_TIMING="1", **env_extra)
    t = time.time()
    cp = subprocess.run([exe, "-r", libp, "-o", out, "-i", inp, "-f", "unstranded"], capture_output=True, text=True, env=env)
    wall = time.time() - t
    assert cp.returncode == 0, cp.stderr[-2000:]
    pipe = [l for l in cp.stderr.splitlines() if "fastq pipeline" in l]
    secs = float(pipe[-1].split(")")[1].split("s,")[0]) if pipe else None
    res[tag] = dict(reads=n, wall_s=round(wall, 3), pipeline_s=secs, reads_per_s_pipeline=n / secs if secs else None,
                    reads_per_s_wall=n / wall)
    print(tag, json.dumps(res[tag]), flush=True)
    return open(out).read()
run("warm", fq, N, {})  # page cache + first-touch of the GPU
a = run("plain_streamed", fq, N, {})
b = run("plain_whole_file", fq, N, {"NIMBLE_FASTQ_BATCH": "0"})
assert a == b
c = run("gz_streamed", fqz, NGZ, {})
e = run("gz_whole_file", fqz, NGZ, {"NIMBLE_FASTQ_BATCH": "0"})
assert c == e
print("E2E " + json.dumps(res))
subprocess.run(["rm", "-rf", d])
