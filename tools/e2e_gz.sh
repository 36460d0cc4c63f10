#!/bin/bash
# .fastq.gz end-to-end rate (development aid) through lib/nimble: a single-member stream of E2E_READS (default 16 M) reads of
# 150 bp with binned qualities written pigz-fashion (synth.gzip_single_stream), and `gzip -6` itself on a quarter of them;
# the many-threaded reader against the one-thread zlib reader, and the stage timers of the reader (NIMBLE_GZIP_DEBUG).
python3 - <<'PY'
import importlib, os, sys, subprocess, time, tempfile
sys.path.insert(0, os.getcwd())
synth = importlib.import_module("nimble-aligner_amd.synth")
d = tempfile.mkdtemp(prefix="nimble_e2e_", dir="/tmp")
names, seqs = synth.make_library(1000)
synth.write_library(d + "/lib.json", names, seqs)
N = int(os.environ.get('E2E_READS', '16000000'))
reads = synth.make_reads(seqs, N)
synth.write_fastq_fast(d + "/r.fastq", reads, qual="binned")
synth.write_fastq_fast(d + "/q.fastq", reads[:N // 4], qual="binned")
raw = os.path.getsize(d + "/r.fastq")
t = time.time()
bg = subprocess.Popen("gzip -6 -c %s/q.fastq > %s/q6.fastq.gz" % (d, d), shell=True)
synth.gzip_single_stream(d + "/r.fastq", d + "/r6.fastq.gz", 6)
assert bg.wait() == 0
print("%d reads, %.2f GB plain, %.2f GB as one gzip member (level 6); gzip -6 of %d reads %.2f GB (%.0f s to compress)" % (
    N, raw / 1e9, os.path.getsize(d + "/r6.fastq.gz") / 1e9, N // 4, os.path.getsize(d + "/q6.fastq.gz") / 1e9, time.time() - t), flush=True)
exe = "nimble-aligner_amd/lib/nimble"
def run(tag, path, n, env, reps=2, show=False):
    e = dict(os.environ, NIMBLE_HOST_TIMING="1", **env)
    best = None
    for rep in range(reps):
        cp = subprocess.run([exe, "-r", d + "/lib.json", "-o", d + "/o.tsv", "-i", path], capture_output=True, text=True, env=e)
        assert cp.returncode == 0, cp.stderr[-500:]
        s = [l for l in cp.stderr.splitlines() if "fastq pipeline" in l][-1]
        t = float(s.split(")")[1].split("s,")[0])
        best = t if best is None else min(best, t)
        tsv = open(d + "/o.tsv", "rb").read()
        os.remove(d + "/o.tsv")
    print("%-50s pipeline %.3f s  %.2f M reads/s" % (tag, best, n / 1e6 / best), flush=True)
    if show:
        print("\n".join("    " + l for l in cp.stderr.splitlines() if l.startswith("[pgzip]") and "chunk " not in l), flush=True)
    return tsv
want = run("plain, %d reads" % N, d + "/r.fastq", N, {})
wantq = run("plain, %d reads" % (N // 4), d + "/q.fastq", N // 4, {})
assert run("gzip -6 (gzip itself), one thread (zlib)", d + "/q6.fastq.gz", N // 4, {"NIMBLE_GZIP_SERIAL": "1"}, reps=1) == wantq
assert run("gzip -6 (gzip itself), defaults", d + "/q6.fastq.gz", N // 4, {"NIMBLE_GZIP_DEBUG": "1"}, show=True) == wantq
p = d + "/r6.fastq.gz"
assert run("one member, defaults", p, N, {"NIMBLE_GZIP_DEBUG": "1"}, show=True) == want
for pt in (2, 4, 8):
    assert run("one member, %d parser threads" % pt, p, N, {"NIMBLE_FASTQ_THREADS": str(pt)}) == want
for gt, pt in ((14, 4), (12, 4), (16, 4), (20, 4)):
    assert run("one member, %d decoder + %d parser threads" % (gt, pt), p, N, {"NIMBLE_GZIP_THREADS": str(gt), "NIMBLE_FASTQ_THREADS": str(pt)}) == want
assert run("one member, 128 MiB windows, 4 parser threads", p, N, {"NIMBLE_GZIP_WINDOW": str(128 << 20), "NIMBLE_FASTQ_THREADS": "4"}) == want
subprocess.run(["rm", "-rf", d])
PY
