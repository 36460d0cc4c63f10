import gzip, os, subprocess, sys, tempfile, importlib
sys.path.insert(0, os.getcwd())
synth = importlib.import_module("nimble-aligner_amd.synth")
d = tempfile.mkdtemp(prefix="nimble_edge_", dir="/tmp")
names, seqs = synth.make_library(16)
synth.write_library(d + "/lib.json", names, seqs)
exe = "nimble-aligner_amd/lib/nimble"
def run(tag, files, env=None):
    cp = subprocess.run([exe, "-r", d + "/lib.json", "-o", d + "/o.tsv", "-i"] + files, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    out = open(d + "/o.tsv").read() if os.path.exists(d + "/o.tsv") else None
    print("%-46s rc=%d  tsv=%r  %s" % (tag, cp.returncode, None if out is None else out[:60], cp.stderr.strip().splitlines()[-1][:90] if cp.returncode else ""), flush=True)
    if os.path.exists(d + "/o.tsv"): os.remove(d + "/o.tsv")
open(d + "/empty.fastq", "w").close()
run("empty plain file", [d + "/empty.fastq"])
with gzip.open(d + "/empty.fastq.gz", "wb") as f: pass
run("empty gzip file", [d + "/empty.fastq.gz"])
open(d + "/short.fastq", "w").write("@a\nACGT\n+\nIIII\n@b\nACGTACGTAC\n+\nIIIIIIIIII\n")
run("reads shorter than a k-mer", [d + "/short.fastq"])
open(d + "/nn.fastq", "w").write("".join("@n%d\n%s\n+\n%s\n" % (i, "N" * 150, "I" * 150) for i in range(5)))
run("all-N reads", [d + "/nn.fastq"])
open(d + "/one.fastq", "w").write("@x\n%s\n+\n%s\n" % (seqs[0][:150].upper(), "I" * 150))
run("one on-target read", [d + "/one.fastq"])
run("one on-target read, no newline at the end", [d + "/one_nn.fastq"]) if False else None
open(d + "/one_nn.fastq", "w").write("@x\n%s\n+\n%s" % (seqs[0][:150].upper(), "I" * 150))
run("one read, file ends without a newline", [d + "/one_nn.fastq"])
open(d + "/long.fastq", "w").write("@x\n%s\n+\n%s\n" % ((seqs[0] * 40)[:70000].upper(), "I" * 70000))
run("a 70 000-base read", [d + "/long.fastq"])
run("R1 one read, R2 empty", [d + "/one.fastq", d + "/empty.fastq"])
run("missing file", [d + "/nothing.fastq"])
run("whole-file mode, empty", [d + "/empty.fastq"], {"NIMBLE_FASTQ_BATCH": "0"})
subprocess.run(["rm", "-rf", d])
