"""One heavy-duplication case for the profiler: tools/dup_one.py <percent of reads that are copies of one read>"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
frac = float(sys.argv[1]) / 100.0
names, seqs = synth.make_library(1000)
path = "/tmp/dup_one_lib.json"
synth.write_library(path, names, seqs)
lib = nim.Library(path, "unstranded").build_index()
n, L = 10_000_000, 150
r = synth.make_reads_torch(seqs, n, L, device="cuda:0")
k = int(n * frac)
if k:
    idx = torch.randperm(n, device="cuda:0")[:k]
    r[idx] = r[12345].clone()
torch.cuda.synchronize()
for _ in range(3):
    lib.score_call_raw(r, None, n=n, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE)
print("dedup ms", lib.device_context().timing()["dedup"])
