"""Adapter-trimmed input: reads of 40..150 bases with offsets, device resident, against the fixed-length form."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
names, seqs = synth.make_library(1000)
path = "/tmp/ragged_probe_lib.json"
synth.write_library(path, names, seqs)
lib = nim.Library(path, "unstranded").build_index()
n, L = 4_000_000, 150
reads = synth.make_reads_torch(seqs, n, L, device="cuda:0")
ctx = lib.device_context()
def best_of(f):
    best = None
    for _ in range(3):
        f()
        t = ctx.timing()
        if best is None or t["total"] < best["total"]:
            best = t
    return best
t = best_of(lambda: lib.score_call_raw(reads, None, n=n, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE))
print("fixed 150          : pack %.3f align %.3f dedup %.3f total %.3f ms" % (t["pack"], t["align"], t["dedup"], t["total"]), flush=True)
g = torch.Generator(device="cuda:0"); g.manual_seed(3)
for lo in (150, 100, 40):
    lens = torch.randint(lo, L + 1, (n,), device="cuda:0", generator=g)
    mask = torch.arange(L, device="cuda:0")[None, :] < lens[:, None]
    flat = reads[mask].contiguous()
    off = torch.zeros(n + 1, dtype=torch.int64, device="cuda:0")
    off[1:] = torch.cumsum(lens, 0)
    off = off.to(torch.uint64) if hasattr(torch, "uint64") else off
    torch.cuda.synchronize()
    t = best_of(lambda: lib.score_call_raw(flat, off, n=n, max_len=L, mem=nim.MEM_DEVICE))
    print("ragged %3d..150    : pack %.3f align %.3f dedup %.3f total %.3f ms  (%.0f MB of bases)" %
          (lo, t["pack"], t["align"], t["dedup"], t["total"], flat.numel() / 1e6), flush=True)
