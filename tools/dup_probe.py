"""Heavy duplication: what do 1 % / 5 % / 20 % copies of ONE read do to the dedup stage?  (same-address atomics)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
names, seqs = synth.make_library(1000)
path = "/tmp/dup_probe_lib.json"
synth.write_library(path, names, seqs)
lib = nim.Library(path, "unstranded").build_index()
n, L = 10_000_000, 150
reads = synth.make_reads_torch(seqs, n, L, device="cuda:0")
ctx = lib.device_context()
for frac in (0.0, 0.01, 0.05, 0.2, 0.5):
    r = reads.clone()
    k = int(n * frac)
    if k:
        idx = torch.randperm(n, device="cuda:0")[:k]
        r[idx] = reads[12345]           # an on-target read
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        rows = lib.score_call_raw(r, None, n=n, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE)
        t = ctx.timing()
        if best is None or t["total"] < best["total"]:
            best = t
    print("copies of one read: %4.0f %%   dedup %.3f ms  align %.3f ms  total %.3f ms" % (100 * frac, best["dedup"], best["align"], best["total"]), flush=True)

# a hot CLASS instead of a hot key: a share of the reads are distinct reads of ONE feature (same callset, different keys)
import numpy as np
feat = np.frombuffer(seqs[0].upper().encode(), dtype=np.uint8)
for frac in (0.05, 0.3, 0.8):
    r = reads.clone()
    k = int(n * frac)
    idx = torch.randperm(n, device="cuda:0")[:k]
    g = torch.Generator(device="cuda:0"); g.manual_seed(7)
    start = torch.randint(0, len(feat) - L + 1, (k,), device="cuda:0", generator=g)
    ft = torch.from_numpy(feat.copy()).to("cuda:0")
    win = ft[start[:, None] + torch.arange(L, device="cuda:0")[None, :]]
    # one substitution at a random place makes the keys distinct far beyond the number of start positions
    pos = torch.randint(0, L, (k,), device="cuda:0", generator=g)
    sub = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda:0")[torch.randint(0, 4, (k,), device="cuda:0", generator=g)]
    win[torch.arange(k, device="cuda:0"), pos] = sub
    r[idx] = win
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        rows = lib.score_call_raw(r, None, n=n, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE)
        t = ctx.timing()
        if best is None or t["total"] < best["total"]:
            best = t
    print("reads of one feature:   %4.0f %%   dedup %.3f ms  align %.3f ms  total %.3f ms" % (100 * frac, best["dedup"], best["align"], best["total"]), flush=True)
