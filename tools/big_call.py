#!/usr/bin/env python3
"""One call far past 2^32 input bytes (development aid): tools/big_call.py [reads=60000000]
   checks call(reads ++ reads) == call(reads) and that a shuffled half + the rest add up, at a size where byte offsets
   into the read buffer need 64 bits (60 M x 150 B = 9 GB)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
names, seqs = synth.make_library(1000)
lib = nim.Library(text=json.dumps(synth.library_json(names, seqs)), strand_filter="unstranded").build_index(0)
half = n // 2
reads = synth.make_reads_torch(seqs, half, device="cuda:0")
torch.cuda.synchronize()
base = lib.score_call(reads, None, n=half, fixed_len=150, mem=nim.MEM_DEVICE)
doubled = torch.cat([reads, reads], dim=0).contiguous()
torch.cuda.synchronize()
t0 = time.time()
big = lib.score_call(doubled, None, n=2 * half, fixed_len=150, mem=nim.MEM_DEVICE)
print("reads", 2 * half, "bytes", doubled.numel(), "rows", len(big), "call %.3fs" % (time.time() - t0), "equal", big == base, flush=True)
ctx = lib.device_context(); ctx.n = 2 * half
print(ctx.timing(), ctx.counters())
# the tail of the buffer (past 4 GiB) really is read: change the last reads and see the table move
doubled[-100000:] = synth.make_reads_torch(seqs, 100000, seed=12345, device="cuda:0")  # new keys, only at the very end
torch.cuda.synchronize()
changed = lib.score_call(doubled, None, n=2 * half, fixed_len=150, mem=nim.MEM_DEVICE)
print("tail edit changes the table:", changed != base)
assert big == base and changed != base
print("OK")
