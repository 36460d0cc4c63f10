"""Ad-hoc stage timing of the device path on synthetic reads (development aid, not the bench contract)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 3
t0 = time.time()
names, seqs = synth.make_library(T)
rn, rs = synth.expand_rows(names, seqs)
idx = nim.Index(rs)
print("index", idx.stats(), "build %.2fs" % (time.time() - t0), flush=True)
t0 = time.time()
reads = synth.make_reads(seqs, N)
print("reads generated %.1fs" % (time.time() - t0), flush=True)
ctx = nim.Context(idx)
p = nim.AlignParams.make(0.33, 50, 0)
flat = reads.reshape(-1)
for rep in range(REP):
    t0 = time.time()
    ctx.call(p, flat, None, n=N, fixed_len=150, max_len=150)
    ctx.synchronize()
    wall = time.time() - t0
    t = ctx.timing()
    print("rep", rep, "wall(incl H2D) %.3fs" % wall, {k: round(v, 3) for k, v in t.items()},
          "Mreads/s(device) %.1f" % (N / t["total"] / 1e3), flush=True)
print(ctx.counters())
print("hist entries", len(ctx.histogram()))
