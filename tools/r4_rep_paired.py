import importlib, json, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
names, seqs = synth.make_library(200)
rn, rs = synth.expand_rows(names, seqs)
idx = nim.Index(rs)
ctx = nim.Context(idx)
ctx.set_counters(True)
nmm = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = nim.AlignParams.make(0.08, 12, nmm)
r1, r2 = synth.make_reads(seqs, 20000, paired=True)
o = synth.fixed_offsets(r1.shape[0], r1.shape[1])
for rep in range(int(os.environ.get("REPS", "8"))):
    ctx.call(p, r1.reshape(-1), o, r2.reshape(-1), o)
    print(rep, ctx.counters(), flush=True)
