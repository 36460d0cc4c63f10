"""One allele-family configuration with work counters: python tools/family_one.py [T] [F] [N]"""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4_000_000
names, seqs = synth.make_family_library(T, F)
lib = nim.Library(text=json.dumps(synth.library_json(names, seqs)), strand_filter="unstranded").build_index(0)
reads = synth.make_reads_torch(seqs, n, 150, device="cuda:0")
torch.cuda.synchronize()
ctx = lib.device_context()
for rep in range(4):
    ctx.set_counters(rep == 0)
    rows = lib.score_call_raw(reads, None, n=n, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE)
    print("rep", rep, {k: round(v, 3) for k, v in ctx.timing().items()}, "rows", len(rows), flush=True)
    if rep == 0:
        ctx.n = n
        c = ctx.counters()
        print("counters", c, "nodes/seeded read %.1f" % (c["nodes"] / max(c["seeded"], 1)), flush=True)
