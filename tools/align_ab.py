"""Stage timing of one call on device-resident synthetic reads (development aid): python tools/align_ab.py [T] [N] [REP]
Run once per build / environment (NIMBLE_ALIGN_V1=1 selects the one-kernel align stage)."""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 4
PAIRED = os.environ.get("AB_PAIRED", "0") != "0"
names, seqs = synth.make_library(T)
lib = nim.Library(text=json.dumps(synth.library_json(names, seqs)), strand_filter="unstranded").build_index(0)
ctx = lib.device_context()
if PAIRED:
    r1, r2 = synth.make_reads(seqs, N, paired=True)
    reads = torch.from_numpy(r1).to("cuda:0")
    mates = torch.from_numpy(r2).to("cuda:0")
else:
    reads = synth.make_reads_torch(seqs, N, L=150, seed=synth.READ_SEED, device="cuda:0")
    mates = None
torch.cuda.synchronize()
if os.environ.get("AB_SORT", "0") != "0" and not PAIRED:
    # upper bound of what grouping reads by graph region buys: order the reads by the first row of their class
    import numpy as np
    lib.score_call_raw(reads, None, n=N, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE)
    ctx.n = N
    cls = ctx.read_records(0)["cls"].astype(np.int64)
    import ctypes as C
    ixh = nim.host_lib().nimble_library_index(lib.h)
    uniq = np.unique(cls)
    first = {}
    for c in uniq:
        c = int(c)
        if c == nim.CLASS_NONE:
            first[c] = 1 << 40
            continue
        one = np.zeros(1, dtype=np.uint32)
        ln = C.c_uint32(0)
        nim.hip_lib().nimble_class_get(ixh, c, one.ctypes.data, 1, C.byref(ln))
        first[c] = int(one[0])
    key = np.array([first[int(c)] for c in uniq], dtype=np.int64)
    order = np.argsort(key[np.searchsorted(uniq, cls)], kind="stable")
    reads = reads[torch.from_numpy(order).to("cuda:0")].contiguous()
    torch.cuda.synchronize()
    print("reads sorted by first class row", flush=True)
sig = None
for rep in range(REP):
    ctx.set_counters(rep == 0)
    rows = lib.score_call_raw(reads, None, mates, None, n=N, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE)
    t = ctx.timing()
    import hashlib
    s = hashlib.sha1(repr(rows.to_list()).encode()).hexdigest()
    sig = sig or s
    assert s == sig
    print("rep", rep, {k: round(v, 4) for k, v in t.items()}, "rows", len(rows), flush=True)
    if rep == 0:
        ctx.n = N
        print("counters", ctx.counters(), flush=True)
print("signature", sig)
