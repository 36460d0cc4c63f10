#!/usr/bin/env python3
"""Many calls of random sizes through every form of the call (plain, slots in flight, streamed, BAM-mode, split):
results must stay equal to a reference computed once, device memory must not creep (development aid)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
names, seqs = synth.make_library(300)
lib = nim.Library(text=json.dumps(synth.library_json(names, seqs)), strand_filter="unstranded").build_index(0)
rng = np.random.default_rng(1)
pool = synth.make_reads(seqs, 400_000, seed=3)
p1, p2 = synth.make_reads(seqs, 200_000, paired=True, seed=4)
ref_cache = {}
def ref(kind, a, b):
    key = (kind, a, b)
    if key not in ref_cache:
        if kind == "se":
            ref_cache[key] = lib.score_call(pool[a:b].reshape(-1), None, n=b - a, fixed_len=150)
        else:
            o = synth.fixed_offsets(b - a, 150)
            ref_cache[key] = lib.score_call(p1[a:b].reshape(-1), o, p2[a:b].reshape(-1), o)
    return ref_cache[key]
free0 = None
t0 = time.time()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for it in range(N):
    a = int(rng.integers(0, 150_000)); b = a + int(rng.integers(1, 50_000))
    mode = it % 6
    if mode == 0:
        got = lib.score_call(pool[a:b].reshape(-1), None, n=b - a, fixed_len=150); want = ref("se", a, b)
    elif mode == 1:
        o = synth.fixed_offsets(b - a, 150)
        got = lib.score_call(p1[a:b].reshape(-1), o, p2[a:b].reshape(-1), o); want = ref("pe", a, b)
    elif mode == 2:
        c = a + (b - a) // 2
        f1, f2 = np.ascontiguousarray(pool[a:c].reshape(-1)), np.ascontiguousarray(pool[c:b].reshape(-1))
        lib.score_call_begin(0, f1, None, n=c - a, fixed_len=150)
        lib.score_call_begin(1, f2, None, n=b - c, fixed_len=150)
        g1, g2 = lib.score_call_end(0), lib.score_call_end(1)
        got, want = (g1, g2), (ref("se", a, c) if c > a else [], ref("se", c, b))
        if c == a: got = ([], g2)
    elif mode == 3:
        lib.stream_begin(False, 150, capacity_hint=int(rng.integers(1, 100_000)))
        x = a
        while x < b:
            y = min(b, x + int(rng.integers(1, 20_000)))
            lib.stream_append(np.ascontiguousarray(pool[x:y].reshape(-1)), None, n=y - x, fixed_len=150); x = y
        got = lib.stream_end(); want = ref("se", a, b)
    elif mode == 4:
        rows, _ = lib.score_call_umis(pool[a:b].reshape(-1), None, n=b - a, fixed_len=150)
        got = [(f, c) for _, f, c, _ in rows]; want = [(f, c) for f, c in ref("se", a, b)]
    else:
        d = torch.from_numpy(pool[a:b].copy()).to("cuda:0"); torch.cuda.synchronize()
        pt = lib.pack(d, None, None, None, n=b - a, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE)
        lib.device_context().synchronize()
        got = lib.score_call_packed(pt); want = ref("se", a, b)
    assert got == want, (it, mode, a, b)
    if it == 200:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
    if it % 250 == 0:
        print("iter", it, "free GiB %.2f" % (torch.cuda.mem_get_info()[0] / 2**30), "%.1fs" % (time.time() - t0), flush=True)
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("done", N, "calls; device memory drift after warm-up: %.1f MiB" % ((free0 - free1) / 2**20))
assert free0 - free1 < 512 * 2**20
print("OK")
