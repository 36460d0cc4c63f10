#!/usr/bin/env python3
"""Device-stage times of the other BASELINE.json configurations (documentation aid; bench.py is the contract).
   configs[1] 1 M SE x 500 features, configs[2] 10 M SE x 1 k, configs[3] PE 2x150 with the mismatch.json settings,
   configs[4] 5 k features."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")

def run(tag, T, n, paired, over):
    names, seqs = synth.make_library(T)
    obj = synth.library_json(names, seqs)
    obj[0].update(over)
    t0 = time.time()
    lib = nim.Library(text=json.dumps(obj), strand_filter="unstranded").build_index(0)
    build = time.time() - t0
    if paired:
        r1, r2 = synth.make_reads(seqs, n, paired=True, seed=5)
        d1, d2 = torch.from_numpy(r1).to("cuda:0"), torch.from_numpy(r2).to("cuda:0")
    else:
        d1, d2 = synth.make_reads_torch(seqs, n, device="cuda:0"), None
    torch.cuda.synchronize()
    ctx = lib.device_context()
    ctx.set_counters(False)
    best = None
    for rep in range(4):
        t0 = time.perf_counter()
        rows = lib.score_call_raw(d1, None, d2, None, n=n, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE)
        wall = time.perf_counter() - t0
        t = ctx.timing()
        if best is None or t["total"] < best[0]["total"]:
            best = (t, wall)
    t, wall = best
    # two calls in flight (what bench.py times): ms per complete call
    for s_ in (0, 1):
        lib.device_context(s_).set_counters(False)
    def begin(slot):
        lib.score_call_begin(slot, d1, None, d2, None, n=n, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE)
    begin(0); begin(1); lib.score_call_end(0, raw=True); lib.score_call_end(1, raw=True)
    K = 20
    t0 = time.perf_counter()
    for i in range(K):
        begin(i % 2)
        if i:
            lib.score_call_end((i - 1) % 2, raw=True)
    lib.score_call_end((K - 1) % 2, raw=True)
    depth2_ms = (time.perf_counter() - t0) * 1e3 / K
    print(json.dumps(dict(config=tag, features=T, reads=n, paired=paired, rows=len(rows), index_build_s=round(build, 2),
                          stage_ms={k: round(v, 3) for k, v in t.items()}, wall_ms=round(wall * 1e3, 3),
                          depth2_ms_per_call=round(depth2_ms, 3),
                          device_reads_per_s=round(n / (t["total"] / 1e3)))), flush=True)

mm = dict(score_percent=0.08, score_threshold=12, num_mismatches=2)
run("configs[1]", 500, 1_000_000, False, {})
run("configs[2]", 1000, 10_000_000, False, {})
run("configs[3] PE, mismatch settings", 1000, 5_000_000, True, mm)
run("configs[4] 5k features", 5000, 10_000_000, False, {})
