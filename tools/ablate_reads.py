"""Ablation: align-kernel time by read class (development aid)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
T, N = 1000, 4_000_000
names, seqs = synth.make_library(T)
rn, rs = synth.expand_rows(names, seqs)
idx = nim.Index(rs); ctx = nim.Context(idx); ctx.set_counters(int(os.environ.get("CNT", "0")))
rng = np.random.default_rng(1)
cat, off = synth._codes(seqs); lens = np.diff(off)
def on_target(n, err):
    f = rng.integers(0, T, size=n); start = (rng.random(n) * (lens[f] - 150 + 1)).astype(np.int64)
    codes = cat[(off[f] + start)[:, None] + np.arange(150)]
    if err > 0:
        m = rng.random((n, 150)) < err
        codes = np.where(m, (codes + rng.integers(1, 4, size=(n, 150), dtype=np.uint8)) % 4, codes).astype(np.uint8)
    return synth.ACGT[codes]
sets = {
  "random(off-target)": synth.ACGT[rng.integers(0, 4, size=(N, 150), dtype=np.uint8)],
  "on-target exact": on_target(N, 0.0),
  "on-target 0.5% err": on_target(N, 0.005),
  "standard mix": synth.make_reads(seqs, N),
}
p = nim.AlignParams.make(0.33, 50, 0)
for name, r in sets.items():
    flat = np.ascontiguousarray(r).reshape(-1)
    for rep in range(2):
        ctx.call(p, flat, None, n=N, fixed_len=150, max_len=150); ctx.synchronize()
    t = ctx.timing()
    print("%-22s" % name, {k: round(v, 3) for k, v in t.items()}, "align ns/read %.2f" % (t["align"] * 1e6 / N), flush=True)
