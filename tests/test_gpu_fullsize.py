"""GPU parity at BASELINE.json's full sizes, through size-independent properties, plus oracle comparisons at the
largest sizes the CPU oracle finishes in seconds.

Properties checked at 10 M reads (configs[2]):
  * idempotence under duplication: call(reads ++ reads) == call(reads)   (dedup by read key, align.rs:576-579,685)
  * permutation invariance: a shuffled read order gives the same table
  * conservation: sum of histogram counts == number of counted representatives == unique kept keys
  * determinism: two calls give identical tables
  * decomposition: per-feature-family disjoint read sets add up (reads of disjoint key sets -> tables add)
"""
import importlib
import json

import numpy as np
import pytest

from oracle import oracle as ora

pytestmark = pytest.mark.gpu

nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
HEADERS = ["reference_genome", "sequence_name", "nt_length", "sequence"]


@pytest.fixture(scope="module")
def lib1000():
    names, seqs = synth.make_library(1000)
    lib = nim.Library(text=json.dumps(synth.library_json(names, seqs)), strand_filter="unstranded").build_index()
    return lib, names, seqs


def table(lib, t, n):
    return lib.score_call(t, None, n=n, fixed_len=150, mem=nim.MEM_DEVICE)


def test_config2_properties_at_10m_reads(lib1000):
    torch = pytest.importorskip("torch")
    lib, names, seqs = lib1000
    n = 10_000_000
    reads = synth.make_reads_torch(seqs, n, device="cuda:0")
    torch.cuda.synchronize()
    base = table(lib, reads, n)
    assert len(base) > 1500
    # determinism
    assert table(lib, reads, n) == base
    # conservation: histogram total == representatives == unique kept keys
    ctx = lib.device_context()
    ctx.n = n
    hist = ctx.histogram()
    rec = ctx.read_records(0)
    total = sum(c for _, _, c in hist)
    assert total == int(rec["counted"].sum()) == ctx.counters()["unique_keys"]
    kept = rec["reason"] == 11
    assert int(rec["counted"].sum()) <= int(kept.sum())
    # every row count is positive and rows are sorted by callset
    assert all(c > 0 for _, c in base) and [f for f, _ in base] == sorted(f for f, _ in base)
    # permutation invariance
    perm = torch.randperm(n, device="cuda:0")
    shuffled = reads[perm].contiguous()
    torch.cuda.synchronize()
    assert table(lib, shuffled, n) == base
    del shuffled, perm
    # idempotence under duplication (20 M reads in one call)
    doubled = torch.cat([reads, reads], dim=0).contiguous()
    torch.cuda.synchronize()
    assert table(lib, doubled, 2 * n) == base
    del doubled
    # decomposition: two halves with disjoint key sets add up (split by a key-content hash)
    # (the key is the converted base string: N reads as A, so the split must hash the converted bases)
    conv = torch.where(reads == ord("N"), torch.full_like(reads, ord("A")), reads)
    h = (conv.to(torch.int64) * torch.arange(1, 151, device="cuda:0")).sum(dim=1)
    del conv
    sel = (h % 2) == 0
    a, b = reads[sel].contiguous(), reads[~sel].contiguous()
    torch.cuda.synchronize()
    ta, tb = table(lib, a, a.shape[0]), table(lib, b, b.shape[0])
    merged = {}
    for f, c in ta + tb:
        merged[tuple(f)] = merged.get(tuple(f), 0) + c
    assert sorted([list(k), v] for k, v in merged.items()) == [[f, c] for f, c in base]


def test_config1_one_million_reads_vs_oracle():
    # BASELINE.json configs[1]: 1 M x 150 bp single-end vs a 500-feature library, table bit-exact vs the CPU path
    names, seqs = synth.make_library(500)
    obj = synth.library_json(names, seqs)
    lib = nim.Library(text=json.dumps(obj), strand_filter="unstranded").build_index()
    reads = synth.make_reads(seqs, 1_000_000, seed=4242)
    got = lib.score_call(reads.reshape(-1), None, n=reads.shape[0], fixed_len=150)
    cols = [["s"] * len(names), names, [str(len(s)) for s in seqs], seqs]
    ref = ora.Reference.from_columns(HEADERS, cols, "")
    cfg = ora.config_from_json(obj[0], len(names), "unstranded")
    exp = ora.call(ora.Index.from_reference(ref), ref, cfg, reads.reshape(-1), synth.fixed_offsets(reads.shape[0], 150),
                   n_threads=8, keep_per_read=True)
    assert [(f, c) for f, c in got] == [(f, c) for f, c in exp.rows]
    ctx = lib.device_context()
    ctx.n = reads.shape[0]
    rec = ctx.read_records(0)
    np.testing.assert_array_equal(rec["reason"], exp.per_read["reason"][0])
    np.testing.assert_array_equal(rec["score"], exp.per_read["score"][0])
    np.testing.assert_array_equal(rec["mismatches"], exp.per_read["mismatches"][0])


@pytest.mark.parametrize("nm", [0, 2])
def test_config3_paired_end_vs_oracle(lib1000, nm):
    # BASELINE.json configs[3] shape (2 x 150 bp, mismatch.rs tolerance settings) at 400 k pairs
    lib, names, seqs = lib1000
    r1, r2 = synth.make_reads(seqs, 400_000, paired=True, seed=31 + nm)
    over = dict(num_mismatches=nm, score_percent=0.08, score_threshold=12)
    lib.update_config(**over)
    o = synth.fixed_offsets(r1.shape[0], 150)
    try:
        got = lib.score_call(r1.reshape(-1), o, r2.reshape(-1), o)
    finally:
        lib.update_config(num_mismatches=0, score_percent=0.33, score_threshold=50)
    cols = [["s"] * len(names), names, [str(len(s)) for s in seqs], seqs]
    ref = ora.Reference.from_columns(HEADERS, cols, "")
    cfg = ora.config_from_json(synth.library_json(names, seqs)[0], len(names), "unstranded").copy(**over)
    exp = ora.call(ora.Index.from_reference(ref), ref, cfg, r1.reshape(-1), o, r2.reshape(-1), o, n_threads=8)
    assert [(f, c) for f, c in got] == [(f, c) for f, c in exp.rows]


def test_config3_at_full_size_properties_and_oracle_sample(lib1000):
    """BASELINE.json configs[3] at its stated size.  "10M paired-end 2x150bp reads" is read as 10 M PAIRS of 2 x 150 bp
    ("10 M x 2x150 bp PE"), 20 M reads in one call; the settings are mismatch.json's (score_percent
    0.08, score_threshold 12) with num_mismatches 0, 1 and 2 as tests/mismatch.rs:45 and tests/basic-cases.rs:83,119 run
    them.  At full size: determinism, conservation, permutation invariance, idempotence under duplication (20 M pairs in
    one call); against the CPU oracle: a 1 M-pair call at num_mismatches 2 and 250 k-pair calls at 0 and 1, tables
    bit-exact."""
    torch = pytest.importorskip("torch")
    lib, names, seqs = lib1000
    n = 10_000_000
    r1, r2 = synth.make_pairs_torch(seqs, n, seed=synth.READ_SEED + 33, device="cuda:0")
    torch.cuda.synchronize()
    a, b = r1[:1_000_000].cpu().numpy(), r2[:1_000_000].cpu().numpy()
    cols = [["s"] * len(names), names, [str(len(s)) for s in seqs], seqs]
    ref = ora.Reference.from_columns(HEADERS, cols, "")
    oidx = ora.Index.from_reference(ref)

    def call(x1, x2, m):
        return lib.score_call(x1, None, x2, None, n=m, fixed_len=150, mem=nim.MEM_DEVICE)

    try:
        for nm, sample in ((2, 1_000_000), (0, 250_000), (1, 250_000)):
            over = dict(num_mismatches=nm, score_percent=0.08, score_threshold=12)
            lib.update_config(**over)
            base = call(r1, r2, n)
            assert len(base) > 1000 and all(c > 0 for _, c in base)
            assert [f for f, _ in base] == sorted(f for f, _ in base)
            ctx = lib.device_context()
            ctx.n = n
            total = sum(c for _, _, c in ctx.histogram())
            rec = ctx.read_records(0)
            assert total == int(rec["counted"].sum()) == ctx.counters()["unique_keys"]
            assert call(r1, r2, n) == base                                   # determinism
            cfg = ora.config_from_json(synth.library_json(names, seqs)[0], len(names), "unstranded").copy(**over)
            o = synth.fixed_offsets(sample, 150)
            exp = ora.call(oidx, ref, cfg, a[:sample].reshape(-1), o, b[:sample].reshape(-1), o, n_threads=16)
            got = call(r1[:sample].contiguous(), r2[:sample].contiguous(), sample)
            assert [(f, c) for f, c in got] == [(f, c) for f, c in exp.rows]
            if nm == 2:
                perm = torch.randperm(n, device="cuda:0")
                p1, p2 = r1[perm].contiguous(), r2[perm].contiguous()
                torch.cuda.synchronize()
                assert call(p1, p2, n) == base                               # permutation invariance
                del p1, p2, perm
                d1, d2 = torch.cat([r1, r1], dim=0).contiguous(), torch.cat([r2, r2], dim=0).contiguous()
                torch.cuda.synchronize()
                assert call(d1, d2, 2 * n) == base                           # idempotence: 20 M pairs in one call
                del d1, d2
    finally:
        lib.update_config(num_mismatches=0, score_percent=0.33, score_threshold=50)


def test_config4_five_thousand_feature_index():
    # BASELINE.json configs[4] library size (5 k features, 10 k index rows) on one GPU at 300 k reads
    names, seqs = synth.make_library(5000)
    obj = synth.library_json(names, seqs)
    lib = nim.Library(text=json.dumps(obj), strand_filter="unstranded").build_index()
    reads = synth.make_reads(seqs, 300_000, seed=99)
    got = lib.score_call(reads.reshape(-1), None, n=reads.shape[0], fixed_len=150)
    cols = [["s"] * len(names), names, [str(len(s)) for s in seqs], seqs]
    ref = ora.Reference.from_columns(HEADERS, cols, "")
    cfg = ora.config_from_json(obj[0], len(names), "unstranded")
    oidx = ora.Index.from_reference(ref)
    exp = ora.call(oidx, ref, cfg, reads.reshape(-1), synth.fixed_offsets(reads.shape[0], 150), n_threads=8)
    assert [(f, c) for f, c in got] == [(f, c) for f, c in exp.rows]
    st = nim.Context(borrowed=nim.host_lib().nimble_library_ctx(lib.h))  # context exists
    assert st.h


def test_config4_properties_at_80m_reads_on_one_gpu():
    """BASELINE.json configs[4] at its stated size -- 80 M x 150 bp reads against the 5 k-feature library -- as ONE call on
    one GPU (12 GB of bases, ~11 GB of workspace of the 288 GB): determinism, conservation, permutation invariance, the
    8-way decomposition by read key that the 8-GPU run makes (each shard = the keys one rank would own; the shards'
    tables must add up to the whole), and the CPU oracle on a 1 M-read subsample."""
    torch = pytest.importorskip("torch")
    names, seqs = synth.make_library(5000)
    obj = synth.library_json(names, seqs)
    lib = nim.Library(text=json.dumps(obj), strand_filter="unstranded").build_index()
    n = 80_000_000
    reads = synth.make_reads_torch(seqs, n, device="cuda:0")
    torch.cuda.synchronize()
    base = table(lib, reads, n)
    assert len(base) > 8000
    ctx = lib.device_context()
    ctx.n = n
    hist_total = sum(c for _, _, c in ctx.histogram())
    assert hist_total == ctx.counters()["unique_keys"] == sum(c for _, c in base)   # group_on is empty: nothing is triaged
    assert all(c > 0 for _, c in base) and [f for f, _ in base] == sorted(f for f, _ in base)
    assert table(lib, reads, n) == base                                              # determinism
    perm = torch.randperm(n, device="cuda:0")
    shuffled = reads[perm].contiguous()
    del perm
    torch.cuda.synchronize()
    assert table(lib, shuffled, n) == base                                           # permutation invariance
    del shuffled
    # the decomposition of the multi-GPU run: shard r = the reads whose KEY (converted bases: N reads as A) hashes to r
    merged = {}
    h = torch.zeros(n, dtype=torch.int64, device="cuda:0")
    w = torch.arange(1, 151, device="cuda:0")
    for lo in range(0, n, 1 << 22):
        part = reads[lo:lo + (1 << 22)]
        conv = torch.where(part == ord("N"), torch.full_like(part, ord("A")), part)
        h[lo:lo + (1 << 22)] = (conv.to(torch.int64) * w).sum(dim=1)
    shard_reads = 0
    for r in range(8):
        part = reads[(h % 8) == r].contiguous()
        torch.cuda.synchronize()
        shard_reads += part.shape[0]
        for f, c in table(lib, part, part.shape[0]):
            merged[tuple(f)] = merged.get(tuple(f), 0) + c
        del part
    assert shard_reads == n
    assert sorted([list(k), v] for k, v in merged.items()) == [[f, c] for f, c in base]
    del h
    # oracle parity on a subsample
    m = 1_000_000
    sub = reads[:m].contiguous()
    torch.cuda.synchronize()
    got = table(lib, sub, m)
    cols = [["s"] * len(names), names, [str(len(s)) for s in seqs], seqs]
    ref = ora.Reference.from_columns(HEADERS, cols, "")
    cfg = ora.config_from_json(obj[0], len(names), "unstranded")
    exp = ora.call(ora.Index.from_reference(ref), ref, cfg, sub.cpu().numpy().reshape(-1), synth.fixed_offsets(m, 150),
                   n_threads=16)
    assert [(f, c) for f, c in got] == [(f, c) for f, c in exp.rows]
