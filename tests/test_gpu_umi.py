"""GPU parity of the BAM pipeline's form of the call: one score::call per UMI group
(src/process/bam.rs:183-226,229-290), reads trimmed for quality before alignment (src/align.rs:866-942), SKIP_ALIGN
dummies (src/align.rs:527-528,549-550) -- a whole batch of groups in ONE device call, checked against the CPU
oracle running one call per group."""
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
def world():
    names, seqs = synth.make_library(200)
    obj = synth.library_json(names, seqs)
    obj[0].update(score_percent=0.08, score_threshold=12, num_mismatches=1, trim_target_length=40, trim_strictness=0.9)
    lib = nim.Library(text=json.dumps(obj), strand_filter="unstranded").build_index()
    cols = [["s"] * len(names), names, [str(len(s)) for s in seqs], seqs]
    ref = ora.Reference.from_columns(HEADERS, cols, "")
    cfg = ora.config_from_json(obj[0], len(names), "unstranded")
    return lib, ref, cfg, ora.Index.from_reference(ref), seqs


def make_batch(seqs, n, seed, contiguous=True):
    rng = np.random.default_rng(seed)
    r1, r2 = synth.make_reads(seqs, n, paired=True, seed=seed)
    r1, r2 = r1.copy(), r2.copy()
    L = r1.shape[1]
    # qualities: high, with a degraded 3' tail of random length on most reads, and some noisy reads
    def quals():
        q = np.full((n, L), ord("I"), dtype=np.uint8)
        tail = rng.integers(0, 90, size=n)
        pos = np.arange(L)[None, :]
        bad = pos >= (L - tail)[:, None]
        q[bad] = rng.integers(0, 12, size=int(bad.sum()), dtype=np.uint8)  # raw byte values: maxinfo does not subtract 33
        noisy = rng.random(n) < 0.1
        q[noisy] = rng.integers(0, 75, size=(int(noisy.sum()), L), dtype=np.uint8)
        return q
    q1, q2 = quals(), quals()
    # UMI groups of 1..30 pairs
    seg = np.zeros(n, dtype=np.uint32)
    i = g = 0
    while i < n:
        k = int(rng.integers(1, 31))
        seg[i:i + k] = g
        i += k
        g += 1
    if not contiguous:
        seg = rng.permutation(seg)              # group members scattered over the batch
        seg = (seg * 7 + 3).astype(np.uint32)   # sparse ids
    # duplicates: same pair again in the same group (dedup), in another group (counted twice), and with other
    # qualities in the same group (different trim: the last one wins)
    src = rng.integers(0, n, size=n // 10)
    dst = rng.integers(0, n, size=n // 10)
    kind = rng.integers(0, 3, size=n // 10)
    for s, d, k in zip(src, dst, kind):
        if s == d:
            continue
        r1[d], r2[d] = r1[s], r2[s]
        if k == 0:
            seg[d] = seg[s]
            q1[d], q2[d] = q1[s], q2[s]
        elif k == 2:
            seg[d] = seg[s]
    skip2 = (rng.random(n) < 0.03).astype(np.uint8)   # unpaired reads get a dummy mate
    skip1 = (rng.random(n) < 0.005).astype(np.uint8)
    return r1, r2, q1, q2, seg, skip1, skip2


@pytest.mark.parametrize("contiguous", [True, False])
def test_umi_batch_matches_one_oracle_call_per_group(world, contiguous):
    lib, ref, cfg, oidx, seqs = world
    n = 30_000
    r1, r2, q1, q2, seg, skip1, skip2 = make_batch(seqs, n, 900 + int(contiguous), contiguous)
    o = synth.fixed_offsets(n, 150)
    exp = ora.call_umi(oidx, ref, cfg, r1.reshape(-1), o, r2.reshape(-1), o, q1=q1.reshape(-1), q2=q2.reshape(-1),
                       skip1=skip1, skip2=skip2, segment=seg, keep_per_read=True)
    rows, filt = lib.score_call_umis(r1.reshape(-1), o, r2.reshape(-1), o, segment=seg,
                                     qual=(q1.reshape(-1), q2.reshape(-1)), skip=(skip1, skip2), per_read=True)
    assert [(s, f, c) for s, f, c, _ in rows] == [(s, f, c) for s, f, c in exp.rows]
    assert len(rows) > 1000
    for s, _, _, rep in rows:
        assert seg[rep] == s                      # the representative is a read of that group
    ctx = lib.device_context()
    ctx.n = n
    for m in (0, 1):
        rec = ctx.read_records(m)
        np.testing.assert_array_equal(rec["reason"], exp.per_read["reason"][m])
        np.testing.assert_array_equal(rec["score"], exp.per_read["score"][m])
        np.testing.assert_array_equal(rec["mismatches"], exp.per_read["mismatches"][m])
        # aligned length = maxinfo(quality) (skipped dummies are never trimmed by the reference)
        al = ctx.read_align_len(m)
        sk = (skip1, skip2)[m].astype(bool)
        np.testing.assert_array_equal(al[~sk], exp.per_read["align_len"][m][~sk])
    np.testing.assert_array_equal(ctx.read_records(0)["counted"], exp.per_read["counted"])
    # filter_reasons entries: reasons as above, scores only for kept alignments
    for m, (rc, sc) in enumerate(((0, 1), (2, 3))):
        np.testing.assert_array_equal(filt[:, rc], exp.per_read["reason"][m])
        kept = exp.per_read["reason"][m] == 11
        np.testing.assert_array_equal(filt[:, sc], np.where(kept, exp.per_read["score"][m], 0))
    assert int((exp.per_read["reason"][1] == 15).sum()) == int(skip2.sum())
    assert int((exp.per_read["align_len"][0] < 150).sum()) > n // 4   # the trim really bites


def test_segments_only_single_end_fixed_length(world):
    # segments without trimming on single-end reads: the fused dedup+count path with the segment in the key
    lib, ref, cfg, oidx, seqs = world
    n = 20_000
    reads = synth.make_reads(seqs, n, seed=77)
    rng = np.random.default_rng(5)
    seg = rng.integers(0, 900, size=n).astype(np.uint32)
    reads[n // 2:] = reads[: n - n // 2]          # every read twice; the copies fall into random groups
    o = synth.fixed_offsets(n, 150)
    exp = ora.call_umi(oidx, ref, cfg, reads.reshape(-1), o, segment=seg)
    rows, _ = lib.score_call_umis(reads.reshape(-1), None, n=n, fixed_len=150, segment=seg)
    assert [(s, f, c) for s, f, c, _ in rows] == [(s, f, c) for s, f, c in exp.rows]
    # one group == the plain call
    plain = lib.score_call(reads.reshape(-1), None, n=n, fixed_len=150)
    rows0, _ = lib.score_call_umis(reads.reshape(-1), None, n=n, fixed_len=150)
    assert [(f, c) for _, f, c, _ in rows0] == [(f, c) for f, c in plain]


def test_maxinfo_kernel_matches_reference_literals_and_oracle(world):
    # src/align.rs:1656-1752 pins maxinfo on literal quality strings; the device kernel must agree with the oracle
    # restatement on those and on random strings (raw byte values, clamped at 60, not Phred)
    lib, ref, cfg, oidx, seqs = world
    rng = np.random.default_rng(11)
    quals = [b"I" * 150, b"#" * 150, b"I" * 75 + b"#" * 75, b"!" * 10, b"I" * 40, b"5" * 39, b"~" * 149 + b"\x00",
             bytes(rng.integers(0, 256, size=150, dtype=np.uint8)) , b"I"]
    quals += [bytes(rng.integers(33, 75, size=int(rng.integers(1, 151)), dtype=np.uint8)) for _ in range(3000)]
    reads = [bytes(rng.choice(list(b"ACGT"), size=len(q)).astype(np.uint8)) for q in quals]
    flat, off = nim.pack_reads(reads)
    qflat, _ = nim.pack_reads(quals)
    ctx = lib.device_context()
    for target, strict in ((40, 0.9), (15, 0.5), (100, 0.1), (0, 1.0)):
        p = lib.align_params()
        ctx.call_ex(p, flat, off, qual=(qflat, None), trim_strictness=strict, trim_target_length=target)
        got = ctx.read_align_len(0)
        want = np.array([ora.maxinfo(q, target, strict) for q in quals])
        np.testing.assert_array_equal(got, np.minimum(want, [len(q) for q in quals]))
