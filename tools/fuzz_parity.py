#!/usr/bin/env python3
"""Randomised differential test, GPU product vs CPU oracle (development aid): tools/fuzz_parity.py [iterations] [seed]
Random small libraries (mutated copies, shared segments, repeats, homopolymers), random alignment settings and
grouping, random reads (errors, junk, N, lower case, ragged lengths, single-end and paired): the final table and the
per-read (reason, score, mismatches) must be identical.  Every third plain case also goes through the multi-GPU forms
on one GPU -- pack / route / calls off the records of every destination, and (single-end) the deferred-dedup form with
three ranks played by the three call slots -- whose merged tables must equal the direct call's."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()   # before the library touches HIP
nim = importlib.import_module("nimble-aligner_amd")
from oracle import oracle as ora

ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
HEADERS = ["reference_genome", "sequence_name", "nt_length", "sequence", "family"]


def rnd(rng, k):
    return ACGT[rng.integers(0, 4, size=k)].tobytes().decode()


def mutate(rng, s, rate):
    a = np.frombuffer(s.encode(), dtype=np.uint8).copy()
    m = rng.random(len(a)) < rate
    a[m] = ACGT[rng.integers(0, 4, size=int(m.sum()))]
    return a.tobytes().decode()


def make_library(rng):
    seqs = []
    n_roots = int(rng.integers(1, 8))
    shared = rnd(rng, int(rng.integers(31, 120)))
    for _ in range(n_roots):
        root = rnd(rng, int(rng.integers(30, 500)))
        kind = rng.integers(0, 6)
        if kind == 0:
            root = root[: len(root) // 2] + shared + root[len(root) // 2:]
        elif kind == 1:
            u = rnd(rng, int(rng.integers(2, 40)))
            root += u * int(rng.integers(2, 12))
        elif kind == 2:
            root += "ACGT"[int(rng.integers(0, 4))] * int(rng.integers(31, 90))
        seqs.append(root)
        for _ in range(int(rng.integers(0, 5))):
            seqs.append(mutate(rng, root, float(rng.choice([0.0, 0.003, 0.01, 0.05]))))
    if rng.random() < 0.3:
        seqs.append(rnd(rng, int(rng.integers(1, 30))))     # shorter than k
    if rng.random() < 0.3:
        seqs.append(seqs[0])                                 # duplicate feature
    names = ["F%03d" % i for i in range(len(seqs))]
    fam = ["G%d" % (i // int(rng.integers(1, 4) + 1)) if rng.random() < 0.8 else "" for i in range(len(seqs))]
    return names, seqs, fam


def make_reads(rng, seqs, n, lo, hi):
    out = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        k = rng.random()
        s = seqs[int(rng.integers(0, len(seqs)))]
        if k < 0.7 and len(s) >= 1:
            st = int(rng.integers(0, max(1, len(s) - L + 1)))
            r = np.frombuffer(s[st:st + L].encode(), dtype=np.uint8).copy()
            if rng.random() < 0.5:
                r = np.frombuffer(r.tobytes().decode()[::-1].translate(str.maketrans("ACGT", "TGCA")).encode(), dtype=np.uint8).copy()
            if len(r) and rng.random() < 0.5:
                for _ in range(int(rng.integers(1, 5))):
                    r[int(rng.integers(0, len(r)))] = ACGT[int(rng.integers(0, 4))]
            if len(r) and rng.random() < 0.1:
                r[int(rng.integers(0, len(r)))] = ord("N")
            if rng.random() < 0.1:
                r = np.frombuffer(r.tobytes().lower(), dtype=np.uint8).copy()
            if rng.random() < 0.15 and len(r) < L:
                r = np.concatenate([r, ACGT[rng.integers(0, 4, size=L - len(r))]])
            out.append(r.tobytes())
        elif k < 0.9:
            out.append(rnd(rng, L).encode())
        else:
            out.append(("ACGT"[int(rng.integers(0, 4))] * L).encode())
    if n > 4:   # exact duplicates
        for _ in range(n // 10):
            out[int(rng.integers(0, n))] = out[int(rng.integers(0, n))]
    return out


def merged_rows(parts):
    acc = {}
    for rows in parts:
        for f, c in rows:
            acc[tuple(f)] = acc.get(tuple(f), 0) + c
    return sorted([list(k), v] for k, v in acc.items())


def split_forms(lib, b1, o1, b2, o2, n, paired, max_len, rng):
    """The same reads through the multi-GPU forms; returns {form name: merged table}."""
    import torch
    out = {}
    world = int(rng.integers(1, 6))
    ctx = lib.device_context(2)
    pt = lib.pack(b1, o1, b2, o2, n=n, max_len=max_len, slot=2)
    rec, counts = pt.route(ctx, world)
    parts, lo = [], 0
    for d in range(world):
        part = rec[lo:lo + counts[d]].contiguous()
        lo += counts[d]
        lib.score_call_records_begin(0, part, pt.max_len, pt.paired)
        parts.append(lib.score_call_end(0))
    out["records, world %d" % world] = merged_rows(parts)
    if not paired:
        slots, kw = (0, 1, 3), pt.key_words
        cuts = np.sort(rng.integers(0, n + 1, size=2))
        bounds = [0, int(cuts[0]), int(cuts[1]), n]
        rank_rec, rank_perm, rank_counts, keep = [], [], [], []
        for r in range(3):
            s, e = bounds[r], bounds[r + 1]
            bb = np.ascontiguousarray(b1[int(o1[s]):int(o1[e])])
            oo = (o1[s:e + 1] - o1[s]).astype(np.uint64)
            if bb.size == 0:
                bb = np.zeros(1, dtype=np.uint8)
            keep.append((bb, oo))
            m = e - s
            rank_rec.append(torch.empty((max(m, 1), kw + 2), dtype=torch.int64, device="cuda:0"))
            rank_perm.append(torch.empty((max(m, 1),), dtype=torch.int32, device="cuda:0"))
            c = lib.device_context(slots[r])
            c.defer_dedup(3, rank_rec[r], rank_perm[r])
            lib.score_call_begin(slots[r], bb, oo, None, None, n=m, max_len=max_len)
            rank_counts.append(c.route_counts(3))
        starts = [np.concatenate([[0], np.cumsum(c)]).astype(np.int64) for c in rank_counts]
        back = [[None] * 3 for _ in range(3)]
        for owner in range(3):
            pieces = [rank_rec[r][starts[r][owner]:starts[r][owner + 1]] for r in range(3)]
            got = torch.cat(pieces).contiguous()
            verdict = torch.zeros((max(int(got.shape[0]), 1),), dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            ctx.dedup_records(got, kw, verdict)
            ctx.synchronize()
            lo = 0
            for r in range(3):
                m = int(pieces[r].shape[0])
                back[r][owner] = verdict[lo:lo + m].clone()
                lo += m
        parts = []
        for r in range(3):
            mine = torch.cat(back[r] + [torch.zeros(1, dtype=torch.uint8, device="cuda:0")]).contiguous()
            torch.cuda.synchronize()
            lib.device_context(slots[r]).count_verdicts(mine)
            parts.append(lib.score_call_end(slots[r]))
        out["deferred dedup, 3 ranks"] = merged_rows(parts)
    return out


t0 = time.time()
for it in range(ITER):
    rng = np.random.default_rng(SEED * 100003 + it)
    names, seqs, fam = make_library(rng)
    cfg_obj = dict(score_percent=float(rng.choice([0.0, 0.05, 0.2, 0.33, 0.8])), score_filter=25,
                   score_threshold=int(rng.choice([0, 12, 30, 50, 100])), num_mismatches=int(rng.integers(0, 4)),
                   discard_multiple_matches=bool(rng.random() < 0.2), require_valid_pair=bool(rng.random() < 0.3),
                   discard_multi_hits=int(rng.choice([0, 0, 2, 3])), intersect_level=int(rng.integers(0, 3)),
                   max_hits_to_report=int(rng.choice([1, 3, 10, 50])), group_on=str(rng.choice(["", "family"])),
                   trim_target_length=40, trim_strictness=0.9)
    if rng.random() < 0.2:
        cfg_obj["discard_nonzero_mismatch"] = True
    strand = str(rng.choice(["unstranded", "fiveprime", "threeprime", "none"]))
    obj = [cfg_obj, {"headers": HEADERS, "columns": [["g"] * len(names), names, [str(len(s)) for s in seqs], seqs, fam]}]
    text = json.dumps(obj)
    try:
        lib = nim.Library(text=text, strand_filter=strand).build_index(0)
    except nim.Panic as e:
        # the reference panics on the same input (e.g. group_on column rules): the oracle must agree
        try:
            ora.Reference.from_columns(HEADERS, obj[1]["columns"], cfg_obj["group_on"])
            raise SystemExit("iteration %d: product panicked (%s) but the oracle accepted the library" % (it, e))
        except ora.OracleError:
            continue
    ref = ora.Reference.from_columns(HEADERS, obj[1]["columns"], cfg_obj["group_on"])
    cfg = ora.config_from_json(cfg_obj, len(names), strand)
    oidx = ora.Index.from_reference(ref)
    paired = rng.random() < 0.5
    n = int(rng.integers(1, 3000))
    lo, hi = (0, 200) if rng.random() < 0.5 else (100, 151)
    r1 = make_reads(rng, seqs, n, lo, hi)
    b1, o1 = nim.pack_reads(r1)
    b2 = o2 = None
    if paired:
        r2 = make_reads(rng, seqs, n, lo, hi)
        if rng.random() < 0.5:   # proper mates: reverse complement of the same template
            r2 = [x.decode().upper().replace("N", "A")[::-1].translate(str.maketrans("ACGT", "TGCA")).encode() for x in r1]
        b2, o2 = nim.pack_reads(r2)
    if rng.random() < 0.3:
        # the BAM pipeline's form of the call: UMI groups, quality trim, SKIP_ALIGN dummies
        seg = rng.integers(0, max(1, n // int(rng.integers(1, 20))) + 1, size=n).astype(np.uint32)
        if rng.random() < 0.5:
            seg = np.sort(seg)
        def quals(off):
            q = rng.integers(0, 90, size=int(off[-1]), dtype=np.uint8)
            if rng.random() < 0.5:
                q[:] = 73
                cut = rng.integers(0, len(q) + 1, size=max(1, len(q) // 200))
                for c in cut:
                    q[c:c + 40] = rng.integers(0, 10, size=len(q[c:c + 40]), dtype=np.uint8)
            return q
        use_q = rng.random() < 0.7
        q1 = quals(o1) if use_q else None
        q2 = quals(o2) if (use_q and paired) else None
        sk1 = (rng.random(n) < 0.05).astype(np.uint8) if rng.random() < 0.5 else None
        sk2 = (rng.random(n) < 0.1).astype(np.uint8) if (paired and rng.random() < 0.5) else None
        expu = ora.call_umi(oidx, ref, cfg, b1, o1, b2, o2, q1=q1, q2=q2, skip1=sk1, skip2=sk2, segment=seg,
                            keep_per_read=True)
        rows, _ = lib.score_call_umis(b1, o1, b2, o2, segment=seg, qual=(q1, q2), skip=(sk1, sk2))
        if [(sg, f, c) for sg, f, c, _ in rows] != [(sg, f, c) for sg, f, c in expu.rows]:
            raise SystemExit("iteration %d: UMI TABLE MISMATCH\n got %s\n exp %s" % (it, rows[:5], expu.rows[:5]))
        ctx = lib.device_context()
        ctx.n = n
        for m in range(2 if paired else 1):
            rec = ctx.read_records(m)
            for k in ("reason", "score", "mismatches"):
                if not np.array_equal(rec[k], expu.per_read[k][m]):
                    bad = int(np.nonzero(rec[k] != expu.per_read[k][m])[0][0])
                    raise SystemExit("iteration %d: UMI %s of mate %d differs at read %d: got %d exp %d" %
                                     (it, k, m, bad, rec[k][bad], expu.per_read[k][m][bad]))
        if it % 25 == 0:
            print("iteration", it, "ok (UMI mode, %d reads, %d rows) %.0fs" % (n, len(rows), time.time() - t0), flush=True)
        continue
    exp = ora.call(oidx, ref, cfg, b1, o1, b2, o2, keep_per_read=True)
    got = lib.score_call(b1, o1, b2, o2)
    if [(f, c) for f, c in got] != [(f, c) for f, c in exp.rows]:
        json.dump(dict(lib=obj, strand=strand, r1=[x.decode() for x in r1], r2=None if not paired else [x.decode() for x in r2]),
                  open("gpurun_out/fuzz_fail_%d_%d.json" % (SEED, it), "w"))
        raise SystemExit("iteration %d: TABLE MISMATCH\n got %s\n exp %s" % (it, got[:5], exp.rows[:5]))
    ctx = lib.device_context()
    ctx.n = n
    for m in range(2 if paired else 1):
        rec = ctx.read_records(m)
        for k, ek in (("reason", "reason"), ("score", "score"), ("mismatches", "mismatches")):
            if not np.array_equal(rec[k], exp.per_read[ek][m]):
                bad = int(np.nonzero(rec[k] != exp.per_read[ek][m])[0][0])
                json.dump(dict(lib=obj, strand=strand, r1=[x.decode() for x in r1], r2=None if not paired else [x.decode() for x in r2]),
                          open("gpurun_out/fuzz_fail_%d_%d.json" % (SEED, it), "w"))
                raise SystemExit("iteration %d: %s of mate %d differs at read %d: got %d exp %d" %
                                 (it, k, m, bad, rec[k][bad], exp.per_read[ek][m][bad]))
    if it % 3 == 0:
        max_len = int(max(np.diff(o1.astype(np.int64)).max(), np.diff(o2.astype(np.int64)).max() if paired else 0, 1))
        for form, rows in split_forms(lib, b1, o1, b2, o2, n, paired, max_len, rng).items():
            if rows != [[f, c] for f, c in got]:
                json.dump(dict(lib=obj, strand=strand, r1=[x.decode() for x in r1], r2=None if not paired else [x.decode() for x in r2]),
                          open("gpurun_out/fuzz_fail_%d_%d.json" % (SEED, it), "w"))
                raise SystemExit("iteration %d: %s: TABLE MISMATCH\n got %s\n exp %s" % (it, form, rows[:5], got[:5]))
    if it % 25 == 0:
        print("iteration", it, "ok (%d features, %d reads, %s, %d rows) %.0fs" % (len(seqs), n, "PE" if paired else "SE", len(got), time.time() - t0), flush=True)
print("FUZZ OK:", ITER, "iterations, seed", SEED)
