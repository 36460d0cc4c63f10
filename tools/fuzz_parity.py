#!/usr/bin/env python3
"""Randomised differential test, GPU product vs CPU oracle (development aid): tools/fuzz_parity.py [iterations] [seed]
Random small libraries (mutated copies, shared segments, repeats, homopolymers), random alignment settings and
grouping, random reads (errors, junk, N, lower case, ragged lengths, single-end and paired): the final table and the
per-read (reason, score, mismatches) must be identical."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
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
    if it % 25 == 0:
        print("iteration", it, "ok (%d features, %d reads, %s, %d rows) %.0fs" % (len(seqs), n, "PE" if paired else "SE", len(got), time.time() - t0), flush=True)
print("FUZZ OK:", ITER, "iterations, seed", SEED)
