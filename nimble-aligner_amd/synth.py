"""Synthetic libraries and read sets for the configs of BASELINE.json.

Deterministic for a given (T, n, seed).  numpy implementation for tests and CPU-side use; bench.py has
the same recipe written with torch ops so that 10 M reads are generated directly in HBM.

Library: allele families of 4 (mirrors tests/test-sequences/libraries/basic.json A02-0/1/2/LC): T/4 roots,
i.i.d. uniform ACGT, length uniform in [600, 2400]; alleles 2-4 = root with substitutions at rate 1 %;
allele 4 is a case-flipped copy of allele 1 in 25 % of the families (duplicate-sequence classes).
Reads (L = 150): 75 % on-target (uniform feature, uniform start, 50 % reverse-complemented, 0.5 %
substitutions), 15 % uniform random, 5 % exact duplicates of an earlier read, 3 % low-complexity
(140 x A + 10 random), 2 % on-target with 1-3 bases overwritten by N.
"""
import json

import os

import numpy as np

LIB_SEED = 0x6E696D626C65  # "nimble"
READ_SEED = 0x52454144     # "READ"
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def make_library(T, seed=LIB_SEED):
    """Returns (names, sequences) of the T features (strings; some lower-case)."""
    assert T % 4 == 0
    rng = np.random.default_rng(seed + T)
    names, seqs = [], []
    for fam in range(T // 4):
        length = int(rng.integers(600, 2401))
        root = rng.integers(0, 4, size=length, dtype=np.uint8)
        alleles = [root]
        for _ in range(3):
            a = root.copy()
            mask = rng.random(length) < 0.01
            a[mask] = (a[mask] + rng.integers(1, 4, size=int(mask.sum()), dtype=np.uint8)) % 4
            alleles.append(a)
        flip = rng.random() < 0.25
        for k, a in enumerate(alleles):
            s = ACGT[a].tobytes().decode()
            if k == 3 and flip:
                s = ACGT[alleles[0]].tobytes().decode().lower()
            names.append("F%05d-%d" % (fam, k))
            seqs.append(s)
    return names, seqs


def library_json(names, seqs, cfg=None):
    """The [config, {headers, columns}] object of a library file (reference_library.rs:28-78)."""
    config = dict(score_percent=0.33, score_filter=25, score_threshold=50, num_mismatches=0,
                  discard_multiple_matches=False, require_valid_pair=False, discard_multi_hits=0, intersect_level=0,
                  max_hits_to_report=10, group_on="", trim_target_length=40, trim_strictness=0.9, data_type="DNA")
    if cfg:
        config.update(cfg)
    body = dict(headers=["reference_genome", "sequence_name", "nt_length", "sequence"],
                columns=[["synthetic"] * len(names), list(names), [str(len(s)) for s in seqs], list(seqs)])
    return [config, body]


def write_library(path, names, seqs, cfg=None):
    with open(path, "w") as f:
        json.dump(library_json(names, seqs, cfg), f)


def expand_rows(names, seqs):
    """Row expansion of reference_library::get_reference_library (reference_library.rs:128-161): per
    feature the row itself, then name + '§rev' with the reverse-complemented sequence (U->T, N kept)."""
    comp = {"a": "t", "c": "g", "t": "a", "g": "c", "u": "a", "A": "T", "C": "G", "T": "A", "G": "C", "U": "A"}
    out_names, out_seqs = [], []
    for nme, s in zip(names, seqs):
        s = s.replace("U", "T").replace("u", "t")
        out_names += [nme, nme + "§rev"]
        out_seqs += [s, "".join(comp.get(ch, "N") for ch in reversed(s))]
    return out_names, out_seqs


def _codes(seqs):
    lut = np.zeros(256, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
        lut[ch + 32] = i
    cat = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return lut[cat], off


def make_reads(seqs, n, L=150, seed=READ_SEED, paired=False, chunk=1 << 20):
    """Returns uint8 ASCII array [n, L] (and a second one for the mates when paired)."""
    rng = np.random.default_rng(seed + n)
    cat, off = _codes(seqs)
    lens = np.diff(off)
    T = len(seqs)
    r1 = np.empty((n, L), dtype=np.uint8)
    r2 = np.empty((n, L), dtype=np.uint8) if paired else None
    kind = rng.random(n)
    # thresholds: on-target 0.75 | off-target 0.15 | duplicate 0.05 | low complexity 0.03 | N reads 0.02
    K_ON, K_OFF, K_DUP, K_LOW = 0.75, 0.90, 0.95, 0.98
    ar = np.arange(L, dtype=np.int64)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        m = hi - lo
        k = kind[lo:hi]
        f = rng.integers(0, T, size=m)
        flen = lens[f]
        strand = rng.random(m) < 0.5
        if not paired:
            start = (rng.random(m) * (flen - L + 1)).astype(np.int64)
            codes = cat[(off[f] + start)[:, None] + ar]
            rc = COMP[codes[:, ::-1]]
            codes = np.where(strand[:, None], rc, codes)
            mates = None
        else:
            frag = np.clip(np.rint(rng.normal(350.0, 50.0, size=m)).astype(np.int64), L, flen)
            start = (rng.random(m) * (flen - frag + 1)).astype(np.int64)
            head = cat[(off[f] + start)[:, None] + ar]                      # fragment[0:L]
            tail = cat[(off[f] + start + frag - L)[:, None] + ar]           # fragment[-L:]
            tail_rc = COMP[tail[:, ::-1]]                                   # revcomp(fragment)[0:L]
            codes = np.where(strand[:, None], tail_rc, head)
            mates = np.where(strand[:, None], head, tail_rc)

        def mutate(c):
            mask = rng.random((m, L)) < 0.005
            bump = rng.integers(1, 4, size=(m, L), dtype=np.uint8)
            return np.where(mask, (c + bump) % 4, c).astype(np.uint8)

        codes = mutate(codes)
        out = ACGT[codes]
        outm = ACGT[mutate(mates)] if paired else None
        # off-target
        sel = (k >= K_ON) & (k < K_OFF)
        cnt = int(sel.sum())
        out[sel] = ACGT[rng.integers(0, 4, size=(cnt, L), dtype=np.uint8)]
        if paired:
            outm[sel] = ACGT[rng.integers(0, 4, size=(cnt, L), dtype=np.uint8)]
        # low complexity: 140 x A then 10 random
        sel = (k >= K_DUP) & (k < K_LOW)
        cnt = int(sel.sum())
        low = np.full((cnt, L), ord("A"), dtype=np.uint8)
        low[:, L - 10:] = ACGT[rng.integers(0, 4, size=(cnt, 10), dtype=np.uint8)]
        out[sel] = low
        if paired:
            outm[sel] = low
        # N reads: 1-3 positions overwritten
        sel = np.nonzero(k >= K_LOW)[0]
        for _ in range(3):
            take = sel[rng.random(sel.size) < (2.0 / 3.0)] if _ else sel
            out[take, rng.integers(0, L, size=take.size)] = ord("N")
        r1[lo:hi] = out
        if paired:
            r2[lo:hi] = outm
    # exact duplicates of an earlier non-duplicate read
    dup = np.nonzero((kind >= K_OFF) & (kind < K_DUP))[0]
    nondup = np.nonzero(~((kind >= K_OFF) & (kind < K_DUP)))[0]
    before = np.searchsorted(nondup, dup)
    ok = before > 0
    src = nondup[(rng.random(dup.size) * np.maximum(before, 1)).astype(np.int64)]
    r1[dup[ok]] = r1[src[ok]]
    if paired:
        r2[dup[ok]] = r2[src[ok]]
    return (r1, r2) if paired else r1


def fixed_offsets(n, L):
    return np.arange(n + 1, dtype=np.uint64) * np.uint64(L)


def write_fastq(path, reads, prefix="r"):
    qual = b"I" * reads.shape[1]
    with open(path, "wb") as f:
        for i in range(reads.shape[0]):
            f.write(b"@" + prefix.encode() + str(i).encode() + b"\n" + reads[i].tobytes() + b"\n+\n" + qual + b"\n")


def write_fastq_fast(path, reads, prefix=b"r", chunk=1 << 20, qual="const"):
    """Vectorised FASTQ writer for large synthetic sets: fixed-width records '@r%010d' / bases / '+' / 'I'*L.
    qual="binned": qualities drawn from four values the way a binning sequencer writes them (mostly 'F', some ':', ',',
    '#') -- what a gzip stream of real reads spends most of its literals on; a constant quality line compresses to
    nothing and flatters a gzip reader."""
    n, L = reads.shape
    head = 1 + len(prefix) + 10 + 1
    rec = head + L + 3 + L + 1
    with open(path, "wb") as f:
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            m = hi - lo
            out = np.empty((m, rec), dtype=np.uint8)
            out[:, 0] = ord("@")
            out[:, 1:1 + len(prefix)] = np.frombuffer(prefix, dtype=np.uint8)
            idx = np.arange(lo, hi, dtype=np.int64)
            for d in range(10):
                out[:, 1 + len(prefix) + 9 - d] = (idx // 10 ** d) % 10 + ord("0")
            out[:, head - 1] = ord("\n")
            out[:, head:head + L] = reads[lo:hi]
            out[:, head + L:head + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
            if qual == "binned":
                rng = np.random.default_rng(lo + 17)
                q = np.frombuffer(b"F:,#", dtype=np.uint8)[rng.choice(4, size=(m, L), p=[0.90, 0.06, 0.03, 0.01])]
                out[:, head + L + 3:head + 2 * L + 3] = q
            else:
                out[:, head + L + 3:head + 2 * L + 3] = ord("I")
            out[:, rec - 1] = ord("\n")
            f.write(out.tobytes())


def _deflate_part(args):
    path, lo, hi, level, last = args
    import zlib
    with open(path, "rb") as f:
        f.seek(max(lo - 32768, 0))
        prime = f.read(lo - max(lo - 32768, 0))
        data = f.read(hi - lo)
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, prime) if prime else \
        zlib.compressobj(level, zlib.DEFLATED, -15, 9)
    return c.compress(data) + c.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH), zlib.crc32(data), len(data)


def gzip_single_stream(src, dst, level=6, part=32 << 20, workers=None):
    """`src` compressed into ONE gzip member by many processes (the way pigz does it: each part is deflated with the 32 KiB
    before it as its dictionary and closed with a sync flush, so the parts concatenate into a single deflate stream whose
    back-references cross the seams).  For large synthetic .fastq.gz inputs, where `gzip` itself would take minutes."""
    import multiprocessing as mp
    import struct
    import zlib
    size = os.path.getsize(src)
    cuts = list(range(0, size, part)) + [size]
    jobs = [(src, cuts[i], cuts[i + 1], level, i + 2 == len(cuts)) for i in range(len(cuts) - 1)]
    if not jobs:
        jobs = [(src, 0, 0, level, True)]
    workers = workers or min(len(jobs), max(1, (os.cpu_count() or 2) // 2), 32)
    crc, total = 0, 0
    with mp.get_context("fork").Pool(workers) as pool, open(dst, "wb") as out:
        out.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for blob, c, n in pool.imap(_deflate_part, jobs):
            out.write(blob)
            crc = _crc32_combine(crc, c, n)
            total += n
        out.write(struct.pack("<II", crc & 0xFFFFFFFF, total & 0xFFFFFFFF))


def _crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (python's zlib does not export it): crc of A+B from crc(A), crc(B), len(B) -- multiplication
    by x^(8 len2) modulo the CRC polynomial, on GF(2) matrices."""
    if len2 == 0:
        return crc1

    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[i]) for i in range(32)]

    odd = [0xEDB88320] + [1 << i for i in range(31)]  # the operator for one zero bit
    even = square(odd)    # two
    odd = square(even)    # four
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def make_reads_torch(seqs, n, L=150, seed=READ_SEED, device="cuda:0", chunk=1 << 20, mix=None, subst=0.005):
    """Same recipe as make_reads (single-end) with torch ops, generating straight into device memory.
    Returns a uint8 tensor [n, L] on `device`.  Deterministic for (n, seed) on a given torch build.
    `mix` = cumulative shares (on-target, off-target, duplicate, low-complexity; the rest carry N) and `subst` = per-base
    substitution rate: the defaults are the bench recipe, other values are for probing the kernel (tools/mix_probe.py)."""
    import torch

    cat_np, off_np = _codes(seqs)
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.init()  # a Generator on a cuda device does not trigger torch's lazy initialisation
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed + n))
    cat = torch.from_numpy(cat_np).to(dev)
    off = torch.from_numpy(off_np).to(dev)
    lens = off[1:] - off[:-1]
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    T = len(seqs)
    out = torch.empty((n, L), dtype=torch.uint8, device=dev)
    kind = torch.rand(n, generator=g, device=dev)
    K_ON, K_OFF, K_DUP, K_LOW = mix if mix is not None else (0.75, 0.90, 0.95, 0.98)
    ar = torch.arange(L, device=dev)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        m = hi - lo
        k = kind[lo:hi]
        f = torch.randint(0, T, (m,), generator=g, device=dev)
        flen = lens[f]
        start = (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (flen - L + 1).double()).long()
        codes = cat[(off[f] + start)[:, None] + ar]
        strand = torch.rand(m, generator=g, device=dev) < 0.5
        rc = 3 - codes.flip(1)
        codes = torch.where(strand[:, None], rc, codes)
        mask = torch.rand((m, L), generator=g, device=dev) < subst
        bump = torch.randint(1, 4, (m, L), generator=g, device=dev, dtype=torch.uint8)
        codes = torch.where(mask, (codes + bump) % 4, codes)
        rnd = torch.randint(0, 4, (m, L), generator=g, device=dev, dtype=torch.uint8)
        offt = (k >= K_ON) & (k < K_OFF)
        codes = torch.where(offt[:, None], rnd, codes)
        low = (k >= K_DUP) & (k < K_LOW)
        lowc = rnd.clone()
        lowc[:, : L - 10] = 0
        codes = torch.where(low[:, None], lowc, codes)
        asc = acgt[codes.long()]
        nsel = torch.nonzero(k >= K_LOW).flatten()
        for rep in range(3):
            take = nsel if rep == 0 else nsel[torch.rand(nsel.numel(), generator=g, device=dev) < (2.0 / 3.0)]
            pos = torch.randint(0, L, (take.numel(),), generator=g, device=dev)
            asc[take, pos] = ord("N")
        out[lo:hi] = asc
    isdup = (kind >= K_OFF) & (kind < K_DUP)
    dup = torch.nonzero(isdup).flatten()
    nondup = torch.nonzero(~isdup).flatten()
    before = torch.searchsorted(nondup, dup)
    ok = before > 0
    pick = (torch.rand(dup.numel(), generator=g, device=dev, dtype=torch.float64) * before.clamp(min=1).double()).long()
    src = nondup[pick]
    out[dup[ok]] = out[src[ok]]
    return out


def make_family_library(T, F, seed=5, divergence=0.01):
    """T features = T // F gene families of F alleles each: a random root of 600..2400 bases and F - 1 copies with
    substitutions at `divergence` -- the shape of an immune-gene library (hundreds of alleles per gene, the domain of the
    reference's own fixtures, tests/test-sequences/libraries/basic.json).  Returns (names, sequences)."""
    rng = np.random.default_rng(seed)
    names, seqs = [], []
    for fam in range(T // F):
        length = int(rng.integers(600, 2401))
        root = rng.integers(0, 4, size=length, dtype=np.uint8)
        for k in range(F):
            a = root.copy()
            if k:
                m = rng.random(length) < divergence
                a[m] = (a[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) % 4
            names.append("G%04d*%03d" % (fam, k))
            seqs.append(ACGT[a].tobytes().decode())
    return names, seqs


def make_pairs_torch(seqs, n, L=150, seed=READ_SEED, device="cuda:0", chunk=1 << 20):
    """The paired recipe of make_reads with torch ops, straight into device memory: fragment length ~ round(N(350, 50^2))
    clipped to [L, feature length], R1 = fragment[0:L], R2 = revcomp(fragment)[0:L] (or the other way round, by strand),
    0.5 % substitutions per mate, the same shares of off-target pairs, duplicates of earlier pairs, low-complexity pairs
    and pairs whose first mate carries N.  Returns two uint8 tensors [n, L].  Deterministic for (n, seed) on a given
    torch build; not the same stream of numbers as the numpy form."""
    import torch

    cat_np, off_np = _codes(seqs)
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.init()
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed + 2 * n + 1))
    cat = torch.from_numpy(cat_np).to(dev)
    off = torch.from_numpy(off_np).to(dev)
    lens = off[1:] - off[:-1]
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    T = len(seqs)
    o1 = torch.empty((n, L), dtype=torch.uint8, device=dev)
    o2 = torch.empty((n, L), dtype=torch.uint8, device=dev)
    kind = torch.rand(n, generator=g, device=dev)
    K_ON, K_OFF, K_DUP, K_LOW = 0.75, 0.90, 0.95, 0.98
    ar = torch.arange(L, device=dev)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        m = hi - lo
        k = kind[lo:hi]
        f = torch.randint(0, T, (m,), generator=g, device=dev)
        flen = lens[f]
        frag = torch.round(torch.randn(m, generator=g, device=dev, dtype=torch.float64) * 50.0 + 350.0).long()
        frag = torch.minimum(torch.clamp(frag, min=L), flen)
        start = (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (flen - frag + 1).double()).long()
        head = cat[(off[f] + start)[:, None] + ar]
        tail_rc = 3 - cat[(off[f] + start + frag - L)[:, None] + ar].flip(1)
        strand = torch.rand(m, generator=g, device=dev) < 0.5
        c1 = torch.where(strand[:, None], tail_rc, head)
        c2 = torch.where(strand[:, None], head, tail_rc)
        rnd = []
        outs = []
        for c in (c1, c2):
            mask = torch.rand((m, L), generator=g, device=dev) < 0.005
            bump = torch.randint(1, 4, (m, L), generator=g, device=dev, dtype=torch.uint8)
            c = torch.where(mask, (c + bump) % 4, c)
            r = torch.randint(0, 4, (m, L), generator=g, device=dev, dtype=torch.uint8)
            offt = (k >= K_ON) & (k < K_OFF)
            c = torch.where(offt[:, None], r, c)
            rnd.append(r)
            outs.append(c)
        low = (k >= K_DUP) & (k < K_LOW)
        lowc = rnd[0].clone()
        lowc[:, : L - 10] = 0
        a1 = acgt[torch.where(low[:, None], lowc, outs[0]).long()]
        a2 = acgt[torch.where(low[:, None], lowc, outs[1]).long()]
        nsel = torch.nonzero(k >= K_LOW).flatten()
        for rep in range(3):
            take = nsel if rep == 0 else nsel[torch.rand(nsel.numel(), generator=g, device=dev) < (2.0 / 3.0)]
            pos = torch.randint(0, L, (take.numel(),), generator=g, device=dev)
            a1[take, pos] = ord("N")
        o1[lo:hi] = a1
        o2[lo:hi] = a2
    isdup = (kind >= K_OFF) & (kind < K_DUP)
    dup = torch.nonzero(isdup).flatten()
    nondup = torch.nonzero(~isdup).flatten()
    before = torch.searchsorted(nondup, dup)
    ok = before > 0
    pick = (torch.rand(dup.numel(), generator=g, device=dev, dtype=torch.float64) * before.clamp(min=1).double()).long()
    src = nondup[pick]
    o1[dup[ok]] = o1[src[ok]]
    o2[dup[ok]] = o2[src[ok]]
    return o1, o2
