"""The BAM reader and the UMI grouping of the BAM pipeline (no GPU): this build's BGZF + BAM decoder and its restatement of
src/parse/sorted_bam_reader.rs + src/parse/bam.rs against an independent Python model (tests/bam_util.py) on BAM files
written by the tests, and the reference's own unit literals of src/process/bam.rs:437-471."""
import importlib
import os

import numpy as np
import pytest

import bam_util

nim = importlib.import_module("nimble-aligner_amd")


def make_records(rng, n_umis=40, seq_of=None, orphan_only_umi=False, good_quals=False):
    """10x-like records: runs of one UMI, several cells per UMI in scrambled order, mates adjacent; plus everything the
    reader has to cope with (unpaired reads, orphans, the poly-A UMI, records without CB / with UR only, reverse strand,
    124-base reads that get clipped, N bases, integer tags)."""
    acgt = "ACGT"
    rnd = lambda k: "".join(acgt[i] for i in rng.integers(0, 4, size=k))
    recs = []
    qn = 0
    for u in range(n_umis):
        umi = "AAAAAAAAAA" if u == 7 else rnd(10)
        cells = [rnd(16) + "-1" for _ in range(int(rng.integers(1, 4)))]
        if orphan_only_umi and u == 41:   # every record of this UMI falls to the pairing filter: the reference stops here
            qn += 1
            recs.append(dict(qname="q%05d" % qn, flag=0x1 | 0x40, seq=rnd(98), qual=bytes([30] * 98),
                             tags={"CB": cells[0], "UB": umi}))
            continue
        for _ in range(int(rng.integers(1, 7))):
            cb = cells[int(rng.integers(0, len(cells)))]
            qn += 1
            L = 124 if rng.random() < 0.6 else int(rng.integers(60, 151))
            kind = rng.random()
            tags = {"NH": 1, "CB": cb, "UR": umi, "GN": "GENE%d" % (qn % 5), "CR": cb[:-2], "nM": 0}
            if rng.random() < 0.8:
                tags["UB"] = umi
                tags = {k: tags[k] for k in ("NH", "CB", "UB", "UR", "GN", "CR", "nM")}
            if rng.random() < 0.03:
                del tags["CB"]
            mk = lambda flag, **kw: dict(qname="q%05d" % qn, flag=flag, seq=(seq_of(L) if seq_of else rnd(L)),
                                         qual=bytes(rng.integers(36 if good_quals else 2, 42, size=L).astype(np.uint8)), tags=dict(tags),
                                         pos=int(rng.integers(0, 9000)), mpos=int(rng.integers(0, 9000)),
                                         tlen=int(rng.integers(-500, 500)), mapq=int(rng.integers(0, 256)), **kw)
            if kind < 0.70:        # a proper pair, mates adjacent; either may be on the reverse strand
                f1 = 0x1 | 0x2 | 0x40 | (0x10 if rng.random() < 0.3 else 0) | (0x20 if rng.random() < 0.5 else 0)
                f2 = 0x1 | 0x2 | 0x80 | (0x10 if rng.random() < 0.5 else 0)
                pair = [mk(f1), mk(f2)]
                if rng.random() < 0.3:
                    pair.reverse()  # the second in template comes first in the file
                recs += pair
            elif kind < 0.90:      # an unpaired read: gets a SKIP_ALIGN dummy (or is dropped with -p)
                recs.append(mk(0x10 if rng.random() < 0.4 else 0))
            else:                  # a paired read whose mate is not there: dropped with a warning
                recs.append(mk(0x1 | 0x40))
    for r in recs[::17]:
        s = list(r["seq"])
        s[len(s) // 2] = "N"
        r["seq"] = "".join(s)
    return recs


@pytest.mark.parametrize("force", [False, True])
@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("batch", [0, 2048, 9000])
def test_umi_groups_equal_the_model(tmp_path, force, seed, batch, monkeypatch):
    """batch: compressed bytes the BGZF reader inflates at once (0 = its own 4 MiB): with a few KB, records -- and their
    size fields -- straddle the batches, whose whole records the inflate helpers describe ahead of the decoder."""
    if batch:
        monkeypatch.setenv("NIMBLE_BGZF_BATCH", str(batch))
    rng = np.random.default_rng(seed)
    recs = make_records(rng, n_umis=60 if seed != 3 else 1, orphan_only_umi=seed == 2)
    path = str(tmp_path / "t.bam")
    bam_util.write_bam(path, recs, block=3000 + 500 * seed)   # records straddle BGZF blocks
    got = nim.bam_umi_groups(path, force)
    want = bam_util.model_groups(recs, force)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[0] == w[0] and g[1] == w[1] and g[2] == w[2]
        assert len(g[3]) == len(w[3])
        for (s1, f1), (s2, f2) in zip(g[3], w[3]):
            assert s1 == s2
            assert f1 == f2, [(bam_util.FIELDS[i], a, b) for i, (a, b) in enumerate(zip(f1, f2)) if a != b]
    n_rec = sum(len(g[3]) for g in got)
    assert seed == 3 or n_rec > 80
    if not force and seed != 3:
        flat = [f for g in got for _, f in g[3]]
        assert any(f[37] == "TRUE" for f in flat) and any(f[2] == "true" for f in flat)
        assert all(len(s) == 111 for g in got for s, f in g[3] if f[16] == "124")          # 13 bases clipped


def test_empty_and_truncated_files(tmp_path):
    p = str(tmp_path / "e.bam")
    bam_util.write_bam(p, [])
    assert nim.bam_umi_groups(p) == [("", "", False, [])]
    rng = np.random.default_rng(5)
    recs = make_records(rng, n_umis=20)
    full = str(tmp_path / "f.bam")
    bam_util.write_bam(full, recs, block=1 << 20)
    cut = str(tmp_path / "c.bam")
    data = open(full, "rb").read()
    open(cut, "wb").write(data[:len(data) // 10])         # a file that ends inside a BGZF block
    with pytest.raises(nim.Panic, match="truncated"):
        nim.bam_umi_groups(cut)
    with pytest.raises(nim.Panic, match="not a BAM file"):
        nim.bam_umi_groups(os.path.join(os.path.dirname(__file__), "golden", "reads", "basic.fastq"))


def test_plain_gzip_and_corrupt_blocks(tmp_path, monkeypatch):
    """A BAM stream inside ordinary gzip (no BGZF extra field) goes through zlib's gz layer and yields the same groups; a
    BGZF member whose payload was damaged is reported (CRC-32 / inflate), at the point where the decoder gets there."""
    import gzip
    import zlib
    rng = np.random.default_rng(9)
    recs = make_records(rng, n_umis=30)
    bg = str(tmp_path / "b.bam")
    bam_util.write_bam(bg, recs, block=5000)
    want = nim.bam_umi_groups(bg, False)
    # the same stream as one ordinary gzip member
    raw = b""
    data = open(bg, "rb").read()
    d = zlib.decompressobj(31)
    while data:
        raw += d.decompress(data)
        data = d.unused_data
        if d.eof and data:
            d = zlib.decompressobj(31)
        elif d.eof:
            break
    plain = str(tmp_path / "p.bam")
    with gzip.open(plain, "wb") as f:
        f.write(raw)
    assert nim.bam_umi_groups(plain, False) == want
    # damage the deflate payload of the SECOND member (a reader that stops early -- the reference's end-of-input quirks --
    # never reads far into the file; what lies in front of the damage is delivered, the error stands behind it)
    data = bytearray(open(bg, "rb").read())
    starts = [i for i in range(len(data) - 16) if data[i:i + 4] == b"\x1f\x8b\x08\x04" and data[i + 12:i + 14] == b"BC"]
    assert len(starts) > 4
    at = starts[1] + 30
    data[at] ^= 0x55
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(data))
    # ... or bytes that are no BGZF member at all between two members, or a member that claims more than 64 KiB
    clean = open(bg, "rb").read()
    junk = str(tmp_path / "junk.bam")
    open(junk, "wb").write(clean[:starts[1]] + bytes(rng.integers(0, 256, size=200, dtype=np.uint8)) + clean[starts[1]:])
    big = bytearray(clean)
    big[starts[2] - 4:starts[2]] = (1 << 20).to_bytes(4, "little")      # ISIZE of the second member
    huge = str(tmp_path / "huge.bam")
    open(huge, "wb").write(bytes(big))
    for batch in ("", "4096"):
        if batch:
            monkeypatch.setenv("NIMBLE_BGZF_BATCH", batch)
        for path in (bad, junk, huge):
            with pytest.raises(nim.Panic, match="corrupt BGZF block|truncated"):
                nim.bam_umi_groups(path, False)


def test_reference_unit_literals():
    # src/process/bam.rs:437-471
    assert nim.reverse_comp_if_needed("ATGC", True) == "GCAT"
    assert nim.reverse_comp_if_needed("ATGC", False) == "ATGC"
    assert nim.parse_str_as_bool("true") is True
    assert nim.parse_str_as_bool("false") is False
    with pytest.raises(nim.Panic, match='Could not parse revcomp field "invalid" as boolean'):
        nim.parse_str_as_bool("invalid")
