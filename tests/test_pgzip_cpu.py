"""The many-threaded gzip reader (host/pgzip.cpp) against zlib, on the CPU.

The reference reads .fastq.gz through niffler/flate2 on one thread (src/parse/fastq.rs:21-43); what has to hold for a
replacement is that the records come out the same, whatever the file's block structure, and that a damaged file is
refused where flate2 refuses it.  Every case decompresses with the chunk workers (forced small chunks so that even
these small files are cut many times) and compares with python's zlib, byte for byte, and then runs the pipeline's batch
reader over the file and compares with the same records read from the plain file.
"""
import gzip
import importlib
import os
import zlib

import numpy as np
import pytest

nim = importlib.import_module("nimble-aligner_amd")


def _fastq(n, seed, read_len=(40, 151)):
    """Records with ragged lengths, real-looking headers and noisy qualities (so the blocks hold many literals)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        L = int(rng.integers(*read_len))
        seq = "".join("ACGTN"[k] for k in rng.choice(5, L, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        qual = "".join(chr(33 + int(q)) for q in np.clip(rng.normal(34, 6, L), 2, 41))
        out.append("@A00%d:%d:HXXXX:%d:%d:%d:%d %d:N:0:ACGT\n%s\n+\n%s\n"
                   % (seed, i % 7, 1 + i % 4, 1101 + i % 90, i * 13 % 30000, i * 7 % 20000, 1 + i % 2, seq, qual))
    return "".join(out).encode()


@pytest.fixture(scope="module")
def raw():
    return _fastq(30000, 11)


@pytest.fixture(scope="module")
def cases(raw):
    return _cases(raw)


@pytest.fixture()
def env():
    keys = ("NIMBLE_GZIP_CHUNK", "NIMBLE_GZIP_THREADS", "NIMBLE_GZIP_WINDOW", "NIMBLE_GZIP_SERIAL", "NIMBLE_FASTQ_SERIAL")
    old = {k: os.environ.get(k) for k in keys}

    def set_(**kw):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update({k: str(v) for k, v in kw.items()})

    yield set_
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _stream(level, data, wbits=31):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits)
    return c.compress(data) + c.flush()


def _cases(raw):
    third = len(raw) // 3
    cut1 = raw.rfind(b"\n@A00", 0, third) + 1
    cut2 = raw.rfind(b"\n@A00", 0, 2 * third) + 1
    sync = zlib.compressobj(6, zlib.DEFLATED, 31)
    flushed = b"".join(sync.compress(raw[lo:lo + 200000]) + sync.flush(zlib.Z_FULL_FLUSH)
                       for lo in range(0, len(raw), 200000)) + sync.flush()
    huff = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_HUFFMAN_ONLY)
    fixed = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    return {
        "level1": _stream(1, raw),
        "level6": _stream(6, raw),
        "level9": _stream(9, raw),
        "three_members": b"".join(gzip.compress(p, 6) for p in (raw[:cut1], raw[cut1:cut2], raw[cut2:])),
        "members_cut_mid_record": b"".join(gzip.compress(raw[lo:lo + 333333], 6) for lo in range(0, len(raw), 333333)),
        "bgzf_like": b"".join(gzip.compress(raw[lo:lo + 60000], 6) for lo in range(0, len(raw), 60000)),
        "stored": _stream(0, raw),
        "full_flush_points": flushed,
        "huffman_only": huff.compress(raw) + huff.flush(),
        "fixed_codes": fixed.compress(raw) + fixed.flush(),
        "empty_member_first": gzip.compress(b"") + _stream(6, raw),
    }


CASES = ["level1", "level6", "level9", "three_members", "members_cut_mid_record", "bgzf_like", "stored",
         "full_flush_points", "huffman_only", "fixed_codes", "empty_member_first"]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("chunk,threads", [(1 << 16, 4), (1 << 18, 3), (1 << 22, 1)])
def test_decompress_equals_zlib(tmp_path, raw, cases, env, case, chunk, threads):
    data = cases[case]
    p = tmp_path / "x.fastq.gz"
    p.write_bytes(data)
    env(NIMBLE_GZIP_CHUNK=chunk)
    out = tmp_path / "x.out"
    pieces = nim.pgzip_decompress(str(p), str(out), threads)
    assert pieces >= 1
    assert out.read_bytes() == raw


@pytest.mark.parametrize("case", CASES)
def test_batch_reader_same_records(tmp_path, raw, cases, env, case):
    plain = tmp_path / "x.fastq"
    plain.write_bytes(raw)
    env()
    want = nim.read_fastq_batched_stats(str(plain), 1 << 12)
    p = tmp_path / "x.fastq.gz"
    p.write_bytes(cases[case])
    env(NIMBLE_GZIP_SERIAL=1)
    serial = nim.read_fastq_batched_stats(str(p), 1 << 12)
    # small windows: records straddle window ends, pieces straddle windows
    env(NIMBLE_GZIP_CHUNK=1 << 16, NIMBLE_GZIP_THREADS=4, NIMBLE_GZIP_WINDOW=300000)
    par = nim.read_fastq_batched_stats(str(p), 1 << 12)
    env(NIMBLE_GZIP_THREADS=3)
    par_default = nim.read_fastq_batched_stats(str(p), 1 << 12)
    for got in (serial, par, par_default):
        assert (got[0], got[1], got[2], got[4]) == (want[0], want[1], want[2], want[4])


def test_damage_is_refused(tmp_path, raw, env):
    good = _stream(6, raw)
    env(NIMBLE_GZIP_CHUNK=1 << 16, NIMBLE_GZIP_THREADS=4)
    # a flipped bit in the middle: either the deflate data stops making sense or the member's CRC-32 is wrong
    for at in (len(good) // 2, len(good) // 5, len(good) - 100):
        bad = bytearray(good)
        bad[at] ^= 0x10
        p = tmp_path / "bad.fastq.gz"
        p.write_bytes(bytes(bad))
        with pytest.raises(nim.Panic):
            nim.read_fastq_batched_stats(str(p), 1 << 12)
    # the trailer: wrong CRC, wrong length
    for k in (8, 4):
        bad = bytearray(good)
        bad[-k] ^= 1
        p = tmp_path / "bad.fastq.gz"
        p.write_bytes(bytes(bad))
        with pytest.raises(nim.Panic):
            nim.read_fastq_batched_stats(str(p), 1 << 12)
    # cut off: in the middle, and inside the trailer
    for cut in (len(good) // 3, len(good) - 5):
        p = tmp_path / "cut.fastq.gz"
        p.write_bytes(good[:cut])
        with pytest.raises(nim.Panic):
            nim.read_fastq_batched_stats(str(p), 1 << 12)


def test_window_references_across_chunks(tmp_path, env):
    """A stream whose every chunk leans on the 32 KiB in front of it: one 20 kB record-like unit repeated, so almost the
    whole output is back-references whose sources lie in the unknown window of each chunk."""
    unit = _fastq(80, 3)
    raw = unit * 400
    p = tmp_path / "rep.fastq.gz"
    p.write_bytes(_stream(9, raw))
    env(NIMBLE_GZIP_CHUNK=1 << 16)
    out = tmp_path / "rep.out"
    nim.pgzip_decompress(str(p), str(out), 4)
    assert out.read_bytes() == raw


def test_empty_and_tiny(tmp_path, env):
    env(NIMBLE_GZIP_CHUNK=1 << 16)
    for name, raw in (("empty", b""), ("one", b"@r\nACGT\n+\nIIII\n")):
        p = tmp_path / (name + ".fastq.gz")
        p.write_bytes(gzip.compress(raw))
        out = tmp_path / (name + ".out")
        nim.pgzip_decompress(str(p), str(out), 4)
        assert out.read_bytes() == raw
    got = nim.read_fastq_batched_stats(str(tmp_path / "one.fastq.gz"), 16)
    assert got[:3] == (1, 4, 4)
