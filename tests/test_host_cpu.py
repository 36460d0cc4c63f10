"""CPU-only tests of the product's host logic and of the C ABI surface (no GPU, no compute calls).

Covers: the libraries load and export every symbol include/*.h declares; the C++ host mirror of
reference_library.rs / utils.rs (reference unit tests restated); the host coercion of class pairs against
the oracle on randomised inputs; the flat index builder against the oracle's graph; FASTQ / TSV plumbing.
"""
import gzip
import importlib
import json
import os
import re

import numpy as np
import pytest

from oracle import oracle as ora

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
SEP = "§"


def lib_path(name):
    return os.path.join(GOLDEN, "libraries", name)


# ---------------- C ABI surface ----------------
def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nimble_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    L = nim.hip_lib()
    names = declared_functions("nimble_hip.h")
    assert len(names) >= 16
    for n in names:
        assert hasattr(L, n), n
    assert sorted(names) == sorted(nim.HIP_SYMBOLS)
    assert L.nimble_abi_version() == 1


def test_integration_binding_lists_every_entry_point():
    # INTEGRATION.md shows the reference-side binding (the Rust extern block): it has to name every function of the header
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index('extern "C" {'):]
    block = block[:block.index("```")]
    bound = set(re.findall(r"pub fn (nimble_\w+)\(", block))
    assert bound == set(declared_functions("nimble_hip.h"))


def test_host_library_exports_every_declared_symbol():
    L = nim.host_lib()
    names = declared_functions("nimble_host.h")
    for n in names:
        assert hasattr(L, n), n
    assert sorted(names) == sorted(nim.HOST_SYMBOLS)


def test_no_device_fails_loudly():
    # the product has no CPU path: without a GPU the index cannot be built
    if nim.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(nim.NimbleError):
        nim.Index(["ACGT" * 20])
    lib = nim.Library(lib_path("basic.json"), "none")
    with pytest.raises(nim.Panic):
        lib.build_index()


# ---------------- flat index builder vs the oracle's graph ----------------
@pytest.mark.parametrize("lib", ["basic.json", "basic-rev.json", "mismatch.json", "strandedness.json"])
def test_flat_index_matches_oracle_graph_fixtures(lib):
    obj = json.load(open(lib_path(lib)))
    _, rows = synth.expand_rows(obj[1]["columns"][1], obj[1]["columns"][3])
    assert nim.flat_index_stats(rows) == ora.Index.from_sequences(rows).stats()


def test_flat_index_matches_oracle_graph_synthetic_and_cycles():
    names, seqs = synth.make_library(40)
    _, rows = synth.expand_rows(names, seqs)
    assert nim.flat_index_stats(rows) == ora.Index.from_sequences(rows).stats()
    # tandem repeats / homopolymers: pure cycles in the graph, cut canonically on both sides
    tricky = ["ACGT" * 30, "A" * 80, "AC" * 50 + "GGTTCA" * 10, "ACGT" * 10 + "TTGACCA" * 9, "ACG", ""]
    assert nim.flat_index_stats(tricky) == ora.Index.from_sequences(tricky).stats()


# ---------------- reference_library.rs:301-480 ----------------
def test_get_reference_library_valid_json():
    lib = nim.Library(lib_path("reference-library-correct.json"), "none")
    c = lib.config
    assert (c.score_percent, c.score_filter, c.score_threshold, c.num_mismatches) == (0.85, 200, 300, 2)
    assert (c.discard_multiple_matches, c.require_valid_pair, c.discard_multi_hits) == (1, 0, 1)
    assert (c.intersect_level, c.max_hits_to_report, c.trim_target_length, c.trim_strictness) == (1, 10, 40, 0.9)
    assert c.discard_nonzero_mismatch == 0 and c.reference_genome_size == 2 and c.strand_filter == 3
    assert lib.group_on == 1
    assert lib.headers == ["id", "feature_id", "sequence_name", "sequence"]
    assert lib.column(0) == ["1", "1", "2", "2"]
    assert lib.column(1) == ["fid1", "fid1", "fid2", "fid2"]
    assert lib.column(2) == ["seq_name1", "seq_name1" + SEP + "rev", "seq_name2", "seq_name2" + SEP + "rev"]
    assert lib.column(3) == ["ATGC", "GCAT", "CGTA", "TACG"]
    assert (lib.sequence_name_idx, lib.sequence_idx) == (2, 3)


@pytest.mark.parametrize("name,msg", [
    ("reference-library-missing-fields.json", "Error -- could not parse score_percent as f64"),
    ("reference-library-types-broken.json", "Error -- could not parse score_percent as f64"),
    ("reference-library-broken-format.json", "Error -- could not parse reference library JSON"),
    ("does-not-exist.json", "Error -- could not read reference library"),
])
def test_get_reference_library_panics(name, msg):
    with pytest.raises(nim.Panic, match=re.escape(msg)):
        nim.Library(lib_path(name), "none")


@pytest.mark.parametrize("lib,exp", [
    ("reference-library-rna.json", ["ATGCTT", "AAGCAT", "tTgcAT", "ATgcAa"]),
    ("reference-library-mixed-case-rna.json", ["atGcTt", "aAgCat", "TtgCAt", "aTGcaA"]),
    ("reference-library-no-rna-bases.json", ["ATGCGT", "ACGCAT", "CGTACG", "CGTACG"]),
])
def test_rna_to_dna_conversion(lib, exp):
    assert nim.Library(lib_path(lib), "none").column(3) == exp


@pytest.mark.parametrize("field,value,msg", [
    ("score_percent", 1.5, "Error -- score_percent must be between 0 and 1"),
    ("score_filter", -10, "Error -- score_filter must be positive"),
    ("trim_strictness", 1.5, "Error -- trim_strictness must be between 0 and 1"),
])
def test_sanity_check_align_config(field, value, msg):
    lib = nim.Library(lib_path("reference-library-correct.json"), "none")
    with pytest.raises(nim.Panic, match=re.escape(msg)):
        lib.update_config(**{field: value})
    lib.update_config(score_percent=0.85)  # valid config passes


def test_library_json_errors_match_reference_messages():
    base = json.load(open(lib_path("basic.json")))

    def with_cfg(**kw):
        obj = json.loads(json.dumps(base))
        obj[0].update(kw)
        return json.dumps(obj)

    with pytest.raises(nim.Panic, match="invalid intersect level"):
        nim.Library(text=with_cfg(intersect_level=3))
    with pytest.raises(nim.Panic, match="could not find column for group_on nope"):
        nim.Library(text=with_cfg(group_on="nope"))
    obj = json.loads(json.dumps(base))
    obj[1]["columns"][1][0] = 5
    with pytest.raises(nim.Panic, match="could not parse column element"):
        nim.Library(text=json.dumps(obj))
    obj = json.loads(json.dumps(base))
    obj[1]["columns"][3][0] = "ACGTX"
    with pytest.raises(nim.Panic, match="Input sequence base is not DNA: X"):
        nim.Library(text=json.dumps(obj))
    obj = json.loads(json.dumps(base))
    obj[1]["headers"][1] = "name"
    with pytest.raises(nim.Panic, match="Could not find header sequence_name"):
        nim.Library(text=json.dumps(obj))


def test_library_agrees_with_oracle_reference():
    for name in ("basic.json", "basic-rev.json", "mismatch.json", "strandedness.json"):
        lib = nim.Library(lib_path(name), "fiveprime")
        cfg, ref = ora.get_reference_library(lib_path(name), "fiveprime")
        assert lib.headers == ref.headers
        for c in range(lib.n_cols):
            assert lib.column(c) == ref.column(c)
        hc = lib.config
        for f, _ in ora.Config._fields_:
            assert getattr(hc, f) == getattr(cfg, f), f


# ---------------- utils.rs tests ----------------
def test_revcomp_entropy_maxinfo_literals():
    assert nim.revcomp("ATGC") == "GCAT" and nim.revcomp("CCGGTTAA") == "TTAACCGG"
    assert nim.revcomp("acgtuUNn") == "NNAaacgt"
    with pytest.raises(nim.Panic, match="Input sequence base is not DNA"):
        nim.revcomp("ATGX")
    for s in ("A", "AT", "ATCG", "AAAT", "ATCGATCGATCG", "ACGGT" * 30, "A" * 140 + "CGTCGTCGTA"):
        assert nim.shannon_entropy(s) == ora.shannon_entropy(s)  # bit-identical doubles

    def adj(q):
        return "".join(chr(ord(c) - 33) for c in q)
    for q, t, s in (("I" * 20, 15, 0.5), ("!" * 20, 15, 0.9), ("IIIIII!!!!!!IIIIII", 15, 0.7), ("I" * 20, 15, 1.0),
                    ("I" * 20, 15, 0.0), ("IIIIII!!!!!!IIIIII", 15, 0.8), ("5" * 124, 40, 0.9)):
        assert nim.maxinfo(adj(q), t, s) == ora.maxinfo(adj(q), t, s)
    assert nim.maxinfo(adj("IIIIII!!!!!!IIIIII"), 15, 0.7) == 6


def test_filter_reason_display_strings():
    # src/align.rs:53-77
    assert nim.filter_reason_text(10) == "Low Entropy"
    assert nim.filter_reason_text(6) == "Required Valid Pair Not Matching"
    assert nim.filter_reason_text(13) == "Equivalence Class Empty After Filters"
    assert nim.filter_reason_text(15) == "SKipped Align Due To Unpaired Dummy Read"


def test_natural_lexical_cmp_matches_oracle():
    rng = np.random.default_rng(3)
    alphabet = list("aAbB019-_.") + [SEP]
    words = ["".join(rng.choice(alphabet, size=int(rng.integers(0, 7)))) for _ in range(300)]
    words += ["A02-0", "A02-LC", "A02-0" + SEP + "rev", "a2", "A10", "x007", "x7", "x07a"]
    for a in words[:120]:
        for b in words[100:220]:
            assert nim.natural_lexical_cmp(a, b) == ora.natural_lexical_cmp(a, b), (a, b)


# ---------------- host coercion vs oracle ----------------
def _coercion_library(group_values=None, headers_group="grp"):
    names = ["F%d-%d" % (f, a) for f in range(6) for a in range(3)]
    seqs = ["ACGT" * 3] * len(names)
    headers = ["reference_genome", "sequence_name", "nt_length", "sequence", headers_group]
    groups = group_values or [("G%d" % (i // 4) if i % 5 else "") for i in range(len(names))]
    cols = [["x"] * len(names), names, ["12"] * len(names), seqs, groups]
    return headers, cols


@pytest.mark.parametrize("strand", ["unstranded", "fiveprime", "threeprime", "none"])
@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("group_on", ["", "grp"])
def test_host_coercion_matches_oracle(strand, level, group_on):
    headers, cols = _coercion_library()
    cfg_obj = dict(score_percent=0.3, score_filter=1, score_threshold=10, num_mismatches=0,
                   discard_multiple_matches=False, require_valid_pair=False, discard_multi_hits=0,
                   intersect_level=level, max_hits_to_report=4, group_on=group_on, trim_target_length=40,
                   trim_strictness=0.9)
    text = json.dumps([cfg_obj, dict(headers=headers, columns=cols)])
    lib = nim.Library(text=text, strand_filter=strand)
    oref = ora.Reference.from_columns(headers, cols, group_on)
    ocfg = ora.config_from_json(cfg_obj, len(cols[0]), strand)
    rng = np.random.default_rng(hash((strand, level, group_on)) % (2 ** 32))
    n_rows = lib.n_rows
    for trial in range(300):
        def rand_class():
            k = int(rng.integers(1, 7))
            return sorted(set(int(x) for x in rng.integers(0, n_rows, size=k)))
        c1 = rand_class() if rng.random() < 0.85 else None
        c2 = rand_class() if rng.random() < 0.7 else None
        if c1 is None and c2 is None:
            continue
        if trial % 3 == 0:
            dmh = int(rng.integers(0, 3))
            lib.update_config(discard_multi_hits=dmh)
            ocfg.discard_multi_hits = dmh
        got, triage = lib.coerce(c1, c2)
        exp, otriage = ora.coerce(oref, ocfg, c1, c2)
        assert got == exp, (c1, c2)
        assert nim.filter_reason_text(triage).replace(" ", "") != "" and (triage == ora.R[otriage])


def test_host_coercion_string_quirks():
    # feature names that end in "rev" without the separator, duplicated names, nt_sequence header branch
    names = ["Trev", "T", "dup", "dup", "Xrevrev"]
    headers = ["nt_sequence", "sequence_name", "sequence"]
    cols = [["g", "g", "h", "", "k"], names, ["ACGT" * 3] * 5]
    cfg_obj = dict(score_percent=0.3, score_filter=1, score_threshold=10, num_mismatches=0,
                   discard_multiple_matches=False, require_valid_pair=False, discard_multi_hits=0, intersect_level=0,
                   max_hits_to_report=10, group_on="nt_sequence", trim_target_length=40, trim_strictness=0.9)
    lib = nim.Library(text=json.dumps([cfg_obj, dict(headers=headers, columns=cols)]), strand_filter="none")
    oref = ora.Reference.from_columns(headers, cols, "nt_sequence")
    ocfg = ora.config_from_json(cfg_obj, 5, "none")
    for c1, c2 in (([0], None), ([2, 4], [6]), ([1, 3], [0]), ([4, 6], [5]), ([8], None), ([9], [0, 2])):
        try:
            exp = ora.coerce(oref, ocfg, c1, c2)
        except ora.OracleError as e:
            with pytest.raises(nim.Panic, match=re.escape(str(e))):
                lib.coerce(c1, c2)
            continue
        got, triage = lib.coerce(c1, c2)
        assert (got, triage) == (exp[0], ora.R[exp[1]]), (c1, c2)


# ---------------- FASTQ / TSV plumbing ----------------
def test_read_fastq_fixtures_and_gzip(tmp_path):
    assert nim.read_fastq_stats(os.path.join(GOLDEN, "reads", "basic.fastq")) == (4, 415, 114)
    # quality strings longer than the sequences must still parse (tests/mismatch.rs relies on it)
    assert nim.read_fastq_stats(os.path.join(GOLDEN, "reads", "mismatch.fastq")) == (3, 310, 104)
    assert nim.read_fastq_stats(os.path.join(GOLDEN, "reads", "fastq_pipeline_test_r1.fastq")) == (2, 16, 8)
    with pytest.raises(nim.Panic, match="Unable to read sequence"):
        nim.read_fastq_stats(os.path.join(GOLDEN, "reads", "fastq_invalid_data.fastq"))
    with pytest.raises(nim.Panic, match="could not determine compression format"):
        nim.read_fastq_stats(str(tmp_path / "missing.fastq"))
    gz = tmp_path / "r.fastq.gz"
    with gzip.open(gz, "wb") as f:
        f.write(open(os.path.join(GOLDEN, "reads", "basic.fastq"), "rb").read())
    assert nim.read_fastq_stats(str(gz)) == (4, 415, 114)
    multi = tmp_path / "multi.fastq"
    multi.write_text("@a desc\nACGT\nAC\n+\nIIII\nII\n@b\nGG\n+\nII")
    assert nim.read_fastq_stats(str(multi)) == (2, 8, 6)
    # rust-bio calls a record incomplete when its raw quality text (terminators included) is empty: a BLANK quality
    # line is text, the record stands; a record that ends before any quality line is incomplete
    blank = tmp_path / "blank.fastq"
    blank.write_text("@a\nACGT\n+\n\n@b\nGG\n+\nII\n")
    assert nim.read_fastq_stats(str(blank)) == (2, 6, 4)
    assert nim.read_fastq_batched_stats(str(blank), 16)[:3] == (2, 6, 4)
    cut = tmp_path / "cut.fastq"
    cut.write_text("@a\nACGT\n+\nIIII\n@b\nGG\n+\n")
    with pytest.raises(nim.Panic, match="Unable to read sequence"):
        nim.read_fastq_stats(str(cut))
    with pytest.raises(nim.Panic, match="Unable to read sequence"):
        nim.read_fastq_batched_stats(str(cut), 16)


def _fnv_records(records):
    h = 1469598103934665603
    for r in records:
        h = ((h ^ len(r)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        for c in r:
            h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_batched_fastq_reader_matches_whole_file_reader(tmp_path, monkeypatch):
    # the pipeline's threaded reader: same records in the same order for every batch size, plain and gzip,
    # lines that straddle the read window, a malformed record after N good ones
    rng = np.random.default_rng(3)
    recs = [bytes(rng.choice(list(b"ACGTN"), size=int(rng.integers(0, 300))).astype(np.uint8)) for _ in range(5000)]
    recs[17] = b""          # empty sequence line
    text = b"".join(b"@r%d some description\n%s\n+\n%s\n" % (i, r, b"I" * max(len(r), 1)) for i, r in enumerate(recs))
    plain = tmp_path / "x.fastq"
    plain.write_bytes(text)
    gz = tmp_path / "x.fastq.gz"
    with gzip.open(gz, "wb") as f:
        f.write(text)
    n, bases, max_len = nim.read_fastq_stats(str(plain))
    assert (n, bases, max_len) == (len(recs), sum(map(len, recs)), max(map(len, recs)))
    want = _fnv_records(recs)
    monkeypatch.setenv("NIMBLE_GZIP_SERIAL", "1")            # the one-thread zlib reader: batches of exactly `batch`
    for batch in (1, 7, 1000, 4999, 5000, 5001, 1 << 20):
        got = nim.read_fastq_batched_stats(str(gz), batch)
        assert got[:3] == (n, bases, max_len) and got[4] == want
        assert got[3] == len(recs) // batch + 1          # a final (possibly empty) batch closes the file
    monkeypatch.delenv("NIMBLE_GZIP_SERIAL")
    # the many-threaded gzip reader (tests/test_pgzip_cpu.py): batches follow its parse chunks, the records do not change
    for batch in (1, 1000, 1 << 20):
        got = nim.read_fastq_batched_stats(str(gz), batch)
        assert got[:3] == (n, bases, max_len) and got[4] == want
    # plain files are parsed in parallel chunks (one batch per chunk); the chunk size must not matter, nor the
    # number of threads, and a chunk boundary may fall anywhere (inside a header, a quality line that starts with '@')
    for chunk, threads in ((64, 3), (100, 1), (997, 8), (4096, 4), (1 << 16, 2), (1 << 30, 4)):
        monkeypatch.setenv("NIMBLE_FASTQ_CHUNK", str(chunk))
        monkeypatch.setenv("NIMBLE_FASTQ_THREADS", str(threads))
        got = nim.read_fastq_batched_stats(str(plain), 1000)
        assert got[:3] == (n, bases, max_len) and got[4] == want, (chunk, threads)
    # qualities that begin with '@' or '+', multi-line records: the guessed record starts are often wrong there and
    # the consumer has to re-parse; the result must still be the sequential reader's
    tricky = tmp_path / "tricky.fastq"
    parts, want_recs = [], []
    for i in range(3000):
        r = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 120))).astype(np.uint8))
        q = bytes(rng.choice(list(b"@+I#"), size=len(r)).astype(np.uint8))
        if i % 5 == 0:   # two sequence lines, two quality lines
            h = len(r) // 2
            parts.append(b"@m%d\n%s\n%s\n+\n%s\n%s\n" % (i, r[:h], r[h:], q[:h] or b"I", q[h:] or b"I"))
        else:
            parts.append(b"@s%d\n%s\n+\n%s\n" % (i, r, q))
        want_recs.append(r)
    tricky.write_bytes(b"".join(parts))
    tn, tb, tm = nim.read_fastq_stats(str(tricky))
    assert (tn, tb) == (len(want_recs), sum(map(len, want_recs)))
    for chunk in (64, 333, 5000, 1 << 30):
        monkeypatch.setenv("NIMBLE_FASTQ_CHUNK", str(chunk))
        got = nim.read_fastq_batched_stats(str(tricky), 100)
        assert got[:3] == (tn, tb, tm) and got[4] == _fnv_records(want_recs), chunk
    monkeypatch.setenv("NIMBLE_FASTQ_CHUNK", "128")
    # a very long line (longer than the 4 MiB window) and CRLF line ends
    big = tmp_path / "big.fastq"
    long_read = b"ACGT" * (3 << 20)
    big.write_bytes(b"@a\r\nAC\r\n+\r\nII\r\n@b\n" + long_read + b"\n+\n" + b"I" * 10 + b"\n")
    got = nim.read_fastq_batched_stats(str(big), 1)
    assert got[:3] == (2, 2 + len(long_read), len(long_read)) and got[4] == _fnv_records([b"AC", long_read])
    # malformed after 3 good records: the panic of the reference, raised after the good records were delivered
    bad = tmp_path / "bad.fastq"
    bad.write_bytes(b"@a\nAC\n+\nII\n@b\nGG\n+\nII\n@c\nTT\n+\nII\nnot a header\nAC\n+\nII\n")
    with pytest.raises(nim.Panic, match="Input R1 data malformed.: Unable to read sequence"):
        nim.read_fastq_batched_stats(str(bad), 2)
    with pytest.raises(nim.Panic, match="could not determine compression format"):
        nim.read_fastq_batched_stats(str(tmp_path / "missing.fastq"), 2)


def test_synthetic_generator_is_deterministic():
    n1, s1 = synth.make_library(8)
    n2, s2 = synth.make_library(8)
    assert (n1, s1) == (n2, s2)
    a = synth.make_reads(s1, 500)
    b = synth.make_reads(s1, 500)
    assert np.array_equal(a, b) and a.shape == (500, 150)
    r1, r2 = synth.make_reads(s1, 300, paired=True)
    assert r1.shape == r2.shape == (300, 150)


def test_host_2bit_packer_against_a_plain_restatement():
    # parse::fastq::pack_reads_2bit (AVX2 on the box, scalar elsewhere): 32 bases a word, first base on top, either case,
    # anything that is no A/C/G/T as A (DnaString::from_acgt_bytes), zero bits behind the last base
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTacgtNn-*RY", dtype=np.uint8)
    reads = [bytes(alphabet[rng.integers(0, len(alphabet), int(L))]) for L in
             list(rng.integers(0, 200, 300)) + [0, 1, 31, 32, 33, 63, 64, 65, 150, 151]]
    code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("a"): 0, ord("c"): 1, ord("g"): 2, ord("t"): 3}
    for stride in (None, 9):
        words, lens, st = nim.pack_reads_2bit(reads, stride)
        assert st == (stride or 7) and list(lens) == [len(r) for r in reads]
        for i, r in enumerate(reads):
            want = [0] * st
            for j, c in enumerate(r):
                want[j >> 5] |= code.get(c, 0) << (62 - 2 * (j & 31))
            assert [int(w) for w in words[i * st:(i + 1) * st]] == want, (i, r)
    with pytest.raises(nim.Panic, match="does not fit"):
        nim.pack_reads_2bit([b"A" * 65], 2)


def _spell(r):
    # what the packed words say of a read: upper case, anything that is no A/C/G/T as A
    t = bytes.maketrans(b"acgt", b"ACGT")
    return bytes(c if c in b"ACGT" else ord("A") for c in r.translate(t))


def test_packed_batch_reader_same_records_as_the_line_reader(tmp_path, monkeypatch):
    # the reader in the mode the FASTQ pipeline runs it in (fused parse + 2-bit pack, no ASCII copy): the records of the
    # line reader, spelt by the words, for plain and gzip input, any chunk size, ordinary and odd records, ragged lengths
    rng = np.random.default_rng(8)
    recs = []
    for i in range(6000):
        L = int(rng.integers(0, 200)) if i % 50 else 150
        r = bytes(rng.choice(list(b"ACGTNacgtn"), size=L, p=[0.22, 0.22, 0.22, 0.22, 0.02, 0.02, 0.02, 0.02, 0.02, 0.02])
                  .astype(np.uint8))
        recs.append(r)
    recs[17] = b""
    text = b"".join(b"@r%d some description\n%s\n+\n%s\n" % (i, r, b"F" * max(len(r), 1)) for i, r in enumerate(recs))
    plain = tmp_path / "x.fastq"
    plain.write_bytes(text)
    want = (len(recs), sum(map(len, recs)), max(map(len, recs)), _fnv_records([_spell(r) for r in recs]))
    for chunk, threads in ((64, 3), (997, 8), (4096, 4), (1 << 16, 2), (1 << 30, 4)):
        monkeypatch.setenv("NIMBLE_FASTQ_CHUNK", str(chunk))
        monkeypatch.setenv("NIMBLE_FASTQ_THREADS", str(threads))
        got = nim.read_fastq_packed_stats(str(plain), 1000)
        assert (got[0], got[1], got[2], got[4]) == want, (chunk, threads)
    monkeypatch.setenv("NIMBLE_FASTQ_NO_AVX2", "1")   # the general way only (parse, then pack)
    got = nim.read_fastq_packed_stats(str(plain), 1000)
    assert (got[0], got[1], got[2], got[4]) == want
    monkeypatch.delenv("NIMBLE_FASTQ_NO_AVX2")
    gz = tmp_path / "x.fastq.gz"
    with gzip.open(gz, "wb") as f:
        f.write(text)
    monkeypatch.setenv("NIMBLE_GZIP_WINDOW", "200000")
    monkeypatch.setenv("NIMBLE_GZIP_CHUNK", "65536")
    got = nim.read_fastq_packed_stats(str(gz), 1000)
    assert (got[0], got[1], got[2], got[4]) == want
    # records that are not of the ordinary shape: two sequence lines, qualities that start with '@' or '+', CRLF, and a
    # longer read appearing late (the stride grows)
    parts, odd = [], []
    for i in range(2000):
        r = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 120))).astype(np.uint8))
        q = bytes(rng.choice(list(b"@+I#"), size=len(r)).astype(np.uint8))
        if i % 5 == 0:
            h = len(r) // 2
            parts.append(b"@m%d\n%s\n%s\n+\n%s\n%s\n" % (i, r[:h], r[h:], q[:h] or b"I", q[h:] or b"I"))
        elif i % 7 == 0:
            parts.append(b"@c%d\r\n%s\r\n+\r\n%s\r\n" % (i, r, q))
        else:
            parts.append(b"@s%d\n%s\n+\n%s\n" % (i, r, q))
        odd.append(r)
    long_read = b"ACGT" * 100
    parts.append(b"@long\n" + long_read + b"\n+\n" + b"I" * 400 + b"\n")
    odd.append(long_read)
    tricky = tmp_path / "tricky.fastq"
    tricky.write_bytes(b"".join(parts))
    want = (len(odd), sum(map(len, odd)), 400, _fnv_records(odd))
    for chunk in (64, 333, 5000, 1 << 30):
        monkeypatch.setenv("NIMBLE_FASTQ_CHUNK", str(chunk))
        got = nim.read_fastq_packed_stats(str(tricky), 100)
        assert (got[0], got[1], got[2], got[4]) == want, chunk
    # a malformed record still ends the file with the reference's panic
    bad = tmp_path / "bad.fastq"
    bad.write_bytes(b"@a\nAC\n+\nII\n@b\nGG\n+\nII\nnot a header\nAC\n+\nII\n")
    with pytest.raises(nim.Panic, match="Input R1 data malformed.: Unable to read sequence"):
        nim.read_fastq_packed_stats(str(bad), 2)


def test_offsets_argument_refuses_byte_arrays():
    # the mates handed over where the offsets belong (two positional byte arrays) are refused in the binding
    lib = nim.Library(os.path.join(GOLDEN, "libraries", "basic.json"), "unstranded")
    a = np.zeros(300, dtype=np.uint8)
    with pytest.raises(TypeError, match="64-bit"):
        lib.score_call(a, a, n=2, fixed_len=150)
    with pytest.raises(TypeError, match="64-bit"):
        lib.score_call_begin(0, a, a, n=2, fixed_len=150)


# ---------------- stretch records of the fast walk (round 4) ----------------
def test_stretch_records_spell_the_unitigs():
    """The index's stretch records (what walk_fast of k_align gathers: csrc/flat_index.h FlatIndex::srec) against the unitig
    records they were cut from -- bases, class, extension bits, neighbours, forks with more than two ways out -- on the
    reference's fixture libraries, the synthetic bench library, and libraries with long unitigs, repeats and 3-4 way forks."""
    import json as _json
    rng = np.random.default_rng(5)
    cases = []
    for lib in ("basic.json", "basic-rev.json", "mismatch.json", "strandedness.json"):
        obj = _json.load(open(lib_path(lib)))
        names, seqs = obj[1]["columns"][1], obj[1]["columns"][3]
        cases.append(synth.expand_rows(names, seqs)[1])
    names, seqs = synth.make_library(120)
    cases.append(synth.expand_rows(names, seqs)[1])
    # long unitigs (unique genes of 3 kb: many continuation records) beside families with 3 and 4 alleles differing at ONE site
    # (forks with three and four ways out), a homopolymer run and a tandem repeat
    genes = ["".join(rng.choice(list("ACGT"), size=3000)) for _ in range(3)]
    root = "".join(rng.choice(list("ACGT"), size=400))
    fork = [root[:200] + b + root[201:] for b in "ACGT"]
    odd = ["A" * 90 + "".join(rng.choice(list("ACGT"), size=60)), ("ACGTTGCA" * 20)[:150] + "".join(rng.choice(list("ACGT"), size=50))]
    cases.append(genes + fork + fork[:3] + odd)
    for rows in cases:
        n_rec = nim.flat_index_selfcheck(rows)
        st = nim.flat_index_stats(rows)
        assert n_rec >= st["nodes"] > 0          # a record or more per unitig
    # an index whose classes do not fit the 64-row mask form takes the general walk: no stretch records
    fam = ["".join(rng.choice(list("ACGT"), size=300))]
    wide = []
    for a in range(80):
        s = list(fam[0])
        for p_ in np.nonzero(rng.random(len(s)) < 0.01)[0]:
            s[p_] = "ACGT"[("ACGT".index(s[p_]) + 1) % 4]
        wide.append("".join(s))
    assert nim.flat_index_selfcheck(wide) == 0
