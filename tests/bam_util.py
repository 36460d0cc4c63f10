"""Test infrastructure for the BAM pipeline: a minimal BGZF + BAM writer (the reference's own test BAMs are git-LFS
pointers in this checkout, so fixtures are written here from the synthetic generators), and an independent Python restatement of
what the reference does between the file and the aligner:

  model_groups()      src/parse/sorted_bam_reader.rs:6-186 + src/parse/bam.rs:51-288 on a list of records
  expected_rows()     src/process/bam.rs:305-405 + the TSV of :92-121, with the CPU oracle's call_umi as the aligner

Nothing here is imported by the product."""
import struct
import zlib

import numpy as np

FIELDS = ["QNAME", "QUAL", "REVERSE", "MATE_REVERSE", "PAIRED", "PROPER_PAIRED", "PAIR_ORIENTATION", "UNMAPPED",
          "MATE_UNMAPPED", "FIRST_IN_TEMPLATE", "LAST_IN_TEMPLATE", "STRAND", "MAPQ", "POS", "MATE_POS", "SEQ", "SEQ_LEN",
          "INSERT_SIZE", "QUALITY_FAILED", "SECONDARY", "DUPLICATE", "SUPPLEMENTARY", "NH", "HI", "AS", "GN", "TX", "AN", "nM",
          "fx", "RE", "CR", "CY", "CB", "UR", "UY", "UB", "SKIP_ALIGN"]
REASON_TEXT = {0: "Score Below Threshold", 1: "Discarded Multiple Match", 2: "Discarded Nonzero Mismatch", 3: "No Match",
               6: "Required Valid Pair Not Matching", 8: "Short Read", 9: "Max Hits Exceeded", 10: "Low Entropy",
               11: "Successful Match", 13: "Equivalence Class Empty After Filters", 14: "Above Mismatch Threshold",
               15: "SKipped Align Due To Unpaired Dummy Read", 16: "None", 7: "Force Intersect Failure"}


def _bgzf_block(data):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = comp.compress(data) + comp.flush()
    bsize = 12 + 6 + len(body) + 8 - 1
    head = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def write_bam(path, records, block=40000):
    """records: dicts with qname, flag, seq (str), qual (bytes of Phred values, b"" = absent), tags (ordered dict: name ->
    str for Z, int for i), and optional tid, pos, mapq, mtid, mpos, tlen."""
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    out = bytearray(b"BAM\x01")
    text = b"@HD\tVN:1.6\tSO:unknown\n@SQ\tSN:chr1\tLN:100000\n"
    out += struct.pack("<i", len(text)) + text + struct.pack("<i", 1)
    out += struct.pack("<i", 5) + b"chr1\x00" + struct.pack("<i", 100000)
    for r in records:
        name = r["qname"].encode() + b"\x00"
        seq = r["seq"].upper()
        packed = bytearray((len(seq) + 1) // 2)
        for i, c in enumerate(seq):
            packed[i // 2] |= code.get(c, 15) << (0 if i & 1 else 4)
        qual = r["qual"] if r["qual"] else b"\xff" * len(seq)
        aux = bytearray()
        for tag, v in r.get("tags", {}).items():
            if isinstance(v, int):
                aux += tag.encode() + b"i" + struct.pack("<i", v)
            else:
                aux += tag.encode() + b"Z" + v.encode() + b"\x00"
        body = struct.pack("<iiBBHHHiiii", r.get("tid", 0), r.get("pos", 0), len(name), r.get("mapq", 255), 4680, 0,
                           r["flag"], len(seq), r.get("mtid", 0), r.get("mpos", 0), r.get("tlen", 0))
        body += name + bytes(packed) + qual + bytes(aux)
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        for lo in range(0, len(out), block):
            f.write(_bgzf_block(bytes(out[lo:lo + block])))
        f.write(_bgzf_block(b""))  # the EOF marker block


def _canon(seq):
    return "".join(c if c in "ACGT" else "A" for c in seq.upper())


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def _orientation(r):
    f = r["flag"]
    if not (f & 1 and not f & 4 and not f & 8 and r.get("tid", 0) == r.get("mtid", 0)):
        return "None"
    pos, mpos = r.get("pos", 0), r.get("mpos", 0)
    if pos == mpos:
        return "None"
    rev, mrev = bool(f & 16), bool(f & 32)
    if f & 64:
        p1, p2, f1, f2 = pos, mpos, not rev, not mrev
    else:
        p1, p2, f1, f2 = mpos, pos, not mrev, not rev
    if p1 < p2:
        return {(True, True): "F1F2", (True, False): "F1R2", (False, True): "R1F2", (False, False): "R1R2"}[(f1, f2)]
    return {(True, True): "F2F1", (True, False): "F2R1", (False, True): "R2F1", (False, False): "R2R1"}[(f2, f1)]


def _fields(r, skip_align):
    f = r["flag"]
    rev = bool(f & 16)
    seq = r["seq"]
    if len(seq) == 124:
        seq = seq[:-13] if rev else seq[13:]
    seq = _canon(seq)
    qual = r["qual"].decode("latin-1") if r["qual"] else ""       # absent qualities (0xFF) are no UTF-8: empty string
    if len(qual) == 124:
        qual = qual[:-13] if rev else qual[13:]
    if rev:
        qual = qual[::-1]
    b = lambda v: "true" if v else "false"
    tags = r.get("tags", {})
    named = {"QNAME": r["qname"], "QUAL": qual, "REVERSE": b(rev), "MATE_REVERSE": b(f & 32), "PAIRED": b(f & 1),
             "PROPER_PAIRED": b(f & 2), "PAIR_ORIENTATION": _orientation(r), "UNMAPPED": b(f & 4), "MATE_UNMAPPED": b(f & 8),
             "FIRST_IN_TEMPLATE": b(f & 64), "LAST_IN_TEMPLATE": b(f & 128), "STRAND": "-" if rev else "+",
             "MAPQ": str(r.get("mapq", 255)), "POS": str(r.get("pos", 0)), "MATE_POS": str(r.get("mpos", 0)), "SEQ": seq,
             "SEQ_LEN": str(len(r["seq"])), "INSERT_SIZE": str(r.get("tlen", 0)), "QUALITY_FAILED": b(f & 512),
             "SECONDARY": b(f & 256), "DUPLICATE": b(f & 1024), "SUPPLEMENTARY": b(f & 2048)}
    out = []
    for name in FIELDS:
        if name == "SKIP_ALIGN":
            out.append(skip_align)
        elif len(name) == 2 and isinstance(tags.get(name), str):
            out.append(tags[name])
        else:
            out.append(named.get(name, ""))       # integer tags and absent tags read as empty
    return seq, out


def model_groups(records, force_bam_paired=False):
    """-> [(umi, cell barcode, dropped, [(sequence, [38 fields])])] exactly as UMIReader yields them."""
    def umi(r):
        t = r.get("tags", {})
        return t["UB"] if isinstance(t.get("UB"), str) else t["UR"]

    kept = [r for r in records
            if not (force_bam_paired and not r["flag"] & 1) and isinstance(r.get("tags", {}).get("CB"), str)
            and umi(r) != "AAAAAAAAAA"]
    # runs of one UMI in file order; every run but the last is stably sorted by the cell barcode
    runs = []
    for r in kept:
        if runs and umi(runs[-1][0]) == umi(r):
            runs[-1].append(r)
        else:
            runs.append([r])
    stream = []
    for k, run in enumerate(runs):
        if k + 1 < len(runs):
            run = sorted(run, key=lambda r: r["tags"]["CB"])
        recs = []
        for r in run:
            recs.append((r, "FALSE" if not force_bam_paired else ""))
            if not force_bam_paired and not r["flag"] & 1:
                recs.append((r, "TRUE"))
        i = 0
        n_before = len(stream)
        while i + 1 < len(recs):
            if recs[i][0]["qname"] == recs[i + 1][0]["qname"]:
                pair = [recs[i], recs[i + 1]] if recs[i][0]["flag"] & 64 else [recs[i + 1], recs[i]]
                stream += pair
                i += 2
            else:
                i += 1
        if len(stream) == n_before:
            break   # a UMI whose records all fall to the pairing filter ends the input: SortedBamReader::next reports its
                    # empty buffer as an error and UMIReader takes any error for the end (sorted_bam_reader.rs:169-186)
    groups = []
    for r, skip in stream:
        cb = r["tags"]["CB"][:-2]
        key = (umi(r), cb)
        if not groups or groups[-1][0] != key:
            groups.append((key, []))
        groups[-1][1].append(_fields(r, skip))
    out = [(k[0], k[1], False, recs) for k, recs in groups]
    if len(out) > 1:
        out[-1] = (out[-1][0], out[-1][1], True, out[-1][3])       # read, never sent (process/bam.rs:163-178)
    if not out:
        out = [("", "", False, [])]
    return out


def bam_data(fields):
    return "\t".join(v for i, v in enumerate(fields) if i not in (1, 15))


def header():
    def h(p):
        return "\t".join("%s_%s" % (p, f) for i, f in enumerate(FIELDS) if i not in (1, 15))
    return ("nimble_features\tnimble_score\t" + h("r1") + "\t" + h("r2") + "\tr1_filter_forward\tr1_forward_score\t"
            "r1_filter_reverse\tr1_reverse_score\tr2_filter_forward\tr2_forward_score\tr2_filter_reverse\tr2_reverse_score\t"
            "triage_reason\taligndirection")


def call_inputs(groups):
    """The arrays of one call over all sent groups: bases / qualities / offsets / skip flags per mate, segment per pair."""
    b, q, off, skip, seg = ([], []), ([], []), ([0], [0]), ([], []), []
    for g, (_, _, dropped, recs) in enumerate(groups):
        if dropped:
            continue
        for k in range(0, len(recs) - 1, 2):
            for m in (0, 1):
                seq, md = recs[k + m]
                s = _revcomp(seq) if md[2] == "true" else seq
                b[m].append(s)
                q[m].append(md[1])
                off[m].append(off[m][-1] + len(s))
                skip[m].append(1 if md[37] == "TRUE" else 0)
            seg.append(g)
    enc = lambda parts: np.frombuffer("".join(parts).encode("latin-1"), dtype=np.uint8).copy()
    return dict(r1=enc(b[0]), r2=enc(b[1]), q1=enc(q[0]), q2=enc(q[1]), o1=np.asarray(off[0], dtype=np.uint64),
                o2=np.asarray(off[1], dtype=np.uint64), skip1=np.asarray(skip[0], dtype=np.uint8),
                skip2=np.asarray(skip[1], dtype=np.uint8), seg=np.asarray(seg, dtype=np.uint32), keys=[a + c for a, c in zip(b[0], b[1])])


def check_output(text, groups, inputs, oracle_result):
    """The gzip TSV of process::bam::process against the model: header; per sent group the oracle's (callset, count) rows in
    callset order, each standing on a pair of the group, then one empty row per pair whose QNAME stands for no callset; the
    filter columns = the records of the LAST pair with the same read key in the group (align.rs:591-600)."""
    if not oracle_result.rows:
        assert text == ""          # the header goes out with the first row (process/bam.rs:86-101)
        return 0
    lines = text.split("\n")
    assert lines[0] == header()
    assert lines[-1] == ""
    rows = [l.split("\t") for l in lines[1:-1]]
    width = 2 + 36 + 36 + 10
    assert all(len(r) == width for r in rows), sorted(set(len(r) for r in rows))
    per = oracle_result.per_read
    kept = per["kept"]
    exp_rows = {}
    for s, f, c in oracle_result.rows:
        exp_rows.setdefault(int(s), []).append((",".join(f), str(c)))
    seg = inputs["seg"]
    at = 0
    pair0 = 0
    n_checked = 0
    for g, (_, _, dropped, recs) in enumerate(groups):
        npairs = len(recs) // 2
        if dropped:
            continue
        idx = list(range(pair0, pair0 + npairs))
        pair0 += npairs
        want = exp_rows.get(g, [])
        if not want:
            continue        # a group without a single call writes nothing at all
        last = {}
        for i in idx:
            last[inputs["keys"][i]] = i
        got = rows[at:at + len(want)]
        assert [(r[0], r[1]) for r in got] == want, (g, [(r[0], r[1]) for r in got], want)
        qn_of = {}
        for k in range(npairs):
            qn_of.setdefault(recs[2 * k][1][0], []).append(k)
        reps = set()

        def check_row(r, k):
            m1, m2 = recs[2 * k][1], recs[2 * k + 1][1]
            assert r[2:38] == bam_data(m2).split("\t") and r[38:74] == bam_data(m1).split("\t")
            j = last[inputs["keys"][idx[k]]]
            sc = [int(per["score"][m][j]) if (per["reason"][m][j] == 11 or (per["reason"][m][j] == 6 and kept[m][j])) else 0
                  for m in (0, 1)]
            assert r[74:82] == [REASON_TEXT[int(per["reason"][1][j])], str(sc[1]), "None", "0",
                                REASON_TEXT[int(per["reason"][0][j])], str(sc[0]), "None", "0"], (g, k, r[74:84])
            assert r[83] == "None"

        for r in got:
            # the pair the row stands on: any pair of the group with that QNAME pair of field blocks
            cands = [k for k in range(npairs) if bam_data(recs[2 * k + 1][1]).split("\t") == r[2:38]
                     and bam_data(recs[2 * k][1]).split("\t") == r[38:74]]
            assert cands, (g, r[:2])
            check_row(r, cands[0])
            reps.add(recs[2 * cands[0]][1][0])
            n_checked += 1
        at += len(want)
        rest = [k for k in range(npairs) if recs[2 * k + 1][1][0] not in reps]
        got = rows[at:at + len(rest)]
        assert [(r[0], r[1]) for r in got] == [("", "0")] * len(rest)
        for r, k in zip(got, rest):
            check_row(r, k)
        at += len(rest)
    assert at == len(rows), (at, len(rows))
    return n_checked
