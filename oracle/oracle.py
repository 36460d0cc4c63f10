"""ctypes binding of the CPU oracle (oracle/libnimble_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under nimble-aligner_amd/ imports this module.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnimble_oracle.so")

REASONS = [
    "ScoreBelowThreshold", "DiscardedMultipleMatch", "DiscardedNonzeroMismatch", "NoMatch",
    "NoMatchAndScoreBelowThreshold", "DifferentFilterReasons", "NotMatchingPair", "ForceIntersectFailure",
    "ShortRead", "MaxHitsExceeded", "HighEntropy", "SuccessfulMatch", "StrandWasWrong",
    "TriageEmptyEquivalenceClass", "AboveMismatchThreshold", "SkippedAlignDueToUnpairedDummy", "None",
]
R = {name: i for i, name in enumerate(REASONS)}
CHEM = {"unstranded": 0, "fiveprime": 1, "threeprime": 2, "none": 3}


class Config(C.Structure):
    """AlignFilterConfig (src/align.rs:79-95)."""
    _fields_ = [
        ("reference_genome_size", C.c_uint64),
        ("score_percent", C.c_double),
        ("score_threshold", C.c_uint64),
        ("num_mismatches", C.c_uint64),
        ("discard_nonzero_mismatch", C.c_int32),
        ("discard_multiple_matches", C.c_int32),
        ("score_filter", C.c_int32),
        ("intersect_level", C.c_int32),
        ("require_valid_pair", C.c_int32),
        ("strand_filter", C.c_int32),
        ("discard_multi_hits", C.c_uint64),
        ("max_hits_to_report", C.c_uint64),
        ("trim_strictness", C.c_double),
        ("trim_target_length", C.c_uint64),
    ]

    def copy(self, **kw):
        c = Config()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(Config))
        for k, v in kw.items():
            setattr(c, k, v)
        return c


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "nimble_oracle.cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "libnimble_oracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, cp, i32, u64, dbl = C.c_void_p, C.c_char_p, C.c_int, C.c_uint64, C.c_double
        pp = C.POINTER(C.c_char_p)
        sig = {
            "ora_last_error": (cp, []),
            "ora_ref_create": (vp, [i32, pp, i32, pp, cp]),
            "ora_ref_create_raw": (vp, [i32, pp, i32, pp, i32, i32, i32]),
            "ora_ref_free": (None, [vp]),
            "ora_ref_n_rows": (i32, [vp]),
            "ora_ref_n_cols": (i32, [vp]),
            "ora_ref_group_on": (i32, [vp]),
            "ora_ref_sequence_name_idx": (i32, [vp]),
            "ora_ref_sequence_idx": (i32, [vp]),
            "ora_ref_header": (cp, [vp, i32]),
            "ora_ref_cell": (cp, [vp, i32, i32]),
            "ora_ref_push_column": (i32, [vp, cp, pp, i32]),
            "ora_ref_set_group_on": (None, [vp, i32]),
            "ora_sanity_check_config": (i32, [C.POINTER(Config)]),
            "ora_index_build_from_ref": (vp, [vp]),
            "ora_index_build": (vp, [i32, pp]),
            "ora_index_free": (None, [vp]),
            "ora_index_stats": (None, [vp, C.POINTER(u64)]),
            "ora_index_node": (i32, [vp, C.c_uint32, cp, i32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32)]),
            "ora_index_class": (i32, [vp, C.c_uint32, C.POINTER(C.c_uint32), i32]),
            "ora_map_read": (i32, [vp, cp, i32, i32, C.POINTER(C.c_uint32), i32, C.POINTER(i32), C.POINTER(i32),
                                   C.POINTER(i32)]),
            "ora_pseudoalign": (i32, [vp, C.POINTER(Config), cp, i32, i32, C.POINTER(C.c_uint32), i32,
                                      C.POINTER(i32), C.POINTER(i32), C.POINTER(dbl), C.POINTER(i32)]),
            "ora_filter_alignment_by_metrics": (i32, [i32, u64, dbl, u64, dbl, i32, u64, u64]),
            "ora_filter_pair": (i32, [C.POINTER(C.c_uint32), i32, C.POINTER(C.c_uint32), i32]),
            "ora_shannon_entropy": (dbl, [cp]),
            "ora_natural_lexical_cmp": (i32, [cp, cp]),
            "ora_maxinfo": (u64, [cp, i32, u64, dbl]),
            "ora_revcomp": (i32, [cp, cp]),
            "ora_coerce": (i32, [vp, C.POINTER(Config), i32, C.POINTER(C.c_uint32), i32, i32,
                                 C.POINTER(C.c_uint32), i32, cp, i32]),
            "ora_filter_read_calls_with_orientation": (i32, [cp, cp, i32]),
            "ora_filter_orientation_on_library_chemistry": (i32, [cp, cp, i32, cp, cp, i32]),
            "ora_process_class_to_features": (i32, [vp, C.POINTER(Config), C.POINTER(C.c_uint32), i32, i32, cp, i32]),
            "ora_parse_calls": (i32, [cp, cp, i32]),
            "ora_unmap": (i32, [vp, cp, C.POINTER(C.c_uint32), i32]),
            "ora_reference_sequence_data": (i32, [vp, cp, i32]),
            "ora_sort_score_vector": (i32, [cp, i32, C.POINTER(C.c_int32)]),
            "ora_call": (vp, [vp, vp, C.POINTER(Config), vp, vp, vp, vp, u64, i32, i32]),
            "ora_result_free": (None, [vp]),
            "ora_result_n_rows": (u64, [vp]),
            "ora_result_row": (cp, [vp, u64, C.POINTER(C.c_int32)]),
            "ora_result_reason": (C.POINTER(C.c_int32), [vp, i32]),
            "ora_result_score": (C.POINTER(C.c_int32), [vp, i32]),
            "ora_result_mismatch": (C.POINTER(C.c_int32), [vp, i32]),
            "ora_result_class_hash": (C.POINTER(C.c_uint64), [vp, i32]),
            "ora_result_kept": (C.POINTER(C.c_uint8), [vp, i32]),
            "ora_result_counted": (C.POINTER(C.c_uint8), [vp]),
            "ora_result_counters": (None, [vp, C.POINTER(u64)]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _err():
    return OracleError(lib().ora_last_error().decode("utf-8", "replace"))


def _strs(seq):
    arr = (C.c_char_p * len(seq))()
    arr[:] = [s.encode("utf-8") if isinstance(s, str) else s for s in seq]
    return arr


def _u32(seq):
    a = np.ascontiguousarray(np.asarray(list(seq), dtype=np.uint32))
    return a, a.ctypes.data_as(C.POINTER(C.c_uint32)), int(a.size)


class Reference:
    """reference_library::Reference (src/reference_library.rs:10-17)."""

    def __init__(self, handle):
        self.h = handle

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:   # (module globals are gone at interpreter exit)
            lib().ora_ref_free(self.h)
            self.h = None

    @classmethod
    def from_columns(cls, headers, columns, group_on=""):
        n_rows = len(columns[0]) if columns else 0
        cells = [v for col in columns for v in col]
        h = lib().ora_ref_create(len(headers), _strs(headers), n_rows, _strs(cells), group_on.encode())
        if not h:
            raise _err()
        return cls(h)

    @classmethod
    def raw(cls, headers, columns, group_on, sequence_name_idx, sequence_idx):
        n_rows = len(columns[0]) if columns else 0
        cells = [v for col in columns for v in col]
        return cls(lib().ora_ref_create_raw(len(headers), _strs(headers), n_rows, _strs(cells), group_on,
                                            sequence_name_idx, sequence_idx))

    @property
    def n_rows(self):
        return lib().ora_ref_n_rows(self.h)

    @property
    def n_cols(self):
        return lib().ora_ref_n_cols(self.h)

    @property
    def group_on(self):
        return lib().ora_ref_group_on(self.h)

    @group_on.setter
    def group_on(self, col):
        lib().ora_ref_set_group_on(self.h, col)

    @property
    def sequence_name_idx(self):
        return lib().ora_ref_sequence_name_idx(self.h)

    @property
    def sequence_idx(self):
        return lib().ora_ref_sequence_idx(self.h)

    @property
    def headers(self):
        return [lib().ora_ref_header(self.h, c).decode() for c in range(self.n_cols)]

    def column(self, c):
        return [lib().ora_ref_cell(self.h, c, r).decode() for r in range(self.n_rows)]

    def push_column(self, header, values):
        return lib().ora_ref_push_column(self.h, header.encode(), _strs(values), len(values))


def config_from_json(obj, n_rows, strand_filter="unstranded"):
    """The config half of reference_library::get_reference_library (reference_library.rs:27-126)."""
    def need(key, typ):
        if key not in obj or not isinstance(obj[key], typ) or (typ is int and isinstance(obj[key], bool)):
            raise OracleError("Error -- could not parse %s" % key)
        return obj[key]
    c = Config()
    c.score_percent = float(need("score_percent", (int, float)))
    c.score_filter = need("score_filter", int)
    c.score_threshold = need("score_threshold", int)
    c.num_mismatches = need("num_mismatches", int)
    c.discard_multiple_matches = int(need("discard_multiple_matches", bool))
    c.require_valid_pair = int(need("require_valid_pair", bool))
    c.discard_multi_hits = need("discard_multi_hits", int)
    lvl = need("intersect_level", int)
    if lvl not in (0, 1, 2):
        raise OracleError("Error -- invalid intersect level in config file. Please choose intersect level 0, 1, or 2.")
    c.intersect_level = lvl
    c.max_hits_to_report = need("max_hits_to_report", int)
    need("group_on", str)
    c.trim_target_length = need("trim_target_length", int)
    c.trim_strictness = float(need("trim_strictness", (int, float)))
    c.discard_nonzero_mismatch = 0  # reference_library.rs:116
    c.reference_genome_size = n_rows
    c.strand_filter = CHEM[strand_filter] if isinstance(strand_filter, str) else strand_filter
    if lib().ora_sanity_check_config(C.byref(c)) != 0:
        raise _err()
    return c


def get_reference_library(path, strand_filter="unstranded"):
    """reference_library::get_reference_library (src/reference_library.rs:20-174)."""
    with open(path) as f:
        v = json.load(f)
    cfg_obj, ref_obj = v[0], v[1]
    headers = ref_obj["headers"]
    columns = ref_obj["columns"]
    if not isinstance(headers, list) or not all(isinstance(h, str) for h in headers):
        raise OracleError("Error -- could not parse headers as array")
    for col in columns:
        if not isinstance(col, list) or not all(isinstance(x, str) for x in col):
            raise OracleError("Error -- could not parse column element as a string")
    name_idx = headers.index("sequence_name") if "sequence_name" in headers else None
    if name_idx is None:
        raise OracleError("Could not find header sequence_name")
    cfg = config_from_json(cfg_obj, len(columns[name_idx]), strand_filter)
    ref = Reference.from_columns(headers, columns, cfg_obj["group_on"])
    return cfg, ref


class Index:
    """align::PseudoAligner (src/align.rs:21)."""

    def __init__(self, handle):
        if not handle:
            raise _err()
        self.h = handle

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().ora_index_free(self.h)
            self.h = None

    @classmethod
    def from_reference(cls, ref):
        return cls(lib().ora_index_build_from_ref(ref.h))

    @classmethod
    def from_sequences(cls, seqs):
        return cls(lib().ora_index_build(len(seqs), _strs(seqs)))

    def stats(self):
        s = (C.c_uint64 * 5)()
        lib().ora_index_stats(self.h, s)
        return dict(kmers=s[0], nodes=s[1], classes=s[2], unitig_bases=s[3], class_entries=s[4])

    def node(self, i):
        buf = C.create_string_buffer(1 << 16)
        col, le, re = C.c_uint32(), C.c_uint32(), C.c_uint32()
        n = lib().ora_index_node(self.h, i, buf, len(buf), C.byref(col), C.byref(le), C.byref(re))
        if n < 0:
            raise IndexError(i)
        return buf.value.decode(), col.value, le.value, re.value

    def eq_class(self, colour):
        n = lib().ora_index_class(self.h, colour, None, 0)
        a = np.zeros(max(n, 1), dtype=np.uint32)
        lib().ora_index_class(self.h, colour, a.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return a[:n].tolist()

    def map_read(self, read, allowed):
        """map_read_with_mismatch -> None | (class, coverage, mismatches)"""
        if isinstance(read, str):
            read = read.encode()
        cap = 1 << 16
        cls = np.zeros(cap, dtype=np.uint32)
        n, sc, mm = C.c_int(), C.c_int(), C.c_int()
        ok = lib().ora_map_read(self.h, read, len(read), allowed, cls.ctypes.data_as(C.POINTER(C.c_uint32)), cap,
                                C.byref(n), C.byref(sc), C.byref(mm))
        if not ok:
            return None
        return cls[:n.value].tolist(), sc.value, mm.value

    def pseudoalign(self, read, cfg, min_read_length=40):
        """align::pseudoalign -> (AlignmentScore | None, Filter | None) as the reference returns them."""
        if isinstance(read, str):
            read = read.encode()
        cap = 1 << 16
        cls = np.zeros(cap, dtype=np.uint32)
        n, reason, sc, norm = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        ok = lib().ora_pseudoalign(self.h, C.byref(cfg), read, len(read), min_read_length,
                                   cls.ctypes.data_as(C.POINTER(C.c_uint32)), cap, C.byref(n), C.byref(reason),
                                   C.byref(norm), C.byref(sc))
        if ok:
            return (cls[:n.value].tolist(), norm.value, sc.value), None
        return None, (REASONS[reason.value], norm.value, sc.value)


def pack_reads(reads):
    """list of str/bytes -> (uint8 buffer, uint64 offsets[n+1])"""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    return buf, off


class CallResult:
    def __init__(self, rows, per_read, counters):
        self.rows = rows  # [(features list, count)], sorted as score::call returns them
        self.per_read = per_read
        self.counters = counters


def call(index, ref, cfg, r1, r1_off, r2=None, r2_off=None, n_threads=1, keep_per_read=False):
    """score::call (src/score.rs:14-46) over in-memory reads given as (uint8 buffer, uint64 offsets)."""
    r1 = np.ascontiguousarray(r1, dtype=np.uint8)
    r1_off = np.ascontiguousarray(r1_off, dtype=np.uint64)
    n = int(r1_off.size - 1)
    p2 = o2 = None
    if r2 is not None:
        r2 = np.ascontiguousarray(r2, dtype=np.uint8)
        r2_off = np.ascontiguousarray(r2_off, dtype=np.uint64)
        if r2_off.size != r1_off.size:
            raise OracleError("Error -- read and reverse read files do not have matching lengths: ")
        p2, o2 = r2.ctypes.data, r2_off.ctypes.data
    h = lib().ora_call(index.h, ref.h, C.byref(cfg), r1.ctypes.data, r1_off.ctypes.data, p2, o2, n, n_threads,
                       1 if keep_per_read else 0)
    if not h:
        raise _err()
    try:
        rows = []
        for i in range(lib().ora_result_n_rows(h)):
            cnt = C.c_int32()
            s = lib().ora_result_row(h, i, C.byref(cnt)).decode()
            rows.append((s.split("\t"), cnt.value))
        per_read = None
        if keep_per_read:
            def arr(ptr, dt):
                return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
            per_read = dict(
                reason=[arr(lib().ora_result_reason(h, m), np.int32) for m in (0, 1)],
                score=[arr(lib().ora_result_score(h, m), np.int32) for m in (0, 1)],
                mismatches=[arr(lib().ora_result_mismatch(h, m), np.int32) for m in (0, 1)],
                class_hash=[arr(lib().ora_result_class_hash(h, m), np.uint64) for m in (0, 1)],
                kept=[arr(lib().ora_result_kept(h, m), np.uint8) for m in (0, 1)],
                counted=arr(lib().ora_result_counted(h), np.uint8),
            )
        c = (C.c_uint64 * 8)()
        lib().ora_result_counters(h, c)
        counters = dict(reads=c[0], unique_keys=c[1], probes=c[2], nodes=c[3], class_entries=c[4], seeded=c[5],
                        prefiltered=c[6], filter_reason_keys=c[7])
        return CallResult(rows, per_read, counters)
    finally:
        lib().ora_result_free(h)


def call_umi(index, ref, cfg, r1, r1_off, r2=None, r2_off=None, q1=None, q2=None, skip1=None, skip2=None,
             segment=None, keep_per_read=False):
    """The BAM pipeline's score::call per UMI group (src/process/bam.rs:183-226,229-290): `segment` groups the
    reads, q1/q2 are quality strings laid out like the bases (quality trim, align.rs:866-871), skip1/skip2 the
    SKIP_ALIGN dummies.  Rows: (segment, features, count) sorted by (segment, callset)."""
    r1 = np.ascontiguousarray(r1, dtype=np.uint8)
    r1_off = np.ascontiguousarray(r1_off, dtype=np.uint64)
    n = int(r1_off.size - 1)

    def opt(a, dt):
        return None if a is None else np.ascontiguousarray(a, dtype=dt)

    r2, r2_off = opt(r2, np.uint8), opt(r2_off, np.uint64)
    q1, q2, skip1, skip2 = opt(q1, np.uint8), opt(q2, np.uint8), opt(skip1, np.uint8), opt(skip2, np.uint8)
    segment = opt(segment, np.uint32)
    ptr = lambda a: None if a is None else a.ctypes.data
    L = lib()
    L.ora_call_umi.restype = C.c_void_p
    L.ora_call_umi.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Config)] + [C.c_void_p] * 9 + [C.c_uint64, C.c_int]
    L.ora_result_row_segment.restype = C.c_uint32
    L.ora_result_row_segment.argtypes = [C.c_void_p, C.c_uint64]
    L.ora_result_align_len.restype = C.POINTER(C.c_int32)
    L.ora_result_align_len.argtypes = [C.c_void_p, C.c_int]
    h = L.ora_call_umi(index.h, ref.h, C.byref(cfg), ptr(r1), ptr(r1_off), ptr(r2), ptr(r2_off), ptr(q1), ptr(q2),
                       ptr(skip1), ptr(skip2), ptr(segment), n, 1 if keep_per_read else 0)
    if not h:
        raise _err()
    try:
        rows = []
        for i in range(L.ora_result_n_rows(h)):
            cnt = C.c_int32()
            s = L.ora_result_row(h, i, C.byref(cnt)).decode()
            rows.append((int(L.ora_result_row_segment(h, i)), s.split("\t"), cnt.value))
        per_read = None
        if keep_per_read:
            def arr(p, dt):
                return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
            per_read = dict(
                reason=[arr(L.ora_result_reason(h, m), np.int32) for m in (0, 1)],
                score=[arr(L.ora_result_score(h, m), np.int32) for m in (0, 1)],
                mismatches=[arr(L.ora_result_mismatch(h, m), np.int32) for m in (0, 1)],
                class_hash=[arr(L.ora_result_class_hash(h, m), np.uint64) for m in (0, 1)],
                kept=[arr(L.ora_result_kept(h, m), np.uint8) for m in (0, 1)],
                align_len=[arr(L.ora_result_align_len(h, m), np.int32) for m in (0, 1)],
                counted=arr(L.ora_result_counted(h), np.uint8),
            )
        c = (C.c_uint64 * 8)()
        L.ora_result_counters(h, c)
        counters = dict(reads=c[0], unique_keys=c[1], probes=c[2], nodes=c[3], class_entries=c[4], seeded=c[5],
                        prefiltered=c[6], filter_reason_keys=c[7])
        return CallResult(rows, per_read, counters)
    finally:
        L.ora_result_free(h)


def get_calls_fastq(index, ref, cfg, reads, mates=None, **kw):
    b1, o1 = pack_reads(reads)
    if mates is not None:
        b2, o2 = pack_reads(mates)
        return call(index, ref, cfg, b1, o1, b2, o2, **kw)
    return call(index, ref, cfg, b1, o1, **kw)


# ---- thin wrappers for the unit-level restatements ----
def filter_alignment_by_metrics(cls, score, normalized, score_threshold, score_percent, discard_multiple_matches,
                                mismatch_threshold, mismatches):
    r = lib().ora_filter_alignment_by_metrics(len(cls), score, normalized, score_threshold, score_percent,
                                              int(discard_multiple_matches), mismatch_threshold, mismatches)
    if r == R["SuccessfulMatch"]:
        return (list(cls), normalized, score), None
    return None, (REASONS[r], normalized, score)


def filter_pair(a, b):
    aa, pa, na = _u32(a)
    bb, pb, nb = _u32(b)
    return bool(lib().ora_filter_pair(pa, na, pb, nb))


def shannon_entropy(s):
    return lib().ora_shannon_entropy(s.encode())


def natural_lexical_cmp(a, b):
    return lib().ora_natural_lexical_cmp(a.encode(), b.encode())


def maxinfo(quality, target_length, strictness):
    q = quality.encode("latin-1") if isinstance(quality, str) else quality
    return lib().ora_maxinfo(q, len(q), target_length, strictness)


def revcomp(seq):
    out = C.create_string_buffer(len(seq.encode()) + 1)
    if lib().ora_revcomp(seq.encode(), out) != 0:
        raise _err()
    return out.value.decode()


def _lines(v):
    return "\n".join(v).encode()


def filter_read_calls_with_orientation(calls):
    out = C.create_string_buffer(1 << 16)
    n = lib().ora_filter_read_calls_with_orientation(_lines(calls), out, len(out))
    if n < 0:
        raise _err()
    s = out.value.decode()
    return s.split("\n") if s else []


def filter_orientation_on_library_chemistry(seq, mate, chem):
    a = C.create_string_buffer(1 << 16)
    b = C.create_string_buffer(1 << 16)
    if lib().ora_filter_orientation_on_library_chemistry(_lines(seq), _lines(mate), CHEM[chem], a, b, len(a)) != 0:
        raise _err()
    sa, sb = a.value.decode(), b.value.decode()
    return (sa.split("\n") if sa else []), (sb.split("\n") if sb else [])


def process_equivalence_class_to_feature_list(cls, ref, cfg, ignore_group_rollup):
    out = C.create_string_buffer(1 << 16)
    a, p, n = _u32(cls)
    r = lib().ora_process_class_to_features(ref.h, C.byref(cfg), p, n, int(ignore_group_rollup), out, len(out))
    if r < 0:
        raise _err()
    s = out.value.decode()
    return s.split("\n") if s else []


def parse_calls(calls):
    """AlignmentOrientation::parse_calls (align.rs:276-285) -> [(feature, is_rev)]"""
    out = C.create_string_buffer(1 << 16)
    if lib().ora_parse_calls(_lines(calls), out, len(out)) < 0:
        raise _err()
    s = out.value.decode()
    return [(l.rsplit("\t", 1)[0], l.rsplit("\t", 1)[1] == "1") for l in s.split("\n")] if s else []


def unmap(features, ref):
    """unmap (align.rs:851-864)"""
    out = (C.c_uint32 * max(1, len(features)))()
    n = lib().ora_unmap(ref.h, _lines(features), out, len(out))
    if n < 0:
        raise _err()
    return [int(out[i]) for i in range(n)]


def get_reference_sequence_data(ref):
    """utils::get_reference_sequence_data (utils.rs:7-24) -> ([DnaString::to_string()], [name])"""
    out = C.create_string_buffer(1 << 20)
    if lib().ora_reference_sequence_data(ref.h, out, len(out)) < 0:
        raise _err()
    s = out.value.decode()
    rows = [l.split("\t") for l in s.split("\n")] if s else []
    return [r[1] for r in rows], [r[0] for r in rows]


def sort_score_vector(scores):
    """utils::sort_score_vector (utils.rs:54-59); scores = [(key list, anything)]"""
    keys = "\n".join("\t".join(k) for k, _ in scores).encode()
    order = (C.c_int32 * max(1, len(scores)))()
    if lib().ora_sort_score_vector(keys, len(scores), order) != 0:
        raise _err()
    return [scores[order[i]] for i in range(len(scores))]


def coerce(ref, cfg, c1, c2):
    """filter_and_coerce_sequence_call_orientations on explicit classes (None = Option::None).
    Returns (callset list, triage reason name)."""
    out = C.create_string_buffer(1 << 18)
    a1, p1, n1 = _u32(c1 if c1 is not None else [])
    a2, p2, n2 = _u32(c2 if c2 is not None else [])
    r = lib().ora_coerce(ref.h, C.byref(cfg), int(c1 is not None), p1, n1, int(c2 is not None), p2, n2, out, len(out))
    if r < 0:
        raise _err()
    s = out.value.decode()
    return (s.split("\t") if s else []), REASONS[r]
