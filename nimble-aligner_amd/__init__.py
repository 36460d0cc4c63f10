"""nimble-aligner_amd -- MI355X-native hot path of nimble-aligner (ctypes binding of the C ABI).

The compute lives in ``lib/libnimble_hip.so`` (hand-written HIP for gfx950, see ``csrc/``) and, above
it, ``lib/libnimble_host.so`` (C++ mirror of the reference's ``reference_library`` / ``score::call`` /
FASTQ pipeline, see ``host/``).  This module only loads them and passes pointers; it contains no
compute and no fallback: if the library is missing, importing the device classes raises.

The directory name has a hyphen, so import it with
``importlib.import_module("nimble-aligner_amd")``.

When torch is used in the same process (device buffers, torch.distributed), import torch BEFORE the first call
into this module: torch bundles its own HIP runtime and must be the first one loaded.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NIMBLE_LIB_DIR selects another build of both libraries (A/B runs of whole builds: libv/<name>/)
LIB_DIR = os.environ.get("NIMBLE_LIB_DIR") or os.path.join(_HERE, "lib")
# NIMBLE_HIP_LIB selects another build of the same library (A/B runs of kernel variants)
HIP_LIB_PATH = os.environ.get("NIMBLE_HIP_LIB") or os.path.join(LIB_DIR, "libnimble_hip.so")
HOST_LIB_PATH = os.path.join(LIB_DIR, "libnimble_host.so")

CLASS_NONE = 0xFFFFFFFF
OPT_COUNTERS, OPT_ALIGN_GRID_PCT, OPT_TAIL_ASIDE = 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1

REASONS = {
    0: "ScoreBelowThreshold", 1: "DiscardedMultipleMatch", 2: "DiscardedNonzeroMismatch", 3: "NoMatch",
    6: "NotMatchingPair", 8: "ShortRead", 10: "HighEntropy", 11: "SuccessfulMatch", 14: "AboveMismatchThreshold",
    16: "None",
}


class NimbleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("nimble error %d: %s" % (code, msg))
        self.code = code


class AlignParams(C.Structure):
    """nimble_align_params (include/nimble_hip.h): the per-read part of AlignFilterConfig."""
    _fields_ = [
        ("score_percent", C.c_double),
        ("score_threshold", C.c_uint64),
        ("num_mismatches", C.c_uint32),
        ("discard_nonzero_mismatch", C.c_uint32),
        ("discard_multiple_matches", C.c_uint32),
        ("require_valid_pair", C.c_uint32),
        ("min_read_length", C.c_uint32),
        ("reserved", C.c_uint32),
    ]

    @classmethod
    def make(cls, score_percent, score_threshold, num_mismatches=0, discard_nonzero_mismatch=False,
             discard_multiple_matches=False, require_valid_pair=False, min_read_length=40):
        return cls(float(score_percent), int(score_threshold), int(num_mismatches), int(discard_nonzero_mismatch),
                   int(discard_multiple_matches), int(require_valid_pair), int(min_read_length), 0)


class NimblePacked(C.Structure):
    """nimble_packed (include/nimble_hip.h): caller-owned device arrays of the packed form of a read set."""
    _fields_ = [
        ("keys", C.c_void_p), ("len", C.c_void_p * 2), ("hash", C.c_void_p), ("pre", C.c_void_p * 2),
        ("key_words", C.c_uint32), ("paired", C.c_uint32),
    ]


class CallExtra(C.Structure):
    """struct nimble_call_extra (include/nimble_hip.h): what the BAM pipeline adds to a call."""
    _fields_ = [("segment", C.c_void_p), ("n_segments", C.c_uint32), ("reserved", C.c_uint32),
                ("qual", C.c_void_p * 2), ("trim_strictness", C.c_double), ("trim_target_length", C.c_uint64),
                ("skip", C.c_void_p * 2)]


class UmiExtra(C.Structure):
    """struct nimble_umi_extra (include/nimble_host.h)."""
    _fields_ = [("segment", C.c_void_p), ("n_segments", C.c_uint32), ("reserved", C.c_uint32),
                ("qual", C.c_void_p * 2), ("skip", C.c_void_p * 2)]


_hip = None

HIP_SYMBOLS = [
    "nimble_abi_version", "nimble_last_error", "nimble_device_count", "nimble_index_build", "nimble_index_free",
    "nimble_index_stats", "nimble_class_get", "nimble_ctx_create", "nimble_ctx_free", "nimble_ctx_synchronize",
    "nimble_call", "nimble_histogram", "nimble_histogram_dense_se", "nimble_read_records", "nimble_call_counters",
    "nimble_call_timing", "nimble_flat_index_stats", "nimble_flat_index_selfcheck", "nimble_ctx_set_option", "nimble_key_words", "nimble_pack",
    "nimble_call_packed", "nimble_call_words", "nimble_ctx_stream", "nimble_stream_begin", "nimble_stream_append", "nimble_stream_append_packed", "nimble_stream_end",
    "nimble_pinned_alloc", "nimble_pinned_free", "nimble_call_ex", "nimble_histogram_seg", "nimble_read_align_len",
    "nimble_route_records", "nimble_unpack_records", "nimble_pinned_register", "nimble_pinned_unregister",
    "nimble_ctx_defer_dedup", "nimble_route_counts", "nimble_dedup_records", "nimble_count_verdicts",
    "nimble_call_records", "nimble_comm_create", "nimble_comm_free", "nimble_comm_size", "nimble_comm_uses_rccl", "nimble_comm_abort",
    "nimble_counts_allreduce", "nimble_counts_allreduce_host", "nimble_records_alltoall", "nimble_sharded_begin",
    "nimble_sharded_append", "nimble_sharded_append_packed", "nimble_sharded_end", "nimble_sharded_grow", "nimble_sharded_abort",
    "nimble_class_table_read", "nimble_class_pool_read", "nimble_steps_begin", "nimble_steps_submit", "nimble_steps_flush", "nimble_steps_end",
]


def hip_lib():
    """Load lib/libnimble_hip.so.  Raises if it has not been built -- there is no other backend."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise ImportError("nimble-aligner_amd: %s is missing; run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (needs hipcc)" % HIP_LIB_PATH)
        L = C.CDLL(HIP_LIB_PATH)
        vp, i32, u32, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
        L.nimble_abi_version.restype = i32
        L.nimble_last_error.restype = C.c_char_p
        L.nimble_device_count.argtypes = [C.POINTER(i32)]
        L.nimble_index_build.argtypes = [vp, vp, u32, i32, C.POINTER(vp)]
        L.nimble_index_free.argtypes = [vp]
        L.nimble_index_free.restype = None
        L.nimble_flat_index_stats.argtypes = [vp, vp, u32, C.POINTER(u64)]
        L.nimble_flat_index_selfcheck.argtypes = [vp, vp, u32, C.POINTER(u64)]
        L.nimble_index_stats.argtypes = [vp, C.POINTER(u64)]
        L.nimble_class_get.argtypes = [vp, u32, vp, u32, C.POINTER(u32)]
        L.nimble_ctx_create.argtypes = [vp, vp, C.POINTER(vp)]
        L.nimble_ctx_free.argtypes = [vp]
        L.nimble_ctx_free.restype = None
        L.nimble_ctx_synchronize.argtypes = [vp]
        L.nimble_ctx_set_option.argtypes = [vp, i32, C.c_int64]
        L.nimble_call.argtypes = [vp, C.POINTER(AlignParams), vp, vp, vp, vp, u64, u32, u32, i32]
        L.nimble_histogram.argtypes = [vp, vp, vp, vp, u64, C.POINTER(u64)]
        L.nimble_histogram_dense_se.argtypes = [vp, vp, u32]
        L.nimble_read_records.argtypes = [vp, i32, vp, vp, vp, vp, vp, u64]
        L.nimble_call_counters.argtypes = [vp, C.POINTER(u64)]
        L.nimble_call_timing.argtypes = [vp, C.POINTER(C.c_float)]
        L.nimble_key_words.argtypes = [u32, i32]
        L.nimble_key_words.restype = u32
        L.nimble_pack.argtypes = [vp, C.POINTER(AlignParams), vp, vp, vp, vp, u64, u32, u32, i32, C.POINTER(NimblePacked)]
        L.nimble_call_packed.argtypes = [vp, C.POINTER(AlignParams), C.POINTER(NimblePacked), u64, u32]
        L.nimble_stream_begin.argtypes = [vp, C.POINTER(AlignParams), i32, u32, u64]
        L.nimble_stream_append.argtypes = [vp, vp, vp, vp, vp, u64, u32, i32]
        L.nimble_stream_append_packed.argtypes = [vp, vp, vp, u32, vp, vp, u32, u64]
        L.nimble_call_words.argtypes = [vp, vp, vp, vp, u32, vp, vp, u32, u64, u32, i32]
        L.nimble_stream_end.argtypes = [vp]
        L.nimble_pinned_alloc.argtypes = [u64, C.POINTER(vp)]
        L.nimble_pinned_free.argtypes = [vp]
        L.nimble_pinned_free.restype = None
        L.nimble_pinned_register.argtypes = [vp, u64]
        L.nimble_pinned_unregister.argtypes = [vp]
        L.nimble_pinned_unregister.restype = None
        L.nimble_call_ex.argtypes = [vp, C.POINTER(AlignParams), vp, vp, vp, vp, u64, u32, u32, i32, C.POINTER(CallExtra)]
        L.nimble_histogram_seg.argtypes = [vp, vp, vp, vp, vp, vp, u64, C.POINTER(u64)]
        L.nimble_read_align_len.argtypes = [vp, i32, vp, u64]
        L.nimble_route_records.argtypes = [vp, C.POINTER(NimblePacked), u64, u32, vp, vp]
        L.nimble_unpack_records.argtypes = [vp, vp, u64, C.POINTER(NimblePacked)]
        L.nimble_ctx_defer_dedup.argtypes = [vp, u32, vp, vp]
        L.nimble_route_counts.argtypes = [vp, vp]
        L.nimble_dedup_records.argtypes = [vp, vp, u64, u32, vp]
        L.nimble_count_verdicts.argtypes = [vp, vp]
        L.nimble_call_records.argtypes = [vp, C.POINTER(AlignParams), vp, u64, u32, i32]
        L.nimble_ctx_stream.argtypes = [vp]
        L.nimble_ctx_stream.restype = vp
        L.nimble_comm_create.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
        L.nimble_comm_free.argtypes = [vp]
        L.nimble_comm_free.restype = None
        L.nimble_comm_size.argtypes = [vp]
        L.nimble_comm_uses_rccl.argtypes = [vp]
        L.nimble_comm_abort.argtypes = [vp]
        L.nimble_comm_abort.restype = None
        L.nimble_counts_allreduce.argtypes = [vp, i32, vp, u64, vp]
        L.nimble_counts_allreduce_host.argtypes = [vp, i32, vp, u64]
        L.nimble_records_alltoall.argtypes = [vp, i32, vp, vp, u32, vp, u64, C.POINTER(u64), vp]
        L.nimble_sharded_begin.argtypes = [vp, i32, vp, C.POINTER(AlignParams), i32, u32]
        L.nimble_sharded_append.argtypes = [vp, i32, vp, vp, vp, vp, u64, u32, i32]
        L.nimble_sharded_append_packed.argtypes = [vp, i32, vp, vp, u32, vp, vp, u32, u64]
        L.nimble_sharded_end.argtypes = [vp, i32, C.POINTER(u64)]
        L.nimble_sharded_grow.argtypes = [vp, i32, u32]
        L.nimble_steps_begin.argtypes = [vp, i32, vp, vp, vp, C.POINTER(AlignParams), i32, u32]
        L.nimble_steps_submit.argtypes = [vp, i32, vp, vp, vp, vp, u64, u32, i32, C.POINTER(vp)]
        L.nimble_steps_flush.argtypes = [vp, i32, C.POINTER(vp)]
        L.nimble_steps_end.argtypes = [vp, i32]
        L.nimble_sharded_abort.argtypes = [vp, i32]
        _hip = L
    return _hip


def _check(rc):
    if rc != 0:
        raise NimbleError(rc, hip_lib().nimble_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    rc = hip_lib().nimble_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def pack_reads(reads):
    """list of str/bytes -> (uint8 buffer, uint64 offsets[n+1])"""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    return buf, off


def _ptr(x):
    """pointer of a numpy array / torch tensor / int / None"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    raise TypeError(type(x))


def _off(x):
    """pointer of a read-offsets argument: 8-byte integers or None.  (Two byte tensors handed over positionally -- mates in
    the place of the offsets -- once went all the way to the device, which read bases as offsets.)"""
    if x is None or isinstance(x, int):
        return x
    size = x.dtype.itemsize if isinstance(x, np.ndarray) else (x.element_size() if hasattr(x, "element_size") else 8)
    if size != 8:
        raise TypeError("read offsets must be 64-bit integers (n + 1 of them); got elements of %d bytes -- are the mates in "
                        "the place of the offsets?" % size)
    return _ptr(x)


def key_words(max_len, paired):
    """u64 words of one packed read key (nimble_key_words)."""
    return int(hip_lib().nimble_key_words(int(max_len), int(bool(paired))))


def flat_index_stats(sequences):
    """Host-only: statistics of the flat index the builder would upload (no GPU needed)."""
    buf, off = pack_reads(sequences)
    s = (C.c_uint64 * 5)()
    _check(hip_lib().nimble_flat_index_stats(buf.ctypes.data, off.ctypes.data, len(sequences), s))
    return dict(kmers=s[0], nodes=s[1], classes=s[2], unitig_bases=s[3], class_entries=s[4])


def flat_index_selfcheck(sequences):
    """Host-only: builds the flat index and checks its stretch records against the unitig records; returns how many records
    there are (0: the index takes the general walk).  Raises NimbleError with what is wrong."""
    buf, off = pack_reads(sequences)
    n = C.c_uint64()
    _check(hip_lib().nimble_flat_index_selfcheck(buf.ctypes.data, off.ctypes.data, len(sequences), C.byref(n)))
    return int(n.value)


class Index:
    """Device-resident pseudoalignment index: replaces align::PseudoAligner (src/align.rs:21), built as
    debruijn_mapping::build_index::<Kmer30> is at src/bin/main.rs:121-128."""

    def __init__(self, sequences, device=0):
        buf, off = pack_reads(sequences)
        h = C.c_void_p()
        _check(hip_lib().nimble_index_build(buf.ctypes.data, off.ctypes.data, len(sequences), device, C.byref(h)))
        self.h = h
        self.device = device
        self._class_cache = {}

    def close(self):
        if getattr(self, "h", None):
            hip_lib().nimble_index_free(self.h)
            self.h = None

    def __del__(self):
        try:  # at interpreter shutdown the module globals may be gone already
            self.close()
        except Exception:
            pass

    def stats(self):
        s = (C.c_uint64 * 8)()
        _check(hip_lib().nimble_index_stats(self.h, s))
        return dict(kmers=s[0], nodes=s[1], classes=s[2], unitig_bases=s[3], class_entries=s[4], hash_slots=s[5],
                    device_bytes=s[6], dynamic_classes=s[7])

    def eq_class(self, class_id):
        class_id = int(class_id)
        got = self._class_cache.get(class_id)
        if got is None:
            n = C.c_uint32(0)
            _check(hip_lib().nimble_class_get(self.h, class_id, None, 0, C.byref(n)))
            a = np.zeros(max(n.value, 1), dtype=np.uint32)
            _check(hip_lib().nimble_class_get(self.h, class_id, a.ctypes.data, n.value, C.byref(n)))
            got = a[:n.value].tolist()
            self._class_cache[class_id] = got
        return got


class Context:
    """One `score::call` in flight (src/score.rs:14-46): workspace + stream over a shared Index."""

    def __init__(self, index=None, stream=None, borrowed=None):
        self.index = index
        self.n = 0
        self._keep = None
        self._owned = borrowed is None
        if borrowed is not None:  # a context owned by a Library (nimble_library_ctx)
            self.h = C.c_void_p(borrowed)
            return
        h = C.c_void_p()
        _check(hip_lib().nimble_ctx_create(index.h, stream, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and self._owned:
            hip_lib().nimble_ctx_free(self.h)
        self.h = None

    def __del__(self):
        try:  # at interpreter shutdown the module globals may be gone already
            self.close()
        except Exception:
            pass

    def call(self, params, r1, r1_off=None, r2=None, r2_off=None, n=None, fixed_len=0, max_len=0, mem=MEM_HOST):
        """nimble_call.  r1/r2: numpy uint8 arrays (host) or torch uint8 tensors / raw pointers (device)."""
        if r1_off is not None:
            n = int(len(r1_off) - 1) if n is None else n
        if n is None:
            raise ValueError("n is required for fixed-length input")
        if max_len == 0:
            if r1_off is not None and isinstance(r1_off, np.ndarray):
                max_len = int(np.diff(r1_off.astype(np.int64)).max()) if n else 1
                if r2_off is not None:
                    max_len = max(max_len, int(np.diff(r2_off.astype(np.int64)).max()) if n else 1)
            else:
                max_len = fixed_len
        self._keep = (r1, r1_off, r2, r2_off)  # keep device inputs alive until the next call
        _check(hip_lib().nimble_call(self.h, C.byref(params), _ptr(r1), _off(r1_off), _ptr(r2), _off(r2_off), n,
                                     fixed_len, max_len, mem))
        self.n = n

    def call_ex(self, params, r1, r1_off=None, r2=None, r2_off=None, n=None, fixed_len=0, max_len=0, mem=MEM_HOST,
                segment=None, n_segments=0, qual=(None, None), skip=(None, None), trim_strictness=0.0,
                trim_target_length=0):
        """nimble_call_ex: the call as the BAM pipeline makes it (UMI segments, quality trim, SKIP_ALIGN)."""
        if r1_off is not None:
            n = int(len(r1_off) - 1) if n is None else n
        if max_len == 0:
            if r1_off is not None and isinstance(r1_off, np.ndarray):
                max_len = int(np.diff(r1_off.astype(np.int64)).max()) if n else 1
                if r2_off is not None:
                    max_len = max(max_len, int(np.diff(r2_off.astype(np.int64)).max()) if n else 1)
            else:
                max_len = fixed_len
        ex = CallExtra()
        ex.segment = _ptr(segment)
        ex.n_segments = n_segments
        for m in range(2):
            ex.qual[m] = _ptr(qual[m])
            ex.skip[m] = _ptr(skip[m])
        ex.trim_strictness = trim_strictness
        ex.trim_target_length = trim_target_length
        self._keep = (r1, r1_off, r2, r2_off, segment, qual, skip)
        _check(hip_lib().nimble_call_ex(self.h, C.byref(params), _ptr(r1), _off(r1_off), _ptr(r2), _off(r2_off), n,
                                        fixed_len, max_len, mem, C.byref(ex)))
        self.n = n

    def histogram_seg(self):
        """[(segment, class R1, class R2, count, representative read)] sorted by (segment, c1, c2)."""
        ne = C.c_uint64()
        _check(hip_lib().nimble_histogram_seg(self.h, None, None, None, None, None, 0, C.byref(ne)))
        k = ne.value
        sg, c1, c2 = (np.zeros(max(k, 1), dtype=np.uint32) for _ in range(3))
        cnt = np.zeros(max(k, 1), dtype=np.uint64)
        rep = np.zeros(max(k, 1), dtype=np.uint32)
        if k:
            _check(hip_lib().nimble_histogram_seg(self.h, sg.ctypes.data, c1.ctypes.data, c2.ctypes.data,
                                                  cnt.ctypes.data, rep.ctypes.data, k, C.byref(ne)))
        return [(int(sg[i]), int(c1[i]), int(c2[i]), int(cnt[i]), int(rep[i])) for i in range(k)]

    def read_align_len(self, mate=0):
        out = np.zeros(max(self.n, 1), dtype=np.uint32)
        _check(hip_lib().nimble_read_align_len(self.h, mate, out.ctypes.data, self.n))
        return out[:self.n]

    def stream_begin(self, params, paired, max_len, capacity_hint=0):
        """nimble_stream_begin: open one call whose reads arrive in batches."""
        _check(hip_lib().nimble_stream_begin(self.h, C.byref(params), int(bool(paired)), max_len, capacity_hint))
        self.n = 0

    def stream_append(self, r1, r1_off=None, r2=None, r2_off=None, n=None, fixed_len=0, mem=MEM_HOST):
        if r1_off is not None and n is None:
            n = int(len(r1_off) - 1)
        self._keep = (r1, r1_off, r2, r2_off)
        _check(hip_lib().nimble_stream_append(self.h, _ptr(r1), _off(r1_off), _ptr(r2), _off(r2_off), n, fixed_len,
                                              mem))
        self.n += n

    def call_words(self, params, w1, len1, stride1, w2=None, len2=None, stride2=0, n=None, max_len=0, mem=MEM_HOST):
        """nimble_call_words: the call on reads that are already packed."""
        if n is None:
            n = int(len(len1))
        self._keep = (w1, len1, w2, len2)
        _check(hip_lib().nimble_call_words(self.h, C.byref(params), _ptr(w1), _ptr(len1), stride1, _ptr(w2), _ptr(len2),
                                           stride2, n, max(max_len, 1), mem))
        self.n = n

    def stream_append_packed(self, w1, len1, stride1, w2=None, len2=None, stride2=0):
        """nimble_stream_append_packed: a batch the host has packed (pack_reads_2bit).  The arrays are kept alive here until
        the stream ends (the copy runs behind the call)."""
        n = int(len(len1))
        self._keep_packed = getattr(self, "_keep_packed", []) + [(w1, len1, w2, len2)]
        _check(hip_lib().nimble_stream_append_packed(self.h, _ptr(w1), _ptr(len1), stride1, _ptr(w2), _ptr(len2), stride2, n))
        self.n += n

    def stream_end(self):
        _check(hip_lib().nimble_stream_end(self.h))
        self._keep_packed = []

    def call_reads(self, params, reads, mates=None):
        b1, o1 = pack_reads(reads)
        if mates is not None:
            b2, o2 = pack_reads(mates)
            return self.call(params, b1, o1, b2, o2)
        return self.call(params, b1, o1)

    def synchronize(self):
        _check(hip_lib().nimble_ctx_synchronize(self.h))

    def set_option(self, option, value):
        _check(hip_lib().nimble_ctx_set_option(self.h, option, int(value)))

    def stream_ptr(self):
        """The HIP stream the context launches on (an integer; torch.cuda.ExternalStream wraps it)."""
        return int(hip_lib().nimble_ctx_stream(self.h) or 0)

    # -- align-where-the-reads-are form of the multi-GPU step (include/nimble_hip.h) --
    def defer_dedup(self, world, records, perm):
        """Arm the next call: route its keys into `records` [n, key_words + 2] int64 / `perm` [n] int32 (device
        tensors) and stop before the dedup; world = 0 disarms."""
        if not world:
            _check(hip_lib().nimble_ctx_defer_dedup(self.h, 0, None, None))
            return
        assert records.is_contiguous() and perm.is_contiguous() and perm.element_size() == 4
        self._defer_keep = (records, perm)
        _check(hip_lib().nimble_ctx_defer_dedup(self.h, world, C.c_void_p(records.data_ptr()),
                                                C.c_void_p(perm.data_ptr())))

    def route_counts(self, world):
        """Records per destination rank of the last routing on this context (a deferred call's, or route(wait=False));
        waits for that routing only."""
        counts = np.zeros(256, dtype=np.uint64)
        _check(hip_lib().nimble_route_counts(self.h, counts.ctypes.data))
        return [int(c) for c in counts[:world]]

    def dedup_records(self, records, key_words, verdict):
        """Owner side: verdict[j] = 1 for one copy of every key among `records` (device tensors; asynchronous)."""
        n = int(records.shape[0])
        assert records.is_contiguous() and verdict.is_contiguous() and verdict.numel() >= n
        assert n == 0 or records.shape[1] == key_words + 2
        self._dedup_keep = (records, verdict)
        _check(hip_lib().nimble_dedup_records(self.h, C.c_void_p(records.data_ptr()), n, key_words,
                                              C.c_void_p(verdict.data_ptr())))

    def count_verdicts(self, verdict):
        """Close the deferred call with the owners' verdicts (uint8 device tensor in this rank's record order)."""
        self._verdict_keep = verdict
        _check(hip_lib().nimble_count_verdicts(self.h, C.c_void_p(verdict.data_ptr())))

    def set_counters(self, on):
        _check(hip_lib().nimble_ctx_set_option(self.h, 1, int(bool(on))))

    def histogram(self):
        """[(class_r1, class_r2, count)] sorted by class ids; CLASS_NONE marks an absent mate call."""
        ne = C.c_uint64(0)
        _check(hip_lib().nimble_histogram(self.h, None, None, None, 0, C.byref(ne)))
        k = ne.value
        c1 = np.zeros(max(k, 1), dtype=np.uint32)
        c2 = np.zeros(max(k, 1), dtype=np.uint32)
        cnt = np.zeros(max(k, 1), dtype=np.uint64)
        if k:
            _check(hip_lib().nimble_histogram(self.h, c1.ctypes.data, c2.ctypes.data, cnt.ctypes.data, k,
                                              C.byref(ne)))
        return [(int(c1[i]), int(c2[i]), int(cnt[i])) for i in range(k)]

    def histogram_dense_se(self, counts_dev_ptr, n_classes):
        _check(hip_lib().nimble_histogram_dense_se(self.h, _ptr(counts_dev_ptr), n_classes))

    def read_records(self, mate=0):
        n = self.n
        reason = np.zeros(max(n, 1), dtype=np.int32)
        score = np.zeros(max(n, 1), dtype=np.int32)
        mism = np.zeros(max(n, 1), dtype=np.int32)
        cls = np.zeros(max(n, 1), dtype=np.uint32)
        counted = np.zeros(max(n, 1), dtype=np.uint8)
        _check(hip_lib().nimble_read_records(self.h, mate, reason.ctypes.data, score.ctypes.data, mism.ctypes.data,
                                             cls.ctypes.data, counted.ctypes.data, n))
        return dict(reason=reason[:n], score=score[:n], mismatches=mism[:n], cls=cls[:n], counted=counted[:n])

    def counters(self):
        c = (C.c_uint64 * 8)()
        _check(hip_lib().nimble_call_counters(self.h, c))
        return dict(reads=c[0], unique_keys=c[1], probes=c[2], nodes=c[3], class_entries=c[4], seeded=c[5],
                    prefiltered=c[6], dynamic_classes=c[7])

    def timing(self):
        """Device ms of the stages of the last call (HIP events on the launch stream)."""
        t = (C.c_float * 6)()
        _check(hip_lib().nimble_call_timing(self.h, t))
        return dict(pack=t[0], align=t[1], intern=t[2], dedup=t[3], count=t[4], total=t[5])


# ------------------------------------------------------------------------------------------------
# Host mirror (lib/libnimble_host.so): reference_library / score::call / FASTQ pipeline in C++
# ------------------------------------------------------------------------------------------------
CHEM = {"unstranded": 0, "fiveprime": 1, "threeprime": 2, "none": 3}

HOST_SYMBOLS = [
    "nimble_host_last_error", "nimble_library_load", "nimble_library_parse", "nimble_library_free",
    "nimble_library_get_config", "nimble_library_set_config", "nimble_library_n_rows", "nimble_library_n_cols",
    "nimble_library_group_on", "nimble_library_set_group_on", "nimble_library_sequence_name_idx",
    "nimble_library_sequence_idx", "nimble_library_header", "nimble_library_cell", "nimble_library_push_column", "nimble_library_from_table",
    "nimble_library_build_index", "nimble_library_index", "nimble_library_ctx", "nimble_score_call",
    "nimble_score_call_fastq", "nimble_rows_free", "nimble_rows_count", "nimble_rows_get", "nimble_fastq_process",
    "nimble_write_to_tsv", "nimble_multi_steps", "nimble_host_coerce", "nimble_host_parse_calls", "nimble_host_unmap", "nimble_host_feature_list",
    "nimble_host_reference_sequence_data", "nimble_host_sort_score_vector", "nimble_host_natural_lexical_cmp", "nimble_host_shannon_entropy",
    "nimble_host_revcomp", "nimble_host_maxinfo", "nimble_host_read_fastq", "nimble_host_filter_reason_text",
    "nimble_library_pack", "nimble_score_call_packed", "nimble_score_call_begin", "nimble_score_call_begin_words", "nimble_score_call_end",
    "nimble_library_ctx_slot", "nimble_library_pack_slot", "nimble_score_call_packed_begin",
    "nimble_score_call_records_begin", "nimble_rows_signature", "nimble_rows_counts", "nimble_host_read_fastq_batched", "nimble_host_read_fastq_packed", "nimble_score_call_umis", "nimble_umi_rows_free",
    "nimble_umi_rows_count", "nimble_umi_rows_get", "nimble_umi_rows_reads", "nimble_umi_rows_filter", "nimble_score_stream_begin", "nimble_score_stream_append", "nimble_score_stream_end",
    "nimble_fastq_process_sharded", "nimble_bam_process", "nimble_host_bam_dump", "nimble_host_pack_reads_2bit", "nimble_host_reverse_comp_if_needed",
    "nimble_host_parse_str_as_bool", "nimble_host_pgzip_decompress",
]


class HostConfig(C.Structure):
    """nimble_host_config == AlignFilterConfig (src/align.rs:79-95)."""
    _fields_ = [
        ("reference_genome_size", C.c_uint64), ("score_percent", C.c_double), ("score_threshold", C.c_uint64),
        ("num_mismatches", C.c_uint64), ("discard_nonzero_mismatch", C.c_int32),
        ("discard_multiple_matches", C.c_int32), ("score_filter", C.c_int32), ("intersect_level", C.c_int32),
        ("require_valid_pair", C.c_int32), ("strand_filter", C.c_int32), ("discard_multi_hits", C.c_uint64),
        ("max_hits_to_report", C.c_uint64), ("trim_strictness", C.c_double), ("trim_target_length", C.c_uint64),
    ]


_host = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError("nimble-aligner_amd: %s is missing; run __graft_entry__.build()" % HOST_LIB_PATH)
        hip_lib()  # dependency (also resolved through the rpath)
        L = C.CDLL(HOST_LIB_PATH)
        vp, i32, u32, u64, cp = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_char_p
        pp = C.POINTER(C.c_char_p)
        L.nimble_host_last_error.restype = cp
        L.nimble_library_load.argtypes = [cp, i32, C.POINTER(vp)]
        L.nimble_library_parse.argtypes = [cp, i32, C.POINTER(vp)]
        L.nimble_library_free.argtypes = [vp]
        L.nimble_library_free.restype = None
        L.nimble_library_get_config.argtypes = [vp, C.POINTER(HostConfig)]
        L.nimble_library_set_config.argtypes = [vp, C.POINTER(HostConfig)]
        for f in ("n_rows", "n_cols", "group_on", "sequence_name_idx", "sequence_idx"):
            getattr(L, "nimble_library_" + f).argtypes = [vp]
        L.nimble_library_set_group_on.argtypes = [vp, i32]
        L.nimble_library_set_group_on.restype = None
        L.nimble_library_header.argtypes = [vp, i32]
        L.nimble_library_header.restype = cp
        L.nimble_library_cell.argtypes = [vp, i32, i32]
        L.nimble_library_cell.restype = cp
        L.nimble_library_push_column.argtypes = [vp, cp, pp, i32]
        L.nimble_library_from_table.argtypes = [i32, pp, vp, pp, i32, i32, i32, C.POINTER(vp)]
        L.nimble_library_build_index.argtypes = [vp, i32]
        L.nimble_library_index.argtypes = [vp]
        L.nimble_library_index.restype = vp
        L.nimble_library_ctx.argtypes = [vp]
        L.nimble_library_ctx.restype = vp
        L.nimble_score_call.argtypes = [vp, vp, vp, vp, vp, u64, u32, u32, i32, C.POINTER(vp)]
        L.nimble_score_call_fastq.argtypes = [vp, cp, cp, C.POINTER(vp)]
        L.nimble_score_call_begin.argtypes = [vp, i32, vp, vp, vp, vp, u64, u32, u32, i32]
        L.nimble_score_call_begin_words.argtypes = [vp, i32, vp, vp, u32, vp, vp, u32, u64, u32, i32]
        L.nimble_score_call_end.argtypes = [vp, i32, C.POINTER(vp)]
        L.nimble_library_ctx_slot.argtypes = [vp, i32]
        L.nimble_score_stream_begin.argtypes = [vp, i32, u32, u64]
        L.nimble_library_pack_slot.argtypes = [vp, i32, vp, vp, vp, vp, u64, u32, u32, i32, C.POINTER(NimblePacked)]
        L.nimble_score_call_packed_begin.argtypes = [vp, i32, C.POINTER(NimblePacked), u64, u32]
        L.nimble_score_call_records_begin.argtypes = [vp, i32, vp, u64, u32, i32]
        L.nimble_rows_signature.argtypes = [vp]
        L.nimble_rows_signature.restype = u64
        L.nimble_rows_counts.argtypes = [vp, vp]
        L.nimble_rows_counts.restype = None
        L.nimble_score_call_umis.argtypes = [vp, vp, vp, vp, vp, u64, u32, u32, i32, C.POINTER(UmiExtra), i32,
                                             C.POINTER(vp)]
        L.nimble_umi_rows_free.argtypes = [vp]
        L.nimble_umi_rows_free.restype = None
        L.nimble_umi_rows_count.argtypes = [vp]
        L.nimble_umi_rows_count.restype = u64
        L.nimble_umi_rows_get.argtypes = [vp, u64, C.POINTER(u32), C.POINTER(i32), C.POINTER(u32)]
        L.nimble_umi_rows_get.restype = cp
        L.nimble_umi_rows_reads.argtypes = [vp]
        L.nimble_umi_rows_reads.restype = u64
        L.nimble_umi_rows_filter.argtypes = [vp, u64, C.POINTER(i32 * 5)]
        L.nimble_host_read_fastq_batched.argtypes = [cp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u32),
                                                     C.POINTER(u64), C.POINTER(u64)]
        L.nimble_host_read_fastq_packed.argtypes = [cp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u32),
                                                     C.POINTER(u64), C.POINTER(u64)]
        L.nimble_score_stream_append.argtypes = [vp, vp, vp, vp, vp, u64, u32, i32]
        L.nimble_score_stream_end.argtypes = [vp, C.POINTER(vp)]
        L.nimble_library_ctx_slot.restype = vp
        L.nimble_library_pack.argtypes = [vp, vp, vp, vp, vp, u64, u32, u32, i32, C.POINTER(NimblePacked)]
        L.nimble_score_call_packed.argtypes = [vp, C.POINTER(NimblePacked), u64, u32, C.POINTER(vp)]
        L.nimble_rows_free.argtypes = [vp]
        L.nimble_rows_free.restype = None
        L.nimble_rows_count.argtypes = [vp]
        L.nimble_rows_count.restype = u64
        L.nimble_rows_get.argtypes = [vp, u64, C.POINTER(C.c_int32)]
        L.nimble_rows_get.restype = cp
        L.nimble_fastq_process.argtypes = [i32, pp, i32, C.POINTER(vp), pp]
        L.nimble_fastq_process_sharded.argtypes = [i32, pp, vp, C.POINTER(i32), i32, cp]
        L.nimble_multi_steps.argtypes = [C.POINTER(vp), C.POINTER(i32), i32, C.POINTER(vp), C.POINTER(vp), i32, C.c_uint64,
                                         C.c_uint32, i32, i32, i32, C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(vp)]
        L.nimble_bam_process.argtypes = [cp, i32, C.POINTER(vp), pp, i32, i32]
        L.nimble_host_bam_dump.argtypes = [cp, i32, cp]
        L.nimble_host_reverse_comp_if_needed.argtypes = [cp, i32, cp, u64]
        L.nimble_host_pack_reads_2bit.argtypes = [vp, vp, u64, u32, vp, vp]
        L.nimble_host_parse_str_as_bool.argtypes = [cp, C.POINTER(i32)]
        L.nimble_host_pgzip_decompress.argtypes = [cp, i32, cp, C.POINTER(u64)]
        L.nimble_write_to_tsv.argtypes = [vp, cp]
        L.nimble_host_coerce.argtypes = [vp, i32, vp, i32, i32, vp, i32, cp, i32]
        L.nimble_host_parse_calls.argtypes = [vp, cp, cp, i32]
        L.nimble_host_unmap.argtypes = [vp, cp, vp, i32]
        L.nimble_host_feature_list.argtypes = [vp, vp, i32, i32, cp, i32]
        L.nimble_host_reference_sequence_data.argtypes = [vp, cp, i32]
        L.nimble_host_sort_score_vector.argtypes = [cp, i32, vp]
        L.nimble_host_natural_lexical_cmp.argtypes = [cp, cp]
        L.nimble_host_shannon_entropy.argtypes = [cp]
        L.nimble_host_shannon_entropy.restype = C.c_double
        L.nimble_host_revcomp.argtypes = [cp, cp]
        L.nimble_host_maxinfo.argtypes = [cp, i32, u64, C.c_double]
        L.nimble_host_maxinfo.restype = u64
        L.nimble_host_read_fastq.argtypes = [cp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u32)]
        L.nimble_host_filter_reason_text.argtypes = [i32]
        L.nimble_host_filter_reason_text.restype = cp
        _host = L
    return _host


class Panic(RuntimeError):
    """A Rust panic of the reference, surfaced with its message."""


def _hcheck(rc):
    if rc != 0:
        raise Panic(host_lib().nimble_host_last_error().decode("utf-8", "replace"))


def _cstrs(seq):
    arr = (C.c_char_p * len(seq))()
    arr[:] = [s.encode("utf-8") if isinstance(s, str) else s for s in seq]
    return arr


def _rows(handle):
    L = host_lib()
    try:
        out = []
        for i in range(L.nimble_rows_count(handle)):
            cnt = C.c_int32()
            s = L.nimble_rows_get(handle, i, C.byref(cnt)).decode()
            out.append((s.split("\t"), cnt.value))
        return out
    finally:
        L.nimble_rows_free(handle)


class PackedTensors:
    """torch-owned device arrays of the packed form (mirror of nimble_packed)."""

    def __init__(self, keys, len0, len1, hash_, pre0, pre1, max_len, paired):
        self.keys, self.len0, self.len1, self.hash, self.pre0, self.pre1 = keys, len0, len1, hash_, pre0, pre1
        self.max_len, self.paired = int(max_len), bool(paired)
        self.key_words = int(keys.shape[0])
        self.n = int(keys.shape[1])

    @classmethod
    def empty(cls, n, max_len, paired, device):
        """Uninitialised arrays for nimble_pack / nimble_unpack_records to fill (they write every element; the
        mate-1 arrays of a single-end set are zeroed).  Returns after torch's stream has finished with them, so a
        kernel on the library's own stream may write them at once."""
        import torch
        kw = int(hip_lib().nimble_key_words(max_len, int(paired)))
        m = max(n, 1)
        mk = torch.empty if paired else torch.zeros
        pt = cls(torch.empty((kw, m), dtype=torch.int64, device=device)[:, :n],
                 torch.empty(m, dtype=torch.int32, device=device)[:n],
                 mk(m, dtype=torch.int32, device=device)[:n],
                 torch.empty(m, dtype=torch.int64, device=device)[:n],
                 torch.empty(m, dtype=torch.uint8, device=device)[:n],
                 mk(m, dtype=torch.uint8, device=device)[:n], max_len, paired)
        if str(device).startswith("cuda"):
            torch.cuda.current_stream(device).synchronize()
        return pt

    def as_struct(self):
        st = NimblePacked()
        assert self.keys.is_contiguous() or self.n == 0
        st.keys = self.keys.data_ptr()
        st.len[0] = self.len0.data_ptr()
        st.len[1] = self.len1.data_ptr() if self.paired else None
        st.hash = self.hash.data_ptr()
        st.pre[0] = self.pre0.data_ptr()
        st.pre[1] = self.pre1.data_ptr() if self.paired else None
        st.key_words = self.key_words
        st.paired = int(self.paired)
        return st

    # one int64 record per read for the exchange: [key words..., hash, len0 | len1 << 16 | pre0 << 32 | pre1 << 40]
    def to_records(self):
        import torch
        meta = (self.len0.to(torch.int64) | (self.len1.to(torch.int64) << 16) | (self.pre0.to(torch.int64) << 32)
                | (self.pre1.to(torch.int64) << 40))
        return torch.cat([self.keys.t(), self.hash[:, None], meta[:, None]], dim=1).contiguous()

    def route(self, ctx, world, out=None, wait=True):
        """Group the reads by destination rank (key hash mod world) into exchange records with the device kernels
        (nimble_route_records).  Returns (records [n, key_words + 2] int64 on the device, counts per rank list).
        out: a records tensor of that shape to fill (re-used buffers keep the allocator out of the pipeline).
        wait=False only enqueues (counts = None); ctx.route_counts(world) waits for the routing and returns them."""
        import torch
        rec = out if out is not None else torch.empty((max(self.n, 1), self.key_words + 2), dtype=torch.int64,
                                                      device=self.keys.device)[:self.n]
        assert tuple(rec.shape) == (self.n, self.key_words + 2) and rec.is_contiguous()
        counts = np.zeros(world, dtype=np.uint64)
        st = self.as_struct()
        if not wait:
            _check(hip_lib().nimble_route_records(ctx.h, C.byref(st), self.n, world, rec.data_ptr(), None))
            return rec, None
        _check(hip_lib().nimble_route_records(ctx.h, C.byref(st), self.n, world, rec.data_ptr(), counts.ctypes.data))
        return rec, [int(c) for c in counts]

    @classmethod
    def unpack(cls, ctx, rec, key_words, max_len, paired, out=None):
        """Received exchange records -> packed arrays (nimble_unpack_records, on the context's stream)."""
        n = int(rec.shape[0])
        pt = out if out is not None and out.n == n else cls.empty(n, max_len, paired, rec.device)
        assert pt.key_words == key_words
        st = pt.as_struct()
        pt._rec = rec  # keep the records alive until the kernel has run
        _check(hip_lib().nimble_unpack_records(ctx.h, rec.data_ptr(), n, C.byref(st)))
        return pt

    @classmethod
    def from_records(cls, rec, key_words, max_len, paired):
        import torch
        keys = rec[:, :key_words].t().contiguous()
        meta = rec[:, key_words + 1]
        return cls(keys, (meta & 0xFFFF).to(torch.int32), ((meta >> 16) & 0xFFFF).to(torch.int32),
                   rec[:, key_words].contiguous(), ((meta >> 32) & 0xFF).to(torch.uint8),
                   ((meta >> 40) & 0xFF).to(torch.uint8), max_len, paired)


class RowsHandle:
    """Owns a nimble_rows result (Vec<(Vec<String>, i32)> on the C++ side)."""

    def __init__(self, h):
        self.h = h

    def __len__(self):
        return int(host_lib().nimble_rows_count(self.h))

    def to_list(self):
        L = host_lib()
        out = []
        for i in range(len(self)):
            cnt = C.c_int32()
            out.append((L.nimble_rows_get(self.h, i, C.byref(cnt)).decode().split("\t"), cnt.value))
        return out

    def signature(self):
        """64-bit digest of the row keys (callsets, in order): equal digests = same keys in the same order."""
        return int(host_lib().nimble_rows_signature(self.h))

    def counts(self):
        """int64 numpy array of the row counts, in row order."""
        out = np.zeros(max(len(self), 1), dtype=np.int64)
        host_lib().nimble_rows_counts(self.h, out.ctypes.data)
        return out[:len(self)]

    def keys(self):
        L = host_lib()
        cnt = C.c_int32()
        return [L.nimble_rows_get(self.h, i, C.byref(cnt)).decode() for i in range(len(self))]

    def close(self):
        if getattr(self, "h", None):
            host_lib().nimble_rows_free(self.h)
            self.h = None

    def __del__(self):
        try:  # at interpreter shutdown the module globals may be gone already
            self.close()
        except Exception:
            pass


class Library:
    """(AlignFilterConfig, Reference) of reference_library::get_reference_library, plus the device index."""

    def __init__(self, path=None, strand_filter="unstranded", text=None):
        sf = CHEM[strand_filter] if isinstance(strand_filter, str) else int(strand_filter)
        h = C.c_void_p()
        if text is not None:
            _hcheck(host_lib().nimble_library_parse(text.encode(), sf, C.byref(h)))
        else:
            _hcheck(host_lib().nimble_library_load(os.fsencode(path), sf, C.byref(h)))
        self.h = h

    @classmethod
    def from_table(cls, headers, columns, group_on, sequence_name_idx, sequence_idx):
        """A Reference written out by hand (src/align.rs:1533-1546): nothing added, columns may differ in length."""
        self = cls.__new__(cls)
        h = C.c_void_p()
        rows_of = np.asarray([len(c) for c in columns], dtype=np.int32)
        cells = [v for col in columns for v in col]
        _hcheck(host_lib().nimble_library_from_table(len(headers), _cstrs(headers), rows_of.ctypes.data, _cstrs(cells),
                                                     int(group_on), int(sequence_name_idx), int(sequence_idx), C.byref(h)))
        self.h = h
        return self

    def close(self):
        if getattr(self, "h", None):
            host_lib().nimble_library_free(self.h)
            self.h = None

    def __del__(self):
        try:  # at interpreter shutdown the module globals may be gone already
            self.close()
        except Exception:
            pass

    @property
    def config(self):
        c = HostConfig()
        host_lib().nimble_library_get_config(self.h, C.byref(c))
        return c

    @config.setter
    def config(self, c):
        _hcheck(host_lib().nimble_library_set_config(self.h, C.byref(c)))

    def update_config(self, **kw):
        c = self.config
        for k, v in kw.items():
            setattr(c, k, v)
        self.config = c

    n_rows = property(lambda self: host_lib().nimble_library_n_rows(self.h))
    n_cols = property(lambda self: host_lib().nimble_library_n_cols(self.h))
    sequence_name_idx = property(lambda self: host_lib().nimble_library_sequence_name_idx(self.h))
    sequence_idx = property(lambda self: host_lib().nimble_library_sequence_idx(self.h))

    @property
    def group_on(self):
        return host_lib().nimble_library_group_on(self.h)

    @group_on.setter
    def group_on(self, col):
        host_lib().nimble_library_set_group_on(self.h, col)

    @property
    def headers(self):
        return [host_lib().nimble_library_header(self.h, c).decode() for c in range(self.n_cols)]

    def column(self, c):
        return [host_lib().nimble_library_cell(self.h, c, r).decode() for r in range(self.n_rows)]

    def push_column(self, header, values):
        return host_lib().nimble_library_push_column(self.h, header.encode(), _cstrs(values), len(values))

    def build_index(self, device=0):
        _hcheck(host_lib().nimble_library_build_index(self.h, device))
        return self

    def score_call(self, r1, r1_off=None, r2=None, r2_off=None, n=None, fixed_len=0, max_len=0, mem=MEM_HOST):
        """score::call (src/score.rs:14-46) -> [(features, count)] sorted by callset."""
        if r1_off is not None and n is None:
            n = int(len(r1_off) - 1)
        if max_len == 0:
            if r1_off is not None and isinstance(r1_off, np.ndarray) and n:
                max_len = int(np.diff(r1_off.astype(np.int64)).max())
                if r2_off is not None:
                    max_len = max(max_len, int(np.diff(r2_off.astype(np.int64)).max()))
            else:
                max_len = fixed_len
        h = C.c_void_p()
        _hcheck(host_lib().nimble_score_call(self.h, _ptr(r1), _off(r1_off), _ptr(r2), _off(r2_off), n, fixed_len,
                                             max(max_len, 1), mem, C.byref(h)))
        return _rows(h)

    def score_call_raw(self, r1, r1_off=None, r2=None, r2_off=None, n=None, fixed_len=0, max_len=0, mem=MEM_HOST):
        """score::call that leaves the rows in the C++ result object (no Python list is built): returns a
        RowsHandle.  Used by bench.py so that the timed step contains no Python-side conversion."""
        if r1_off is not None and n is None:
            n = int(len(r1_off) - 1)
        h = C.c_void_p()
        _hcheck(host_lib().nimble_score_call(self.h, _ptr(r1), _off(r1_off), _ptr(r2), _off(r2_off), n, fixed_len,
                                             max(max_len or fixed_len, 1), mem, C.byref(h)))
        return RowsHandle(h)

    def score_call_begin(self, slot, r1, r1_off=None, r2=None, r2_off=None, n=None, fixed_len=0, max_len=0,
                         mem=MEM_HOST):
        """First half of score::call for streamed batches: enqueue the device work of one batch in `slot` (0/1)
        and return.  The buffers must stay alive until score_call_end(slot)."""
        if r1_off is not None and n is None:
            n = int(len(r1_off) - 1)
        if max_len == 0 and r1_off is not None and isinstance(r1_off, np.ndarray) and n:
            max_len = int(np.diff(r1_off.astype(np.int64)).max())
            if r2_off is not None:
                max_len = max(max_len, int(np.diff(r2_off.astype(np.int64)).max()))
        _hcheck(host_lib().nimble_score_call_begin(self.h, slot, _ptr(r1), _off(r1_off), _ptr(r2), _off(r2_off), n,
                                                   fixed_len, max(max_len or fixed_len, 1), mem))

    def score_call_begin_words(self, slot, w1, len1, stride1, w2=None, len2=None, stride2=0, n=None, max_len=0,
                               mem=MEM_HOST):
        """score_call_begin with the reads already packed (pack_reads_2bit / a device tensor of the same layout): the form
        score::call itself receives its reads in.  The arrays must stay alive until score_call_end(slot)."""
        if n is None:
            n = int(len(len1))
        self._keep_words = (w1, len1, w2, len2)
        _hcheck(host_lib().nimble_score_call_begin_words(self.h, slot, _ptr(w1), _ptr(len1), stride1, _ptr(w2), _ptr(len2),
                                                         stride2, n, max(max_len, 1), mem))

    def score_call_end(self, slot, raw=False):
        """Second half: wait for the slot's batch, return its sorted rows (a RowsHandle when raw)."""
        h = C.c_void_p()
        _hcheck(host_lib().nimble_score_call_end(self.h, slot, C.byref(h)))
        return RowsHandle(h) if raw else _rows(h)

    def score_call_umis(self, r1, r1_off=None, r2=None, r2_off=None, n=None, fixed_len=0, max_len=0, mem=MEM_HOST,
                        segment=None, n_segments=0, qual=(None, None), skip=(None, None), per_read=False):
        """The BAM pipeline's per-UMI score::call for a whole batch of UMI groups in one device call.
        Returns (rows, filters): rows = [(segment, features, count, representative read)] sorted by (segment,
        callset); filters = int32 array [n, 5] = (r1 reason, r1 score, r2 reason, r2 score, triage) or None."""
        if r1_off is not None and n is None:
            n = int(len(r1_off) - 1)
        if max_len == 0 and r1_off is not None and isinstance(r1_off, np.ndarray) and n:
            max_len = int(np.diff(r1_off.astype(np.int64)).max())
            if r2_off is not None:
                max_len = max(max_len, int(np.diff(r2_off.astype(np.int64)).max()))
        ex = UmiExtra()
        ex.segment = _ptr(segment)
        ex.n_segments = n_segments
        for m in range(2):
            ex.qual[m] = _ptr(qual[m])
            ex.skip[m] = _ptr(skip[m])
        h = C.c_void_p()
        L = host_lib()
        _hcheck(L.nimble_score_call_umis(self.h, _ptr(r1), _off(r1_off), _ptr(r2), _off(r2_off), n, fixed_len,
                                         max(max_len or fixed_len, 1), mem, C.byref(ex), int(per_read), C.byref(h)))
        try:
            rows = []
            for i in range(L.nimble_umi_rows_count(h)):
                sg, cnt, rep = C.c_uint32(), C.c_int32(), C.c_uint32()
                f = L.nimble_umi_rows_get(h, i, C.byref(sg), C.byref(cnt), C.byref(rep)).decode()
                rows.append((sg.value, f.split("\t"), cnt.value, rep.value))
            filt = None
            if per_read:
                m = L.nimble_umi_rows_reads(h)
                filt = np.zeros((m, 5), dtype=np.int32)
                buf = (C.c_int32 * 5)()
                for i in range(m):
                    L.nimble_umi_rows_filter(h, i, C.byref(buf))
                    filt[i] = buf[:]
            return rows, filt
        finally:
            L.nimble_umi_rows_free(h)

    def stream_begin(self, paired, max_len, capacity_hint=0):
        """score::call over reads that arrive in batches (one call: dedup over everything appended)."""
        _hcheck(host_lib().nimble_score_stream_begin(self.h, int(bool(paired)), max_len, capacity_hint))

    def stream_append(self, r1, r1_off=None, r2=None, r2_off=None, n=None, fixed_len=0, mem=MEM_HOST):
        if r1_off is not None and n is None:
            n = int(len(r1_off) - 1)
        self._keep = (r1, r1_off, r2, r2_off)
        _hcheck(host_lib().nimble_score_stream_append(self.h, _ptr(r1), _off(r1_off), _ptr(r2), _off(r2_off), n,
                                                      fixed_len, mem))

    def stream_end(self, raw=False):
        h = C.c_void_p()
        _hcheck(host_lib().nimble_score_stream_end(self.h, C.byref(h)))
        return RowsHandle(h) if raw else _rows(h)

    def score_call_reads(self, reads, mates=None):
        b1, o1 = pack_reads(reads)
        if mates is not None:
            b2, o2 = pack_reads(mates)
            return self.score_call(b1, o1, b2, o2)
        return self.score_call(b1, o1)

    def pack(self, r1, r1_off=None, r2=None, r2_off=None, n=None, fixed_len=0, max_len=0, mem=MEM_HOST,
             device="cuda:0", slot=0, out=None):
        """First half of the split call: 2-bit packed keys, lengths, key hash and prefilter verdicts written into
        torch tensors on `device` (a PackedTensors), ready to be exchanged between ranks.  slot: the context that
        runs the kernel (2 = the utility context, for use while calls are in flight on slots 0 / 1)."""
        if r1_off is not None and n is None:
            n = int(len(r1_off) - 1)
        max_len = max(max_len or fixed_len, 1)
        pt = out if out is not None and out.n == n else PackedTensors.empty(n, max_len, r2 is not None, device)
        st = pt.as_struct()
        _hcheck(host_lib().nimble_library_pack_slot(self.h, slot, _ptr(r1), _off(r1_off), _ptr(r2), _off(r2_off), n,
                                                    fixed_len, max_len, mem, C.byref(st)))
        return pt

    def score_call_packed_begin(self, slot, pt):
        """Enqueue the second half of the split call on slot 0 / 1; collect it with score_call_end(slot).  `pt` must
        stay alive until then."""
        st = pt.as_struct()
        _hcheck(host_lib().nimble_score_call_packed_begin(self.h, slot, C.byref(st), pt.n, pt.max_len))

    def score_call_records_begin(self, slot, rec, max_len, paired):
        """The same straight off received exchange records ([n, key_words + 2] int64 device tensor; it must stay
        alive and untouched until score_call_end(slot))."""
        assert rec.is_contiguous()
        self._rec_keep = getattr(self, "_rec_keep", {})
        self._rec_keep[slot] = rec
        _hcheck(host_lib().nimble_score_call_records_begin(self.h, slot, C.c_void_p(rec.data_ptr()),
                                                           int(rec.shape[0]), max_len, int(bool(paired))))

    def score_call_packed(self, pt, raw=False):
        """Second half of the split call: score::call from packed arrays (possibly received from other ranks)."""
        st = pt.as_struct()
        h = C.c_void_p()
        _hcheck(host_lib().nimble_score_call_packed(self.h, C.byref(st), pt.n, pt.max_len, C.byref(h)))
        self._keep_packed = pt  # the arrays must outlive the asynchronous device call
        return RowsHandle(h) if raw else _rows(h)

    def device_context(self, slot=0):
        """The device context behind this library's PseudoAligner (stage timings, counters, records)."""
        p = host_lib().nimble_library_ctx_slot(self.h, slot)
        if not p:
            raise Panic("the library has no index")
        return Context(borrowed=p)

    def align_params(self):
        c = self.config
        return AlignParams.make(c.score_percent, c.score_threshold, c.num_mismatches, c.discard_nonzero_mismatch,
                                c.discard_multiple_matches, c.require_valid_pair)

    def score_call_fastq(self, r1_path, r2_path=None):
        h = C.c_void_p()
        _hcheck(host_lib().nimble_score_call_fastq(self.h, os.fsencode(r1_path),
                                                   os.fsencode(r2_path) if r2_path else None, C.byref(h)))
        return _rows(h)

    def parse_calls(self, calls):
        """AlignmentOrientation::parse_calls (src/align.rs:276-285) -> [(feature, is_rev)]."""
        out = C.create_string_buffer(1 << 16)
        _hcheck(host_lib().nimble_host_parse_calls(self.h, "\n".join(calls).encode(), out, len(out)))
        s = out.value.decode()
        return [(l.rsplit("\t", 1)[0], l.rsplit("\t", 1)[1] == "1") for l in s.split("\n")] if s else []

    def unmap(self, features):
        """unmap (src/align.rs:851-864)."""
        out = np.zeros(max(1, len(features)), dtype=np.uint32)
        n = host_lib().nimble_host_unmap(self.h, "\n".join(features).encode(), out.ctypes.data, out.size)
        if n < 0:
            raise Panic(host_lib().nimble_host_last_error().decode("utf-8", "replace"))
        return [int(v) for v in out[:n]]

    def feature_list(self, cls, ignore_group_rollup):
        """process_equivalence_class_to_feature_list (src/align.rs:802-849)."""
        a = np.ascontiguousarray(np.asarray(list(cls), dtype=np.uint32))
        out = C.create_string_buffer(1 << 16)
        _hcheck(host_lib().nimble_host_feature_list(self.h, a.ctypes.data, a.size, int(ignore_group_rollup), out, len(out)))
        s = out.value.decode()
        return s.split("\n") if s else []

    def reference_sequence_data(self):
        """utils::get_reference_sequence_data (src/utils.rs:7-24) -> (sequences, names)."""
        out = C.create_string_buffer(1 << 22)
        _hcheck(host_lib().nimble_host_reference_sequence_data(self.h, out, len(out)))
        s = out.value.decode()
        rows = [l.split("\t") for l in s.split("\n")] if s else []
        return [r[1] for r in rows], [r[0] for r in rows]

    def coerce(self, c1, c2):
        """Host coercion of one (class R1, class R2) pair -> (callset, triage reason code)."""
        a1 = np.ascontiguousarray(np.asarray(list(c1 or []), dtype=np.uint32))
        a2 = np.ascontiguousarray(np.asarray(list(c2 or []), dtype=np.uint32))
        out = C.create_string_buffer(1 << 18)
        r = host_lib().nimble_host_coerce(self.h, int(c1 is not None), a1.ctypes.data, a1.size, int(c2 is not None),
                                          a2.ctypes.data, a2.size, out, len(out))
        if r < 0:
            raise Panic(host_lib().nimble_host_last_error().decode("utf-8", "replace"))
        s = out.value.decode()
        return (s.split("\t") if s else []), r


def dna_to_string(seq):
    """DnaString::from_acgt_bytes(seq).to_string(): through the product's 2-bit packing (nimble_host_pack_reads_2bit) and back."""
    words, lens, _ = pack_reads_2bit([seq])
    n = int(lens[0])
    w = np.asarray(words).reshape(-1)
    return "".join("ACGT"[(int(w[i >> 5]) >> (62 - 2 * (i & 31))) & 3] for i in range(n))


def sort_score_vector(scores):
    """utils::sort_score_vector (src/utils.rs:54-59); scores = [(key list, anything)]."""
    keys = "\n".join("\t".join(k) for k, _ in scores).encode()
    order = np.zeros(max(1, len(scores)), dtype=np.int32)
    _hcheck(host_lib().nimble_host_sort_score_vector(keys, len(scores), order.ctypes.data))
    return [scores[int(order[i])] for i in range(len(scores))]


def fastq_process(input_files, libraries, output_paths):
    """process::fastq::process (src/process/fastq.rs:7-30)."""
    arr = (C.c_void_p * len(libraries))(*[l.h for l in libraries])
    _hcheck(host_lib().nimble_fastq_process(len(input_files), _cstrs(input_files), len(libraries), arr,
                                            _cstrs(output_paths)))


def multi_steps(libraries, devices, r1_ptrs, r2_ptrs, n, fixed_len, warmup, steps, align_grid_pct=87):
    """process::multi::run_steps: successive calls over device-resident read sets spread across the GPUs of one node,
    one native host thread per rank (include/nimble_host.h nimble_multi_steps).  libraries[rank] has its index on
    devices[rank]; r1_ptrs[rank][set] are device pointers (ints) to n reads of fixed_len bases; r2_ptrs likewise or None.
    Returns (ms per timed step, over RCCL?, rows of the last step)."""
    W = len(devices)
    n_sets = len(r1_ptrs[0])
    libs = (C.c_void_p * W)(*[l.h for l in libraries])
    dev = (C.c_int * W)(*devices)
    a1 = (C.c_void_p * (W * n_sets))(*[p for r in r1_ptrs for p in r])
    a2 = (C.c_void_p * (W * n_sets))(*[p for r in r2_ptrs for p in r]) if r2_ptrs else None
    ms, rccl, rows = C.c_double(0.0), C.c_int(0), C.c_void_p()
    _hcheck(host_lib().nimble_multi_steps(libs, dev, W, a1, a2, n_sets, n, fixed_len, warmup, steps, align_grid_pct,
                                          C.byref(ms), C.byref(rccl), C.byref(rows)))
    return ms.value, bool(rccl.value), RowsHandle(rows)


def fastq_process_sharded(input_files, library, devices, output_path):
    """process::fastq::process over several ranks (one per entry of `devices`; an ordinal may repeat)."""
    L = host_lib()
    ins = _cstrs(input_files)
    dev = (C.c_int * len(devices))(*devices)
    _hcheck(L.nimble_fastq_process_sharded(len(input_files), ins, library.h, dev, len(devices), output_path.encode()))


def bam_process(input_file, libraries, output_paths, cores=1, force_bam_paired=False):
    """process::bam::process (src/process/bam.rs:45-243)."""
    arr = (C.c_void_p * len(libraries))(*[l.h for l in libraries])
    _hcheck(host_lib().nimble_bam_process(input_file.encode(), len(libraries), arr, _cstrs(output_paths), cores,
                                          int(force_bam_paired)))


def bam_umi_groups(input_file, force_bam_paired=False):
    """[(umi, cell barcode, dropped, [(sequence, [38 fields])])]: what UMIReader hands to the aligner (no GPU)."""
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".txt") as t:
        _hcheck(host_lib().nimble_host_bam_dump(input_file.encode(), int(force_bam_paired), t.name.encode()))
        groups = []
        for line in open(t.name, "rb").read().decode("latin-1").split("\n"):
            if not line:
                continue
            f = line.split("\t")
            if f[0] in ("G", "G*"):
                groups.append((f[1], f[2], f[0] == "G*", []))
            else:
                fields = f[2:]
                fields[1] = bytes.fromhex(fields[1]).decode("latin-1")
                groups[-1][3].append((f[1], fields))
    return groups


def pgzip_decompress(path, out_path, threads=4):
    """The parallel inflate alone: writes the decompressed stream, returns the number of pieces it came in."""
    k = C.c_uint64(0)
    _hcheck(host_lib().nimble_host_pgzip_decompress(os.fsencode(path), threads, os.fsencode(out_path), C.byref(k)))
    return k.value


def reverse_comp_if_needed(seq, reverse_comp):
    out = C.create_string_buffer(len(seq) + 8)
    _hcheck(host_lib().nimble_host_reverse_comp_if_needed(seq.encode(), int(reverse_comp), out, len(out)))
    return out.value.decode()


def parse_str_as_bool(v):
    r = C.c_int(0)
    _hcheck(host_lib().nimble_host_parse_str_as_bool(v.encode(), C.byref(r)))
    return bool(r.value)


def natural_lexical_cmp(a, b):
    return host_lib().nimble_host_natural_lexical_cmp(a.encode(), b.encode())


def shannon_entropy(s):
    return host_lib().nimble_host_shannon_entropy(s.encode())


def revcomp(s):
    out = C.create_string_buffer(len(s.encode()) + 1)
    _hcheck(host_lib().nimble_host_revcomp(s.encode(), out))
    return out.value.decode()


def maxinfo(quality, target_length, strictness):
    q = quality.encode("latin-1") if isinstance(quality, str) else quality
    return host_lib().nimble_host_maxinfo(q, len(q), target_length, strictness)


def read_fastq_stats(path):
    n, b, m = C.c_uint64(), C.c_uint64(), C.c_uint32()
    _hcheck(host_lib().nimble_host_read_fastq(os.fsencode(path), C.byref(n), C.byref(b), C.byref(m)))
    return n.value, b.value, m.value


def pack_reads_2bit(reads, stride=None):
    """The host's packer (parse::fastq::pack_reads_2bit) on a list of reads: (words uint64 [n * stride], lens uint32 [n],
    stride).  32 bases a word, first base in the highest bit pair, A=0 C=1 G=2 T=3, anything else as A."""
    flat, off = pack_reads(reads)
    n = len(off) - 1
    longest = int((off[1:] - off[:-1]).max()) if n else 0
    stride = stride or max(1, (longest + 31) // 32)
    words = np.zeros(n * stride, dtype=np.uint64)
    lens = np.zeros(n, dtype=np.uint32)
    _hcheck(host_lib().nimble_host_pack_reads_2bit(_ptr(flat), _ptr(off), n, stride, _ptr(words), _ptr(lens)))
    return words, lens, stride


def read_fastq_batched_stats(path, batch_reads, checksum=True):
    """The pipeline's threaded batch reader run to the end: (records, bases, max_len, batches, checksum).  The checksum
    is a serial pass over every base; `checksum=False` leaves it out (0) when the reader itself is being timed."""
    n, b, m, nb, h = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint64(), C.c_uint64()
    _hcheck(host_lib().nimble_host_read_fastq_batched(os.fsencode(path), batch_reads, C.byref(n), C.byref(b),
                                                      C.byref(m), C.byref(nb), C.byref(h) if checksum else None))
    return n.value, b.value, m.value, nb.value, h.value


def read_fastq_packed_stats(path, batch_reads):
    """The batch reader in the pipeline's packed mode run to the end: (records, bases, max_len, batches, checksum); the
    checksum spells the bases from the packed words (upper case, foreign bytes as A)."""
    n, b, m, nb, h = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint64(), C.c_uint64()
    _hcheck(host_lib().nimble_host_read_fastq_packed(os.fsencode(path), batch_reads, C.byref(n), C.byref(b), C.byref(m),
                                                     C.byref(nb), C.byref(h)))
    return n.value, b.value, m.value, nb.value, h.value


def filter_reason_text(code):
    return host_lib().nimble_host_filter_reason_text(code).decode()
