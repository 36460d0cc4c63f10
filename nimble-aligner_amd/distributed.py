"""Multi-GPU driver of the hot path: one process per GPU, torch.distributed for the collectives
(backend "nccl" is RCCL over xGMI on ROCm; "gloo" for CPU rehearsal).

Reads are independent except for the dedup-by-sequence of `score_map` (src/align.rs:496-505,685): a read
key must be counted once per call, so all copies of a key have to meet on one rank.  The path therefore
has one real exchange step and one reduction:

  0. pack      : ASCII -> 2-bit keys + canonical key hash, where the reads are   (nimble_pack)
  1. partition : dest = key hash mod world                            (per rank, on device)
  2. exchange  : all_to_all of the PACKED records, bucketed by dest   (RCCL all-to-all, 56 B/read)
  3. compute   : rest of score::call on the received shard            (nimble_call_packed, no collective)
  4. reduce    : per-callset counts summed over ranks                 (RCCL all-reduce of a dense
                                                                       int64 vector over the union of
                                                                       callsets; tiny, latency-bound)

The replicated index makes class contents identical on every rank; callsets (lists of feature names)
are therefore global keys, and their union is agreed on with one all_gather_object of the key lists.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

# NIMBLE_DIST_TIMING=1: accumulate wall milliseconds per phase of sharded_step (adds synchronisation; diagnostics)
_TIMING = {} if os.environ.get("NIMBLE_DIST_TIMING") else None


class _Staging:
    """Small host <-> device transfers through page-locked buffers on the current stream.  Pageable copies
    (torch.tensor(list, device=...), .tolist(), .item()) were measured to wait for whatever runs on the library's
    launch stream -- 2.5 ms behind a call in flight -- page-locked non_blocking copies do not."""

    def __init__(self, device):
        self.device = device
        self.pins = {}

    def _pin(self, tag, n):
        t = self.pins.get(tag)
        if t is None or t.numel() < n:
            t = torch.empty(max(n, 16), dtype=torch.int64)
            if torch.device(self.device).type == "cuda":
                t = t.pin_memory()
            self.pins[tag] = t
        return t[:n]

    def to_device(self, tag, values, out=None):
        """values: sequence / int64 numpy array -> int64 device tensor."""
        n = len(values)
        pin = self._pin(("h2d", tag), n)
        if n:
            pin.copy_(torch.as_tensor(values, dtype=torch.int64))
        dst = out if out is not None else torch.empty(n, dtype=torch.int64, device=self.device)
        dst.copy_(pin, non_blocking=True)
        return dst

    def to_host(self, tag, tensor):
        """int64 device tensor -> list, after a synchronisation of the current stream only."""
        n = tensor.numel()
        pin = self._pin(("d2h", tag), n)
        pin.copy_(tensor.reshape(-1), non_blocking=True)
        if tensor.is_cuda:
            torch.cuda.current_stream(tensor.device).synchronize()
        return pin.tolist()


def _weights(width, device):
    g = torch.Generator(device="cpu")
    g.manual_seed(0x6E696D62)
    w = torch.randint(1, 1 << 30, (width,), generator=g, dtype=torch.int64) * 2 + 1
    return w.to(device)


def _canonical(r):
    """Bases as the read key sees them (DnaString::from_acgt_bytes + to_string): upper-case A/C/G/T, anything
    else reads as 'A'.  The partition must be a function of the KEY, not of the raw bytes."""
    x = r & 0xDF
    ok = (x == 65) | (x == 67) | (x == 71) | (x == 84)
    return torch.where(ok, x, torch.full_like(x, 65))


def key_partition(r1, r2, world):
    """dest rank per read(-pair): a function of the read key only, so equal keys share a rank.
    r1/r2: uint8 tensors [n, L] (r2 may be None).  Generic ASCII form (CPU rehearsal); the GPU path routes
    on the key hash computed by the pack kernel (sharded_call_packed)."""
    h = (_canonical(r1).to(torch.int64) * _weights(r1.shape[1], r1.device)).sum(dim=1)
    if r2 is not None:
        h = h * 1000003 + (_canonical(r2).to(torch.int64) * _weights(r2.shape[1], r2.device)).sum(dim=1)
    h = h ^ (h >> 29)
    h = h * 0x2545F4914F6CDD1D  # wraps in int64; only equality of equal keys matters
    h = h ^ (h >> 32)
    return torch.remainder(h, world)


def exchange_reads(r1, r2, dest, group=None):
    """all_to_all of fixed-length reads by destination rank.  Returns the received (r1, r2)."""
    world = dist.get_world_size(group)
    order = torch.argsort(dest, stable=True)
    counts = torch.bincount(dest, minlength=world)
    recv = torch.empty_like(counts)
    dist.all_to_all_single(recv, counts, group=group)
    in_split, out_split = counts.tolist(), recv.tolist()
    out = []
    for r in (r1, r2):
        if r is None:
            out.append(None)
            continue
        send = r[order].contiguous()
        got = torch.empty((sum(out_split), r.shape[1]), dtype=r.dtype, device=r.device)
        dist.all_to_all_single(got, send, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
        out.append(got)
    return out[0], out[1]


def exchange_records(rec, dest, group=None):
    """all_to_all of fixed-width records (2-D tensor, one row per read) by destination rank."""
    world = dist.get_world_size(group)
    order = torch.argsort(dest.to(torch.uint8) if world <= 256 else dest, stable=True)  # one radix pass
    counts = torch.bincount(dest, minlength=world)
    recv = torch.empty_like(counts)
    dist.all_to_all_single(recv, counts, group=group)
    in_split, out_split = counts.tolist(), recv.tolist()
    send = rec[order].contiguous()
    got = torch.empty((sum(out_split), rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_to_all_single(got, send, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
    return got


def hash_partition(key_hash, world):
    """dest rank from the 64-bit key hash of the pack kernel (stored in an int64 tensor)."""
    return torch.remainder(key_hash & 0x7FFFFFFFFFFFFFFF, world)


def reduce_tables(rows, device, group=None):
    """Sum per-callset counts over ranks.  rows: [(features list, count)].  Every rank gets the merged
    table sorted by callset (the order of utils::sort_score_vector, src/utils.rs:54-59)."""
    world = dist.get_world_size(group)
    keys = ["\t".join(f) for f, _ in rows]
    gathered = [None] * world
    dist.all_gather_object(gathered, keys, group=group)
    universe = sorted(set(k for ks in gathered for k in ks), key=lambda s: s.split("\t"))
    pos = {k: i for i, k in enumerate(universe)}
    vec = torch.zeros(max(len(universe), 1), dtype=torch.int64, device=device)
    if rows:
        idx = torch.tensor([pos[k] for k in keys], dtype=torch.int64, device=device)
        val = torch.tensor([c for _, c in rows], dtype=torch.int64, device=device)
        vec.index_add_(0, idx, val)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    counts = vec.tolist()
    return [(k.split("\t"), int(counts[i])) for i, k in enumerate(universe) if counts[i]]


def exchange_routed(rec, counts, group=None, alloc=None, staging=None, with_counts=False):
    """all_to_all of records already grouped by destination rank (`counts` records per rank).  alloc(rows) may
    supply the receive buffer; staging (a _Staging) keeps the small host <-> device copies page-locked.
    with_counts: also return the number of records received from each rank."""
    st = staging or _Staging(rec.device)
    send_counts = st.to_device("send_counts", counts)
    recv = torch.empty_like(send_counts)
    dist.all_to_all_single(recv, send_counts, group=group)
    out_split = st.to_host("recv_counts", recv)
    rows = sum(out_split)
    got = alloc(rows) if alloc else torch.empty((rows, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_to_all_single(got, rec, output_split_sizes=out_split, input_split_sizes=list(counts), group=group)
    return (got, out_split) if with_counts else got


class TableReducer:
    """Per-callset counts summed over ranks, call after call.  Every rank holds the same table of callsets (the
    universe); a call moves one dense int64 vector (one all_reduce) whose last element flags a rank that met a callset
    the table does not hold.  Only then the table grows, and only by what is new: each rank contributes the callsets
    ITS table lacks -- since the tables are identical, that is exactly what nobody has -- as one byte string (two small
    all_gathers: lengths, then bytes); nothing that was agreed before travels again."""

    def __init__(self, device, group=None):
        self.device, self.group = device, group
        self.sig = None          # digest of this rank's key list at the last agreement
        self.universe = []       # sorted union of the keys of all ranks (Vec<String> order)
        self.pos = {}            # key -> position in the universe
        self.local2uni = None    # position of each local row in the universe
        self.staging = _Staging(device)
        self.agreements = 0      # how often the table grew / how many key bytes this rank sent for it (diagnostics)
        self.bytes_sent = 0

    def _agree(self, keys):
        world = dist.get_world_size(self.group)
        fresh = [k for k in keys if k not in self.pos]
        blob = "\n".join(fresh).encode("utf-8")     # callsets are '\t'-joined feature names: no newline inside
        size = torch.tensor([len(blob)], dtype=torch.int64, device=self.device)
        sizes = torch.empty(world, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(sizes, size, group=self.group)
        sizes = sizes.tolist()
        width = max(sizes)
        self.agreements += 1
        self.bytes_sent += len(blob)
        if width:
            mine = torch.zeros(width, dtype=torch.uint8, device=self.device)
            if blob:
                mine[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(self.device)
            every = torch.empty(world * width, dtype=torch.uint8, device=self.device)
            dist.all_gather_into_tensor(every, mine, group=self.group)
            every = every.cpu().numpy().reshape(world, width)
            added = set()
            for r in range(world):
                if sizes[r]:
                    added.update(every[r, :sizes[r]].tobytes().decode("utf-8").split("\n"))
            added.difference_update(self.pos)
            if added:
                self.universe = sorted(self.universe + list(added), key=lambda s: s.split("\t"))
                self.pos = {k: i for i, k in enumerate(self.universe)}
        self.local2uni = np.asarray([self.pos[k] for k in keys], dtype=np.int64)

    def reduce(self, keys_fn, counts, sig):
        """keys_fn() -> list of '\t'-joined callsets (only called when an agreement is needed); counts: int64
        numpy array in the same order; sig: digest of the key list.  Returns (universe keys, summed counts list)."""
        changed = sig != self.sig
        for attempt in range(2):
            # the dense vector is laid out on the host: a device kernel for it would queue for a CU beside the
            # persistent align grid (measured 0.1-0.5 ms for this 20 kB scatter) with the host waiting behind it
            dense = np.zeros(len(self.universe) + 1, dtype=np.int64)
            if changed:
                dense[-1] = 1
            elif counts.size:
                dense[self.local2uni] = counts
            vec = self.staging.to_device("dense", dense)
            dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.group)
            host = self.staging.to_host("vec", vec)
            if host[-1] == 0:
                return self.universe, host[:-1]
            # some rank has new keys: every rank re-agrees, then the counts go round again
            self._agree(keys_fn())
            self.sig = sig
            changed = False
        raise RuntimeError("TableReducer: key agreement did not settle")

    def rows(self, universe, counts):
        counts = counts.tolist() if hasattr(counts, "tolist") else counts
        return [(k.split("\t"), int(counts[i])) for i, k in enumerate(universe) if counts[i]]


def sharded_step(lib, r1, r2, n, fixed_len, device, reducer, group=None):
    """One multi-GPU step with device-side routing and the cached key table: pack -> route (device kernels) ->
    all_to_all of the records -> unpack -> rest of score::call -> all_reduce of the counts.  Returns
    (universe keys, summed counts tensor); reducer.rows() turns them into the merged table."""
    nim = __import__("importlib").import_module("nimble-aligner_amd")
    world = dist.get_world_size(group)
    ctx = lib.device_context()
    timing = _TIMING is not None
    t = [time.perf_counter()] if timing else None

    def mark():
        if timing:
            torch.cuda.synchronize()
            ctx.synchronize()
            t.append(time.perf_counter())

    pt = lib.pack(r1, None, r2, None, n=n, fixed_len=fixed_len, max_len=fixed_len, mem=nim.MEM_DEVICE,
                  device=str(device))
    mark()
    rec, counts = pt.route(ctx, world)                  # complete on return
    mark()
    got = exchange_routed(rec, counts, group)
    torch.cuda.current_stream().synchronize()           # the records arrive on torch's stream
    mark()
    shard = nim.PackedTensors.unpack(ctx, got, pt.key_words, pt.max_len, pt.paired)
    mark()
    rows = lib.score_call_packed(shard, raw=True)
    mark()
    out = reducer.reduce(rows.keys, rows.counts(), rows.signature())
    mark()
    if timing:
        names = ("pack", "route", "exchange", "unpack", "call", "reduce")
        for k, (a, b) in zip(names, zip(t[:-1], t[1:])):
            _TIMING[k] = _TIMING.get(k, 0.0) + (b - a) * 1e3
        _TIMING["steps"] = _TIMING.get("steps", 0) + 1
    return out


def _rows_buffer(bufs, tag, rows, width, device, dtype=torch.int64):
    """A [rows, width] view of a re-used buffer with head-room: batches differ a little in how many records a rank
    receives, and an allocation inside the pipeline waits for the call in flight (and frees what a kernel may still
    read) -- so the buffer only ever grows, by a quarter at a time."""
    t = bufs.get(tag)
    if t is None or t.shape[0] < rows or t.shape[1] != width:
        t = torch.empty((max(int(rows * 1.25) + 1024, 1), width), dtype=dtype, device=device)
        bufs[tag] = t
    return t[:rows]


class ShardedPipeline:
    """The multi-GPU step, software-pipelined over successive batches.  For batch b:
         P(b) pack + route            launch stream (utility context)
         X(b) all_to_all of records   RCCL, overlaps C(b-1) on the device
         C(b) unpack + packed call    launch stream, call slot b % 2
         F(b) rows + count all-reduce host + one small RCCL all_reduce
    submit(b) runs P(b), enqueues C(b-1), runs X(b) and returns F(b-2): the launch stream always holds the next
    kernels, the exchange and the host work hide behind a call.  flush() drains the last two batches."""

    def __init__(self, lib, device, reducer, group=None, align_grid_pct=87):
        self.nim = __import__("importlib").import_module("nimble-aligner_amd")
        self.lib, self.device, self.reducer, self.group = lib, device, reducer, group
        self.world = dist.get_world_size(group)
        self.util = lib.device_context(2)
        # every torch / RCCL operation of the pipeline runs on this side stream: work on torch's default (null)
        # stream waits for whatever is in flight on the library's launch stream (measured: even a .tolist() of
        # eight numbers takes the length of the running call), a side stream does not
        self.comm = torch.cuda.Stream(device=device)
        self.launch = torch.cuda.ExternalStream(lib.device_context(0).stream_ptr(), device=device)
        align_grid_pct = int(os.environ.get("NIMBLE_ALIGN_GRID_PCT", align_grid_pct))
        for slot in (0, 1):  # leave room beside the persistent align grid for RCCL's kernels
            lib.device_context(slot).set_option(self.nim.OPT_ALIGN_GRID_PCT, align_grid_pct)
            # the tail of a call stays on the launch stream here: one stream fewer next to RCCL's (the gain of moving
            # it aside is 1.5 % in this pipeline, 6 % in the single-GPU one)
            lib.device_context(slot).set_option(self.nim.OPT_TAIL_ASIDE, int(os.environ.get("NIMBLE_DEDUP_ASIDE_SHARDED", 0)))
        self.from_records = os.environ.get("NIMBLE_UNPACK_RECORDS") is None
        self.i = 0
        self.arrived = None        # (records, key_words, max_len, paired) of batch i-1, exchanged, not yet begun
        self.inflight = {}         # slot -> shard tensors of the call in flight (kept alive)
        self._bufs = {}            # re-used device buffers: allocation inside the loop stalls behind the running call

    def _tensor(self, tag, shape, dtype=torch.int64):
        t = self._bufs.get(tag)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[tag] = t
        return t

    def _packed(self, tag, n, max_len, paired):
        pt = self._bufs.get(tag)
        if pt is None or pt.n != n or pt.max_len != max_len or pt.paired != paired:
            pt = self.nim.PackedTensors.empty(n, max_len, paired, self.device)
            self._bufs[tag] = pt
        return pt

    def _begin(self, b):
        got, kw, max_len, paired, arrived_ev = self.arrived
        self.arrived = None
        slot = b % 2
        self.launch.wait_event(arrived_ev)   # the launch stream, not the host, waits for the exchange
        if self.from_records:
            # the call reads the received records as they are (no unpack pass)
            self.lib.score_call_records_begin(slot, got, max_len, paired)
            self.inflight[slot] = got
            return
        shard = self.nim.PackedTensors.unpack(self.lib.device_context(slot), got, kw, max_len, paired,
                                              out=self._packed(("shard", b % 3), int(got.shape[0]), max_len, paired))
        self.lib.score_call_packed_begin(slot, shard)
        self.inflight[slot] = shard

    def _finish(self, b):
        slot = b % 2
        t0 = time.perf_counter()
        rows = self.lib.score_call_end(slot, raw=True)
        self.inflight.pop(slot, None)
        t1 = time.perf_counter()
        counts, sig = rows.counts(), rows.signature()
        t2 = time.perf_counter()
        out = self.reducer.reduce(rows.keys, counts, sig)
        if _TIMING is not None:
            for k, v in (("f_end", t1 - t0), ("f_counts", t2 - t1), ("f_reduce", time.perf_counter() - t2)):
                _TIMING[k] = _TIMING.get(k, 0.0) + v * 1e3
        return out

    def submit(self, r1, r2, n, fixed_len):
        """Feed batch i; returns the reduced table of batch i-2 (None for the first two calls)."""
        with torch.cuda.stream(self.comm):
            return self._submit(r1, r2, n, fixed_len)

    def _submit(self, r1, r2, n, fixed_len):
        b = self.i
        t0 = time.perf_counter()
        pt = self.lib.pack(r1, None, r2, None, n=n, fixed_len=fixed_len, max_len=fixed_len, mem=self.nim.MEM_DEVICE,
                           device=str(self.device), slot=2, out=self._packed("pack", n, fixed_len, r2 is not None))
        # two send buffers: X(b-1) may still be reading the other one (the launch stream waits for X(b-1) at the start
        # of C(b-1), i.e. before the routing of batch b+1 writes this buffer again)
        rec, _ = pt.route(self.util, self.world, out=self._tensor(("rec", b % 2), (n, pt.key_words + 2)), wait=False)
        t1 = time.perf_counter()
        if self.arrived is not None:
            self._begin(b - 1)                           # C(b-1) queues up behind P(b): no gap on the launch stream
        t2 = time.perf_counter()
        # F(b-2) before X(b) is enqueued: RCCL work is ordered on the comm stream, and the small count all-reduce must
        # not queue behind the record exchange
        out = self._finish(b - 2) if b >= 2 else None
        t3 = time.perf_counter()
        counts = self.util.route_counts(self.world)     # the host waits for P(b) only
        t4 = time.perf_counter()
        got = exchange_routed(rec, counts, self.group,     # X(b) while C(b-1) runs; nobody on the host waits for it
                              alloc=lambda rows: _rows_buffer(self._bufs, ("got", b % 3), rows, pt.key_words + 2,
                                                              self.device),
                              staging=self.reducer.staging)
        ev = torch.cuda.Event()
        ev.record()
        self.arrived = (got, pt.key_words, pt.max_len, pt.paired, ev)
        t5 = time.perf_counter()
        if _TIMING is not None:
            for k, v in (("p_pack", t1 - t0), ("p_begin", t2 - t1), ("p_finish", t3 - t2), ("p_route_wait", t4 - t3),
                         ("p_exchange", t5 - t4)):
                _TIMING[k] = _TIMING.get(k, 0.0) + v * 1e3
            _TIMING["steps"] = _TIMING.get("steps", 0) + 1
        self.i += 1
        return out

    def flush(self):
        """Drain: returns the reduced tables of the batches still in the pipeline, oldest first."""
        with torch.cuda.stream(self.comm):
            return self._flush()

    def _flush(self):
        b = self.i
        outs = []
        if self.arrived is not None:
            self._begin(b - 1)
        if b >= 2:
            outs.append(self._finish(b - 2))
        if b >= 1:
            outs.append(self._finish(b - 1))
        self.i = 0
        return outs


class LocalAlignPipeline:
    """The multi-GPU step in its second form: every rank aligns the reads it was given; only the KEYS travel, to the
    rank that owns them (hash mod world), and one verdict byte per read comes back (include/nimble_hip.h,
    nimble_ctx_defer_dedup).  Against ShardedPipeline this drops the unpack, keeps the per-read records on the rank
    that read the input, and lets the key exchange run beside the alignment of the same batch.  For batch b:
         P(b) head + pack + route     launch stream
         A(b) align + interning       launch stream;   X1(b) all_to_all of the key records runs beside it (RCCL)
         D(b) owner's dedup of the received records -> verdict bytes;   X2(b) all_to_all of the verdicts
         C(b) count with the verdicts  launch stream, enqueued behind A(b+1) so that X2(b) is never waited for
         F(b) rows + count all-reduce  host + one small RCCL all_reduce, two batches later
    Three call slots are open at a time.  submit(b) returns the reduced table of batch b-2; flush() drains."""

    SLOTS = (0, 1, 3)

    def __init__(self, lib, device, reducer, group=None, align_grid_pct=87):
        self.nim = __import__("importlib").import_module("nimble-aligner_amd")
        self.lib, self.device, self.reducer, self.group = lib, device, reducer, group
        self.world = dist.get_world_size(group)
        self.util = lib.device_context(2)
        self.ctxs = [lib.device_context(s) for s in self.SLOTS]
        self.comm = torch.cuda.Stream(device=device)     # see ShardedPipeline: nothing may run on the null stream
        self.launch = torch.cuda.ExternalStream(self.ctxs[0].stream_ptr(), device=device)
        align_grid_pct = int(os.environ.get("NIMBLE_ALIGN_GRID_PCT", align_grid_pct))
        for c in self.ctxs:
            c.set_option(self.nim.OPT_ALIGN_GRID_PCT, align_grid_pct)
        self.i = 0
        self.waiting = None      # (batch, event after X2, verdict tensor): its count is enqueued by the next submit
        self._bufs = {}

    def _tensor(self, tag, shape, dtype=torch.int64):
        t = self._bufs.get(tag)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[tag] = t
        return t

    def _count_waiting(self):
        """X2 + C of the batch whose owner-side dedup was enqueued by the previous submit."""
        if self.waiting is None:
            return
        b, d, owner, recv_counts, counts, mine, n = self.waiting
        self.waiting = None
        torch.cuda.current_stream().wait_event(d)
        dist.all_to_all_single(mine[:n], owner, output_split_sizes=list(counts), input_split_sizes=list(recv_counts),
                               group=self.group)                                 # X2: one verdict byte per read
        x2 = torch.cuda.Event()
        x2.record()
        self.launch.wait_event(x2)
        self.ctxs[b % 3].count_verdicts(mine)

    def _finish(self, b):
        rows = self.lib.score_call_end(self.SLOTS[b % 3], raw=True)
        return self.reducer.reduce(rows.keys, rows.counts(), rows.signature())

    def submit(self, r1, r2, n, fixed_len):
        """Feed batch i (uint8 device tensors [n, fixed_len]); returns the reduced table of batch i-2 or None."""
        with torch.cuda.stream(self.comm):
            return self._submit(r1, r2, n, fixed_len)

    def _submit(self, r1, r2, n, fixed_len):
        b, k = self.i, self.i % 3
        ctx = self.ctxs[k]
        kw = self.nim.key_words(fixed_len, r2 is not None)
        t0 = time.perf_counter()
        rec = self._tensor(("rec", k), (max(n, 1), kw + 2))
        perm = self._tensor(("perm", k), (max(n, 1),), torch.int32)
        ctx.defer_dedup(self.world, rec, perm)
        self.lib.score_call_begin(self.SLOTS[k], r1, None, r2, None, n=n, fixed_len=fixed_len,
                                  mem=self.nim.MEM_DEVICE)                       # P(b) A(b)
        # X2(b-1) C(b-1): the verdict exchange waits for D(b-1), which ran before P(b); the count lands behind A(b).
        # (Enqueued here and not at the end of submit(b-1): RCCL work is ordered on the comm stream, and the count
        # all-reduce of F(b-3) must not queue behind an exchange that waits for A(b-1).)
        self._count_waiting()
        t1 = time.perf_counter()
        out = self._finish(b - 2) if b >= 2 else None                            # F(b-2), before X1(b) is enqueued
        counts = ctx.route_counts(self.world)                                    # the host waits for P(b) only
        t2 = time.perf_counter()
        got, recv_counts = exchange_routed(rec[:n], counts, self.group, staging=self.reducer.staging, with_counts=True,
                                           alloc=lambda rows: _rows_buffer(self._bufs, ("got", k), rows, kw + 2,
                                                                           self.device))
        x1 = torch.cuda.Event()
        x1.record()                                                              # X1(b), beside A(b)
        self.launch.wait_event(x1)
        owner = _rows_buffer(self._bufs, ("owner", k), int(got.shape[0]), 1, self.device, torch.uint8)[:, 0]
        self.util.dedup_records(got, kw, owner)                                  # D(b)
        d = torch.cuda.Event()
        d.record(self.launch)
        mine = self._tensor(("mine", k), (max(n, 1),), torch.uint8)
        self.waiting = (b, d, owner, recv_counts, counts, mine, n)
        t3 = time.perf_counter()
        t4 = t3
        if _TIMING is not None:
            for name, v in (("l_begin", t1 - t0), ("l_route_wait", t2 - t1), ("l_exchange", t3 - t2),
                            ("l_finish", t4 - t3)):
                _TIMING[name] = _TIMING.get(name, 0.0) + v * 1e3
            _TIMING["steps"] = _TIMING.get("steps", 0) + 1
        self.i += 1
        return out

    def flush(self):
        """Drain: the reduced tables of the batches still in the pipeline, oldest first."""
        with torch.cuda.stream(self.comm):
            b = self.i
            self._count_waiting()
            outs = []
            if b >= 2:
                outs.append(self._finish(b - 2))
            if b >= 1:
                outs.append(self._finish(b - 1))
            self.i = 0
            return outs


def sharded_call_packed(lib, r1, r2, n, fixed_len, device, group=None, raw=False):
    """The multi-GPU step on the HIP path: pack locally (2 bits per base, canonical key hash), exchange the
    packed records by key hash (40-56 B per read instead of 150-300 B of ASCII), finish score::call on the
    receiving rank, all-reduce the counts.  r1/r2: uint8 device tensors [n, fixed_len]."""
    nim = __import__("importlib").import_module("nimble-aligner_amd")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    pt = lib.pack(r1, None, r2, None, n=n, fixed_len=fixed_len, max_len=fixed_len, mem=nim.MEM_DEVICE,
                  device=str(device))
    if world > 1:
        lib.device_context().synchronize()  # pack ran on the library's stream; torch reads its output next
        rec = pt.to_records()
        got = exchange_records(rec, hash_partition(pt.hash, world), group)
        pt = nim.PackedTensors.from_records(got, pt.key_words, pt.max_len, pt.paired)
        torch.cuda.current_stream().synchronize()
    rows = lib.score_call_packed(pt, raw=raw and world == 1)
    if world > 1:
        rows = reduce_tables(rows, device, group)
    return rows


def sharded_call(compute, r1, r2, device, group=None):
    """The whole multi-GPU step.  `compute(r1, r2) -> rows` runs score::call on one rank's shard
    (the HIP path in production; tests inject the CPU oracle to rehearse the collectives on gloo)."""
    world = dist.get_world_size(group)
    if world > 1:
        dest = key_partition(r1, r2, world)
        r1, r2 = exchange_reads(r1, r2, dest, group)
    rows = compute(r1, r2)
    if world > 1:
        rows = reduce_tables(rows, device, group)
    return rows
