"""Multi-GPU driver of the hot path: one process per GPU, torch.distributed for the collectives
(backend "nccl" is RCCL over xGMI on ROCm; "gloo" for CPU rehearsal).

Reads are independent except for the dedup-by-sequence of `score_map` (src/align.rs:496-505,685): a read
key must be counted once per call, so all copies of a key have to meet on one rank.  The path therefore
has one real exchange step and one reduction:

  1. partition : dest = hash(read key) mod world                     (per rank, on device)
  2. exchange  : all_to_all of the reads, bucketed by dest            (RCCL all-to-all)
  3. compute   : score::call on the received shard                    (HIP path, no collective)
  4. reduce    : per-callset counts summed over ranks                 (RCCL all-reduce of a dense
                                                                       int64 vector over the union of
                                                                       callsets; tiny, latency-bound)

The replicated index makes class contents identical on every rank; callsets (lists of feature names)
are therefore global keys, and their union is agreed on with one all_gather_object of the key lists.
"""
import torch
import torch.distributed as dist


def _weights(width, device):
    g = torch.Generator(device="cpu")
    g.manual_seed(0x6E696D62)
    w = torch.randint(1, 1 << 30, (width,), generator=g, dtype=torch.int64) * 2 + 1
    return w.to(device)


def key_partition(r1, r2, world):
    """dest rank per read(-pair): a function of the base content only, so equal keys share a rank.
    r1/r2: uint8 tensors [n, L] (r2 may be None)."""
    h = (r1.to(torch.int64) * _weights(r1.shape[1], r1.device)).sum(dim=1)
    if r2 is not None:
        h = h * 1000003 + (r2.to(torch.int64) * _weights(r2.shape[1], r2.device)).sum(dim=1)
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
