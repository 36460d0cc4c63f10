#!/usr/bin/env python3
"""bench.py -- reads/s of the hot path (score::call: pack -> align -> intern -> dedup -> count -> rows) on MI355X.

Workload at N=1: BASELINE.json configs[2] -- 10 M synthetic 150 bp single-end reads against a 1 k-feature
(2 k index rows) allele-family library with the basic.json alignment settings; reads are resident in HBM when
the timed region starts.  A "step" is one complete score::call over the batch, ending with the sorted
(callset -> count) rows on the host.  At N>1 every rank holds its own 10 M reads (weak scaling) and a step is
partition -> all-to-all exchange by read key -> score::call per rank -> all-reduce of the count vector (RCCL).

Prints ONE JSON line on rank 0 (contract in the task description), with `roofline` for the dominant kernel
(k_align, HBM-bound integer work) and `cpu_baseline` (the CPU oracle, a port, on this box's host cores).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s achievable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--features", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=2, choices=(1, 2),
                    help="calls in flight at N=1: 2 = step i+1 runs on the GPU while the host finishes step i")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="multi-GPU: run every step to completion before the next one starts (no overlap of the "
                         "exchange with the previous call)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="rehearsal: run the multi-GPU step (route, all-to-all, reduce over RCCL) even with one rank")
    ap.add_argument("--form", choices=("sharded", "local"), default=os.environ.get("NIMBLE_MULTI_GPU_FORM", "sharded"),
                    help="multi-GPU step: 'sharded' moves the packed reads to the key's owner, 'local' aligns where "
                         "the reads are and moves only keys and verdict bytes")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000,
                    help="reads timed on the CPU oracle, all host cores (0 = skip); about 11 s at the default")
    ap.add_argument("--cpu-threads", type=int, default=0, help="oracle threads (0 = all host cores, at most 64)")
    args = ap.parse_args()

    # HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that share a queue run
    # in order: with the launch stream, the result stream, torch's streams and RCCL's, the exchange of the multi-GPU
    # pipeline would queue behind the align kernel it is meant to overlap.  Must be set before HIP initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    import torch
    import torch.distributed as dist

    nim = importlib.import_module("nimble-aligner_amd")
    synth = importlib.import_module("nimble-aligner_amd.synth")
    nd = importlib.import_module("nimble-aligner_amd.distributed")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    L = 150
    n = args.reads
    names, seqs = synth.make_library(args.features)
    lib_obj = synth.library_json(names, seqs)
    lib = nim.Library(text=json.dumps(lib_obj), strand_filter="unstranded").build_index(local_rank)
    ctx = lib.device_context()
    reads = synth.make_reads_torch(seqs, n, L=L, seed=synth.READ_SEED + 7919 * rank, device=str(device))
    torch.cuda.synchronize()

    def compute(a, b=None):
        torch.cuda.current_stream().synchronize()  # reads produced on torch's stream; the call runs on its own
        return lib.score_call(a, None, n=a.shape[0], fixed_len=L, max_len=L, mem=nim.MEM_DEVICE)

    reducer = nd.TableReducer(device) if sharded else None

    def step():
        """One score::call: ends with the sorted rows materialised on the host (C++ result object)."""
        if sharded:
            # pack -> route by key hash (device kernels) -> all-to-all of the records -> unpack -> finish per rank
            # -> all-reduce of the counts over the agreed callset table
            return nd.sharded_step(lib, reads, None, n, L, device, reducer)
        return lib.score_call_raw(reads, None, n=n, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE)

    # one untimed pass with the work counters on: gives P / U / sum(E) for the algorithmic byte count
    ctx.set_counters(True)
    rows = step()
    ctx.n = n
    counters = ctx.counters()   # multi-GPU: of the shard this rank received (statistically the same reads)
    ctx.set_counters(False)

    depth = args.depth if not sharded else 1
    ctxs = [lib.device_context(s) for s in range(depth)]
    for c in ctxs:
        c.set_counters(False)  # the work counters cost ~30 % of k_align; collected once, above

    def begin(slot):
        lib.score_call_begin(slot, reads, None, n=n, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE)

    def end(slot):
        r = lib.score_call_end(slot, raw=True)
        for k, v in ctxs[slot].timing().items():
            stage[k] += v
        return r

    stage = {k: 0.0 for k in ("pack", "align", "intern", "dedup", "count", "total")}
    for _ in range(args.warmup):
        rows = step()
    if depth > 1:
        for s in range(depth):  # the second slot allocates its buffers on first use
            begin(s)
        for s in range(depth):
            end(s)
    pipe = None
    if sharded and not args.no_pipeline:
        pipe = (nd.LocalAlignPipeline if args.form == "local" else nd.ShardedPipeline)(lib, device, reducer)
        for s_ in range(4 if args.form == "local" else 3):
            lib.device_context(s_).set_counters(False)
        for _ in range(max(args.warmup, 3)):   # allocates the other call slots and the utility context
            pipe.submit(reads, None, n, L)
        for r in pipe.flush():
            rows = r
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    stage = {k: 0.0 for k in stage}
    t0 = time.perf_counter()
    if sharded and not args.no_pipeline:
        # software-pipelined multi-GPU steps: K submits + the drain all end inside the timed region
        for _ in range(args.steps):
            r = pipe.submit(reads, None, n, L)
            rows = r if r is not None else rows
        for r in pipe.flush():
            rows = r
        for k, v in ctx.timing().items():
            stage[k] += v * args.steps
    elif depth == 1:
        for _ in range(args.steps):
            rows = step()
            for k, v in ctx.timing().items():
                stage[k] += v
    else:
        # every step is a complete score::call (begin + end); two are in flight, all K end inside the timed region
        for i in range(args.steps):
            begin(i % 2)
            if i:
                rows = end((i - 1) % 2)
        if args.steps:
            rows = end((args.steps - 1) % 2)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    steps = max(args.steps, 1)
    stage = {k: v / steps for k, v in stage.items()}
    total_reads = n * world * steps
    value = total_reads / elapsed
    # device time per call: the event span of one call, unless calls overlap on the device (then the step time)
    device_ms = stage["total"] if (depth == 1 and not sharded) else 1000.0 * elapsed / steps

    out = {
        "metric": "reads/sec aligned (whole job), counts bit-exact vs the CPU path",
        "value": value,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE.json configs[2]: %d x %dbp single-end reads per GPU vs %d-feature library "
                        "(%d index rows), basic.json settings (score_percent 0.33, score_threshold 50, "
                        "num_mismatches 0), unstranded" % (n, L, args.features, 2 * args.features),
            "reads_per_gpu": n, "read_len": L, "features": args.features,
            "parallelism": (("1 process/GPU; reads aligned where they are, keys to their owner by hash (all-to-all), "
                             "verdict bytes back (all-to-all) + count all-reduce (RCCL); key exchange of step i "
                             "beside its own alignment" if args.form == "local" and not args.no_pipeline else
                             "1 process/GPU; packed reads routed by key hash (all-to-all) + count all-reduce (RCCL)"
                             + ("" if args.no_pipeline else "; exchange of step i overlaps the call of step i-1")))
                           if sharded else "single GPU",
            "rows": len(rows) if not sharded else len(reducer.rows(*rows)),
            "calls_in_flight": depth,
        },
        "stage_ms": {k: round(v, 4) for k, v in stage.items()},
        "stage_note": ("per call, HIP events; with calls in flight the dedup / count / compaction of call i run on a "
                       "side stream beside the pack of call i+1, so pack and dedup read longer than alone "
                       "(0.37 / 0.53 ms) and 'total' is the latency of one call, not the time per call"),
        "device_reads_per_s": n / (device_ms / 1000.0) if device_ms > 0 else None,
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (k_align): algorithmic bytes of one launch / its duration
        P, U, E = counters["probes"], counters["nodes"], counters["class_entries"]
        hit = counters["seeded"]
        key_bytes = 8 * ((L + 31) // 32)
        n_call = int(counters["reads"]) if sharded else n   # reads of the launch the counters describe
        align_bytes = n_call * key_bytes + 16 * P + hit * key_bytes + 16 * U + 4 * E + 16 * n_call
        pipe_bytes = n_call * L + 16 * P + hit * (key_bytes + 96) + 16 * U + 4 * E + 16 * n_call
        align_s = stage["align"] / 1000.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("reads") == n and tj.get("features") == args.features and not sharded:
                    traffic = tj.get("k_align_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # measured device stream copy on this box: read + write of 1 GiB, best of 5
        a_ = torch.empty(1 << 27, dtype=torch.int64, device=device)
        b_ = torch.empty_like(a_)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        copy_ms = None
        for _ in range(5):
            e0.record()
            b_.copy_(a_)
            e1.record()
            e1.synchronize()
            t_ = e0.elapsed_time(e1)
            copy_ms = t_ if copy_ms is None else min(copy_ms, t_)
        stream_copy_gbps = 2 * a_.numel() * 8 / (copy_ms / 1e3) / 1e9
        del a_, b_
        out["roofline"] = {
            "bound": "hbm",
            "kernel": "k_align",
            "achieved": align_bytes / align_s / 1e9,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": align_bytes / align_s / 1e9 / HBM_PEAK_GBPS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": align_bytes,
            "kernel_ms": stage["align"],
            "bytes_per_read": align_bytes / max(n_call, 1),
            "pipeline": {"algorithmic_bytes_per_read": pipe_bytes / max(n_call, 1),
                         "achieved_GBps": pipe_bytes / (device_ms / 1000.0) / 1e9,
                         "frac": pipe_bytes / (device_ms / 1000.0) / 1e9 / HBM_PEAK_GBPS},
            "counters": {"probes": P, "nodes": U, "class_entries": E, "seeded": hit},
            "stream_copy_GBps": stream_copy_gbps,
            "frac_of_stream_copy": align_bytes / align_s / 1e9 / stream_copy_gbps,
        }
        # ---- CPU baseline: the oracle (a port of the reference path) on a bounded sample of the same reads
        if args.cpu_sample > 0 and world == 1:
            from oracle import oracle as ora
            S = min(args.cpu_sample, n)
            sample = reads[:S].cpu().numpy()
            cols = [["s"] * len(names), names, [str(len(s)) for s in seqs], seqs]
            ref = ora.Reference.from_columns(lib_obj[1]["headers"], cols, "")
            cfg = ora.config_from_json(lib_obj[0], len(names), "unstranded")
            oidx = ora.Index.from_reference(ref)
            threads = args.cpu_threads or min(os.cpu_count() or 1, 64)
            offs = synth.fixed_offsets(S, L)
            t1 = time.perf_counter()
            ores = ora.call(oidx, ref, cfg, sample.reshape(-1), offs, n_threads=threads)
            cpu_s = time.perf_counter() - t1
            S1 = min(S, 1_000_000)
            t1 = time.perf_counter()
            ora.call(oidx, ref, cfg, sample[:S1].reshape(-1), synth.fixed_offsets(S1, L), n_threads=1)
            cpu1_s = time.perf_counter() - t1
            # parity on the sample: the GPU table must equal the oracle's table
            got = lib.score_call(reads[:S].contiguous(), None, n=S, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE)
            parity = [(f, c) for f, c in got] == [(f, c) for f, c in ores.rows]
            if not parity:
                raise SystemExit("bench.py: GPU table differs from the CPU oracle on the sample")
            out["cpu_baseline"] = {
                "value": S / cpu_s, "unit": "reads/s", "cores": threads, "kind": "port",
                "sample": "first %d of the %d reads, oracle/libnimble_oracle.so, %d threads (hash-partitioned by key); "
                          "index build and read generation excluded" % (S, n, threads),
                "single_thread_value": S1 / cpu1_s,
                "parity_on_sample": "bit-exact table (%d rows)" % len(got),
            }
            out["speedup_vs_cpu_baseline"] = value / (S / cpu_s)
    if rank == 0 and getattr(nd, "_TIMING", None):
        k = max(nd._TIMING.get("steps", 1), 1)
        out["sharded_phase_ms"] = {a: round(b / k, 3) for a, b in nd._TIMING.items() if a != "steps"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
