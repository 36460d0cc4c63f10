#!/usr/bin/env python3
"""bench.py -- reads/s of the hot path (score::call: pack -> align -> intern -> dedup -> count -> rows) on MI355X.

Default workload (every N): BASELINE.json configs[2] -- 10 M synthetic 150 bp single-end reads per GPU against a
1 k-feature (2 k index rows) allele-family library with the basic.json alignment settings; reads are resident in HBM when
the timed region starts.  A "step" is one complete score::call over one batch, ending with the sorted (callset -> count)
rows on the host.  Successive steps take DIFFERENT read sets (--read-sets, default 3, rotated), so the class table, the
host's coercion memo and the multi-GPU key agreement keep meeting new material as they do in production.  At N>1 every
rank holds its own reads (weak scaling) and a step is partition -> all-to-all exchange by read key -> score::call per rank
-> all-reduce of the count vector (RCCL).

Other workloads (--workload): `configs4` = BASELINE.json configs[4], 80 M reads in total split over the ranks against a
5 k-feature library; `configs3` = configs[3], paired-end 2x150 with the mismatch.json settings; `families100` = a library of
gene families of 100 alleles at 1 % divergence (the shape of the reference's own MHC fixtures), same reads recipe;
`families500` = families of 500 alleles (classes beyond the 256-row register window: the LDS row window).

Prints ONE JSON line on rank 0 (contract in the task description), with `roofline` for the dominant kernel (k_align,
HBM-bound integer work) and `cpu_baseline` (the CPU oracle, a port, on this box's host cores), plus `step_ms` (min / median
/ max over the timed steps) and, at N=1, `e2e_fastq_reads_per_s` / `e2e_fastq_gz_reads_per_s`: the whole FASTQ pipeline
(lib/nimble: parse, H2D, call, TSV) on a bounded file, and `e2e_bam_reads_per_s`: the BAM pipeline on a synthetic 10x-style
BAM, all outside the timed value.
"""
import argparse
import importlib
import json
import os
import statistics
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s achievable

WORKLOADS = {
    "configs2": dict(features=1000, family=4, paired=False, reads=10_000_000, total=False,
                     text="BASELINE.json configs[2]: %(n)d x 150bp single-end reads per GPU vs %(T)d-feature library "
                          "(%(rows)d index rows), basic.json settings (score_percent 0.33, score_threshold 50, "
                          "num_mismatches 0), unstranded"),
    "configs3": dict(features=1000, family=4, paired=True, reads=10_000_000, total=False,
                     text="BASELINE.json configs[3]: %(n)d paired-end 2x150bp read pairs per GPU vs %(T)d-feature library "
                          "(%(rows)d index rows), mismatch.json settings (score_percent 0.08, score_threshold 12, "
                          "num_mismatches 2), unstranded"),
    "configs4": dict(features=5000, family=4, paired=False, reads=80_000_000, total=True,
                     text="BASELINE.json configs[4]: 80 M x 150bp single-end reads in total, %(n)d per GPU, vs %(T)d-feature "
                          "library (%(rows)d index rows), basic.json settings, unstranded"),
    "families100": dict(features=1000, family=100, paired=False, reads=4_000_000, total=False,
                        text="allele families of 100 (not a BASELINE.json config): %(n)d x 150bp single-end reads per GPU vs "
                             "%(T)d features = %(fam)d gene families of 100 alleles at 1 %% divergence (%(rows)d index rows), "
                             "basic.json settings, unstranded"),
    "families500": dict(features=2000, family=500, paired=False, reads=2_000_000, total=False,
                        text="allele families of 500 (not a BASELINE.json config): %(n)d x 150bp single-end reads per GPU vs "
                             "%(T)d features = %(fam)d gene families of 500 alleles at 1 %% divergence (%(rows)d index rows), "
                             "basic.json settings, unstranded"),
}


def usable_cpus():
    """Host CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box shows 256 CPUs
    and grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, round(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, round(q / per)))
        except (OSError, ValueError):
            pass
    return n


def e2e_fastq(names, seqs, reads_np, n_plain, n_gz):
    """The FASTQ pipeline end to end through the argv-compatible CLI (parse + H2D + call + TSV), page cache warm."""
    synth = importlib.import_module("nimble-aligner_amd.synth")
    exe = os.path.join(ROOT, "nimble-aligner_amd", "lib", "nimble")
    out = {}
    d = tempfile.mkdtemp(prefix="nimble_e2e_", dir="/tmp")
    try:
        libp = os.path.join(d, "lib.json")
        synth.write_library(libp, names, seqs)
        fq = os.path.join(d, "reads.fastq")
        synth.write_fastq_fast(fq, reads_np[:n_plain], qual="binned")
        fz = os.path.join(d, "reads_gz.fastq")
        if n_gz == n_plain:
            fz = fq
        else:
            synth.write_fastq_fast(fz, reads_np[:n_gz], qual="binned")
        # one gzip member, level 6, deflated by several processes the way pigz writes it (gzip itself would take minutes)
        # (in a fresh interpreter: this process holds the GPU, and the writer forks a pool)
        subprocess.run([sys.executable, "-c",
                        "import importlib, sys; sys.path.insert(0, %r); importlib.import_module('nimble-aligner_amd.synth')"
                        ".gzip_single_stream(%r, %r, 6, workers=16)" % (ROOT, fz, os.path.join(d, "reads_gz.fastq.gz"))],
                       check=True)
        fz = os.path.join(d, "reads_gz.fastq")
        for tag, path, n in (("warm", fq, n_plain), ("e2e_fastq_reads_per_s", fq, n_plain),
                             ("e2e_fastq_gz_reads_per_s", fz + ".gz", n_gz)):
            env = dict(os.environ, NIMBLE_HOST_TIMING="1")
            t0 = time.perf_counter()
            cp = subprocess.run([exe, "-r", libp, "-o", os.path.join(d, tag + ".tsv"), "-i", path, "-f", "unstranded"],
                                capture_output=True, text=True, env=env)
            wall = time.perf_counter() - t0
            if cp.returncode != 0:
                raise RuntimeError(cp.stderr[-500:])
            pipe = [l for l in cp.stderr.splitlines() if "fastq pipeline" in l]
            secs = float(pipe[-1].split(")")[1].split("s,")[0]) if pipe else wall
            if tag != "warm":
                out[tag] = n / secs
                out[tag.replace("reads_per_s", "wall_s")] = wall
        out["e2e_note"] = ("lib/nimble on %d (plain) / %d (.gz: one member, level 6) of the bench reads written as FASTQ with "
                           "binned qualities: reads/s of the pipeline itself (mapped file -> [inflate ->] parse -> pinned "
                           "batches -> H2D -> one call -> TSV) on the box's CPU share; the wall figure adds process start, "
                           "index build and HIP initialisation" % (n_plain, n_gz))
    finally:
        subprocess.run(["rm", "-rf", d])
    return out


def e2e_bam(pairs):
    """The BAM pipeline end to end through the CLI (BGZF -> records -> UMI groups -> device calls -> one gzip-compressed TSV row
    per read pair) on a synthetic 10x-style BAM written by tools/e2e_bam.py: wall time of the whole process."""
    exe = os.path.join(ROOT, "nimble-aligner_amd", "lib", "nimble")
    d = tempfile.mkdtemp(prefix="nimble_bam_", dir="/tmp")
    try:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_bam.py"), "write", d, str(pairs), "8"], check=True,
                       capture_output=True)
        walls = []
        for _ in range(2):  # (the first run warms the page cache)
            t0 = time.perf_counter()
            cp = subprocess.run([exe, "-r", d + "/lib.json", "-o", d + "/out.tsv.gz", "-i", d + "/in.bam"], capture_output=True,
                                text=True)
            walls.append(time.perf_counter() - t0)
            if cp.returncode != 0:
                raise RuntimeError(cp.stderr[-500:])
        return {"e2e_bam_reads_per_s": 2 * pairs / walls[-1], "e2e_bam_wall_s": walls[-1],
                "e2e_bam_note": "lib/nimble on a BAM of %d proper pairs (runs of one UMI of 8 pairs, every pair its own cell "
                                "barcode, 10x-style tags): wall time of the process incl. start, HIP initialisation and index "
                                "build (about 0.75 s of it); reads = 2 x pairs; the output holds a row of 36 + 36 BAM fields per "
                                "pair" % pairs}
    finally:
        subprocess.run(["rm", "-rf", d])


def native_leg(nim, synth, torch, lib_obj, seqs, devices, n, L, n_sets, warmup, steps):
    """The multi-GPU step through the C ABI alone (include/nimble_hip.h nimble_comm_* / nimble_steps_*, RCCL inside the
    device library; include/nimble_host.h nimble_multi_steps): ONE process, one native host thread per device, reads
    resident on their devices.  The table of the last step must equal ONE call over the union of the ranks' reads of that
    step (run here on the first device).  Returns a dict for the JSON line."""
    W = len(devices)
    libs = [nim.Library(text=json.dumps(lib_obj), strand_filter="unstranded").build_index(d) for d in devices]
    sets = [[synth.make_reads_torch(seqs, n, L=L, seed=synth.READ_SEED + 7919 * r + 104729 * k, device="cuda:%d" % devices[r])
             for k in range(n_sets)] for r in range(W)]
    for d in sorted(set(devices)):
        torch.cuda.synchronize(d)
    ptrs = [[t.data_ptr() for t in rs] for rs in sets]
    ms, rccl, rows = nim.multi_steps(libs, devices, ptrs, None, n, L, warmup, steps)
    got = rows.to_list()
    last = (warmup + steps - 1) % n_sets
    union = torch.cat([sets[r][last].to("cuda:%d" % devices[0]) for r in range(W)], dim=0).contiguous()
    torch.cuda.synchronize(devices[0])
    want = libs[0].score_call_raw(union, None, None, None, n=W * n, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE).to_list()
    del union
    return {"ms_per_step": ms, "reads_per_s": n * W / (ms / 1000.0), "ranks": W, "devices": list(devices), "rccl": rccl,
            "rows": len(got),
            "parity_on_union": ("bit-exact table (%d rows) vs one call over the %d reads of all ranks" % (len(got), W * n))
            if got == want else "MISMATCH",
            "note": "one process, one native host thread per rank: pack + route of step i ahead of the call of step i-1, "
                    "all-to-all of step i (RCCL send/recv inside libnimble_hip.so) beside it, rows + count all-reduce of "
                    "step i-2 on the host meanwhile; fill and drain of the pipeline inside the timed steps"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="configs2")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (0 = the workload's own size)")
    ap.add_argument("--features", type=int, default=0, help="library features (0 = the workload's own size)")
    ap.add_argument("--read-sets", type=int, default=3, help="distinct read sets rotated through the steps")
    ap.add_argument("--depth", type=int, default=2, choices=(1, 2),
                    help="calls in flight at N=1: 2 = step i+1 runs on the GPU while the host finishes step i")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="multi-GPU: run every step to completion before the next one starts (no overlap of the "
                         "exchange with the previous call)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="rehearsal: run the multi-GPU step (route, all-to-all, reduce over RCCL) even with one rank")
    ap.add_argument("--form", choices=("sharded", "local", "native"), default=os.environ.get("NIMBLE_MULTI_GPU_FORM", "sharded"),
                    help="multi-GPU step: 'sharded' moves the packed reads to the key's owner, 'local' aligns where "
                         "the reads are and moves only keys and verdict bytes (both: one process per GPU, torch.distributed "
                         "for the collectives); 'native' = ONE process, one native host thread per GPU, the C ABI's own "
                         "RCCL collectives (launch without torch.distributed.run: python bench.py --gpus N --form native)")
    ap.add_argument("--native", type=int, default=1,
                    help="multi-GPU under torch.distributed.run: after the torch form, rank 0 also runs the native form over "
                         "all N devices (0 = skip); the line's value is the faster of the forms that ran and passed parity")
    ap.add_argument("--native-child", type=int, default=0,
                    help="(internal) run ONLY the native form over that many devices and print its JSON object: how rank 0 of "
                         "a torch.distributed.run launch starts it, as a child process with a time limit")
    ap.add_argument("--native-timeout", type=float, default=600.0, help="time limit of that child, seconds")
    ap.add_argument("--native-virtual", type=int, default=0,
                    help="rehearsal of the native form on ONE GPU: that many ranks share device 0 (device copies instead of RCCL)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000,
                    help="reads timed on the CPU oracle, all host cores (0 = skip); about 11 s at the default")
    ap.add_argument("--cpu-threads", type=int, default=0, help="oracle threads (0 = the host CPUs this process is granted, at most 64)")
    ap.add_argument("--packed-input", type=int, default=1,
                    help="also time the steps with the reads handed over as 2-bit words (0 = skip); an extra key, not `value`")
    ap.add_argument("--e2e-reads", type=int, default=16_000_000,
                    help="reads of the end-to-end FASTQ runs at N=1, plain and .gz (0 = skip)")
    ap.add_argument("--e2e-bam-pairs", type=int, default=1_000_000,
                    help="read pairs of the end-to-end BAM run at N=1 (0 = skip); skipped with --e2e-reads 0")
    args = ap.parse_args()
    # under a profiler the end-to-end leg is left out: it writes a FASTQ file, gzips it and starts lib/nimble children that
    # inherit the profiler's preload (their kernels would land in the counter files, and the run outlasts its timeout)
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        args.e2e_reads = 0

    # `python bench.py --gpus N` with N > 1 and no launcher around it: the ranks are started here, as fresh child processes
    # of torch.distributed.run, BEFORE this process imports torch or touches a GPU (a process that has initialised the GPU
    # must not exec or be replaced; a child is a new process).  Rank 0's JSON line comes through on stdout, the exit code
    # is the launcher's.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.form != "native" and not args.native_child:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (counting the devices does not initialise the GPU; a node with fewer devices than ranks gets one line that says so
        # instead of N tracebacks)
        try:
            import torch
            have = torch.cuda.device_count()
        except Exception:
            have = -1
        if 0 <= have < args.gpus:
            print(json.dumps({"error": "bench.py --gpus %d: this node shows %d device(s)" % (args.gpus, have), "n_gpus": args.gpus}))
            raise SystemExit(2)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

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
    native_only = args.form == "native" and world == 1
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and not native_only:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d "
                             "(or with --form native as one process)" % (args.gpus, args.gpus))
    if native_only:
        args.form = "sharded"  # (the single-GPU part of this run -- counters, roofline -- is the ordinary one)
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
    ctrl = dist.new_group(backend="gloo") if world > 1 else None  # host-side waits (an RCCL barrier would spin on the GPUs)

    wl = WORKLOADS[args.workload]
    L = 150
    T = args.features or wl["features"]
    job_ranks = world if not native_only else max(args.gpus, args.native_virtual, 1)
    n = args.reads or (wl["reads"] // job_ranks if wl["total"] else wl["reads"])
    paired = wl["paired"]
    if paired and sharded:
        raise SystemExit("bench.py: the paired-end workload is a single-GPU configuration (BASELINE.json configs[3])")
    if T < wl["family"] or T % wl["family"]:
        raise SystemExit("bench.py: --features must be a multiple of the workload's family size (%d)" % wl["family"])
    if wl["family"] > 4:
        names, seqs = synth.make_family_library(T, wl["family"])
    else:
        names, seqs = synth.make_library(T)
    lib_obj = synth.library_json(names, seqs)
    if args.workload == "configs3":  # mismatch.json settings with num_mismatches 2 (tests/mismatch.rs:45, basic-cases.rs:119)
        lib_obj[0].update(score_percent=0.08, score_threshold=12, num_mismatches=2)
    n_sets = max(1, args.read_sets)
    if args.native_child:
        devs = [0] * args.native_virtual if args.native_virtual else list(range(args.native_child))
        print(json.dumps(native_leg(nim, synth, torch, lib_obj, seqs, devs, n, L, n_sets, max(args.warmup, 3), args.steps)),
              flush=True)
        return
    lib = nim.Library(text=json.dumps(lib_obj), strand_filter="unstranded").build_index(local_rank)
    ctx = lib.device_context()
    if paired:
        sets = []
        for k in range(n_sets):
            sets.append(synth.make_pairs_torch(seqs, n, L=L, seed=synth.READ_SEED + 7919 * rank + 104729 * k,
                                               device=str(device)))
    else:
        sets = [(synth.make_reads_torch(seqs, n, L=L, seed=synth.READ_SEED + 7919 * rank + 104729 * k, device=str(device)),
                 None) for k in range(n_sets)]
    torch.cuda.synchronize()
    reducer = nd.TableReducer(device) if sharded else None

    def step(k=0):
        """One score::call: ends with the sorted rows materialised on the host (C++ result object)."""
        r1, r2 = sets[k % n_sets]
        if sharded:
            # pack -> route by key hash (device kernels) -> all-to-all of the records -> finish per rank
            # -> all-reduce of the counts over the agreed callset table
            return nd.sharded_step(lib, r1, r2, n, L, device, reducer)
        return lib.score_call_raw(r1, None, r2, None, n=n, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE)

    # one untimed pass with the work counters on: gives P / U / sum(E) for the algorithmic byte count (read set 0; the
    # sets are draws of one recipe and agree to a fraction of a percent)
    ctx.set_counters(True)
    rows = step(0)
    ctx.n = n
    counters = ctx.counters()   # multi-GPU: of the shard this rank received (statistically the same reads)
    ctx.set_counters(False)

    depth = args.depth if not sharded else 1
    ctxs = [lib.device_context(s) for s in range(depth)]
    for c in ctxs:
        c.set_counters(False)  # the work counters cost ~30 % of k_align; collected once, above

    def begin(slot, k):
        r1, r2 = sets[k % n_sets]
        lib.score_call_begin(slot, r1, None, r2, None, n=n, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE)

    def end(slot):
        r = lib.score_call_end(slot, raw=True)
        for k, v in ctxs[slot].timing().items():
            stage[k] += v
        return r

    stage = {k: 0.0 for k in ("pack", "align", "intern", "dedup", "count", "total")}
    for w in range(args.warmup):
        rows = step(w)
    if depth > 1:
        for s in range(depth):  # the second slot allocates its buffers on first use
            begin(s, s)
        for s in range(depth):
            end(s)
    pipe = None
    if sharded and not args.no_pipeline:
        pipe = (nd.LocalAlignPipeline if args.form == "local" else nd.ShardedPipeline)(lib, device, reducer)
        for s_ in range(4 if args.form == "local" else 3):
            lib.device_context(s_).set_counters(False)
        for w in range(max(args.warmup, 3)):   # allocates the other call slots and the utility context
            pipe.submit(sets[w % n_sets][0], sets[w % n_sets][1], n, L)
        for r in pipe.flush():
            rows = r
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    stage = {k: 0.0 for k in stage}
    marks = []  # host time at which each step's rows were in hand
    t0 = time.perf_counter()
    if sharded and not args.no_pipeline:
        # software-pipelined multi-GPU steps: K submits + the drain all end inside the timed region
        for i in range(args.steps):
            r = pipe.submit(sets[i % n_sets][0], sets[i % n_sets][1], n, L)
            if r is not None:
                rows = r
                marks.append(time.perf_counter())
        for r in pipe.flush():
            rows = r
            marks.append(time.perf_counter())
        for k, v in ctx.timing().items():
            stage[k] += v * args.steps
    elif depth == 1:
        for i in range(args.steps):
            rows = step(i)
            marks.append(time.perf_counter())
            for k, v in ctx.timing().items():
                stage[k] += v
    else:
        # every step is a complete score::call (begin + end); two are in flight, all K end inside the timed region
        for i in range(args.steps):
            begin(i % 2, i)
            if i:
                rows = end((i - 1) % 2)
                marks.append(time.perf_counter())
        if args.steps:
            rows = end((args.steps - 1) % 2)
            marks.append(time.perf_counter())
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
    units = n * world * steps
    value = units / elapsed
    # device time per call: the event span of one call, unless calls overlap on the device (then the step time)
    device_ms = stage["total"] if (depth == 1 and not sharded) else 1000.0 * elapsed / steps
    # per-step host intervals (completion to completion; with calls in flight the first one carries the fill)
    per = [1000.0 * (b - a) for a, b in zip([t0] + marks[:-1], marks)]
    steady = per[1:] if len(per) > 2 else per

    # ---- multi-GPU (torch form): the merged table of the LAST step against ONE call over the union of the ranks' reads of
    # that step, run on rank 0 (after the timed region; the native form checks itself the same way).  A failure here is
    # reported in the line, it never takes the line down.
    union_parity = None
    if sharded and not paired:
        try:
            mine = sets[(args.steps - 1) % n_sets][0]
            parts = [mine]
            if world > 1:
                parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
                dist.gather(mine, parts, dst=0)
            if rank == 0:
                union = torch.cat(parts, dim=0).contiguous() if world > 1 else mine
                torch.cuda.synchronize()
                want = lib.score_call_raw(union, None, None, None, n=world * n, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE).to_list()
                got = reducer.rows(*rows)
                norm = lambda t: sorted((tuple(f), int(c)) for f, c in t)
                union_parity = ("bit-exact table (%d rows) vs one call over the %d reads of all ranks" % (len(got), world * n)
                                if norm(got) == norm(want) else "MISMATCH")
                del union
        except Exception as ex:
            union_parity = "not checked: %s" % str(ex)[:200]

    fam = T // wl["family"]
    out = {
        "metric": "reads/sec aligned (whole job), counts bit-exact vs the CPU path",
        "value": value * (2 if paired else 1),
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / steps,
        "higher_is_better": True,
        "scaling": "weak" if not wl["total"] else "strong",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": wl["text"] % dict(n=n, T=T, rows=2 * T, fam=fam),
            "workload_id": args.workload,
            "reads_per_gpu": n, "read_len": L, "features": T, "paired": paired,
            "read_sets_rotated": n_sets,
            "parallelism": (("1 process/GPU; reads aligned where they are, keys to their owner by hash (all-to-all), "
                             "verdict bytes back (all-to-all) + count all-reduce (RCCL); key exchange of step i "
                             "beside its own alignment" if args.form == "local" and not args.no_pipeline else
                             "1 process/GPU; packed reads routed by key hash (all-to-all) + count all-reduce (RCCL)"
                             + ("" if args.no_pipeline else "; exchange of step i overlaps the call of step i-1")))
                           if sharded else "single GPU",
            "rows": len(rows) if not sharded else len(reducer.rows(*rows)),
            "calls_in_flight": depth,
        },
        # (the pipelined multi-GPU form hands steps back in bursts -- the drain returns two at once -- so its smallest
        # interval says nothing; it is left out there)
        "rccl": bool(sharded and dist.get_backend() == "nccl") if sharded else None,
        "parity_on_union": union_parity,
        "step_ms": {"min": None if (sharded and not args.no_pipeline) else min(steady),
                    "median": statistics.median(steady), "max": max(steady),
                    "note": "host interval between successive completed steps (the first, which carries the pipeline fill, "
                            "left out)"} if steady else None,
        "stage_ms": {k: round(v, 4) for k, v in stage.items()},
        "stage_note": ("per call, HIP events; with calls in flight the dedup / count / compaction of call i run on a "
                       "side stream beside the pack of call i+1, so pack and dedup read longer than alone and 'total' is "
                       "the latency of one call, not the time per call"),
        "device_reads_per_s": n / (device_ms / 1000.0) if device_ms > 0 else None,
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (k_align): algorithmic bytes of one launch / its duration
        P, U, E = counters["probes"], counters["nodes"], counters["class_entries"]
        hit = counters["seeded"]
        nm = 2 if paired else 1
        key_bytes = 8 * ((nm * L + 31) // 32)
        n_call = int(counters["reads"]) if sharded else n   # reads of the launch the counters describe
        align_bytes = n_call * key_bytes + 16 * P + hit * key_bytes + 16 * U + 4 * E + 16 * n_call * nm
        pipe_bytes = n_call * L * nm + 16 * P + hit * (key_bytes + 96) + 16 * U + 4 * E + 16 * n_call * nm
        align_s = stage["align"] / 1000.0
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if (tj.get("reads") == n and tj.get("features") == T and tj.get("workload", "configs2") == args.workload
                        and not sharded):
                    traffic = tj.get("k_align_hbm_bytes_per_launch")
                    traffic_source = ("replayed from %s (rocprofv3 --pmc pass of this command at this size: %s); PMC "
                                      "counters cannot be collected inside a timed run" %
                                      ("profiles/traffic_latest.json", tj.get("source", "see profiles/")))
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
            "traffic_source": traffic_source,
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
        if traffic:
            # what the kernel really moved on the memory side of L2 (Infinity-Cache hits included), per second and against
            # the bytes the algorithm needs: traffic well above 1 x algorithmic = lines fetched more than once
            out["roofline"]["traffic_GBps"] = traffic / align_s / 1e9
            out["roofline"]["traffic_over_algorithmic"] = traffic / max(align_bytes, 1)
        if out["roofline"]["frac"] > 1.0:
            # the SURVEY 8(d) formula prices what the REFERENCE algorithm reads: every colour-list entry of every visited
            # class (4 * class_entries).  With allele families of 100 a class has ~100 entries and a read visits ~120 of
            # them; the kernel intersects 64-row bitmap words instead and never reads those lists.
            out["roofline"]["note"] = ("frac above 1: the algorithmic bytes follow the reference's formula (4 B per colour-list "
                                       "entry of every visited class, %d entries per read here), which this kernel does not "
                                       "read (row bitmaps in a register window); without that term the figure is %.3f"
                                       % (E // max(n_call, 1), (align_bytes - 4 * E) / align_s / 1e9 / HBM_PEAK_GBPS))
        # ---- CPU baseline: the oracle (a port of the reference path) on a bounded sample of the same reads
        if args.cpu_sample > 0 and world == 1:
            from oracle import oracle as ora
            S = min(args.cpu_sample, n)
            r1, r2 = sets[0]
            sample = r1[:S].cpu().numpy()
            sample2 = r2[:S].cpu().numpy() if paired else None
            cols = [["s"] * len(names), names, [str(len(s)) for s in seqs], seqs]
            ref = ora.Reference.from_columns(lib_obj[1]["headers"], cols, "")
            cfg = ora.config_from_json(lib_obj[0], len(names), "unstranded")
            oidx = ora.Index.from_reference(ref)
            threads = args.cpu_threads or min(usable_cpus(), 64)
            offs = synth.fixed_offsets(S, L)
            t1 = time.perf_counter()
            ores = ora.call(oidx, ref, cfg, sample.reshape(-1), offs, None if not paired else sample2.reshape(-1),
                            None if not paired else offs, n_threads=threads)
            cpu_s = time.perf_counter() - t1
            S1 = min(S, 1_000_000)
            t1 = time.perf_counter()
            ora.call(oidx, ref, cfg, sample[:S1].reshape(-1), synth.fixed_offsets(S1, L),
                     None if not paired else sample2[:S1].reshape(-1),
                     None if not paired else synth.fixed_offsets(S1, L), n_threads=1)
            cpu1_s = time.perf_counter() - t1
            # parity on the sample: the GPU table must equal the oracle's table
            got = lib.score_call(r1[:S].contiguous(), None, None if not paired else r2[:S].contiguous(), None, n=S, fixed_len=L,
                                 max_len=L, mem=nim.MEM_DEVICE)
            parity = [(f, c) for f, c in got] == [(f, c) for f, c in ores.rows]
            if not parity:
                raise SystemExit("bench.py: GPU table differs from the CPU oracle on the sample")
            per_unit = 2 if paired else 1
            out["cpu_baseline"] = {
                "value": per_unit * S / cpu_s, "unit": "reads/s", "cores": threads, "kind": "port",
                "sample": "first %d of the %d reads(-pairs) of read set 0, oracle/libnimble_oracle.so, %d threads "
                          "(hash-partitioned by key); index build and read generation excluded" % (S, n, threads),
                "single_thread_value": per_unit * S1 / cpu1_s,
                "parity_on_sample": "bit-exact table (%d rows)" % len(got),
            }
            out["speedup_vs_cpu_baseline"] = out["value"] / (per_unit * S / cpu_s)
        # ---- the same steps with the reads handed over already packed (what score::call itself receives: DnaStrings,
        # src/score.rs:14-31; the ASCII -> 2-bit conversion is the reader's, src/parse/fastq.rs:32): an extra figure, not
        # `value`, which keeps the ASCII boundary of the earlier rounds
        if args.packed_input and world == 1 and not paired and not sharded and depth > 1:
            W = (L + 31) // 32
            lut = torch.zeros(256, dtype=torch.int64, device=device)
            for ch, code in ((b"C", 1), (b"G", 2), (b"T", 3), (b"c", 1), (b"g", 2), (b"t", 3)):
                lut[ch[0]] = code
            shifts = (62 - 2 * torch.arange(32, device=device, dtype=torch.int64))
            wsets = []
            for r1, _ in sets:
                words = torch.zeros((n, W), dtype=torch.int64, device=device)
                for lo in range(0, n, 1 << 20):
                    hi = min(n, lo + (1 << 20))
                    codes = torch.zeros((hi - lo, W * 32), dtype=torch.int64, device=device)
                    codes[:, :L] = lut[r1[lo:hi].long()]
                    words[lo:hi] = (codes.view(hi - lo, W, 32) << shifts).sum(dim=2)  # disjoint bit fields: the sum is their OR
                    del codes
                wsets.append(words)
            lens = torch.full((n,), L, dtype=torch.int32, device=device)
            torch.cuda.synchronize()

            def begin_w(slot, k):
                lib.score_call_begin_words(slot, wsets[k % n_sets], lens, W, n=n, max_len=L, mem=nim.MEM_DEVICE)

            for s_ in range(2):
                begin_w(s_, s_)
            rows_w = None
            for s_ in range(2):
                rows_w = lib.score_call_end(s_, raw=True)
            torch.cuda.synchronize()
            tw = time.perf_counter()
            for i in range(args.steps):
                begin_w(i % 2, i)
                if i:
                    rows_w = lib.score_call_end((i - 1) % 2, raw=True)
            if args.steps:
                rows_w = lib.score_call_end((args.steps - 1) % 2, raw=True)
            torch.cuda.synchronize()
            tw = time.perf_counter() - tw
            if rows_w is not None and rows is not None and hasattr(rows, "signature") and (
                    rows_w.signature() != rows.signature() or not np.array_equal(rows_w.counts(), rows.counts())):
                raise SystemExit("bench.py: the table from packed input differs from the table from ASCII input")
            out["packed_input"] = {
                "reads_per_s": n * max(args.steps, 1) / tw, "ms_per_step": tw * 1000.0 / max(args.steps, 1),
                "note": "the same %d steps with the reads handed over as 2-bit words (32 bases a u64, the layout of the "
                        "reference's DnaString, which is what score::call receives): k_pack_words instead of k_pack; the "
                        "table equals the ASCII run's" % args.steps}
            del wsets
        # ---- end to end from a FASTQ file (SURVEY 8(d) figure B), outside the timed value
        if args.e2e_reads > 0 and world == 1 and not paired and not args.force_sharded:
            try:
                m = min(args.e2e_reads, n)
                out.update(e2e_fastq(names, seqs, sets[0][0][:m].cpu().numpy(), m, m))
            except Exception as ex:  # the bench line must not die with the side measurement
                out["e2e_error"] = str(ex)[:300]
            if args.e2e_bam_pairs > 0 and args.workload == "configs2":
                try:
                    out.update(e2e_bam(args.e2e_bam_pairs))
                except Exception as ex:
                    out["e2e_bam_error"] = str(ex)[:300]
    if rank == 0 and getattr(nd, "_TIMING", None):
        k = max(nd._TIMING.get("steps", 1), 1)
        out["sharded_phase_ms"] = {a: round(b / k, 3) for a, b in nd._TIMING.items() if a != "steps"}
    if rank == 0:
        # ---- the native form of the multi-GPU step (C ABI collectives, one process): at N > 1 beside the torch form, or
        # alone (python bench.py --gpus N --form native); never for the paired workload (a single-GPU configuration)
        want_native = not paired and ((native_only and (args.gpus > 1 or args.native_virtual or args.force_sharded)) or
                                      (world > 1 and args.native))
        if want_native:
            devs = [0] * args.native_virtual if args.native_virtual else list(range(args.gpus if native_only else world))
            try:
                if world > 1:
                    # a process of its own with a time limit: a fault or a hang inside the second form (its RCCL path has
                    # its first N-device run in exactly this place) must not take the measured line of the first with it
                    env = {k: v for k, v in os.environ.items()
                           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                        "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
                    cmd = [sys.executable, os.path.abspath(__file__), "--native-child", str(world), "--gpus", str(world),
                           "--form", "native", "--workload", args.workload, "--reads", str(n), "--features", str(T),
                           "--read-sets", str(n_sets), "--steps", str(args.steps), "--warmup", str(args.warmup)]
                    cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.native_timeout)
                    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                    nat = json.loads(lines[-1]) if cp.returncode == 0 and lines else {
                        "error": "native child rc %d: %s" % (cp.returncode, cp.stderr.strip()[-300:])}
                else:
                    nat = native_leg(nim, synth, torch, lib_obj, seqs, devs, n, L, n_sets, max(args.warmup, 3), args.steps)
            except subprocess.TimeoutExpired:
                nat = {"error": "native child: no result within %.0f s (stopped)" % args.native_timeout}
            except Exception as ex:  # the line of the torch form must not die with the second form
                nat = {"error": str(ex)[:400]}
            out["native"] = nat
            ok = "error" not in nat and nat.get("parity_on_union", "").startswith("bit-exact")
            if ok and (native_only or nat["reads_per_s"] > out["value"]):
                out["torch_form"] = None if native_only else {"value": out["value"], "ms_per_step": out["ms_per_step"]}
                out["value"] = nat["reads_per_s"]
                out["ms_per_step"] = nat["ms_per_step"]
                out["n_gpus"] = len(devs)
                out["config"]["parallelism"] = ("ONE process, one native host thread per GPU; packed reads routed by key hash, "
                                                "all-to-all + count all-reduce over RCCL inside the C ABI (nimble_steps_*); "
                                                "exchange of step i beside the call of step i-1")
                out["config"]["reads_per_gpu"] = n
        print(json.dumps(out), flush=True)
    if ctrl is not None:
        dist.barrier(group=ctrl)  # the other ranks wait here (on the host) while rank 0 runs the native form
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
