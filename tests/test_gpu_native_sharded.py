"""The multi-GPU path of the C ABI (include/nimble_hip.h nimble_comm_* / nimble_sharded_*) and the C++ pipeline over it
(process::fastq::process_sharded, `lib/nimble -d 0,1,...`): one process, one rank = one host thread per device, no torch.
On a one-GPU box the ranks share the device (devices = [0, 0, ...]: records and counts move by device copies) and a
communicator of ONE rank runs the real RCCL calls; every configuration must write the single-GPU table, which equals the
CPU oracle's."""
import ctypes as C
import importlib
import os
import subprocess
import threading

import numpy as np
import pytest

from oracle import oracle as ora

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")


@pytest.fixture(scope="module")
def lib_and_reads(tmp_path_factory):
    d = tmp_path_factory.mktemp("sharded")
    names, seqs = synth.make_library(48)
    path = str(d / "lib.json")
    synth.write_library(path, names, seqs)
    r1, r2 = synth.make_reads(seqs, 60_000, paired=True, seed=77)
    # copies of earlier reads far apart in the file: they land in different batches and on different ranks, and must
    # still count once (the dedup scope is the whole call)
    r1[50_000:50_400], r2[50_000:50_400] = r1[100:500], r2[100:500]
    f1, f2 = str(d / "r1.fastq"), str(d / "r2.fastq")
    synth.write_fastq(f1, r1)
    synth.write_fastq(f2, r2)
    return path, names, seqs, r1, r2, f1, f2, d


def oracle_tsv(path, strand, r1, r2=None):
    import json
    obj = json.load(open(path))
    names, seqs = obj[1]["columns"][1], obj[1]["columns"][3]
    cols = [obj[1]["columns"][0], names, obj[1]["columns"][2], seqs]
    ref = ora.Reference.from_columns(obj[1]["headers"], cols, obj[0].get("group_on", ""))
    cfg = ora.config_from_json(obj[0], len(names), strand)
    o = synth.fixed_offsets(r1.shape[0], r1.shape[1])
    res = ora.call(ora.Index.from_reference(ref), ref, cfg, r1.reshape(-1), o, None if r2 is None else r2.reshape(-1),
                   None if r2 is None else o)
    return "feature\tscore\n" + "".join("\t".join(f) + "\t%d\n" % c for f, c in res.rows)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("paired", [False, True])
def test_sharded_fastq_pipeline_equals_one_call(lib_and_reads, devices, paired, monkeypatch):
    path, names, seqs, r1, r2, f1, f2, d = lib_and_reads
    monkeypatch.setenv("NIMBLE_FASTQ_BATCH", "8192")        # several rounds, ragged last one
    want = oracle_tsv(path, "unstranded", r1, r2 if paired else None)
    lib = nim.Library(path, "unstranded")
    out = str(d / ("out_%d_%d.tsv" % (len(devices), int(paired))))
    if os.path.exists(out):
        os.remove(out)
    nim.fastq_process_sharded([f1, f2] if paired else [f1], lib, devices, out)
    assert open(out).read() == want


def test_cli_device_list(lib_and_reads):
    path, names, seqs, r1, r2, f1, f2, d = lib_and_reads
    exe = os.path.join(ROOT, "nimble-aligner_amd", "lib", "nimble")
    out = str(d / "cli.tsv")
    cp = subprocess.run([exe, "-r", path, "-o", out, "-i", f1, "-f", "unstranded", "-d", "0,0"], capture_output=True,
                        text=True, timeout=300, env=dict(os.environ, NIMBLE_FASTQ_BATCH="16384", NIMBLE_HOST_TIMING="1"))
    assert cp.returncode == 0, cp.stderr
    assert "2 ranks" in cp.stderr
    assert open(out).read() == oracle_tsv(path, "unstranded", r1)


def test_collectives_of_the_c_abi_with_three_ranks_on_one_device():
    """nimble_counts_allreduce (device and host form) and nimble_records_alltoall called as a Rust host would: one
    thread per rank."""
    torch = pytest.importorskip("torch")
    L = nim.hip_lib()
    W = 3
    comm = C.c_void_p()
    dev = (C.c_int * W)(0, 0, 0)
    assert L.nimble_comm_create(dev, W, C.byref(comm)) == 0
    assert L.nimble_comm_size(comm) == W and L.nimble_comm_uses_rccl(comm) == 0
    rw = 7
    rng = np.random.default_rng(3)
    counts = rng.integers(0, 50, size=(W, W)).astype(np.uint64)      # [src][dst]
    send = [rng.integers(0, 1 << 60, size=(int(counts[s].sum()), rw)).astype(np.uint64) for s in range(W)]
    vecs = [rng.integers(0, 1000, size=257).astype(np.int64) for _ in range(W)]
    got, errs = {}, []

    def rank(r):
        try:
            s = torch.cuda.Stream()
            v = torch.from_numpy(vecs[r].copy()).to("cuda:0")
            torch.cuda.synchronize()
            assert L.nimble_counts_allreduce(comm, r, v.data_ptr(), v.numel(), s.cuda_stream) == 0
            hv = vecs[r].copy()
            assert L.nimble_counts_allreduce_host(comm, r, hv.ctypes.data, hv.size) == 0
            sd = torch.from_numpy(send[r].view(np.int64).copy()).to("cuda:0")
            cap = int(counts[:, r].sum())
            rv = torch.zeros((max(cap, 1), rw), dtype=torch.int64, device="cuda:0")
            torch.cuda.synchronize()
            n = C.c_uint64(0)
            cr = counts[r].copy()
            assert L.nimble_records_alltoall(comm, r, sd.data_ptr(), cr.ctypes.data, rw, rv.data_ptr(), cap, C.byref(n),
                                             s.cuda_stream) == 0
            s.synchronize()
            got[r] = (v.cpu().numpy(), hv, rv[:n.value].cpu().numpy().view(np.uint64), n.value)
        except Exception as e:               # noqa: BLE001
            errs.append((r, e))

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(W)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    total = sum(vecs)
    for r in range(W):
        v, hv, rv, n = got[r]
        np.testing.assert_array_equal(v, total)
        np.testing.assert_array_equal(hv, total)
        parts = []
        for s in range(W):
            lo = int(counts[s, :r].sum())
            parts.append(send[s][lo:lo + int(counts[s, r])])
        np.testing.assert_array_equal(rv, np.concatenate(parts))
        assert n == int(counts[:, r].sum())
    L.nimble_comm_free(comm)


def test_rccl_communicator_of_one_rank():
    """the real RCCL calls (all-reduce, grouped send/recv to oneself) on the one device this box has"""
    torch = pytest.importorskip("torch")
    L = nim.hip_lib()
    comm = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert L.nimble_comm_create(dev, 1, C.byref(comm)) == 0
    assert L.nimble_comm_uses_rccl(comm) == 1
    s = torch.cuda.Stream()
    v = torch.arange(1000, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    assert L.nimble_counts_allreduce(comm, 0, v.data_ptr(), v.numel(), s.cuda_stream) == 0
    s.synchronize()
    assert torch.equal(v.cpu(), torch.arange(1000, dtype=torch.int64))
    sd = torch.arange(35, dtype=torch.int64, device="cuda:0")
    rv = torch.zeros(35, dtype=torch.int64, device="cuda:0")
    cnt = np.array([5], dtype=np.uint64)
    n = C.c_uint64(0)
    torch.cuda.synchronize()
    assert L.nimble_records_alltoall(comm, 0, sd.data_ptr(), cnt.ctypes.data, 7, rv.data_ptr(), 5, C.byref(n),
                                     s.cuda_stream) == 0
    s.synchronize()
    assert n.value == 5 and torch.equal(rv, sd)
    L.nimble_comm_free(comm)


def _write_ragged_fastq(path, reads, lens):
    with open(path, "wb") as f:
        for i in range(reads.shape[0]):
            L = int(lens[i])
            f.write(b"@r%d\n" % i + reads[i, :L].tobytes() + b"\n+\n" + b"I" * L + b"\n")


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_longer_read_in_a_later_batch_widens_the_kept_records(lib_and_reads, devices, monkeypatch):
    """Batch 1 holds reads of at most 96 bases, later batches reads of up to 150: the sharded call was opened for 96, every
    rank widens the records it has kept (nimble_sharded_grow) and the run goes on -- the table is the one of a single call
    over all reads.  (Round 2 re-opened the call here and died on 'the context holds a call in flight'.)"""
    path, names, seqs, r1, r2, f1, f2, d = lib_and_reads
    monkeypatch.setenv("NIMBLE_FASTQ_BATCH", "4096")
    n = 20_000
    reads = r1[:n]
    rng = np.random.default_rng(5)
    lens = np.where(np.arange(n) < 6000, rng.integers(60, 97, size=n), rng.integers(60, 151, size=n))
    lens[17_000] = 150
    f = str(d / "ragged.fastq")
    _write_ragged_fastq(f, reads, lens)
    # oracle over the ragged reads
    import json
    obj = json.load(open(path))
    cols = [obj[1]["columns"][0], obj[1]["columns"][1], obj[1]["columns"][2], obj[1]["columns"][3]]
    ref = ora.Reference.from_columns(obj[1]["headers"], cols, obj[0].get("group_on", ""))
    cfg = ora.config_from_json(obj[0], len(cols[1]), "unstranded")
    flat = np.concatenate([reads[i, :lens[i]] for i in range(n)])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    res = ora.call(ora.Index.from_reference(ref), ref, cfg, flat, off)
    want = "feature\tscore\n" + "".join("\t".join(fe) + "\t%d\n" % c for fe, c in res.rows)
    lib = nim.Library(path, "unstranded")
    out = str(d / ("ragged_%d.tsv" % len(devices)))
    if os.path.exists(out):
        os.remove(out)
    nim.fastq_process_sharded([f], lib, devices, out)
    assert open(out).read() == want
    # and the single-device pipeline agrees
    out1 = str(d / "ragged_single.tsv")
    if os.path.exists(out1):
        os.remove(out1)
    nim.fastq_process([f], [nim.Library(path, "unstranded").build_index(0)], [out1])
    assert open(out1).read() == want


def test_a_bad_argument_on_one_rank_reaches_every_rank():
    """nimble_records_alltoall with a NULL count table on rank 1: rank 1 gets its error, the other ranks get 'a rank
    failed' -- and nobody is left standing in a barrier (round 2: the bad rank returned before the first agree())."""
    torch = pytest.importorskip("torch")
    L = nim.hip_lib()
    W = 3
    comm = C.c_void_p()
    dev = (C.c_int * W)(0, 0, 0)
    assert L.nimble_comm_create(dev, W, C.byref(comm)) == 0
    rcs = {}

    def rank(r):
        s = torch.cuda.Stream()
        sd = torch.zeros(8 * 7, dtype=torch.int64, device="cuda:0")
        rv = torch.zeros(64 * 7, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        cnt = np.array([1, 1, 1], dtype=np.uint64)
        n = C.c_uint64(0)
        rcs[r] = L.nimble_records_alltoall(comm, r, sd.data_ptr(), None if r == 1 else cnt.ctypes.data, 7, rv.data_ptr(), 64,
                                           C.byref(n), s.cuda_stream)

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(W)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in ts), "a rank is still waiting for a peer that failed"
    assert rcs[1] != 0 and rcs[0] != 0 and rcs[2] != 0
    L.nimble_comm_free(comm)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("paired", [False, True])
def test_pipelined_native_steps_equal_one_call_over_the_union(lib_and_reads, devices, paired):
    """nimble_steps_* through process::multi::run_steps (include/nimble_host.h nimble_multi_steps): successive calls over
    device-resident read sets spread across the ranks, one native thread per rank, pack + route of batch b ahead of the
    call of b-1, the exchange of b beside it.  The table of the last step is the table of ONE call over the union of the
    ranks' reads of that step's set -- at world 1 over real RCCL (ncclSend/ncclRecv to oneself), at world 2 and 3 with
    virtual ranks on this one device.  Read sets differ per step, so a pipeline that mixed batches up would show."""
    torch = pytest.importorskip("torch")
    path, names, seqs, r1, r2, f1, f2, d = lib_and_reads
    W = len(devices)
    n_sets, n = 3, 6000
    libs = [nim.Library(path, "unstranded").build_index(0) for _ in range(W)]
    sets1, sets2, keep = [], [], []
    for r in range(W):
        p1, p2 = [], []
        for s_ in range(n_sets):
            lo = (r * n_sets + s_) * n
            a = torch.from_numpy(np.ascontiguousarray(r1[lo:lo + n])).to("cuda:0")
            b = torch.from_numpy(np.ascontiguousarray(r2[lo:lo + n])).to("cuda:0")
            keep += [a, b]
            p1.append(a.data_ptr())
            p2.append(b.data_ptr())
        sets1.append(p1)
        sets2.append(p2)
    torch.cuda.synchronize()
    warmup, steps = 2, 5
    ms, rccl, rows = nim.multi_steps(libs, devices, sets1, sets2 if paired else None, n, 150, warmup, steps)
    assert rccl == (W == 1) and ms > 0
    last_set = (warmup + steps - 1) % n_sets
    sel = np.concatenate([np.arange((r * n_sets + last_set) * n, (r * n_sets + last_set + 1) * n) for r in range(W)])
    want = oracle_tsv(path, "unstranded", r1[sel], r2[sel] if paired else None)
    got = "feature\tscore\n" + "".join("\t".join(f) + "\t%d\n" % c for f, c in rows.to_list())
    assert got == want


def test_rank_threads_that_cannot_be_started_are_an_error_not_an_abort(lib_and_reads):
    """Round 3's native driver ended the process (std::terminate) when a rank failed; a thread the system refuses for one
    of the ranks must come back as an ordinary error of nimble_multi_steps / the sharded FASTQ pipeline -- all ranks or
    none (csrc/threads.h run_all_or_none), nobody left in a barrier.  In a child process: the refusal hook is process-wide."""
    import subprocess
    import sys
    import textwrap
    path, names, seqs, r1, r2, f1, f2, d = lib_and_reads
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = textwrap.dedent("""
        import importlib, os, sys
        sys.path.insert(0, %r)
        import numpy as np, torch
        nim = importlib.import_module("nimble-aligner_amd")
        synth = importlib.import_module("nimble-aligner_amd.synth")
        path, f1 = %r, %r
        libs = [nim.Library(path, "unstranded").build_index(0) for _ in range(2)]
        a = torch.zeros((4000, 150), dtype=torch.uint8, device="cuda:0") + 65
        torch.cuda.synchronize()
        try:
            nim.multi_steps(libs, [0, 0], [[a.data_ptr()], [a.data_ptr()]], None, 4000, 150, 1, 2)
            print("STEPS RAN")
        except nim.Panic as e:
            print("STEPS PANIC", e)
        try:
            nim.fastq_process_sharded([f1], nim.Library(path, "unstranded"), [0, 0], %r)
            print("SHARDED RAN")
        except nim.Panic as e:
            print("SHARDED PANIC", e)
    """) % (root, path, f1, str(d / "refused.tsv"))
    env = dict(os.environ, NIMBLE_FAIL_SPAWN_AT="1", NIMBLE_INDEX_THREADS="1", NIMBLE_CPUS="1")
    cp = subprocess.run([sys.executable, "-c", body], env=env, capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0, (cp.returncode, cp.stdout[-2000:], cp.stderr[-2000:])   # never SIGABRT
    assert "STEPS PANIC" in cp.stdout and "thread" in cp.stdout, cp.stdout[-2000:]
    assert "SHARDED PANIC" in cp.stdout or "SHARDED RAN" in cp.stdout, cp.stdout[-2000:]
