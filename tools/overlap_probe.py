"""Which torch / RCCL operation waits for a call that is in flight on the library's launch stream? (development aid)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
names, seqs = synth.make_library(1000)
lib = nim.Library(text=json.dumps(synth.library_json(names, seqs)), strand_filter="unstranded").build_index(0)
n = 10_000_000
reads = synth.make_reads_torch(seqs, n, device="cuda:0")
torch.cuda.synchronize()
lib.device_context(0).set_counters(False)
lib.score_call_raw(reads, None, n=n, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE)
a = torch.empty((n, 7), dtype=torch.int64, device=dev)
b = torch.empty_like(a)
small = torch.arange(8, device=dev)
side = torch.cuda.Stream()
def probe(name, fn):
    torch.cuda.synchronize(); lib.device_context(0).synchronize()
    t0 = time.perf_counter()
    lib.score_call_begin(0, reads, None, n=n, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE)
    t1 = time.perf_counter()
    fn()
    t2 = time.perf_counter()
    lib.score_call_end(0, raw=True)
    t3 = time.perf_counter()
    print("%-34s begin %.2f ms  op %.2f ms  end-wait %.2f ms" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
for _ in range(2):
    probe("nothing", lambda: None)
    probe("tolist(small)", lambda: small.tolist())
    probe("d2d copy + stream sync", lambda: (b.copy_(a), torch.cuda.current_stream().synchronize()))
    def side_copy():
        with torch.cuda.stream(side):
            b.copy_(a)
        side.synchronize()
    probe("d2d copy on a side stream", side_copy)
    probe("torch.empty 560MB", lambda: torch.empty((n, 7), dtype=torch.int64, device=dev))
    probe("all_to_all_single small", lambda: (dist.all_to_all_single(torch.empty_like(small), small), torch.cuda.current_stream().synchronize()))
    probe("all_to_all_single 560MB", lambda: (dist.all_to_all_single(b, a), torch.cuda.current_stream().synchronize()))
    probe("all_reduce small + item", lambda: (dist.all_reduce(small), small[0].item()))
    def on_side(fn):
        def run():
            with torch.cuda.stream(side):
                fn()
            side.synchronize()
        return run
    probe("side: tolist(small)", on_side(lambda: small.tolist()))
    probe("side: torch.tensor(list) H2D", on_side(lambda: torch.tensor([1, 2, 3], dtype=torch.int64, device=dev)))
    pinned = torch.tensor([1, 2, 3], dtype=torch.int64).pin_memory()
    probe("side: pinned H2D non_blocking", on_side(lambda: pinned.to(dev, non_blocking=True)))
    probe("side: all_to_all_single small", on_side(lambda: dist.all_to_all_single(torch.empty_like(small), small)))
    probe("side: all_to_all_single 560MB", on_side(lambda: dist.all_to_all_single(b, a)))
    probe("side: all_reduce small", on_side(lambda: dist.all_reduce(small)))
    hp = torch.empty(8, dtype=torch.int64).pin_memory()
    probe("side: D2H into pinned, non_blocking", on_side(lambda: hp.copy_(small, non_blocking=True)))
dist.destroy_process_group()
