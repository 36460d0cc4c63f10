"""Step-by-step timing of one pipelined submit while a packed call is in flight (development aid)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29549", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
names, seqs = synth.make_library(1000)
lib = nim.Library(text=json.dumps(synth.library_json(names, seqs)), strand_filter="unstranded").build_index(0)
n = 10_000_000
reads = synth.make_reads_torch(seqs, n, device="cuda:0")
torch.cuda.synchronize()
for s in range(3):
    lib.device_context(s).set_counters(False)
util = lib.device_context(2)
side = torch.cuda.Stream()
T = lambda: time.perf_counter()
with torch.cuda.stream(side):
    pt = lib.pack(reads, None, None, None, n=n, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE, slot=2)
    rec, counts = pt.route(util, 1)
    shard = nim.PackedTensors.unpack(util, rec, pt.key_words, pt.max_len, pt.paired)
    util.synchronize()
    for rep in range(3):
        lib.score_call_packed_begin(0, shard); lib.score_call_end(0, raw=True)   # warm
    for variant in ("packed_begin", "plain_begin", "packed_begin+pack_first"):
        for rep in range(2):
            torch.cuda.synchronize(); util.synchronize()
            t = [T()]
            if variant == "packed_begin+pack_first":
                pt2 = lib.pack(reads, None, None, None, n=n, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE, slot=2, out=pt)
                rec2, counts2 = pt.route(util, 1, out=rec)
            t.append(T())
            if variant == "plain_begin":
                lib.score_call_begin(0, reads, None, n=n, fixed_len=150, max_len=150, mem=nim.MEM_DEVICE)
            else:
                lib.score_call_packed_begin(0, shard)
            t.append(T())
            x = torch.tensor([n], dtype=torch.int64, device=dev)
            t.append(T())
            y = torch.empty_like(x)
            dist.all_to_all_single(y, x)
            t.append(T())
            l = y.tolist()
            t.append(T())
            lib.score_call_end(0, raw=True)
            t.append(T())
            print(variant, ["%.2f" % ((b - a) * 1e3) for a, b in zip(t[:-1], t[1:])], "[pack+route, begin, torch.tensor, a2a, tolist, end]", flush=True)
dist.destroy_process_group()
