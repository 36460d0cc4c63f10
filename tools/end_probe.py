"""How long does score_call_end take on the host: with the GPU idle, and with the next call running behind it?"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
names, seqs = synth.make_library(1000)
path = "/tmp/end_probe_lib.json"
synth.write_library(path, names, seqs)
lib = nim.Library(path, "unstranded").build_index()
n, L = 10_000_000, 150
reads = synth.make_reads_torch(seqs, n, L, device="cuda:0")
torch.cuda.synchronize()
def begin(s): lib.score_call_begin(s, reads, None, n=n, fixed_len=L, max_len=L, mem=nim.MEM_DEVICE)
for s in (0, 1):
    begin(s)
for s in (0, 1):
    lib.score_call_end(s, raw=True)
# (a) idle: begin, wait for the GPU, then end
ts = []
for _ in range(5):
    begin(0); lib.device_context(0).synchronize(); time.sleep(0.01)
    t = time.perf_counter(); lib.score_call_end(0, raw=True); ts.append(time.perf_counter() - t)
print("end with the GPU idle      : %.3f ms" % (1e3 * sorted(ts)[len(ts) // 2]))
# (b) the call is complete, but the next one is running
ts = []
for _ in range(5):
    begin(0); lib.device_context(0).synchronize(); begin(1); time.sleep(0.0003)
    t = time.perf_counter(); lib.score_call_end(0, raw=True); ts.append(time.perf_counter() - t)
    lib.score_call_end(1, raw=True)
print("end beside a running call  : %.3f ms" % (1e3 * sorted(ts)[len(ts) // 2]))
