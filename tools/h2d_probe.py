"""Host-to-device copy rates on this box (development aid): page-locked by allocation vs by registration, by size."""
import time
import torch

dev = torch.device("cuda:0")
rt = torch.cuda.cudart()
for mb in (16, 64, 256, 1024):
    n = mb << 20
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    a = torch.empty(n, dtype=torch.uint8).pin_memory()
    a.fill_(7)
    b = torch.empty(n, dtype=torch.uint8)
    b.fill_(7)
    rc = rt.cudaHostRegister(b.data_ptr(), n, 0)
    c = torch.empty(n, dtype=torch.uint8)
    c.fill_(7)
    for tag, src in (("hipHostMalloc", a), ("hipHostRegister rc=%s" % rc, b), ("pageable", c)):
        best = 0
        for rep in range(4):
            torch.cuda.synchronize()
            t = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            best = max(best, n / (time.perf_counter() - t) / 1e9)
        print("%5d MiB  %-28s %6.1f GB/s" % (mb, tag, best), flush=True)
    rt.cudaHostUnregister(b.data_ptr())
# two copies in flight on two streams (does a second DMA engine add up?)
n = 256 << 20
srcs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
dsts = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
st = [torch.cuda.Stream() for _ in range(2)]
torch.cuda.synchronize()
t = time.perf_counter()
for rep in range(4):
    for k in range(2):
        with torch.cuda.stream(st[k]):
            dsts[k].copy_(srcs[k], non_blocking=True)
torch.cuda.synchronize()
print("two streams, 256 MiB each: %.1f GB/s together" % (8 * n / (time.perf_counter() - t) / 1e9))
