// Scattered atomic rate on MI355X by operation (table 120 MB / 2 MB).  hipcc --offload-arch=gfx950 -O3 atomic_rate.hip
// Round-1 finding: 64-bit CAS runs at 26.7 G/s whatever the table size (120 MB .. 2 MB), the scope (agent /
// workgroup) or an XCD-local partition of the table: the bound is the per-CU atomic path, not the memory side.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
template <int OP>
__global__ void k_op(uint64_t *table, uint32_t slots, uint64_t n, uint64_t *sink) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = mix64(i + 1);
  const uint32_t pos = __umulhi((uint32_t)h, slots);
  uint32_t *t32 = reinterpret_cast<uint32_t *>(table);
  if (OP == 0) {  // 64-bit CAS, value used
    unsigned long long old = atomicCAS((unsigned long long *)&table[pos], 0ULL, (unsigned long long)(h | 1));
    if (old == 12345ULL) sink[0] = i;
  } else if (OP == 1) {  // 32-bit CAS, value used
    uint32_t old = atomicCAS(&t32[pos], 0u, (uint32_t)h | 1u);
    if (old == 12345u) sink[0] = i;
  } else if (OP == 2) {  // 64-bit add, no return
    atomicAdd((unsigned long long *)&table[pos], 1ULL);
  } else if (OP == 3) {  // 64-bit max, no return
    atomicMax((unsigned long long *)&table[pos], (unsigned long long)h);
  } else if (OP == 4) {  // 32-bit max, no return
    atomicMax(&t32[pos], (uint32_t)h);
  } else if (OP == 5) {  // 32-bit or, no return
    atomicOr(&t32[pos], 1u << (h & 31));
  } else if (OP == 6) {  // plain 64-bit load of the slot (the gather an insert-free probe would do)
    if (table[pos] == 12345ULL) sink[0] = i;
  } else if (OP == 7) {  // plain 64-bit store
    table[pos] = h;
  }
}
template <int OP>
static void run(const char *name, uint64_t *table, uint32_t slots, uint64_t n, uint64_t *sink) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipMemset(table, 0, (size_t)slots * 8);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k_op<OP>), dim3((n + 255) / 256), dim3(256), 0, 0, table, slots, n, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  printf("%-28s slots %9u (%6.1f MB)  %.3f ms  %6.1f G ops/s\n", name, slots, slots * 8 / 1e6, best, n / best / 1e6);
}
int main() {
  const uint64_t n = 10000000;
  uint64_t *table, *sink;
  (void)hipMalloc(&table, (size_t)(1u << 24) * 8); (void)hipMalloc(&sink, 8);
  for (uint32_t slots : {15000000u, 1u << 18}) {
    run<0>("CAS 64, value used", table, slots, n, sink);
    run<1>("CAS 32, value used", table, slots, n, sink);
    run<2>("add 64, no return", table, slots, n, sink);
    run<3>("max 64, no return", table, slots, n, sink);
    run<4>("max 32, no return", table, slots, n, sink);
    run<5>("or 32, no return", table, slots, n, sink);
    run<6>("load 64", table, slots, n, sink);
    run<7>("store 64", table, slots, n, sink);
  }
  return 0;
}
