// Scattered CAS rate vs number of workgroups (persistent blocks striding over 10 M operations): can a few CUs
// saturate the chip's atomic rate?  hipcc --offload-arch=gfx950 -O3 atomic_rate_cus.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
__global__ __launch_bounds__(256) void k_cas(uint64_t *table, uint32_t slots, uint64_t n, uint64_t *sink) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t h = mix64(i + 1);
    const uint32_t pos = __umulhi((uint32_t)h, slots);
    unsigned long long old = atomicCAS((unsigned long long *)&table[pos], 0ULL, (unsigned long long)(h | 1));
    if (old == 12345ULL) sink[0] = i;
  }
}
int main() {
  const uint64_t n = 10000000;
  const uint32_t slots = 15000000;
  uint64_t *table, *sink;
  (void)hipMalloc(&table, (size_t)slots * 8); (void)hipMalloc(&sink, 8);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int blocks : {8, 16, 32, 64, 128, 256, 512, 1024, 2048, 8192}) {
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      (void)hipMemset(table, 0, (size_t)slots * 8);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(a);
      hipLaunchKernelGGL(k_cas, dim3(blocks), dim3(256), 0, 0, table, slots, n, sink);
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms = 0; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("blocks %5d  %.3f ms  %6.1f G CAS/s\n", blocks, best, n / best / 1e6);
  }
  return 0;
}
