// How long does a 12-bit partial radix sort of 8 M (u64 hash, u64 payload) pairs take (rocPRIM)?  The partition step of a
// dedup that is not a global hash insert.  hipcc --offload-arch=gfx950 -O3 sort_rate.hip -o sort_rate.bin
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void fill(uint64_t *k, uint64_t *v, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = i * 0x9E3779B97F4A7C15ULL;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
  k[i] = x;
  v[i] = i;
}

int main() {
  const uint64_t n = 8100000;
  uint64_t *k0, *k1, *v0, *v1;
  hipMalloc(&k0, n * 8); hipMalloc(&k1, n * 8); hipMalloc(&v0, n * 8); hipMalloc(&v1, n * 8);
  hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, 0, k0, v0, n);
  for (int bits : {12, 13, 16}) {
    size_t tmp = 0;
    rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, v0, v1, n, 64 - bits, 64, 0);
    void *t; hipMalloc(&t, tmp);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rocprim::radix_sort_pairs(t, tmp, k0, k1, v0, v1, n, 64 - bits, 64, 0);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) rocprim::radix_sort_pairs(t, tmp, k0, k1, v0, v1, n, 64 - bits, 64, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("radix_sort_pairs u64/u64, %llu items, top %d bits: %.3f ms per sort (temp %zu bytes)\n", (unsigned long long)n, bits, ms / 10, tmp);
    hipFree(t);
  }
  return 0;
}
