// skeleton.hip (round 4 probe) -- what a persistent-block tile loop costs on MI355X when the tiles do nothing:
// k_align over reads that stop at the prefilter took 0.5 ms per 10 M reads (26 us per 256-read tile and block) with its
// key loads and result stores compiled out.  Variants add the loop's parts one at a time.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/skel tools/probes/skeleton.hip && /tmp/skel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int V, bool STATIC>
__global__ __launch_bounds__(256, 8) void k_skel(const uint32_t *__restrict__ len, const uint8_t *__restrict__ pre,
                                                 const uint64_t *__restrict__ keys, uint64_t stride, uint32_t *__restrict__ out,
                                                 unsigned long long *counter, uint64_t n) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  __shared__ unsigned long long s_tile;
  __shared__ uint32_t s_cnt[8];
  __shared__ uint16_t s_perm[256];
  const uint32_t tid = threadIdx.x;
  const uint64_t n_tiles = (n + 255) / 256;
  uint64_t *col = lds + tid;
  uint64_t prev = ~0ULL;
  uint32_t acc = 0;
  for (;;) {
    if (V >= 1) __syncthreads();
    if (V >= 5 && prev != ~0ULL) {
      const uint64_t r = prev * 256 + tid;
      if (r < n) __builtin_nontemporal_store((uint32_t)col[0], out + r);
    }
    if (tid == 0) s_tile = STATIC ? (prev == ~0ULL ? (unsigned long long)blockIdx.x : prev + gridDim.x) : atomicAdd(counter, 1ULL);
    __syncthreads();
    const uint64_t tile = s_tile;
    if (tile >= n_tiles) break;
    prev = tile;
    const uint64_t r = tile * 256 + tid;
    uint32_t kind = 2;
    if (V >= 2 && r < n) {
      const uint32_t l = len[r], p = pre[r];
      if (V >= 3) {
        for (int w = 0; w < 5; ++w) col[w * 256] = __builtin_nontemporal_load(keys + (uint64_t)w * stride + r);
        col[5 * 256] = 0;
      }
      kind = (p == 255 && l >= 30) ? (l & 1) : 2;
      acc += l;
    }
    if (V >= 4) {
      const uint64_t b0 = __ballot(kind == 0), b1 = __ballot(kind == 1);
      const uint32_t wv = tid >> 6, lane = tid & 63;
      if (lane == 0) { s_cnt[wv * 2] = __popcll(b0); s_cnt[wv * 2 + 1] = __popcll(b1); }
      __syncthreads();
      uint32_t tot0 = 0, tot1 = 0, pre0 = 0, pre1 = 0;
      for (uint32_t w = 0; w < 4; ++w) { const uint32_t c0 = s_cnt[w * 2], c1 = s_cnt[w * 2 + 1]; if (w < wv) { pre0 += c0; pre1 += c1; } tot0 += c0; tot1 += c1; }
      const uint64_t below = (1ULL << lane) - 1;
      const uint32_t r0 = __popcll(b0 & below), r1 = __popcll(b1 & below), r2 = lane - r0 - r1, pre2 = wv * 64 - pre0 - pre1;
      const uint32_t pos = kind == 0 ? pre0 + r0 : (kind == 1 ? tot0 + pre1 + r1 : tot0 + tot1 + pre2 + r2);
      s_perm[pos] = (uint16_t)tid;
      __syncthreads();
      acc += s_perm[tid];
      if (V >= 5) lds[s_perm[tid]] = acc;
    }
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

template <int V, bool STATIC>
static int run(const char *what, const uint32_t *len, const uint8_t *pre, const uint64_t *keys, uint32_t *out, unsigned long long *counter,
               uint64_t n, int grid, size_t lds) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipMemsetAsync(counter, 0, 8, 0));
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_skel<V, STATIC>), dim3(grid), dim3(256), lds, 0, len, pre, keys, n, out, counter, n);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("%-58s grid %5d lds %6zu  %.3f ms\n", what, grid, lds, best);
  return 0;
}

int main() {
  const uint64_t n = 10000000;
  uint32_t *len, *out;
  uint8_t *pre;
  uint64_t *keys;
  unsigned long long *counter;
  CHECK(hipMalloc(&len, n * 4));
  CHECK(hipMalloc(&out, n * 4));
  CHECK(hipMalloc(&pre, n));
  CHECK(hipMalloc(&keys, n * 5 * 8));
  CHECK(hipMalloc(&counter, 8));
  CHECK(hipMemset(len, 0, n * 4));
  CHECK(hipMemset(pre, 3, n));
  CHECK(hipMemset(keys, 1, n * 5 * 8));
  const size_t lds = 6 * 256 * 8 + 4 * 256 * 4 + 2640;
  for (int grid : {2048, 1024, 4096, 39063}) {
    printf("-- grid %d\n", grid);
    run<0, false>("V0 tile counter + one barrier", len, pre, keys, out, counter, n, grid, lds);
    run<0, true>("V0 static tiles + one barrier", len, pre, keys, out, counter, n, grid, lds);
    run<1, false>("V1 + barrier at the loop head", len, pre, keys, out, counter, n, grid, lds);
    run<2, false>("V2 + lengths and prefilter verdicts", len, pre, keys, out, counter, n, grid, lds);
    run<3, false>("V3 + key words into LDS", len, pre, keys, out, counter, n, grid, lds);
    run<4, false>("V4 + partition (two barriers)", len, pre, keys, out, counter, n, grid, lds);
    run<5, false>("V5 + staged stores", len, pre, keys, out, counter, n, grid, lds);
    run<5, true>("V5 static tiles", len, pre, keys, out, counter, n, grid, lds);
  }
  run<5, false>("V5 small LDS", len, pre, keys, out, counter, n, 2048, 6 * 256 * 8);
  return 0;
}
