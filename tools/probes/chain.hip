// chain.hip (round 4 probe) -- what bounds a wave that walks a chain of dependent gathers (the align kernel's walk):
// every lane follows its own chain through a table of 32- or 64-byte records (next index = f(record, lane state)), with
// some integer work per step.  Variants: table size (L2-resident or not), 16-byte loads per step, work per step, chains
// per lane (ILP), polluting stream beside it.  Prints ns per step and wave (at 8 waves / SIMD, persistent grid).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/chain tools/probes/chain.hip && /tmp/chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// LOADS: 16-byte loads per record (1, 2 or 4; the record is LOADS * 16 bytes, 64-byte aligned slots)
// WORK: dependent integer ops per step (x8); CHAINS: independent chains per lane
template <int LOADS, int WORK, int CHAINS>
__global__ __launch_bounds__(256, 8) void k_chain(const uint4 *__restrict__ tab, uint32_t mask, uint32_t steps, uint32_t *out,
                                                  const uint4 *__restrict__ pollute, uint64_t pollute_n) {
  uint32_t idx[CHAINS], acc[CHAINS];
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) {
    idx[c] = (g * 2654435761u + c * 40503u) & mask;
    acc[c] = g + c;
  }
  uint64_t pp = (uint64_t)g;
  uint32_t junk = 0;
  for (uint32_t s = 0; s < steps; ++s) {
    uint4 r[CHAINS][LOADS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
      for (int l = 0; l < LOADS; ++l) r[c][l] = tab[(size_t)idx[c] * 4 + l];
    if (pollute) {  // a streaming read beside the walk (the dictionary probes / key stream of the real kernel)
      const uint64_t v = __builtin_nontemporal_load(reinterpret_cast<const uint64_t *>(pollute + (pp % pollute_n)));
      pp += (uint64_t)gridDim.x * blockDim.x;
      junk ^= (uint32_t)v;
    }
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      uint32_t x = r[c][0].x ^ acc[c];
#pragma unroll
      for (int l = 1; l < LOADS; ++l) x += r[c][l].y;
#pragma unroll
      for (int w = 0; w < WORK * 8; ++w) x = x * 1664525u + 1013904223u + (x >> 7);
      acc[c] = x;
      idx[c] = (r[c][0].w ^ (x & 0xFFu)) & mask;   // next record: from the record and a little of the lane's state
    }
  }
  uint32_t t = junk;
  for (int c = 0; c < CHAINS; ++c) t ^= acc[c];
  if (t == 0x12345678u) out[0] = t;
}

// The same chain (2 loads of 16 bytes per step, 32 ops, one chain) with explicit cache-policy bits on the gathers:
// does a policy that does not allocate the line in the CU's L1 raise the rate of distinct lines per CU?
#define POLICY_KERNEL(NAME, BITS)                                                                                       \
  __global__ __launch_bounds__(256, 8) void NAME(const uint4 *__restrict__ tab, uint32_t mask, uint32_t steps, uint32_t *out) { \
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;                                                           \
    uint32_t idx = (g * 2654435761u) & mask, acc = g;                                                                   \
    for (uint32_t s = 0; s < steps; ++s) {                                                                              \
      uint4 r0, r1;                                                                                                     \
      const uint4 *p = tab + (size_t)idx * 4;                                                                           \
      asm volatile("global_load_dwordx4 %0, %2, off " BITS "\n\tglobal_load_dwordx4 %1, %2, off offset:16 " BITS      \
                   "\n\ts_waitcnt vmcnt(0)"                                                                           \
                   : "=&v"(r0), "=&v"(r1)                                                                               \
                   : "v"(p)                                                                                             \
                   : "memory");                                                                                         \
      uint32_t x = r0.x ^ acc;                                                                                          \
      x += r1.y;                                                                                                        \
      _Pragma("unroll") for (int w = 0; w < 32; ++w) x = x * 1664525u + 1013904223u + (x >> 7);                         \
      acc = x;                                                                                                          \
      idx = (r0.w ^ (x & 0xFFu)) & mask;                                                                                \
    }                                                                                                                   \
    if (acc == 0x12345678u) out[0] = acc;                                                                               \
  }
POLICY_KERNEL(k_pol_none, "")
POLICY_KERNEL(k_pol_nt, "nt")
POLICY_KERNEL(k_pol_sc0, "sc0")
POLICY_KERNEL(k_pol_sc1, "sc1")
POLICY_KERNEL(k_pol_sc0sc1, "sc0 sc1")
POLICY_KERNEL(k_pol_sc0nt, "sc0 nt")
POLICY_KERNEL(k_pol_sc1nt, "sc1 nt")
POLICY_KERNEL(k_pol_all, "sc0 sc1 nt")

template <class K>
static int run_policy(const char *what, K kern, const uint4 *tab, uint32_t records, uint32_t steps, uint32_t *out) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e9f;
  const int grid = 2048;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, tab, records - 1, steps, out);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("policy %-12s table %7.1f MiB  %.3f ms  %.1f G lane-steps/s\n", what, records * 64.0 / (1 << 20), best,
         (double)grid * 256 * steps / best / 1e6);
  return 0;
}

template <int LOADS, int WORK, int CHAINS>
static int run(const char *what, const uint4 *tab, uint32_t records, uint32_t steps, uint32_t *out, const uint4 *pollute, uint64_t pn) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e9f;
  const int grid = 2048;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_chain<LOADS, WORK, CHAINS>), dim3(grid), dim3(256), 0, 0, tab, records - 1, steps, out, pollute, pn);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double chain_steps = (double)grid * 256 * steps * CHAINS;
  printf("%-64s table %7.1f MiB  %.3f ms  %.2f us per step and wave  %.1f G lane-steps/s\n", what, records * 64.0 / (1 << 20), best,
         best * 1e3 / steps, chain_steps / best / 1e6);
  return 0;
}

int main() {
  const uint32_t max_records = 1u << 24;  // 1 GiB of 64-byte slots
  uint4 *tab;
  uint32_t *out;
  CHECK(hipMalloc(&tab, (size_t)max_records * 64));
  CHECK(hipMalloc(&out, 64));
  std::vector<uint32_t> h((size_t)max_records * 16);
  uint64_t s = 88172645463325252ULL;
  for (size_t i = 0; i < h.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s >> 16); }
  CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  uint4 *pollute;
  const uint64_t pn = (1ull << 30) / 16;
  CHECK(hipMalloc(&pollute, pn * 16));
  CHECK(hipMemset(pollute, 1, pn * 16));
  const uint32_t steps = 200;
  if (getenv("CHAIN_POLICIES")) {
    for (uint32_t rec : {1u << 15, 1u << 20}) {
      run_policy("none", k_pol_none, tab, rec, steps, out);
      run_policy("nt", k_pol_nt, tab, rec, steps, out);
      run_policy("sc0", k_pol_sc0, tab, rec, steps, out);
      run_policy("sc1", k_pol_sc1, tab, rec, steps, out);
      run_policy("sc0 sc1", k_pol_sc0sc1, tab, rec, steps, out);
      run_policy("sc0 nt", k_pol_sc0nt, tab, rec, steps, out);
      run_policy("sc1 nt", k_pol_sc1nt, tab, rec, steps, out);
      run_policy("sc0 sc1 nt", k_pol_all, tab, rec, steps, out);
    }
    return 0;
  }
  for (uint32_t rec : {1u << 15, 1u << 17, 1u << 20, 1u << 24}) {  // 2 MiB, 8 MiB, 64 MiB, 1 GiB
    run<2, 4, 1>("2 loads, 32 ops, 1 chain", tab, rec, steps, out, nullptr, 0);
    run<4, 4, 1>("4 loads, 32 ops, 1 chain", tab, rec, steps, out, nullptr, 0);
    run<1, 4, 1>("1 load, 32 ops, 1 chain", tab, rec, steps, out, nullptr, 0);
    run<2, 16, 1>("2 loads, 128 ops, 1 chain", tab, rec, steps, out, nullptr, 0);
    run<2, 4, 2>("2 loads, 32 ops, 2 chains", tab, rec, steps, out, nullptr, 0);
    run<2, 16, 2>("2 loads, 128 ops, 2 chains", tab, rec, steps, out, nullptr, 0);
    run<2, 4, 4>("2 loads, 32 ops, 4 chains", tab, rec, steps, out, nullptr, 0);
    run<2, 16, 1>("2 loads, 128 ops, 1 chain + polluting stream", tab, rec, steps, out, pollute, pn);
  }
  return 0;
}
