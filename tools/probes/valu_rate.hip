// Issue cost of the VALU instructions the walk leans on (gfx950): cycles per wave-instruction, measured with 8 waves per SIMD
// so that latency is hidden and only the issue rate shows.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 64
template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t s0, int iters) {
  uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ULL + 1, b = a ^ 0x1234567ULL, c = a + 77, d = b + 99;
  asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555" ::: "s20", "s21");
  uint32_t x = (uint32_t)a, y = (uint32_t)b, z = (uint32_t)c, w = (uint32_t)d;
  const uint32_t s = s0 + (threadIdx.x & 1);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      if (OP == 0) { asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s)); }
      if (OP == 1) { asm volatile("v_lshlrev_b32 %0, %4, %0\n v_lshlrev_b32 %1, %4, %1\n v_lshlrev_b32 %2, %4, %2\n v_lshlrev_b32 %3, %4, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(s)); }
      if (OP == 2) { asm volatile("v_alignbit_b32 %0, %0, %1, %4\n v_alignbit_b32 %1, %1, %2, %4\n v_alignbit_b32 %2, %2, %3, %4\n v_alignbit_b32 %3, %3, %0, %4" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(s)); }
      if (OP == 3) { asm volatile("v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %1, %2, %1\n v_bcnt_u32_b32 %2, %3, %2\n v_bcnt_u32_b32 %3, %0, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); }
      if (OP == 4) { asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); }
      if (OP == 5) { asm volatile("v_ffbh_u32 %0, %1\n v_ffbh_u32 %1, %2\n v_ffbh_u32 %2, %3\n v_ffbh_u32 %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); }
      if (OP == 6) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : : "vcc"); }
      if (OP == 7) { asm volatile("v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s)); }
      if (OP == 8) { asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); }
      if (OP == 9) { asm volatile("v_lshl_add_u64 %0, %0, 3, %1\n v_lshl_add_u64 %1, %1, 3, %2\n v_lshl_add_u64 %2, %2, 3, %3\n v_lshl_add_u64 %3, %3, 3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
      if (OP == 10) { asm volatile("v_cmp_ne_u32 vcc, %0, %1\n v_cmp_ne_u32 vcc, %1, %2\n v_cmp_ne_u32 vcc, %2, %3\n v_cmp_ne_u32 vcc, %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : : "vcc"); }
      if (OP == 12) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %0, s[20:21]" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : : "s20", "s21"); }
      if (OP == 13) { asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(s)); }
      if (OP == 14) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : : "vcc"); }
      if (OP == 15) { asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_u32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : : "vcc"); }
      if (OP == 16) { asm volatile("v_max_u32 %0, %0, %1\n v_max_u32 %1, %1, %2\n v_max_u32 %2, %2, %3\n v_max_u32 %3, %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); }
      if (OP == 17) { asm volatile("v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %3, %3, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); }
      if (OP == 18) { asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); }
      if (OP == 11) { asm volatile("v_min3_u32 %0, %0, %1, %2\n v_min3_u32 %1, %1, %2, %3\n v_min3_u32 %2, %2, %3, %0\n v_min3_u32 %3, %3, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ x ^ y ^ z ^ w;
}

template <int OP>
double run(const char *name, uint64_t *out) {
  const int iters = 2000, blocks = 256 * 8;  // 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 3u, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 3u, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: blocks * 4 waves * iters * REP * 4 / (256 CUs * 4 SIMDs)
  const double per_simd = (double)blocks * 4 * iters * REP * 4 / (256.0 * 4);
  const double cyc = ms * 1e-3 * 2.4e9 / per_simd;
  printf("%-16s %.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, cyc);
  return cyc;
}

int main() {
  uint64_t *out;
  hipMalloc(&out, 256 * 8 * 256 * 8);
  run<8>("v_add_u32", out);
  run<1>("v_lshlrev_b32", out);
  run<0>("v_lshlrev_b64", out);
  run<7>("v_lshrrev_b64", out);
  run<2>("v_alignbit_b32", out);
  run<3>("v_bcnt_u32_b32", out);
  run<4>("v_mul_lo_u32", out);
  run<5>("v_ffbh_u32", out);
  run<6>("v_cndmask_b32", out);
  run<9>("v_lshl_add_u64", out);
  run<10>("v_cmp_ne_u32", out);
  run<11>("v_min3_u32", out);
  run<12>("v_cndmask e64 sgpr", out);
  run<13>("v_bfi_b32", out);
  run<14>("1 cndmask + 3 add", out);
  run<15>("cmp+cndmask x2", out);
  run<16>("v_max_u32", out);
  run<17>("v_and_or_b32", out);
  run<18>("v_xor_b32", out);
  return 0;
}
