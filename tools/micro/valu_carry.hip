// Microbenchmark: does the carry-out of "x + x" (the Myers shift) make a cheaper score update than
// "shift right 31, add"?  8 independent (vector, score) chains per lane, wave64, 16 waves/CU.
// hipcc --offload-arch=gfx950 -O3 -o valu_carry valu_carry.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OPS(X)                                                                                              \
  X(0, 3, "v_lshrrev_b32 %2, 31, %0\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %0")                          \
  X(1, 2, "v_add_co_u32 %0, vcc, %0, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc")                               \
  X(2, 2, "v_add_co_u32 %0, vcc, %0, %0\n v_subbrev_co_u32 %1, vcc, 0, %1, vcc")                            \
  X(3, 1, "v_add_co_u32 %0, vcc, %0, %0")                                                                    \
  X(4, 1, "v_addc_co_u32 %1, vcc, 0, %1, vcc")                                                               \
  X(5, 1, "v_add_u32 %0, %0, %0")                                                                            \
  X(6, 2, "v_add_co_u32 %0, vcc, %0, %0\n v_addc_co_u32 %1, vcc, %1, %1, vcc")                              \
  X(7, 2, "v_add_co_u32 %0, s[10:11], %0, %0\n v_addc_co_u32 %1, s[12:13], 0, %1, s[10:11]")                \
  X(8, 2, "v_alignbit_b32 %1, %1, %0, 31\n v_add_u32 %0, %0, %0")                                            \
  X(9, 4, "v_add_co_u32 %0, vcc, %0, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_and_b32 %2, %0, %1\n v_or_b32 %0, %0, %2") \
  X(10, 4, "v_lshrrev_b32 %2, 31, %0\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %0\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf8")

template <int KIND>
__global__ void k(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a[8], b[8], c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = threadIdx.x * (2 * i + 3) + seed;
    b[i] = a[i] * 77u + 1u;
    c[i] = a[i] ^ 0x55aa55aau;
    asm volatile("" : "+v"(b[i]), "+v"(c[i]));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#define X(ID, N, TEXT) \
  if (KIND == ID) asm volatile(TEXT : "+v"(a[i]), "+v"(b[i]), "+v"(c[i]) : : "vcc", "s10", "s11", "s12", "s13");
        OPS(X)
#undef X
      }
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r ^= a[i] ^ b[i] ^ c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND>
void run(const char *name, int n_ins, uint32_t *d, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 1000, wpc = 16;
  int blocks = 256 * wpc;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  double n = (double)iters * 64 * blocks;  // sequences per wave
  printf("%.3f seq/clk/SIMD  (%d instr: %.3f instr/clk/SIMD)  %s\n", n / (ms * 1e-3) / (1024 * 2.4e9), n_ins,
         n_ins * n / (ms * 1e-3) / (1024 * 2.4e9), name);
}
int main() {
  uint32_t *d;
  (void)hipMalloc(&d, 256 * 32 * 64 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
#define X(ID, N, TEXT) run<ID>(TEXT, N, d, e0, e1);
  OPS(X)
#undef X
  return 0;
}
