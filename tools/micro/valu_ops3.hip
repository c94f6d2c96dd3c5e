// Microbenchmark: issue rate per VALU opcode on gfx950 (8 independent chains per lane, wave64,
// 16 waves/CU).  hipcc --offload-arch=gfx950 -O3 -o valu_ops valu_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OPS(X)                                                                                   \
  X(0, "v_cndmask_b32 %0, %0, %1, vcc")                                                          \
  X(1, "v_cndmask_b32 %0, %1, %2, vcc")                                                          \
  X(2, "v_cndmask_b32_e64 %0, %0, %1, vcc")                                                      \
  X(3, "v_cndmask_b32_e64 %0, %1, %2, s[10:11]")                                                 \
  X(4, "v_cmp_lt_i32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc")                               \
  X(5, "v_cmp_lt_i32_e64 s[10:11], %1, %2\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]")             \
  X(6, "v_cmp_lt_i32 vcc, %1, %2\n s_nop 0\n v_cndmask_b32 %0, %0, %1, vcc")                     \
  X(7, "v_sub_u32 %0, %0, %1\n v_ashrrev_i32 %0, 31, %0\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca") \
  X(8, "v_cmp_lt_i32 vcc, %1, %2\n v_cmp_lt_i32 vcc, %1, %2")                                    \
  X(9, "v_cmp_lt_i32 vcc, %0, %2\n v_addc_co_u32 %0, vcc, %0, %1, vcc")                          \
  X(10, "v_mov_b32 %0, %1")

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
  asm volatile("s_mov_b64 s[10:11], 0x5555\n s_mov_b64 vcc, 0x3333\n s_mov_b32 s12, 0x77" ::: "s10", "s11", "s12", "vcc");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#define X(ID, TEXT) \
  if (KIND == ID) asm volatile(TEXT : "+v"(a[i]) : "v"(b[i]), "v"(c[i]) : "vcc", "s10", "s11");
        OPS(X)
#undef X
      }
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND>
void run(const char *name, uint32_t *d, hipEvent_t e0, hipEvent_t e1) {
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
  double n = (double)iters * 64 * blocks;
  printf("%.3f instr/clk/SIMD  %s\n", n / (ms * 1e-3) / (1024 * 2.4e9), name);
}
int main() {
  uint32_t *d;
  (void)hipMalloc(&d, 256 * 32 * 64 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
#define X(ID, TEXT) run<ID>(TEXT, d, e0, e1);
  OPS(X)
#undef X
  return 0;
}
