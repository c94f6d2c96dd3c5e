// Microbenchmark: issue rate of 2-source vs 3-source VALU ops on gfx950 (8 independent chains per
// lane, so latency is hidden).  hipcc --offload-arch=gfx950 -O3 -o valu_3src valu_3src.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
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
        if (KIND == 0) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        if (KIND == 1) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (KIND == 2) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc4" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (KIND == 3) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b[i]));
        if (KIND == 4) asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(a[i]) : "s"(seed));
        if (KIND == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        if (KIND == 6) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
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
  const int iters = 2000;
  for (int wpc : {8, 16, 32}) {
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
    printf("%-14s waves/CU=%2d  %.3f ms  %.3f instr/clk/SIMD @2.4GHz\n", name, wpc, ms, n / (ms * 1e-3) / (1024 * 2.4e9));
  }
}
int main() {
  uint32_t *d;
  (void)hipMalloc(&d, 256 * 32 * 64 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  run<0>("v_and_b32", d, e0, e1);
  run<1>("v_and_or_b32", d, e0, e1);
  run<2>("v_bitop3_b32", d, e0, e1);
  run<3>("v_lshl_or_b32", d, e0, e1);
  run<4>("v_bfe_u32", d, e0, e1);
  run<5>("v_add_u32", d, e0, e1);
  run<6>("v_min3_i32", d, e0, e1);
  return 0;
}
