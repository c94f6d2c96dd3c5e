// Microbenchmark: integer VALU issue rate on gfx950 (wave64).  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void k(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 ^ 11u, a5 = a0 ^ 13u, a6 = a0 + 17u, a7 = a0 + 19u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) { a0 = (a0 & a1) + a2; a1 = (a1 | a2) ^ a3; a2 = (a2 + a3) & a4; a3 = (a3 ^ a4) | a5; a4 = (a4 & a5) + a6; a5 = (a5 | a6) ^ a7; a6 = (a6 + a7) & a0; a7 = (a7 ^ a0) | a1; }
      if (KIND == 1) { float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1), f2 = __uint_as_float(a2), f3 = __uint_as_float(a3);
        f0 = fmaf(f0, f1, f2); f1 = fmaf(f1, f2, f3); f2 = fmaf(f2, f3, f0); f3 = fmaf(f3, f0, f1);
        float f4 = __uint_as_float(a4), f5 = __uint_as_float(a5), f6 = __uint_as_float(a6), f7 = __uint_as_float(a7);
        f4 = fmaf(f4, f5, f6); f5 = fmaf(f5, f6, f7); f6 = fmaf(f6, f7, f4); f7 = fmaf(f7, f4, f5);
        a0 = __float_as_uint(f0); a1 = __float_as_uint(f1); a2 = __float_as_uint(f2); a3 = __float_as_uint(f3);
        a4 = __float_as_uint(f4); a5 = __float_as_uint(f5); a6 = __float_as_uint(f6); a7 = __float_as_uint(f7); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
int main() {
  uint32_t *d; hipMalloc(&d, 256 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int kind = 0; kind < 2; ++kind)
    for (int wpc : {4, 8, 16, 32}) {       // waves per CU
      int blocks = 256 * wpc / 4;          // 256-thread blocks
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double ops_per_wave = (double)iters * 8 * (kind == 0 ? 16 : 8);   // VALU wave-instructions
      double total = ops_per_wave * blocks * 4;
      printf("kind=%s waves/CU=%2d  %.3f ms  %.1f G wave-instr/s  = %.3f wave-instr/clk/SIMD @2.4GHz\n", kind ? "fma" : "int", wpc, ms, total / ms / 1e6, total / (ms * 1e-3) / (1024 * 2.4e9));
    }
  return 0;
}
