// Microbenchmark: VALU throughput on gfx950 as a function of per-wave ILP (independent dependency
// chains) and occupancy.  hipcc --offload-arch=gfx950 -O3 -o valu_latency valu_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int CHAINS>
__global__ void k(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) a[c] = threadIdx.x * (2 * c + 3) + seed;
  uint32_t m = seed | 1u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        a[c] = (a[c] ^ m) + 0x9e3779b9u;  // two dependent ops (xor, add) unless fused into v_xad_u32
        asm volatile("" : "+v"(a[c]));
      }
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) r ^= a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int CHAINS>
void run(uint32_t *d, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 2000;
  for (int wpc : {4, 8, 16, 24, 32}) {
    int blocks = 256 * wpc;  // one wave per block
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(64), 0, 0, d, iters, 1u);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    double steps = (double)iters * 16 * CHAINS * blocks;  // chain steps (1 or 2 VALU each)
    printf("chains=%d waves/CU=%2d  %.3f ms  %.3f chain-steps/clk/SIMD @2.4GHz\n", CHAINS, wpc, ms,
           steps / (ms * 1e-3) / (1024 * 2.4e9));
  }
}
int main() {
  uint32_t *d;
  hipMalloc(&d, 256 * 32 * 64 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  run<1>(d, e0, e1);
  run<2>(d, e0, e1);
  run<4>(d, e0, e1);
  run<8>(d, e0, e1);
  return 0;
}
