// Microbenchmark: sustained rate of the v_xor_b32 + v_bcnt_u32_b32 pair on gfx950, to pin
// the integer-VALU roofline used in DESIGN.md / bench.py (assumption to verify: both ops
// issue at full rate, 4 SIMD-32 per CU -> 256*4*32*clk lane-ops/s).
//   hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// MODE 0: xor+bcnt pair (the bconv inner product); 1: v_xor_b32 only; 2: v_bcnt_u32_b32 only;
// 3: v_add_u32 only; 4: v_xor_b32 + v_add_u32 pair (two full-rate ops, for comparison)
template <int ACCS, int MODE = 0>
__global__ __launch_bounds__(256) void xor_bcnt_loop(uint32_t* out, const uint32_t* seed, int iters) {
  uint32_t a[ACCS];
  int acc[ACCS];
  const uint32_t s0 = seed[blockIdx.x & 7];           // wave-uniform -> SGPR operand
#pragma unroll
  for (int i = 0; i < ACCS; ++i) { a[i] = threadIdx.x * 2654435761u + i; acc[i] = 0; }
  for (int it = 0; it < iters; ++it) {
    uint32_t w = __builtin_amdgcn_readfirstlane(s0 + it);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < ACCS; ++i) {
        uint32_t t;
        if (MODE == 0) asm volatile("v_xor_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]), "=&v"(t) : "s"(w), "v"(a[i]));
        else if (MODE == 1) asm volatile("v_xor_b32 %0, %2, %0\n\tv_xor_b32 %0, %3, %0" : "+v"(acc[i]), "=&v"(t) : "s"(w), "v"(a[i]));
        else if (MODE == 2) asm volatile("v_bcnt_u32_b32 %0, %3, %0\n\tv_bcnt_u32_b32 %0, %3, %0" : "+v"(acc[i]), "=&v"(t) : "s"(w), "v"(a[i]));
        else if (MODE == 3) asm volatile("v_add_u32 %0, %3, %0\n\tv_add_u32 %0, %2, %0" : "+v"(acc[i]), "=&v"(t) : "s"(w), "v"(a[i]));
        else asm volatile("v_xor_b32 %1, %2, %3\n\tv_add_u32 %0, %1, %0" : "+v"(acc[i]), "=&v"(t) : "s"(w), "v"(a[i]));
      }
    }
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < ACCS; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACCS, int MODE = 0>
void run(int blocks_per_cu, const char* label) {
  const int iters = 4000;
  const int blocks = 256 * blocks_per_cu;
  uint32_t *out, *seed;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&seed, 64);
  hipMemset(seed, 1, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  xor_bcnt_loop<ACCS, MODE><<<blocks, 256>>>(out, seed, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  xor_bcnt_loop<ACCS, MODE><<<blocks, 256>>>(out, seed, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double pairs = (double)blocks * 256 * iters * 16 * ACCS;   // xor+bcnt pairs (lane granularity)
  const double lane_ops = pairs * 2;
  printf("{\"label\":\"%s\",\"accs\":%d,\"blocks_per_cu\":%d,\"ms\":%.3f,\"lane_ops_per_s\":%.4e,\"bmac_per_s\":%.4e}\n",
         label, ACCS, blocks_per_cu, ms, lane_ops / (ms * 1e-3), pairs * 32 / (ms * 1e-3));
  hipFree(out); hipFree(seed);
}

int main() {
  run<8>(1, "1 block/CU (1 wave/SIMD)");
  run<8>(2, "2 blocks/CU");
  run<8>(4, "4 blocks/CU");
  run<8>(8, "8 blocks/CU");
  run<16>(4, "16 accs, 4 blocks/CU");
  run<4>(8, "4 accs, 8 blocks/CU");
  run<8, 1>(8, "MODE1 v_xor_b32 x2");
  run<8, 2>(8, "MODE2 v_bcnt_u32_b32 x2");
  run<8, 3>(8, "MODE3 v_add_u32 x2");
  run<8, 4>(8, "MODE4 v_xor_b32 + v_add_u32");
  return 0;
}
