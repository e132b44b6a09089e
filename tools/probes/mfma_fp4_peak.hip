// Peak rate of v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4) and v_mfma_i32_32x32x32_i8 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void fp4_loop(float* out, int iters) {
  v8i a = {0x22222222, 0x2A2A2A2A, (int)0xA2A2A2A2, 0x22222222, 0, 0, 0, 0};
  v8i b = {0x2222AAAA, 0x2A2A2A2A, 0x22222222, (int)0xAAAA2222, 0, 0, 0, 0};
  a[0] += threadIdx.x & 1;
  v16f c[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void i8_loop(int* out, int iters) {
  v4i a = {0x01FF01FF, 0x0101FFFF, (int)0xFF01FF01, 0x01010101};
  v4i b = {0x01FF01FF, (int)0xFFFF0101, 0x01FF01FF, 0x01010101};
  a[0] += threadIdx.x & 1;
  v16i c[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
  }
  int s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int bpc : {1, 2}) {
    const int blocks = 256 * bpc;
    fp4_loop<4><<<blocks, 256>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); fp4_loop<4><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double macs = (double)blocks * 4 * iters * 4 * 32.0 * 32 * 64;
    printf("fp4 32x32x64: %d blocks/CU  %.3f ms  %.3e MAC/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", bpc, ms, macs / (ms * 1e-3),
           ms * 1e-3 * 2.4e9 / (iters * 4.0 * bpc));
    i8_loop<4><<<blocks, 256>>>((int*)out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); i8_loop<4><<<blocks, 256>>>((int*)out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    macs = (double)blocks * 4 * iters * 4 * 32.0 * 32 * 32;
    printf("i8  32x32x32: %d blocks/CU  %.3f ms  %.3e MAC/s\n", bpc, ms, macs / (ms * 1e-3));
  }
  return 0;
}
