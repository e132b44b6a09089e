// How many instructions of which kind does ONE wave per SIMD hide behind its own FP4 MFMA stream?
// (The streaming LceBconv2d kernel, lce_kernels_stream.h, runs one wave per SIMD: two dependent MFMA chains that
// alternate, N filler instructions behind each MFMA.)  One 256-thread block per CU (100 KiB of LDS); every wave runs
// ITERS x { MFMA(acc0), N fillers, MFMA(acc1), N fillers } and wave 0 reports cycles per MFMA (s_memtime).
// Build twice:  hipcc -O3 --offload-arch=gfx950 -o mfma_gap_a mfma_gap.hip                                  (AGPR accumulators)
//               hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o mfma_gap_v mfma_gap.hip     (VGPR accumulators)
#include <hip/hip_runtime.h>
// -DLCE_PROBE_SCALE=0: the compiler then selects the UNSCALED v_mfma_f32_32x32x64_f8f6f4 (8-byte encoding) instead of the
// v_mfma_scale_... pair (16 bytes: v_mfma_ld_scale + v_mfma)
#ifndef LCE_PROBE_SCALE
#define LCE_PROBE_SCALE 0x7F7F7F7F
#endif
#include <cstdint>
#include <cstdio>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

enum { F_VALU_MUL, F_VALU_PERM, F_DS_READ128, F_DS_WRITE2, F_STORE16, F_PK_MUL, F_SALU, F_VALU_OTHER_ACC, F_DS_WRITE128, F_WAIT, F_NOP,
       F_VALU_LIT, F_KINDS };
static const char* kNames[] = {"v_mul_f32", "v_perm_b32", "ds_read_b128", "ds_write2_b32", "buffer_store_dwordx4", "v_pk_mul_f32",
                               "s_add_u32", "v_mul_f32 reading the idle accumulator set", "ds_write_b128",
                               "s_waitcnt lgkmcnt(0) (nothing pending)", "s_nop 0", "v_and_b32 with a 32-bit literal"};

__device__ unsigned long long g_cycles[1024];

template <int KIND>
__device__ __forceinline__ void filler(float (&f)[8], unsigned (&x)[8], u4& lv, char* lds, int lane, __amdgpu_buffer_rsrc_t r, unsigned& sacc,
                                       const v16f& idle, int i) {
  if constexpr (KIND == F_VALU_MUL) {
    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i & 7]) : "v"(f[(i + 1) & 7]));
  } else if constexpr (KIND == F_VALU_PERM) {
    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(x[(i + 3) & 7]), "s"(0x05010400u));
  } else if constexpr (KIND == F_DS_READ128) {
    u4 t;
    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(lane * 16 + (i & 7) * 1024)));
    lv = t;   // (waited for at the end of the iteration)
  } else if constexpr (KIND == F_DS_WRITE2) {
    asm volatile("ds_write2_b32 %0, %1, %2 offset1:64" ::"v"((unsigned)(lane * 4 + (i & 7) * 1024)), "v"(f[i & 7]), "v"(f[(i + 1) & 7]));
  } else if constexpr (KIND == F_DS_WRITE128) {
    asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(lane * 16 + (i & 7) * 1024)), "v"(lv));
  } else if constexpr (KIND == F_STORE16) {
    __builtin_amdgcn_raw_buffer_store_b128(lv, r, (unsigned)(lane * 16 + (i & 7) * 1024), 0, 2);
  } else if constexpr (KIND == F_PK_MUL) {
    v2f a = {f[(2 * i) & 6], f[((2 * i) & 6) + 1]};
    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(a));
    f[(2 * i) & 6] = a[0];
    f[((2 * i) & 6) + 1] = a[1];
  } else if constexpr (KIND == F_SALU) {
    asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
  } else if constexpr (KIND == F_WAIT) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else if constexpr (KIND == F_NOP) {
    asm volatile("s_nop 0");
  } else if constexpr (KIND == F_VALU_LIT) {
    asm volatile("v_and_b32 %0, 0x12345677, %0" : "+v"(x[i & 7]));
  } else if constexpr (KIND == F_VALU_OTHER_ACC) {
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(f[i & 7]) : "v"(idle[i & 15]), "v"(f[(i + 1) & 7]));
  }
}

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void gap(float* out, int iters) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v8i a = {0x22222222, 0x2A2A2A2A, (int)0xA2A2A2A2, 0x22222222, 0, 0, 0, 0};
  v8i b = {0x2222AAAA, 0x2A2A2A2A, 0x22222222, (int)0xAAAA2222, 0, 0, 0, 0};
  a[0] ^= (lane * 0x01010101) & 0x88888888;
  v16f c0, c1, idle;
#pragma unroll
  for (int j = 0; j < 16; ++j) { c0[j] = 0.f; c1[j] = 0.f; idle[j] = (float)(lane + j); }
  asm volatile("" : "+v"(idle));
  float f[8];
  unsigned x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { f[j] = 1.0f + lane * 1e-6f * j; x[j] = lane * 2654435761u + j; }
  u4 lv = {x[0], x[1], x[2], x[3]};
  unsigned sacc = 0;
  float* mine = out + 4096 + ((size_t)blockIdx.x * 4 + wave) * 4096;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 16384, 0x00020000);
#ifdef LCE_PROBE_B_AGPR   // the B operand lives in accumulation registers (as the streaming kernel's filter bank does)
  asm volatile("" : "+a"(b));
#endif
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, LCE_PROBE_SCALE, 0, LCE_PROBE_SCALE);
      asm volatile("" : "+v"(c0));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N; ++i) filler<KIND>(f, x, lv, lds, lane, r, sacc, idle, s * 16 + i);
      __builtin_amdgcn_sched_barrier(0);
      c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, LCE_PROBE_SCALE, 0, LCE_PROBE_SCALE);
      asm volatile("" : "+v"(c1));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N; ++i) filler<KIND>(f, x, lv, lds, lane, r, sacc, idle, s * 16 + 8 + i);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == F_DS_READ128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) sum += c0[j] + c1[j];
#pragma unroll
  for (int j = 0; j < 8; ++j) sum += f[j] + (float)x[j];
  sum += (float)lv[0] + (float)sacc;
  if (sum == 12345.678f) out[threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_cycles[blockIdx.x] = t1 - t0;
}

template <int KIND, int N>
static void run(float* out, int iters) {
  hipFuncSetAttribute((const void*)gap<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  gap<KIND, N><<<256, 256, 100 * 1024>>>(out, iters);
  hipDeviceSynchronize();
  gap<KIND, N><<<256, 256, 100 * 1024>>>(out, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> c(256);
  hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_cycles), 256 * sizeof(unsigned long long));
  double s = 0;
  for (auto v : c) s += (double)v;
  printf(" %6.1f", s / 256 / ((double)iters * 8));
  fflush(stdout);
}

template <int KIND>
static void kind(float* out, int iters) {
  printf("%-44s", kNames[KIND]);
  run<KIND, 0>(out, iters); run<KIND, 1>(out, iters); run<KIND, 2>(out, iters); run<KIND, 3>(out, iters);
  run<KIND, 4>(out, iters); run<KIND, 5>(out, iters); run<KIND, 6>(out, iters); run<KIND, 8>(out, iters);
  printf("\n");
}

int main() {
  float* out;
  hipMalloc(&out, (4096 + 256 * 4 * 4096) * sizeof(float));
  const int iters = 400;
  printf("cycles per MFMA with N fillers behind every MFMA;  N =     0      1      2      3      4      5      6      8\n");
  kind<F_VALU_MUL>(out, iters);
  kind<F_VALU_PERM>(out, iters);
  kind<F_VALU_OTHER_ACC>(out, iters);
  kind<F_PK_MUL>(out, iters);
  kind<F_SALU>(out, iters);
  kind<F_WAIT>(out, iters);
  kind<F_NOP>(out, iters);
  kind<F_VALU_LIT>(out, iters);
  kind<F_DS_READ128>(out, iters);
  kind<F_DS_WRITE2>(out, iters);
  kind<F_DS_WRITE128>(out, iters);
  kind<F_STORE16>(out, iters);
  return 0;
}
