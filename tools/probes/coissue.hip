// What does a wave that streams FP4 MFMAs lose when the OTHER wave on its SIMD runs the kind of work
// a neighbouring block's prologue / epilogue consists of?  (The direct LceBconv2d kernel keeps two
// 4-wave blocks per CU; measured with tools/phases.py they run in anti-phase: one in its K loop while
// the other expands its input halo or transforms and stores its output.)
//
// Two 256-thread blocks per CU (forced by 74 KiB of dynamic LDS each).  The first block to arrive on a
// CU (per-CU ticket) runs ITERS x 8 independent v_mfma_scale_f32_32x32x64_f8f6f4 and reports cycles per
// MFMA (s_memtime); the second runs the partner work in a loop until the first is done.
// Build: hipcc -O3 --offload-arch=gfx950 -o coissue coissue.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

enum { P_NONE, P_VALU, P_VALU_MULLO, P_LDS_WRITE128, P_LDS_READ128, P_STORE16, P_SALU, P_LDS_WRITE32, P_VALU_F32 };

__device__ unsigned int g_ticket[4096];
__device__ unsigned int g_done[4096];
__device__ unsigned long long g_cycles[4096];

template <int PARTNER>
__global__ __launch_bounds__(256, 4) void coissue(float* out, int iters) {
  extern __shared__ char lds[];
  __shared__ unsigned role_s, cu_s;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const unsigned cu = ((xcc & 15) << 8) | ((hw >> 8) & 0xFF);      // xcc, se, sh, cu
    cu_s = cu;
    role_s = atomicAdd(&g_ticket[cu], 1u);
  }
  __syncthreads();
  const unsigned cu = cu_s;
  if (role_s == 0) {
    v8i a = {0x22222222, 0x2A2A2A2A, (int)0xA2A2A2A2, 0x22222222, 0, 0, 0, 0};
    v8i b = {0x2222AAAA, 0x2A2A2A2A, 0x22222222, (int)0xAAAA2222, 0, 0, 0, 0};
    a[0] ^= (lane * 0x01010101) & 0x88888888;
    v16f c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    // let the partner get going
    __builtin_amdgcn_s_sleep(100);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        c[i & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i & 3], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) sum += c[i][j];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (sum == 12345.678f) out[threadIdx.x] = sum;
    if (threadIdx.x == 0) {
      g_cycles[cu] = t1 - t0;
      __hip_atomic_store(&g_done[cu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (role_s >= 1) {
    unsigned x0 = lane, x1 = lane * 3, x2 = lane * 5, x3 = lane * 7;
    float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3;
    u4 v = {x0, x1, x2, x3};
    float* mine = out + 4096 + (size_t)blockIdx.x * 65536;
    unsigned spin = 0;
    while (true) {
      if (PARTNER == P_NONE) {
        __builtin_amdgcn_s_sleep(64);
      } else if (PARTNER == P_VALU) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          asm volatile("v_perm_b32 %0, %0, %1, %4\n\tv_perm_b32 %1, %1, %2, %4\n\tv_perm_b32 %2, %2, %3, %4\n\tv_perm_b32 %3, %3, %0, %4\n\t"
                       "v_and_b32 %0, %5, %0\n\tv_and_b32 %1, %5, %1\n\tv_and_b32 %2, %5, %2\n\tv_and_b32 %3, %5, %3"
                       : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(0x04000400u), "s"(0x03030303u));
      } else if (PARTNER == P_VALU_F32) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          asm volatile("v_med3_f32 %0, %0, 1.0, %4\n\tv_med3_f32 %1, %1, 1.0, %4\n\tv_med3_f32 %2, %2, 1.0, %4\n\tv_med3_f32 %3, %3, 1.0, %4\n\t"
                       "v_fma_f32 %0, %0, %5, 0.5\n\tv_fma_f32 %1, %1, %5, 0.5\n\tv_fma_f32 %2, %2, %5, 0.5\n\tv_fma_f32 %3, %3, %5, 0.5"
                       : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "s"(9.0f), "s"(1.0001f));
      } else if (PARTNER == P_VALU_MULLO) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          asm volatile("v_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %1, %1, %2\n\tv_mul_lo_u32 %2, %2, %3\n\tv_mul_lo_u32 %3, %3, %0"
                       : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
      } else if (PARTNER == P_LDS_WRITE128) {
#pragma unroll
        for (int k = 0; k < 16; ++k) *(u4*)(lds + ((threadIdx.x * 16 + k * 4096) & 0xFFFF)) = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (PARTNER == P_LDS_WRITE32) {
#pragma unroll
        for (int k = 0; k < 32; ++k) *(unsigned*)(lds + ((threadIdx.x * 4 + k * 1024) & 0xFFFF)) = x0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (PARTNER == P_LDS_READ128) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          u4 r = *(volatile u4*)(lds + ((threadIdx.x * 16 + k * 4096) & 0xFFFF));
          x0 ^= r[0];
        }
      } else if (PARTNER == P_STORE16) {
        const v4f fv = {f0, f1, f2, f3};
#pragma unroll
        for (int k = 0; k < 8; ++k)
          __builtin_nontemporal_store(fv, (v4f*)(mine + ((spin * 8 + k) & 63) * 1024 + threadIdx.x * 4));
      } else if (PARTNER == P_SALU) {
        unsigned s = __builtin_amdgcn_readfirstlane(spin);
#pragma unroll
        for (int k = 0; k < 64; ++k) asm volatile("s_mul_i32 %0, %0, 3\n\ts_add_u32 %0, %0, 1" : "+s"(s));
        x0 ^= s;
      }
      ++spin;
      if ((spin & 3) == 0 && __hip_atomic_load(&g_done[cu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      if (spin > (1u << 22)) break;
    }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u && f0 + f1 + f2 + f3 == 1.5f) out[threadIdx.x] = f0;
    if (threadIdx.x == 0 && role_s == 1) g_cycles[2048 + cu] = spin;
  }
}

template <int PARTNER>
static void run(const char* name, float* out, int iters, int per_cu = 2) {
  void *t, *d, *c;
  hipGetSymbolAddress(&t, HIP_SYMBOL(g_ticket));
  hipGetSymbolAddress(&d, HIP_SYMBOL(g_done));
  hipGetSymbolAddress(&c, HIP_SYMBOL(g_cycles));
  hipFuncSetAttribute((const void*)coissue<PARTNER>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(t, 0, 4096 * 4); hipMemset(d, 0, 4096 * 4); hipMemset(c, 0, 4096 * 8);
    coissue<PARTNER><<<256 * per_cu, 256, (160 / per_cu - 6) * 1024>>>(out, iters);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> cyc(4096);
  std::vector<unsigned> tk(4096);
  hipMemcpy(cyc.data(), c, 4096 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(tk.data(), t, 4096 * 4, hipMemcpyDeviceToHost);
  double sum = 0, spins = 0; int n = 0, paired = 0;
  for (int i = 0; i < 2048; ++i) {
    if ((int)tk[i] == per_cu && cyc[i]) { sum += (double)cyc[i]; spins += (double)cyc[2048 + i]; ++paired; }
    if (tk[i]) ++n;
  }
  printf("{\"partner\": \"%s\", \"cus_seen\": %d, \"cus_with_2_blocks\": %d, \"cycles_per_mfma_per_simd\": %.2f, \"partner_loops\": %.0f}\n",
         name, n, paired, paired ? sum / paired / (iters * 8.0) : 0.0, paired ? spins / paired : 0.0);
}

int main() {
  float* out;
  hipMalloc(&out, (4096 + 512 * 65536) * sizeof(float));
  const int iters = 2000;
  run<P_VALU>("3 per CU: 1 MFMA + 2 x VALU v_perm+v_and", out, iters, 3);
  run<P_VALU>("4 per CU: 1 MFMA + 3 x VALU v_perm+v_and", out, iters, 4);
  run<P_NONE>("idle (s_sleep)", out, iters);
  run<P_VALU>("VALU v_perm+v_and, 4 chains", out, iters);
  run<P_VALU_F32>("VALU med3+mul+add f32, 4 chains", out, iters);
  run<P_VALU_MULLO>("VALU v_mul_lo_u32 (quarter rate)", out, iters);
  run<P_SALU>("SALU s_mul chain", out, iters);
  run<P_LDS_WRITE128>("ds_write_b128 x16 + wait", out, iters);
  run<P_LDS_WRITE32>("ds_write_b32 x32 + wait", out, iters);
  run<P_LDS_READ128>("ds_read_b128 x16", out, iters);
  run<P_STORE16>("global_store_dwordx4 nt x8", out, iters);
  return 0;
}
