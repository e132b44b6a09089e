// Synthetic K-step models: what matrix-pipe utilisation does a block structure reach on gfx950 BEFORE
// it is built?  A 256-thread block runs STEPS K-steps; a step is
//     s_barrier, NMFMA independent v_mfma_scale_f32_32x32x64_f8f6f4 per wave, interleaved with
//     NR128 ds_read_b128 (fragments), NR32 ds_read_b32 + NVALU VALU (in-register bit->FP4 expansion),
//     NW128 ds_write_b128, NDMA buffer_load_dwordx4 ... lds (1 KiB weight pieces from an L2-resident buffer),
// and the block ends with an "epilogue" of NSTORE 16-byte non-temporal stores per wave after EVALU VALU
// instructions; a "prologue" of PVALU VALU + PW128 ds_write_b128 starts it.  Blocks per CU are set by
// the dynamic-LDS size.  Output: cycles of wall time per MFMA per SIMD (32.0 = the pipe's rate).
// Build: hipcc -O3 --offload-arch=gfx950 -o kstep_model kstep_model.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#define VALU4(x0, x1, x2, x3)                                                                            \
  asm volatile("v_perm_b32 %0, %0, %1, %4\n\tv_perm_b32 %1, %1, %2, %4\n\tv_and_b32 %2, %5, %2\n\tv_and_b32 %3, %5, %3" \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(0x04000400u), "s"(0x03030303u))

template <int NMFMA, int NR128, int NR32, int NVALU, int NW128, int NDMA, int PVALU, int PW128, int EVALU, int NSTORE>
__global__ __launch_bounds__(256, 2) void kstep(float* out, const uint8_t* wts, int steps, int tiles) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wts), 0, 256 * 1024, 0x00020000);
  unsigned x0 = lane, x1 = lane * 3, x2 = lane * 5, x3 = lane * 7;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    v8i a = {0x22222222, 0x2A2A2A2A, (int)0xA2A2A2A2, 0x22222222, 0, 0, 0, 0};
    v8i b = {0x2222AAAA, 0x2A2A2A2A, 0x22222222, (int)0xAAAA2222, 0, 0, 0, 0};
    a[0] ^= (lane * 0x01010101) & 0x88888888;
    v16f c[NMFMA];
#pragma unroll
    for (int i = 0; i < NMFMA; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    // ---- prologue
#pragma unroll
    for (int k = 0; k < PVALU / 4; ++k) VALU4(x0, x1, x2, x3);
#pragma unroll
    for (int k = 0; k < PW128; ++k) *(u4*)(lds + ((threadIdx.x * 16 + k * 4096) & 0x3FFF)) = u4{x0, x1, x2, x3};
    // ---- K loop
    for (int ks = 0; ks < steps; ++ks) {
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      u4 fr[NR128 > 0 ? NR128 : 1];
#pragma unroll
      for (int k = 0; k < NR128; ++k) fr[k] = *(volatile u4*)(lds + ((threadIdx.x * 16 + k * 4096 + ks * 64) & 0x3FFF));
      unsigned rb[NR32 > 0 ? NR32 : 1];
#pragma unroll
      for (int k = 0; k < NR32; ++k) rb[k] = *(volatile unsigned*)(lds + ((threadIdx.x * 36 + k * 1024 + ks * 8) & 0x3FFC));
#pragma unroll
      for (int k = 0; k < NDMA; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(lds + 16384 + ((wave * NDMA + k) & 7) * 1024), 16,
                                                 (unsigned)(((ks * 8 + wave * NDMA + k) & 255) * 1024 + lane * 16), 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NMFMA; ++i) {
        if (NR128 > 0) { a[1] ^= (int)fr[i % NR128][0] & 0x88888888; }
        c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
#pragma unroll
      for (int k = 0; k < NR32; ++k) x0 ^= rb[k];
#pragma unroll
      for (int k = 0; k < NVALU / 4; ++k) VALU4(x0, x1, x2, x3);
#pragma unroll
      for (int k = 0; k < NW128; ++k) *(u4*)(lds + ((threadIdx.x * 16 + k * 4096) & 0x3FFF)) = u4{x0, x1, x2, x3};
      if (NVALU > 0) b[2] ^= (int)x0 & 0x88888888;
    }
    __builtin_amdgcn_s_waitcnt(0);
    // ---- epilogue
#pragma unroll
    for (int k = 0; k < EVALU / 4; ++k) VALU4(x0, x1, x2, x3);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NMFMA; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) sum += c[i][j];
    const v4f fv = {sum, (float)x0, (float)x1, (float)x2};
    float* base = out + (size_t)t * (NSTORE > 0 ? NSTORE : 1) * 1024;
#pragma unroll
    for (int s = 0; s < NSTORE; ++s) __builtin_nontemporal_store(fv, (v4f*)(base + (size_t)((s * 4 + wave) * 256 + lane * 4)));
    if (NSTORE == 0 && sum == 12345.678f) out[threadIdx.x] = sum + x3;
  }
}

template <int NMFMA, int NR128, int NR32, int NVALU, int NW128, int NDMA, int PVALU, int PW128, int EVALU, int NSTORE>
static void run(const char* name, float* out, const uint8_t* wts, int steps, int tiles, int lds_kb) {
  auto fn = kstep<NMFMA, NR128, NR32, NVALU, NW128, NDMA, PVALU, PW128, EVALU, NSTORE>;
  (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  fn<<<tiles, 256, lds_kb * 1024>>>(out, wts, steps, tiles);
  (void)hipDeviceSynchronize();
  const int reps = 5;
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) fn<<<tiles, 256, lds_kb * 1024>>>(out, wts, steps, tiles);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double macs = (double)tiles * 4 * steps * NMFMA * 32.0 * 32 * 64;
  printf("{\"model\": \"%s\", \"blocks_per_cu_by_lds\": %d, \"ms\": %.4f, \"mac_per_s\": %.3e, \"mfma_pipe_frac_of_4.3e15\": %.3f, \"store_TBps\": %.2f}\n",
         name, 160 / lds_kb, ms, macs / (ms * 1e-3), macs / (ms * 1e-3) / 4.3e15,
         (double)tiles * NSTORE * 4 * 1024.0 / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  float* out;
  uint8_t* wts;
  (void)hipMalloc(&out, (size_t)12800 * 32 * 4096 + 4096);
  (void)hipMalloc(&wts, 512 * 1024);
  (void)hipMemset(wts, 0x2A, 512 * 1024);
  // L0 = 6272 tiles of 128 px x 256 ch (36 K-steps of 8 MFMAs per wave, 32 stores per wave) -- or 12544 half tiles
  // today's kernel, idealised: 2 blocks/CU, 8 MFMA + 6 frag reads + 2 DMA per step, prologue 200 VALU + 12 LDS writes, epilogue 384 VALU + 32 stores
  run<8, 6, 0, 0, 0, 2, 200, 12, 384, 32>("128x256 today (2/CU): frags 6, dma 2, P 200 valu, E 384 valu + 32 st", out, wts, 36, 6272, 74);
  run<8, 6, 0, 0, 0, 2, 200, 12, 384, 0>("128x256 today, no stores", out, wts, 36, 6272, 74);
  run<8, 6, 0, 0, 0, 2, 200, 12, 384, 32>("128x256 at 3 blocks/CU (LDS 52 KB; needs <=168 VGPR)", out, wts, 36, 6272, 52);
  // 128x128 blocks, 4 per CU: phased FP4 halo (frags from LDS as today), twice the prologues
  run<4, 4, 0, 0, 0, 1, 200, 12, 192, 16>("128x128 phased halo (4/CU): frags 4, dma 1", out, wts, 36, 12544, 40);
  run<4, 4, 0, 0, 0, 1, 200, 12, 192, 16>("128x128 same at 2/CU", out, wts, 36, 12544, 74);
  // 128x128, bit halo, A fragments expanded in registers (2 x (ds_read_b32 + 17 VALU)), B fragments from LDS
  run<4, 2, 2, 34, 0, 1, 40, 3, 192, 16>("128x128 bit halo, in-register expansion (4/CU)", out, wts, 36, 12544, 40);
  run<4, 2, 2, 34, 0, 1, 40, 3, 192, 16>("128x128 bit halo, in-register expansion (5/CU)", out, wts, 36, 12544, 32);
  // 128x128, bit halo, A ring filled by in-block expansion (1 word per thread per step)
  run<4, 4, 1, 17, 1, 1, 40, 3, 192, 16>("128x128 bit halo + A ring (4/CU)", out, wts, 36, 12544, 40);
  // 128x256 with bit halo + in-register expansion, 2/CU (VGPR-bound) -- prologue shrinks only
  run<8, 4, 2, 34, 0, 2, 40, 3, 384, 32>("128x256 bit halo, in-register expansion (2/CU)", out, wts, 36, 6272, 74);
  return 0;
}
