// What does one LDS-DMA instruction (buffer_load_dwordx4 ... lds, 1 KiB per wave) cost the wave that issues
// it between its MFMAs?  One wave per SIMD (1 block per CU), steps of 8 independent FP4 MFMAs with NDMA copies
// woven in after the first MFMAs, as in the LceBconv2d K-step; the weights come from an L2-resident buffer.
// Forms: per-lane VGPR offset (offen) + scalar offset; the same with an instruction-immediate offset for the
// second copy; no VGPR at all (the resource adds TID * 16 itself: ADD_TID_ENABLE).
// Output: cycles per K-step (s_memtime) -- 8 MFMAs alone are ~262.
// Build: hipcc -O3 --offload-arch=gfx950 -o dma_cost dma_cost.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <type_traits>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ unsigned long long g_cyc[1024];
__device__ unsigned g_sink[256 * 1024 * 1024 / 4];   // 1 MiB per block for the store forms
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int NDMA, int FORM, int WAIT>
__global__ __launch_bounds__(256, 1) void dma_cost(const uint8_t* wts, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // FORM 2: stride 16 + ADD_TID_ENABLE (word3 bit 23): the hardware adds TID * stride to the address
  const rsrc_t r = FORM == 2 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wts), 16, 512 * 1024, 0x00020000 | (1 << 23))
                             : __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wts), 0, 512 * 1024, 0x00020000);
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g_sink + (size_t)blockIdx.x * (1024 * 1024 / 4), 0, 1024 * 1024, 0x00020000);
  v8i a = {0x22222222, 0x2A2A2A2A, (int)0xA2A2A2A2, 0x22222222, 0, 0, 0, 0};
  v8i b = {0x2222AAAA, 0x2A2A2A2A, 0x22222222, (int)0xAAAA2222, 0, 0, 0, 0};
  a[0] ^= (lane * 0x01010101) & 0x88888888;
  v16f c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
  const unsigned voff = lane * 16;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int ks = 0; ks < steps; ++ks) {
    const unsigned soff = (unsigned)((ks & 31) * 8192 + wave * 2048);
#define lds_at(k) ((__attribute__((address_space(3))) void*)(lds + (ks % 3) * 8192 + wave * 2048 + (k) * 1024))
    if (FORM == 7) {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        if (k == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 0, 0);
        else if (k == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 1024, 0);
        else if (k == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 2048, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 3072, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (!(FORM == 4 && wave == 0))
        c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      if (i < NDMA && FORM < 3) {
        if (FORM == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(i), 16, voff, soff + i * 1024, 0, 0);
        else if (FORM == 1) {
          if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 0, 0);
          else if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 1024, 0);
          else if (i == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 2048, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 3072, 0);
        } else if (FORM == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(i), 16, 0, soff + i * 1024, 0, 0);
      }
      // FORM 5/6/7: the wave's NDMA copies BACK TO BACK -- after the first MFMA / after the last MFMA / (see below) before the first
      if ((FORM == 5 && i == 0) || (FORM == 6 && i == 7)) {
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
          if (k == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 0, 0);
          else if (k == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 1024, 0);
          else if (k == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 2048, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 3072, 0);
        }
      }
      // FORM 8: staggered by wave -- wave w issues its copies after MFMAs 2w and 2w+1, so the four waves' copies
      // never meet in the address path; FORM 9: only wave 0 copies (NDMA per step), nobody to collide with
      if (FORM == 8 && NDMA == 2 && (i >> 1) == wave) {
        if ((i & 1) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 1024, 0);
      }
      if (FORM == 8 && NDMA == 1 && i == 2 * wave) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(0), 16, voff, soff, 0, 0);
      if (FORM == 9 && wave == 0 && i < NDMA) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(i), 16, voff, soff + i * 1024, 0, 0);
      // FORM 10 / 11: one 16-byte-per-lane non-temporal STORE per step woven in after the third MFMA (the drain of a
      // previous tile through the K loop), without / with the step's two copies
      if ((FORM == 10 || FORM == 11) && i == 2)
        __builtin_amdgcn_raw_buffer_store_b128(v4u{(unsigned)ks, 2u, 3u, (unsigned)lane}, rs, (unsigned)(((ks * 4 + wave) & 1023) * 1024 + lane * 16), 0, 2);
      if (FORM == 11 && i < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(i), 16, voff, soff + i * 1024, 0, 0);
      if (FORM == 3 && wave == 0 && i < NDMA) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(4 * i), 16, voff, soff + i * 4096, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(4 * i), 16, voff, soff + i * 4096, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(4 * i), 16, voff, soff + i * 4096, 2048, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(4 * i), 16, voff, soff + i * 4096, 3072, 0);
      }
      if (FORM == 4 && wave == 0 && i < NDMA) {   // wave 0 is a pure loader: no MFMAs of its own (see below)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(4 * i), 16, voff, soff + i * 4096, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(4 * i), 16, voff, soff + i * 4096, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(4 * i), 16, voff, soff + i * 4096, 2048, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_at(4 * i), 16, voff, soff + i * 4096, 3072, 0);
      }
    }
    if (WAIT == 1) __builtin_amdgcn_s_waitcnt((((NDMA * 2) >> 4) & 3) << 14 | (0xF << 8) | (0x7 << 4) | ((NDMA * 2) & 0xF));
    if (WAIT == 2) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); }
#undef lds_at
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += c[i][j];
  if (sum == 12345.678f) out[threadIdx.x] = sum + lds[lane];
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_cyc[blockIdx.x] = t1 - t0;
}

// Register-staged alternative: buffer_load_dwordx4 into VGPRs now, ds_write_b128 of the piece loaded TWO steps
// ago (three rotating register sets), counted vmcnt -- the classic global -> register -> LDS pipeline.
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int NDMA>
__global__ __launch_bounds__(256, 1) void reg_staged(const uint8_t* wts, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wts), 0, 512 * 1024, 0x00020000);
  v8i a = {0x22222222, 0x2A2A2A2A, (int)0xA2A2A2A2, 0x22222222, 0, 0, 0, 0};
  v8i b = {0x2222AAAA, 0x2A2A2A2A, 0x22222222, (int)0xAAAA2222, 0, 0, 0, 0};
  a[0] ^= (lane * 0x01010101) & 0x88888888;
  v16f c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
  const unsigned voff = lane * 16;
  u4 st[3][NDMA > 0 ? NDMA : 1];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int k = 0; k < NDMA; ++k) st[s][k] = u4{0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  auto one_step = [&](int ks, auto sc) {
    constexpr int S = decltype(sc)::value;   // register set loaded this step; set (S + 1) % 3 was loaded two steps ago
    const unsigned soff = (unsigned)((ks & 31) * 8192 + wave * 2048);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      if (i < NDMA) st[S][i] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff + i * 1024, 0);
      if (i >= 4 && i - 4 < NDMA)
        *(u4*)(lds + (ks % 3) * 8192 + wave * 2048 + (i - 4) * 1024 + lane * 16) = st[(S + 1) % 3][i - 4];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
  };
  for (int ks = 0; ks + 2 < steps; ks += 3) {
    one_step(ks, std::integral_constant<int, 0>{});
    one_step(ks + 1, std::integral_constant<int, 1>{});
    one_step(ks + 2, std::integral_constant<int, 2>{});
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += c[i][j];
  if (sum == 12345.678f) out[threadIdx.x] = sum + lds[lane];
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_cyc[blockIdx.x] = t1 - t0;
}

// A fifth wave as the block's loader, running a 3-deep ring ahead of the four MFMA waves: per step it issues all
// NCOPY copies, waits until only the last two steps' copies are in flight, and meets the others at the barrier.
template <int NCOPY>
__global__ __launch_bounds__(320, 1) void loader_wave(const uint8_t* wts, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wts), 0, 512 * 1024, 0x00020000);
  v8i a = {0x22222222, 0x2A2A2A2A, (int)0xA2A2A2A2, 0x22222222, 0, 0, 0, 0};
  v8i b = {0x2222AAAA, 0x2A2A2A2A, 0x22222222, (int)0xAAAA2222, 0, 0, 0, 0};
  a[0] ^= (lane * 0x01010101) & 0x88888888;
  v16f c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
  const unsigned voff = lane * 16;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave == 4) {
    for (int ks = 0; ks < steps; ++ks) {
      const unsigned soff = (unsigned)((ks & 31) * 8192);
#pragma unroll
      for (int k = 0; k < NCOPY; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + (ks % 3) * 8192 + k * 1024), 16, voff, soff + k * 1024, 0, 0);
      __builtin_amdgcn_s_waitcnt((((NCOPY * 2) >> 4) & 3) << 14 | (0xF << 8) | (0x7 << 4) | ((NCOPY * 2) & 0xF));
      __builtin_amdgcn_s_barrier();
    }
  } else {
    for (int ks = 0; ks < steps; ++ks) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      __builtin_amdgcn_s_barrier();
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += c[i][j];
  if (sum == 12345.678f) out[threadIdx.x] = sum + lds[lane];
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_cyc[blockIdx.x] = t1 - t0;
}

template <int NCOPY>
static void run_loader(const char* name, const uint8_t* w, float* out) {
  const int steps = 2000;
  void (*fn)(const uint8_t*, float*, int) = loader_wave<NCOPY>;
  (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  void* c;
  (void)hipGetSymbolAddress(&c, HIP_SYMBOL(g_cyc));
  for (int rep = 0; rep < 2; ++rep) {
    loader_wave<NCOPY><<<256, 320, 150 * 1024>>>(w, out, steps);
    (void)hipDeviceSynchronize();
  }
  std::vector<unsigned long long> cyc(256);
  (void)hipMemcpy(cyc.data(), c, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : cyc) s += (double)v;
  printf("{\"probe\": \"%s\", \"copies_per_block_step\": %d, \"cycles_per_step\": %.1f}\n", name, NCOPY, s / 256 / steps);
  fflush(stdout);
}

template <int NDMA>
static void run_reg(const char* name, const uint8_t* w, float* out) {
  const int steps = 1998;
  void (*fn)(const uint8_t*, float*, int) = reg_staged<NDMA>;
  (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  void* c;
  (void)hipGetSymbolAddress(&c, HIP_SYMBOL(g_cyc));
  for (int rep = 0; rep < 2; ++rep) {
    reg_staged<NDMA><<<256, 256, 150 * 1024>>>(w, out, steps);
    (void)hipDeviceSynchronize();
  }
  std::vector<unsigned long long> cyc(256);
  (void)hipMemcpy(cyc.data(), c, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : cyc) s += (double)v;
  printf("{\"probe\": \"%s\", \"dma_per_step\": %d, \"cycles_per_step\": %.1f}\n", name, NDMA, s / 256 / steps);
  fflush(stdout);
}

template <int NDMA, int FORM, int WAIT>
static void run(const char* name, const uint8_t* w, float* out) {
  const int steps = 2000;
  void (*fn)(const uint8_t*, float*, int) = dma_cost<NDMA, FORM, WAIT>;
  (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  void* c;
  (void)hipGetSymbolAddress(&c, HIP_SYMBOL(g_cyc));
  for (int rep = 0; rep < 2; ++rep) {
    dma_cost<NDMA, FORM, WAIT><<<256, 256, 150 * 1024>>>(w, out, steps);
    (void)hipDeviceSynchronize();
  }
  std::vector<unsigned long long> cyc(256);
  (void)hipMemcpy(cyc.data(), c, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : cyc) s += (double)v;
  printf("{\"probe\": \"%s\", \"dma_per_step\": %d, \"cycles_per_step\": %.1f}\n", name, NDMA, s / 256 / steps);
  fflush(stdout);
}

int main() {
  uint8_t* w;
  float* out;
  (void)hipMalloc(&w, 1024 * 1024);
  (void)hipMemset(w, 0x2A, 1024 * 1024);
  (void)hipMalloc(&out, 4096);
  run<0, 0, 0>("8 MFMAs alone", w, out);
  run<0, 0, 2>("8 MFMAs + barrier", w, out);
  run<1, 0, 1>("offen + soffset, counted wait", w, out);
  run<2, 0, 1>("offen + soffset, counted wait", w, out);
  run<4, 0, 1>("offen + soffset, counted wait", w, out);
  run<2, 1, 1>("second copy by immediate offset (one M0)", w, out);
  run<4, 1, 1>("copies 2-4 by immediate offset (one M0)", w, out);
  run<2, 0, 2>("offen, lgkmcnt(0) + barrier per step (no vmcnt wait)", w, out);
  run<2, 1, 2>("imm pair, lgkmcnt(0) + barrier per step (no vmcnt wait)", w, out);
  run<2, 5, 2>("2 copies back to back after the FIRST MFMA, barrier per step", w, out);
  run<2, 6, 2>("2 copies back to back after the LAST MFMA, barrier per step", w, out);
  run<2, 7, 2>("2 copies back to back BEFORE the first MFMA, barrier per step", w, out);
  run<4, 5, 2>("4 copies back to back after the FIRST MFMA, barrier per step", w, out);
  run<4, 6, 2>("4 copies back to back after the LAST MFMA, barrier per step", w, out);
  run<1, 6, 2>("1 copy after the LAST MFMA, barrier per step", w, out);
  run<2, 3, 2>("wave 0 issues all 8 copies (and its MFMAs), barrier per step", w, out);
  run<2, 4, 2>("wave 0 is a pure loader (8 copies, no MFMAs), waves 1-3 compute, barrier per step", w, out);
  run<1, 4, 2>("pure loader wave, 4 copies per step", w, out);
  run<2, 8, 2>("staggered: wave w issues its 2 copies after MFMAs 2w, 2w+1, barrier per step", w, out);
  run<1, 8, 2>("staggered: wave w issues its 1 copy after MFMA 2w, barrier per step", w, out);
  run<2, 8, 1>("staggered: 2 copies per wave, counted wait, no barrier", w, out);
  run<1, 9, 2>("only wave 0 copies, 1 per step (no one to collide with), barrier per step", w, out);
  run<2, 9, 2>("only wave 0 copies, 2 per step, barrier per step", w, out);
  run<4, 9, 2>("only wave 0 copies, 4 per step, barrier per step", w, out);
  run<0, 10, 2>("8 MFMAs + one 16-byte nt store per step, barrier per step", w, out);
  run<2, 11, 2>("8 MFMAs + 2 copies + one 16-byte nt store per step, barrier per step", w, out);
  run<2, 0, 2>("8 MFMAs + 2 copies (reference for the line above)", w, out);
  run_loader<4>("fifth wave = loader, 3-deep ring, barrier per step", w, out);
  run_loader<8>("fifth wave = loader, 3-deep ring, barrier per step", w, out);
  run_loader<12>("fifth wave = loader, 3-deep ring, barrier per step", w, out);
  run_reg<1>("register-staged: buffer_load_dwordx4 -> VGPR, ds_write_b128 two steps later", w, out);
  run_reg<2>("register-staged: buffer_load_dwordx4 -> VGPR, ds_write_b128 two steps later", w, out);
  run_reg<4>("register-staged: buffer_load_dwordx4 -> VGPR, ds_write_b128 two steps later", w, out);
  return 0;
}
