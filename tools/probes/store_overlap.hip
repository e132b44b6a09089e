// Ceilings for the float-output LceBconv2d layer (L0: 822 MB of float stores, 4.7e11 FP4 MACs):
//   1. what the chip writes when NOTHING else runs, by store pattern
//        linear   : thread i writes 16 bytes at 16*i (grid-stride)                      [plain / non-temporal]
//        tile16   : the epilogue's pattern -- a block owns a contiguous 128-KiB tile, a wave
//                   instruction writes 2 rows x 512 B (8 full 128-byte lines)            [nt]
//        scatter16: a lane owns ONE pixel: 64 x 16-byte pieces at 1-KiB stride per
//                   instruction, four instructions complete a line (the layout an
//                   A<->B-swapped MFMA would store without a transpose)                  [nt]
//        tile4    : 4 bytes per lane, 2 full lines per instruction                       [nt]
//        scatter16_plain / tile16_plain: the same addresses with write-back stores (round 3: does L2 merge the
//                   four 32-byte pieces of a line that four consecutive instructions of one wave write?),
//                   one block per tile and 256 persistent blocks
//   2. what a kernel with L0's matrix work AND L0's stores reaches when the stores are
//      (a) a burst after the K loop, (b) spread one per K-step through the K loop --
//      no LDS, no barriers, constant operands: the upper bound of any overlap scheme.
// Build: hipcc -O3 --offload-arch=gfx950 -o store_overlap store_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kTilePix = 128, kCh = 256;                 // one block tile = 128 pixels x 256 channels
constexpr size_t kTileFloats = (size_t)kTilePix * kCh;   // 32768 floats = 128 KiB

template <bool NT>
__global__ __launch_bounds__(256) void fill_linear(v4f* out, size_t n16) {
  const v4f v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
  }
}

// MODE 0 tile16, 1 scatter16, 2 tile4
template <int MODE>
__global__ __launch_bounds__(256) void fill_tile(float* out, int tiles) {
  extern __shared__ char lds_dummy[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave & 1, wn = wave >> 1;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    float* base = out + (size_t)t * kTileFloats;
    const v4f v = {1.f, 2.f, 3.f, (float)lane};
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const int row = wm * 64 + (s >> 4) * 32 + (lane >> 5) + 2 * (s & 15);
        const int col = wn * 128 + (lane & 31) * 4;
        __builtin_nontemporal_store(v, (v4f*)(base + (size_t)row * kCh + col));
      }
    } else if (MODE == 3) {      // tile16 addresses, plain stores
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const int row = wm * 64 + (s >> 4) * 32 + (lane >> 5) + 2 * (s & 15);
        const int col = wn * 128 + (lane & 31) * 4;
        *(v4f*)(base + (size_t)row * kCh + col) = v;
      }
    } else if (MODE == 4) {      // scatter16 addresses, plain stores
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const int i = s >> 4, j = (s >> 2) & 3, g = s & 3;
        const int row = wm * 64 + i * 32 + (lane & 31);
        const int col = wn * 128 + j * 32 + 8 * g + 4 * (lane >> 5);
        *(v4f*)(base + (size_t)row * kCh + col) = v;
      }
    } else if (MODE == 1) {
      // lane = pixel (lane & 31) of a 32-pixel row block, half h = lane >> 5; instruction (i, j, g):
      // channels j*32 + 8*g + 4*h .. +3
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const int i = s >> 4, j = (s >> 2) & 3, g = s & 3;
        const int row = wm * 64 + i * 32 + (lane & 31);
        const int col = wn * 128 + j * 32 + 8 * g + 4 * (lane >> 5);
        __builtin_nontemporal_store(v, (v4f*)(base + (size_t)row * kCh + col));
      }
    } else {
#pragma unroll 16
      for (int s = 0; s < 128; ++s) {
        const int i = s >> 6, j = (s >> 4) & 3, r = s & 15;
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wn * 128 + j * 32 + (lane & 31);
        __builtin_nontemporal_store(v[0] + s, base + (size_t)row * kCh + col);
      }
    }
  }
}

// The streaming kernel's store pattern (round 3): 256 persistent blocks, block step s = one 32-pixel block of 256 channels, wave w
// stores channels [64w, 64w + 64) of 4 rows per instruction (8 instructions per step).  WHICH pixel block a block writes at
// step s: IMAGE (block b owns image b: its 98 pixel blocks are consecutive, the 256 blocks write at the same offset of
// images 3,211,264 B = 49 x 64 KiB apart, in lock-step -- what the kernel does) or INTERLEAVED (pixel block s * 256 + b: at any
// moment the 256 blocks write one contiguous 8 MiB window).  Do the lock-step writers collide on HBM channels?
// LPR = lanes per row piece of one instruction: 16 (the kernel: a wave owns 64 channels, 4 rows x 256 B per instruction),
// 32 (2 rows x 512 B: a wave would own 128 channels of 16 rows), 64 (1 row x 1 KiB: a wave would own whole rows -- needs a
// block-level transpose)
// SLAB > 0 (with IMAGE): the block's steps come in runs of SLAB consecutive pixel blocks (one segment of the streaming
// kernel: 14 pixel blocks = 8 rows of a 56-wide image), run k of block b = segment k * 256 + b -- at any moment the 256
// blocks write 256 consecutive segments (a 117 MB window) instead of 256 different images (822 MB)
// ROT > 0 (with IMAGE): block b starts ROT * b steps into its own image and wraps -- the lock-step writers are then at
// different offsets of their images (round 4: does that make the rate independent of the box?)
template <bool IMAGE, bool NT, int LPR = 16, int SLAB = 0, int ROT = 0>
__global__ __launch_bounds__(256) void fill_stream_pattern(float* out, int steps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const v4f v = {1.f, 2.f, 3.f, (float)lane};
  constexpr int RPI = 64 / LPR;            // rows per instruction
  constexpr int WPR = 64 / LPR;            // waves side by side in a row (4, 2, 1)
  const int wcol = wave % WPR, wrow = wave / WPR;
  for (int s0 = 0; s0 < steps; ++s0) {
    const int s = ROT > 0 ? (s0 + (int)blockIdx.x * ROT) % steps : s0;
    size_t pb = IMAGE ? (size_t)blockIdx.x * steps + s : (size_t)s * gridDim.x + blockIdx.x;
    if (SLAB > 0) pb = ((size_t)(s / SLAB) * gridDim.x + blockIdx.x) * SLAB + s % SLAB;
    float* base = out + pb * 32 * kCh + wcol * (LPR * 4) + (lane % LPR) * 4;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int row = wrow * (8 * RPI) + k * RPI + lane / LPR;      // every wave writes 8 KiB per step
      v4f* p = (v4f*)(base + (size_t)row * kCh);
      if (NT) __builtin_nontemporal_store(v, p);
      else *p = v;
    }
  }
}

// L0's work per block: 36 K-steps x 8 MFMAs per wave, 32 16-byte stores per wave.
// STORES: 0 none, 1 burst after the K loop, 2 one per K-step inside it.  PERSIST: grid-stride over tiles.
template <int STORES, bool MFMA>
__global__ __launch_bounds__(256, 2) void mfma_store(float* out, int tiles, int ksteps) {
  extern __shared__ char lds_dummy[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave & 1, wn = wave >> 1;
  v8i a = {0x22222222, 0x2A2A2A2A, (int)0xA2A2A2A2, 0x22222222, 0, 0, 0, 0};
  v8i b = {0x2222AAAA, 0x2A2A2A2A, 0x22222222, (int)0xAAAA2222, 0, 0, 0, 0};
  a[0] ^= (lane * 0x01010101) & 0x88888888;   // random-ish sign bits: data-dependent power
  b[1] ^= (lane * 0x11111111) & 0x88888888;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    float* base = out + (size_t)t * kTileFloats;
    v16f c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    v4f v = {1.f, 2.f, 3.f, (float)lane};
    auto store_s = [&](int s) {
      const int row = wm * 64 + (s >> 4) * 32 + (lane >> 5) + 2 * (s & 15);
      const int col = wn * 128 + (lane & 31) * 4;
      __builtin_nontemporal_store(v, (v4f*)(base + (size_t)row * kCh + col));
    };
    for (int ks = 0; ks < ksteps; ++ks) {
      if (MFMA) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
      if (STORES == 2 && ks < 32) {
        v[0] += 1.0f;
        store_s(ks);
      }
    }
    if (STORES == 1) {
#pragma unroll
      for (int s = 0; s < 32; ++s) store_s(s);
    }
    if (MFMA) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += c[i][j];
      if (sum == 12345.678f) base[threadIdx.x] = sum;   // keeps the MFMAs alive, never true
    }
  }
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) {
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const int tiles = 256 * 3136 / kTilePix;                  // 6272 tiles = L0's output at batch 256
  const size_t bytes = (size_t)tiles * kTileFloats * 4;     // 822 MB
  float* out;
  if (hipMalloc(&out, bytes + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 10;
  auto report = [&](const char* name, float ms, double macs) {
    printf("{\"probe\": \"%s\", \"ms\": %.4f, \"store_TBps\": %.3f, \"mac_per_s\": %.3e}\n", name, ms / reps,
           bytes / (ms / reps * 1e-3) / 1e12, macs / (ms / reps * 1e-3));
    fflush(stdout);
  };
#define TIME(name, macs, ...)                                    \
  do {                                                           \
    __VA_ARGS__; hipDeviceSynchronize();                         \
    hipEventRecord(e0);                                          \
    for (int r = 0; r < reps; ++r) { __VA_ARGS__; }              \
    hipEventRecord(e1); hipDeviceSynchronize();                  \
    report(name, time_ms(e0, e1), macs);                         \
  } while (0)

  const size_t n16 = bytes / 16;
  for (int bpc : {2, 8, 32}) {
    char nm[64];
    snprintf(nm, sizeof nm, "linear_plain_%dblk_per_cu", bpc);
    TIME(nm, 0.0, (fill_linear<false><<<256 * bpc, 256>>>((v4f*)out, n16)));
    snprintf(nm, sizeof nm, "linear_nt_%dblk_per_cu", bpc);
    TIME(nm, 0.0, (fill_linear<true><<<256 * bpc, 256>>>((v4f*)out, n16)));
  }
  // tile patterns: one block per tile (6272 blocks), 2 and 8 blocks resident per CU (LDS-limited)
  for (int lds : {74 * 1024, 16 * 1024}) {
    char nm[64];
    hipFuncSetAttribute((const void*)fill_tile<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)fill_tile<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)fill_tile<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    snprintf(nm, sizeof nm, "tile16_lds%dk", lds / 1024);
    TIME(nm, 0.0, (fill_tile<0><<<tiles, 256, lds>>>(out, tiles)));
    snprintf(nm, sizeof nm, "scatter16_lds%dk", lds / 1024);
    TIME(nm, 0.0, (fill_tile<1><<<tiles, 256, lds>>>(out, tiles)));
    snprintf(nm, sizeof nm, "tile4_lds%dk", lds / 1024);
    TIME(nm, 0.0, (fill_tile<2><<<tiles, 256, lds>>>(out, tiles)));
    hipFuncSetAttribute((const void*)fill_tile<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)fill_tile<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    snprintf(nm, sizeof nm, "tile16_plain_lds%dk", lds / 1024);
    TIME(nm, 0.0, (fill_tile<3><<<tiles, 256, lds>>>(out, tiles)));
    snprintf(nm, sizeof nm, "scatter16_plain_lds%dk", lds / 1024);
    TIME(nm, 0.0, (fill_tile<4><<<tiles, 256, lds>>>(out, tiles)));
    snprintf(nm, sizeof nm, "tile16_plain_persistent256_lds%dk", lds / 1024);
    TIME(nm, 0.0, (fill_tile<3><<<256, 256, lds>>>(out, tiles)));
    snprintf(nm, sizeof nm, "scatter16_plain_persistent256_lds%dk", lds / 1024);
    TIME(nm, 0.0, (fill_tile<4><<<256, 256, lds>>>(out, tiles)));
    snprintf(nm, sizeof nm, "scatter16_nt_persistent256_lds%dk", lds / 1024);
    TIME(nm, 0.0, (fill_tile<1><<<256, 256, lds>>>(out, tiles)));
  }
  TIME("stream_pattern_image_blocked_nt", 0.0, (fill_stream_pattern<true, true><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_interleaved_nt", 0.0, (fill_stream_pattern<false, true><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_image_blocked_rot1_nt", 0.0, (fill_stream_pattern<true, true, 16, 0, 1><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_image_blocked_rot7_nt", 0.0, (fill_stream_pattern<true, true, 16, 0, 7><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_image_blocked_rot37_nt", 0.0, (fill_stream_pattern<true, true, 16, 0, 37><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_image_blocked_nt_again", 0.0, (fill_stream_pattern<true, true><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_slab14_strided_nt", 0.0, (fill_stream_pattern<true, true, 16, 14><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_slab7_strided_nt", 0.0, (fill_stream_pattern<true, true, 16, 7><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_slab2_strided_nt", 0.0, (fill_stream_pattern<true, true, 16, 2><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_slab49_strided_nt", 0.0, (fill_stream_pattern<true, true, 16, 49><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_image_blocked_nt_512B_rows", 0.0, (fill_stream_pattern<true, true, 32><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_image_blocked_nt_1KiB_rows", 0.0, (fill_stream_pattern<true, true, 64><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_interleaved_nt_1KiB_rows", 0.0, (fill_stream_pattern<false, true, 64><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_image_blocked_plain_1KiB_rows", 0.0, (fill_stream_pattern<true, false, 64><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_image_blocked_plain", 0.0, (fill_stream_pattern<true, false><<<256, 256>>>(out, 98)));
  TIME("stream_pattern_interleaved_plain", 0.0, (fill_stream_pattern<false, false><<<256, 256>>>(out, 98)));
  // how fast can ONE CU store when the rest of the chip leaves HBM alone?  G persistent blocks (150 KiB of LDS:
  // one per CU), 24 tiles each, tile16 pattern
  for (int g : {8, 32, 64, 128, 256}) {
    const int t = g * 24;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) fill_tile<0><<<g, 256, 150 * 1024>>>(out, t);
    hipEventRecord(e1); hipDeviceSynchronize();
    const double sec = time_ms(e0, e1) / reps * 1e-3;
    printf("{\"probe\": \"tile16_on_%d_CUs\", \"ms\": %.4f, \"GBps_per_CU\": %.1f, \"store_TBps\": %.3f}\n", g, sec * 1e3,
           24.0 * kTileFloats * 4 / sec / 1e9, (double)t * kTileFloats * 4 / sec / 1e12);
  }
  // one wave per SIMD (1 block per CU, 150 KiB of LDS) vs two: can a lone wave keep the matrix pipe busy?
  {
    hipFuncSetAttribute((const void*)mfma_store<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const double m1 = (double)256 * 4 * 2000 * 8 * 32.0 * 32 * 64;
    TIME("mfma_only_1_wave_per_simd", m1, (mfma_store<0, true><<<256, 256, 150 * 1024>>>(out, 256, 2000)));
    TIME("mfma_only_2_waves_per_simd", 2 * m1, (mfma_store<0, true><<<512, 256, 74 * 1024>>>(out, 512, 2000)));
  }
  // matrix work + stores, 2 blocks per CU as in the real kernel (74 KiB of LDS each)
  const double macs = (double)tiles * 4 * 36 * 8 * 32.0 * 32 * 64;
  const int lds = 74 * 1024;
  hipFuncSetAttribute((const void*)mfma_store<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)mfma_store<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)mfma_store<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)mfma_store<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)mfma_store<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int round = 0; round < 2; ++round) {
    TIME("mfma_only_block_per_tile", macs, (mfma_store<0, true><<<tiles, 256, lds>>>(out, tiles, 36)));
    TIME("mfma_then_store_burst_block_per_tile", macs, (mfma_store<1, true><<<tiles, 256, lds>>>(out, tiles, 36)));
    TIME("mfma_store_interleaved_block_per_tile", macs, (mfma_store<2, true><<<tiles, 256, lds>>>(out, tiles, 36)));
    TIME("store_burst_only_block_per_tile", 0.0, (mfma_store<1, false><<<tiles, 256, lds>>>(out, tiles, 36)));
    TIME("mfma_only_persistent", macs, (mfma_store<0, true><<<512, 256, lds>>>(out, tiles, 36)));
    TIME("mfma_then_store_burst_persistent", macs, (mfma_store<1, true><<<512, 256, lds>>>(out, tiles, 36)));
    TIME("mfma_store_interleaved_persistent", macs, (mfma_store<2, true><<<512, 256, lds>>>(out, tiles, 36)));
  }
  return 0;
}
