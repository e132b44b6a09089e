// Kernel entry-point types and the per-translation-unit lookups of the product build.
//
// The product library compiles each kernel family in its own translation unit (csrc/Makefile: parallel builds, and
// -amdgpu-mfma-vgpr-form only where it is wanted -- the streaming kernel); lce_hip_api.hip sees the kernels only through
// the lookup_* functions below, which hand out host-side launch stubs.  lce_dispatch.h holds the (inline) instance
// tables those lookups are made of; the host simulation of the CPU tests includes it directly.
#pragma once
#include <stdint.h>
#include "../../include/lce_hip.h"
#include "lce_kernel_args.h"

namespace lce {

typedef void (*tiled_fn)(const ConvArgs, const uint32_t*, const uint32_t*, const float*,
                         const float*, const int32_t*, const int32_t*, const float*, void*);
typedef void (*general_fn)(const ConvArgs, const uint32_t*, const uint32_t*, const float*,
                           const float*, const int32_t*, const float*, void*);
typedef void (*mfma_fn)(const ConvArgs, const MfmaArgs, const uint8_t*, const uint8_t*, const float*,
                        const float*, const float*, const float*, void*, uint32_t*);
typedef void (*pointwise_fn)(const PwArgs, const uint32_t*, const uint8_t*, const float*, const float*, const float*, void*, uint32_t*);
typedef void (*stream_fn)(const StreamArgs, const uint8_t*, const uint8_t*, const float*, const float*, const float*,
                          const uint32_t*, void*, uint32_t*);

typedef void (*wstream_fn)(const WsArgs, const uint8_t*, const uint8_t*, const float*, const float*, const float*,
                           const uint32_t*, void*, uint32_t*);

// lce_tu_valu.hip
tiled_fn lookup_tiled(int dst, int tm, int tn, int ch);
general_fn lookup_general(int dst);
// lce_tu_mfma_ws.hip / lce_tu_mfma_direct.hip / lce_tu_mfma_2d.hip
mfma_fn lookup_mfma_workspace(int dst, int bm, int bn, bool zero_pad_correction);
mfma_fn lookup_mfma_direct(int dst, int bm, int bn, bool zero_pad_correction);
mfma_fn lookup_mfma_direct_2d(int dst, int bm, int bn, bool zero_pad_correction);
inline mfma_fn lookup_mfma(int dst, int bm, int bn, bool zero_pad_correction, bool direct, bool tile2d) {
  if (direct && tile2d) return lookup_mfma_direct_2d(dst, bm, bn, zero_pad_correction);
  return direct ? lookup_mfma_direct(dst, bm, bn, zero_pad_correction) : lookup_mfma_workspace(dst, bm, bn, zero_pad_correction);
}
// the workspace variant's expansion pass (bits -> FP4 word planes); returns the launch's hipError_t as an int
int launch_expand_fp4(unsigned grid_x, void* stream, const uint32_t* in, void* workspace, const MfmaArgs& G, uint64_t chunks);
// lce_tu_pointwise.hip
pointwise_fn lookup_pointwise(int dst, int nc, int nj, bool strided, bool i8_floor);
// lce_tu_stream_{f32,f32_clamp,i8,i8_floor,bitpacked}.hip: the weight-stationary kernel's instance table, one part per translation unit
// (lce_dispatch_stream.h, find_stream_part)
stream_fn lookup_stream_f32(int kch, bool fast, bool sign, bool strips);         // float output, the clamp is the identity
stream_fn lookup_stream_f32_clamp(int kch, bool fast, bool sign, bool strips);   // float output with a fused activation's clamp
stream_fn lookup_stream_i8(int kch, bool fast, bool sign, bool strips);          // int8 output, the reference's rounding sequence
stream_fn lookup_stream_i8_floor(int kch, bool fast, bool sign, bool strips);    // int8 output, the proven one-instruction forms
stream_fn lookup_stream_bitpacked(int kch, bool fast, bool strips);
inline stream_fn lookup_stream(int dst, int kch, bool fast, bool clamp, bool sign, bool strips, bool i8_floor) {
  switch (dst) {
    case LCE_HIP_F32: return clamp ? lookup_stream_f32_clamp(kch, fast, sign, strips) : lookup_stream_f32(kch, fast, sign, strips);
    case LCE_HIP_I8: return i8_floor ? lookup_stream_i8_floor(kch, fast, sign, strips) : lookup_stream_i8(kch, fast, sign, strips);
    default: return lookup_stream_bitpacked(kch, fast, strips);
  }
}
// lce_tu_wstream_{f32,i8,i8_floor,bitpacked}.hip: the weight-streaming kernel's, likewise (lce_dispatch_wstream.h, find_wstream_part)
wstream_fn lookup_wstream_f32(int kch, int nb, bool sign);
wstream_fn lookup_wstream_i8(int kch, int nb, bool sign);
wstream_fn lookup_wstream_i8_floor(int kch, int nb, bool sign);
wstream_fn lookup_wstream_bitpacked(int kch, int nb);
inline wstream_fn lookup_wstream(int dst, int kch, int nb, bool sign, bool i8_floor) {
  switch (dst) {
    case LCE_HIP_F32: return lookup_wstream_f32(kch, nb, sign);
    case LCE_HIP_I8: return i8_floor ? lookup_wstream_i8_floor(kch, nb, sign) : lookup_wstream_i8(kch, nb, sign);
    default: return lookup_wstream_bitpacked(kch, nb);
  }
}
int mfma_selftest_wstream();
// known-answer test of the unscaled FP4 MFMA as each of those two translation units compiled it (lce_mfma_selftest.h):
// 0 = as assumed, 1 = wrong products, < 0 = -(hipError_t)
int mfma_selftest_pointwise();
int mfma_selftest_stream();

// the streaming kernel's FAST variant's precondition (lce_kernels_stream.h)
inline bool stream_fast(const StreamArgs& G) {     // (64 / 128 / 256 / 512 channels: 192 runs the 256-channel instance, two of whose words do not exist)
  return (G.Cin == 64 || G.Cin == 128 || G.Cin == 256 || G.Cin == 512) && !G.zero_border;
}
// its clamp is the identity on [0, 2 * K_bt] (activation NONE)
inline bool stream_clamps(const StreamArgs& G) { return !(G.cmin <= 0.0f && G.cmax >= 2.0f * G.a_bt); }

}  // namespace lce
