// Kernel-instance lookup shared by the C ABI (lce_hip_api.hip) and the host simulation
// used by the CPU-only tests (tests/hostsim/).
#pragma once
#include "../../include/lce_hip.h"
#include "lce_kernels.h"
#include "lce_kernels_mfma.h"
#include "lce_kernels_pointwise.h"
#include "lce_kernels_stream.h"

namespace lce {

typedef void (*tiled_fn)(const ConvArgs, const uint32_t*, const uint32_t*, const float*,
                         const float*, const int32_t*, const int32_t*, const float*, void*);
typedef void (*general_fn)(const ConvArgs, const uint32_t*, const uint32_t*, const float*,
                           const float*, const int32_t*, const float*, void*);

template <int DST, int TM, int TN>
tiled_fn tiled_by_ch(int ch) {
  switch (ch) {
    case 4: return bconv2d_tiled<DST, TM, TN, 4>;
    case 2: return bconv2d_tiled<DST, TM, TN, 2>;
    default: return bconv2d_tiled<DST, TM, TN, 1>;
  }
}

template <int DST>
tiled_fn tiled_by_tile(int tm, int tn, int ch) {
  if (tm == 2 && tn == 32) return tiled_by_ch<DST, 2, 32>(ch);
  if (tm == 1 && tn == 32) return tiled_by_ch<DST, 1, 32>(ch);
  if constexpr (DST != kDstBitpacked) {
    if (tm == 4 && tn == 16) return tiled_by_ch<DST, 4, 16>(ch);
    if (tm == 2 && tn == 16) return tiled_by_ch<DST, 2, 16>(ch);
    if (tm == 1 && tn == 16) return tiled_by_ch<DST, 1, 16>(ch);
  }
  return nullptr;
}

inline tiled_fn find_tiled(int dst, int tm, int tn, int ch) {
  switch (dst) {
    case LCE_HIP_F32: return tiled_by_tile<kDstFloat>(tm, tn, ch);
    case LCE_HIP_I8: return tiled_by_tile<kDstInt8>(tm, tn, ch);
    default: return tiled_by_tile<kDstBitpacked>(tm, tn, ch);
  }
}

inline general_fn find_general(int dst) {
  switch (dst) {
    case LCE_HIP_F32: return bconv2d_general<kDstFloat>;
    case LCE_HIP_I8: return bconv2d_general<kDstInt8>;
    default: return bconv2d_general<kDstBitpacked>;
  }
}



typedef void (*mfma_fn)(const ConvArgs, const MfmaArgs, const uint8_t*, const uint8_t*, const float*,
                        const float*, const float*, const float*, void*, uint32_t*);

// CORR = the optimized kernels' SAME-zero float correction in the epilogue (float output only)
// DIRECT = LDS-resident input halo instead of the FP4 workspace; T2D = its 2-D tiles for wide images (lce_kernels_mfma.h)
template <int DST, bool CORR, bool DIRECT, bool T2D>
mfma_fn mfma_by_tile(int bm, int bn) {
  constexpr int ST = DIRECT ? 3 : 4;
  if (bm == 256 && bn == 256) return bconv2d_mfma<DST, 4, 2, 2, 4, CORR, DIRECT, ST, T2D>;
  if (bm == 256 && bn == 128) return bconv2d_mfma<DST, 4, 2, 2, 2, CORR, DIRECT, ST, T2D>;
  if (bm == 512 && bn == 64) return bconv2d_mfma<DST, 8, 1, 2, 2, CORR, DIRECT, ST, T2D>;
  if (bm == 128 && bn == 256) return bconv2d_mfma<DST, 2, 2, 2, 4, CORR, DIRECT, ST, T2D>;
  if (bm == 128 && bn == 128) return bconv2d_mfma<DST, 2, 2, 2, 2, CORR, DIRECT, ST, T2D>;
  if (bm == 256 && bn == 64) return bconv2d_mfma<DST, 4, 1, 2, 2, CORR, DIRECT, ST, T2D>;
  if (bm == 128 && bn == 64) return bconv2d_mfma<DST, 2, 1, 2, 2, CORR, DIRECT, ST, T2D>;
  return nullptr;
}

template <bool DIRECT, bool T2D>
mfma_fn find_mfma_v(int dst, int bm, int bn, bool zero_pad_correction) {
  switch (dst) {
    case LCE_HIP_F32:
      return zero_pad_correction ? mfma_by_tile<kDstFloat, true, DIRECT, T2D>(bm, bn)
                                 : mfma_by_tile<kDstFloat, false, DIRECT, T2D>(bm, bn);
    case LCE_HIP_I8: return mfma_by_tile<kDstInt8, false, DIRECT, T2D>(bm, bn);
    default: return mfma_by_tile<kDstBitpacked, false, DIRECT, T2D>(bm, bn);
  }
}

inline mfma_fn find_mfma(int dst, int bm, int bn, bool zero_pad_correction = false, bool direct = false, bool tile2d = false) {
  if (direct && tile2d) return find_mfma_v<true, true>(dst, bm, bn, zero_pad_correction);
  return direct ? find_mfma_v<true, false>(dst, bm, bn, zero_pad_correction)
                : find_mfma_v<false, false>(dst, bm, bn, zero_pad_correction);
}

typedef void (*pointwise_fn)(const PwArgs, const uint32_t*, const uint8_t*, const float*, const float*, const float*, void*, uint32_t*);

template <int DST, int NC, bool STRIDED>
pointwise_fn pointwise_by_nj(int nj) {
  switch (nj) {
    case 4:
      // (8 K-steps x 4 tiles, or a float tile's 16 row stores x 4 tiles, would not fit 256 VGPRs)
      if constexpr (NC <= 4 && DST != kDstFloat) return bconv2d_pointwise<DST, NC, 4, STRIDED>;
      else return nullptr;
    case 2: return bconv2d_pointwise<DST, NC, 2, STRIDED>;
    case 1: return bconv2d_pointwise<DST, NC, 1, STRIDED>;
    default: return nullptr;
  }
}
template <int DST, bool STRIDED>
pointwise_fn pointwise_by_nc(int nc, int nj) {
  switch (nc) {
    case 8: return pointwise_by_nj<DST, 8, STRIDED>(nj);
    case 4: return pointwise_by_nj<DST, 4, STRIDED>(nj);
    case 2: return pointwise_by_nj<DST, 2, STRIDED>(nj);
    case 1: return pointwise_by_nj<DST, 1, STRIDED>(nj);
    default: return nullptr;
  }
}
template <int DST>
pointwise_fn pointwise_by_stride(int nc, int nj, bool strided) {
  return strided ? pointwise_by_nc<DST, true>(nc, nj) : pointwise_by_nc<DST, false>(nc, nj);
}
inline pointwise_fn find_pointwise(int dst, int nc, int nj, bool strided) {
  switch (dst) {
    case LCE_HIP_F32: return pointwise_by_stride<kDstFloat>(nc, nj, strided);
    case LCE_HIP_I8: return pointwise_by_stride<kDstInt8>(nc, nj, strided);
    default: return pointwise_by_stride<kDstBitpacked>(nc, nj, strided);
  }
}


typedef void (*stream_fn)(const StreamArgs, const uint8_t*, const uint8_t*, const float*, const float*, const float*,
                          const uint32_t*, void*, uint32_t*);

// 3x3 filters over 64 / 128 / 256 (padded) input channels; FAST = every padded word exists and padding is +1;
// CLAMP = the float transform's clamp is not the identity; SIGN = the epilogue also writes the output's LceQuantize
template <int DST, bool FAST, bool CLAMP, bool SIGN>
stream_fn stream_by_kch(int kch) {
  switch (kch) {
    case 4: return bconv2d_stream<DST, 3, 3, 4, FAST, CLAMP, SIGN>;
    case 2: return bconv2d_stream<DST, 3, 3, 2, FAST, CLAMP, SIGN>;
    case 1: return bconv2d_stream<DST, 3, 3, 1, FAST, CLAMP, SIGN>;
    default: return nullptr;
  }
}
template <int DST, bool CLAMP, bool SIGN>
stream_fn stream_by_fast(int kch, bool fast) {
  return fast ? stream_by_kch<DST, true, CLAMP, SIGN>(kch) : stream_by_kch<DST, false, CLAMP, SIGN>(kch);
}
inline stream_fn find_stream(int dst, int kch, bool fast, bool clamp, bool sign) {
  switch (dst) {
    case LCE_HIP_F32:
      if (clamp) return sign ? stream_by_fast<kDstFloat, true, true>(kch, fast) : stream_by_fast<kDstFloat, true, false>(kch, fast);
      return sign ? stream_by_fast<kDstFloat, false, true>(kch, fast) : stream_by_fast<kDstFloat, false, false>(kch, fast);
    case LCE_HIP_I8:
      return sign ? stream_by_fast<kDstInt8, false, true>(kch, fast) : stream_by_fast<kDstInt8, false, false>(kch, fast);
    default: return stream_by_fast<kDstBitpacked, false, false>(kch, fast);
  }
}
// the FAST variant's precondition (lce_kernels_stream.h)
inline bool stream_fast(const StreamArgs& G) { return G.Cin % 64 == 0 && !G.zero_border; }
// the clamp is the identity on [0, 2 * K_bt] (activation NONE)
inline bool stream_clamps(const StreamArgs& G) { return !(G.cmin <= 0.0f && G.cmax >= 2.0f * G.a_bt); }

}  // namespace lce
