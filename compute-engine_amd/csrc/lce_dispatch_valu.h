// Instance table of the xor-popcount kernels (lce_kernels.h) (one translation unit of the product build instantiates it: see lce_kernel_types.h;
// the host simulation of the CPU tests includes all four tables through lce_dispatch.h).
#pragma once
#include "../../include/lce_hip.h"
#include "lce_kernel_types.h"
#include "lce_kernels.h"

namespace lce {

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

}  // namespace lce
