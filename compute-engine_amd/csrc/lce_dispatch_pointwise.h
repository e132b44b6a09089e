// Instance table of the 1x1 streaming kernel (lce_kernels_pointwise.h) (one translation unit of the product build instantiates it: see lce_kernel_types.h;
// the host simulation of the CPU tests includes all four tables through lce_dispatch.h).
#pragma once
#include "../../include/lce_hip.h"
#include "lce_kernel_types.h"
#include "lce_kernels_pointwise.h"

namespace lce {

template <int DST, int NC, bool STRIDED, bool I8F>
pointwise_fn pointwise_by_nj(int nj) {
  switch (nj) {
    case 4:
      // (8 K-steps x 4 tiles, or a float tile's 16 row stores x 4 tiles, would not fit 256 VGPRs)
      if constexpr (NC <= 4 && DST != kDstFloat) return bconv2d_pointwise<DST, NC, 4, STRIDED, I8F>;
      else return nullptr;
    case 2: return bconv2d_pointwise<DST, NC, 2, STRIDED, I8F>;
    case 1: return bconv2d_pointwise<DST, NC, 1, STRIDED, I8F>;
    default: return nullptr;
  }
}
template <int DST, bool STRIDED, bool I8F>
pointwise_fn pointwise_by_nc(int nc, int nj) {
  switch (nc) {
    case 8: return pointwise_by_nj<DST, 8, STRIDED, I8F>(nj);
    case 4: return pointwise_by_nj<DST, 4, STRIDED, I8F>(nj);
    case 2: return pointwise_by_nj<DST, 2, STRIDED, I8F>(nj);
    case 1: return pointwise_by_nj<DST, 1, STRIDED, I8F>(nj);
    default: return nullptr;
  }
}
template <int DST, bool I8F>
pointwise_fn pointwise_by_stride(int nc, int nj, bool strided) {
  return strided ? pointwise_by_nc<DST, true, I8F>(nc, nj) : pointwise_by_nc<DST, false, I8F>(nc, nj);
}
// i8_floor: the int8 instances whose rounding is floor(x + 0.5) (the planner's int8_floor_ok)
inline pointwise_fn find_pointwise(int dst, int nc, int nj, bool strided, bool i8_floor = false) {
  switch (dst) {
    case LCE_HIP_F32: return pointwise_by_stride<kDstFloat, false>(nc, nj, strided);
    case LCE_HIP_I8: return i8_floor ? pointwise_by_stride<kDstInt8, true>(nc, nj, strided) : pointwise_by_stride<kDstInt8, false>(nc, nj, strided);
    default: return pointwise_by_stride<kDstBitpacked, false>(nc, nj, strided);
  }
}

}  // namespace lce
