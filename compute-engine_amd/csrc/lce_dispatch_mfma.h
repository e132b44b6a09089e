// Instance table of the block GEMM (lce_kernels_mfma.h) (one translation unit of the product build instantiates it: see lce_kernel_types.h;
// the host simulation of the CPU tests includes all four tables through lce_dispatch.h).
#pragma once
#include "../../include/lce_hip.h"
#include "lce_kernel_types.h"
#include "lce_kernels_mfma.h"

namespace lce {

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

// (the three variants are stitched together by lookup_mfma, lce_kernel_types.h: an inline function HERE that named all three would
//  instantiate -- and emit -- every variant's kernels in each of the three translation units; until round 6 one did, and the library
//  carried the block GEMM three times)

}  // namespace lce
