// Instance table of the weight-stationary streaming kernel (lce_kernels_stream.h) (one translation unit of the product build instantiates it: see lce_kernel_types.h;
// the host simulation of the CPU tests includes all four tables through lce_dispatch.h).
#pragma once
#include "../../include/lce_hip.h"
#include "lce_kernel_types.h"
#include "lce_kernels_stream.h"

namespace lce {

// 3x3 filters over 64 / 128 / 256 / 512 (padded) input channels; FAST = every padded word exists and padding is +1;
// CLAMP = the float transform's clamp is not the identity; SIGN = the epilogue also writes the output's LceQuantize
template <int DST, bool FAST, bool CLAMP, bool SIGN, bool I8F>
stream_fn stream_by_kch(int kch, bool strips) {
  if (strips)     // column strips of wide images: built for the 256-channel bank (the north star's 224 x 224 x 256 maps)
    return kch == 4 ? bconv2d_stream<DST, 3, 3, 4, FAST, CLAMP, SIGN, false, true, I8F> : nullptr;
  switch (kch) {
    case 8: return bconv2d_stream<DST, 3, 3, 8, FAST, CLAMP, SIGN, true, false, I8F>;     // 512 input channels: K split over wave pairs
    case 4: return bconv2d_stream<DST, 3, 3, 4, FAST, CLAMP, SIGN, false, false, I8F>;
    case 2: return bconv2d_stream<DST, 3, 3, 2, FAST, CLAMP, SIGN, false, false, I8F>;
    case 1: return bconv2d_stream<DST, 3, 3, 1, FAST, CLAMP, SIGN, false, false, I8F>;
    default: return nullptr;
  }
}
template <int DST, bool CLAMP, bool SIGN, bool I8F = false>
stream_fn stream_by_fast(int kch, bool fast, bool strips) {
  return fast ? stream_by_kch<DST, true, CLAMP, SIGN, I8F>(kch, strips) : stream_by_kch<DST, false, CLAMP, SIGN, I8F>(kch, strips);
}
// One PART of the table = the instances of one (output type, float clamp / int8 rounding form): the product build compiles each part in
// its own translation unit (lce_tu_stream_*.hip; the 90 instances in one unit took 8 minutes to compile, the parts build side by side).
template <int DST, bool CLAMP, bool I8F>
stream_fn find_stream_part(int kch, bool fast, bool sign, bool strips) {
  if constexpr (DST == kDstBitpacked) return stream_by_fast<kDstBitpacked, false, false>(kch, fast, strips);
  else return sign ? stream_by_fast<DST, CLAMP, true, I8F>(kch, fast, strips) : stream_by_fast<DST, CLAMP, false, I8F>(kch, fast, strips);
}
// (No function here names ALL parts: a non-template inline function that did -- the former find_stream -- instantiates every kernel in
//  every translation unit that includes this header, used or not: __global__ instantiations are always emitted.  The lookups that
//  stitch the parts together are lookup_stream / lookup_wstream / lookup_mfma in lce_kernel_types.h.)

}  // namespace lce
