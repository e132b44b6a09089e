// Instance table of the weight-streaming kernel (lce_kernels_wstream.h) (one translation unit of the product build instantiates it: see
// lce_kernel_types.h; the host simulation of the CPU tests includes all tables through lce_dispatch.h).
#pragma once
#include "../../include/lce_hip.h"
#include "lce_kernel_types.h"
#include "lce_kernels_wstream.h"

namespace lce {

// 3x3 filters over 128 / 256 / 512 (padded) input channels; NB = the most pixel blocks a block owns; SIGN = the epilogue also
// writes the output's LceQuantize
template <int DST, int KCH, bool SIGN, bool I8F>
wstream_fn wstream_by_nb(int nb) {
  switch (nb) {
    case 1: return bconv2d_wstream<DST, KCH, 1, SIGN, I8F>;
    case 2: return bconv2d_wstream<DST, KCH, 2, SIGN, I8F>;
    case 3: return bconv2d_wstream<DST, KCH, 3, SIGN, I8F>;
    case 4: return bconv2d_wstream<DST, KCH, 4, SIGN, I8F>;
    default: return nullptr;
  }
}
template <int DST, bool SIGN, bool I8F>
wstream_fn wstream_by_kch(int kch, int nb) {
  switch (kch) {
    case 8: return wstream_by_nb<DST, 8, SIGN, I8F>(nb);
    case 4: return wstream_by_nb<DST, 4, SIGN, I8F>(nb);
    case 2: return wstream_by_nb<DST, 2, SIGN, I8F>(nb);
    default: return nullptr;
  }
}
// One PART of the table = the instances of one (output type, int8 rounding form); a translation unit of the product build each
// (lce_tu_wstream_*.hip).
template <int DST, bool I8F>
wstream_fn find_wstream_part(int kch, int nb, bool sign) {
  if constexpr (DST == kDstBitpacked) return wstream_by_kch<kDstBitpacked, false, false>(kch, nb);
  else return sign ? wstream_by_kch<DST, true, I8F>(kch, nb) : wstream_by_kch<DST, false, I8F>(kch, nb);
}
}  // namespace lce
