// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_mfma.h"

namespace lce {
mfma_fn lookup_mfma_direct(int dst, int bm, int bn, bool zero_pad_correction) {
  return find_mfma_v<true, false>(dst, bm, bn, zero_pad_correction);
}
}  // namespace lce
