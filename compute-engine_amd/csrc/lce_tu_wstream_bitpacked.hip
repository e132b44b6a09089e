// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_wstream.h"
#include "lce_mfma_selftest.h"

namespace lce {
wstream_fn lookup_wstream_bitpacked(int kch, int nb) { return find_wstream_part<kDstBitpacked, false>(kch, nb, false); }
int mfma_selftest_wstream() { return run_mfma_unscaled_selftest<2>(); }
}  // namespace lce
