// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_wstream.h"
#include "lce_mfma_selftest.h"

namespace lce {
wstream_fn lookup_wstream(int dst, int kch, int nb, bool sign, bool i8_floor) { return find_wstream(dst, kch, nb, sign, i8_floor); }
int mfma_selftest_wstream() { return run_mfma_unscaled_selftest<2>(); }
}  // namespace lce
