// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_pointwise.h"
#include "lce_mfma_selftest.h"

namespace lce {
pointwise_fn lookup_pointwise(int dst, int nc, int nj, bool strided, bool i8_floor) { return find_pointwise(dst, nc, nj, strided, i8_floor); }
int mfma_selftest_pointwise() { return run_mfma_unscaled_selftest<2>(); }
}  // namespace lce
