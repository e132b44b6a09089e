// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_pointwise.h"

namespace lce {
pointwise_fn lookup_pointwise(int dst, int nc, int nj, bool strided) { return find_pointwise(dst, nc, nj, strided); }
}  // namespace lce
