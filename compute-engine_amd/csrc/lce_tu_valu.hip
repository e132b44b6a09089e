// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_valu.h"

namespace lce {
tiled_fn lookup_tiled(int dst, int tm, int tn, int ch) { return find_tiled(dst, tm, tn, ch); }
general_fn lookup_general(int dst) { return find_general(dst); }
}  // namespace lce
