// Host simulation -- TEST ONLY: one part of the kernel instance tables, compiled for the CPU against tests/hostsim/lce_device_intrinsics.h
// (the product build has the same cut: compute-engine_amd/csrc/lce_tu_*.hip, lce_kernel_types.h).
#include "lce_dispatch_pointwise.h"

namespace lce {
pointwise_fn lookup_pointwise(int dst, int nc, int nj, bool strided, bool i8_floor) { return find_pointwise(dst, nc, nj, strided, i8_floor); }
}  // namespace lce
