// Host simulation -- TEST ONLY: one part of the kernel instance tables, compiled for the CPU against tests/hostsim/lce_device_intrinsics.h
// (the product build has the same cut: compute-engine_amd/csrc/lce_tu_*.hip, lce_kernel_types.h).
#include "lce_dispatch_valu.h"

namespace lce {
tiled_fn lookup_tiled(int dst, int tm, int tn, int ch) { return find_tiled(dst, tm, tn, ch); }
general_fn lookup_general(int dst) { return find_general(dst); }
}  // namespace lce
