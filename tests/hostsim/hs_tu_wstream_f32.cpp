// Host simulation -- TEST ONLY: one part of the kernel instance tables, compiled for the CPU against tests/hostsim/lce_device_intrinsics.h
// (the product build has the same cut: compute-engine_amd/csrc/lce_tu_*.hip, lce_kernel_types.h).
#include "lce_dispatch_wstream.h"

namespace lce {
wstream_fn lookup_wstream_f32(int kch, int nb, bool sign) { return find_wstream_part<kDstFloat, false>(kch, nb, sign); }
}  // namespace lce
