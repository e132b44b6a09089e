// Host simulation -- TEST ONLY: one part of the kernel instance tables, compiled for the CPU against tests/hostsim/lce_device_intrinsics.h
// (the product build has the same cut: compute-engine_amd/csrc/lce_tu_*.hip, lce_kernel_types.h).
#include "lce_dispatch_wstream.h"

namespace lce {
wstream_fn lookup_wstream_i8_floor(int kch, int nb, bool sign) { return find_wstream_part<kDstInt8, true>(kch, nb, sign); }
}  // namespace lce
