// Host simulation -- TEST ONLY: one part of the kernel instance tables, compiled for the CPU against tests/hostsim/lce_device_intrinsics.h
// (the product build has the same cut: compute-engine_amd/csrc/lce_tu_*.hip, lce_kernel_types.h).
#include "lce_dispatch_mfma.h"

namespace lce {
mfma_fn lookup_mfma_workspace(int dst, int bm, int bn, bool zero_pad_correction) { return find_mfma_v<false, false>(dst, bm, bn, zero_pad_correction); }
}  // namespace lce
