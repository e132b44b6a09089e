// Host simulation -- TEST ONLY: one part of the kernel instance tables, compiled for the CPU against tests/hostsim/lce_device_intrinsics.h
// (the product build has the same cut: compute-engine_amd/csrc/lce_tu_*.hip, lce_kernel_types.h).
#include "lce_dispatch_stream.h"

namespace lce {
stream_fn lookup_stream_f32(int kch, bool fast, bool sign, bool strips) { return find_stream_part<kDstFloat, false, false>(kch, fast, sign, strips); }
}  // namespace lce
