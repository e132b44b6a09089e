// Per-block time stamps of the streaming kernels (lce_kernels_stream.h, lce_kernels_wstream.h) for tools/stream_phases.py.
#pragma once
#include <lce_device_intrinsics.h>

namespace lce {

#ifdef LCE_STREAM_PHASES
// Profiling aid (never defined in the product build): wave 0 of every block stamps s_memtime at entry, first rows
// resident, and after every tile step (up to 60), the last slot at exit.
__device__ unsigned long long lce_stream_tl[512 * 64];
#define LCE_SPH(slot)                                                                               \
  do {                                                                                              \
    const int sph_b = block_idx_y() * grid_dim_x() + block_idx_x();                                 \
    if (thread_idx_x() == 0 && sph_b < 512 && (slot) < 64)                                          \
      lce_stream_tl[sph_b * 64 + (slot)] = __builtin_readcyclecounter();                            \
  } while (0)
#else
#define LCE_SPH(slot) do {} while (0)
#endif

}  // namespace lce
