// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_stream.h"
#include "lce_mfma_selftest.h"

namespace lce {
stream_fn lookup_stream(int dst, int kch, bool fast, bool clamp, bool sign, bool strips, bool i8_floor) { return find_stream(dst, kch, fast, clamp, sign, strips, i8_floor); }
int mfma_selftest_stream() { return run_mfma_unscaled_selftest<1>(); }
}  // namespace lce
