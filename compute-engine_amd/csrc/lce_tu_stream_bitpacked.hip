// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_stream.h"
#include "lce_mfma_selftest.h"

namespace lce {
stream_fn lookup_stream_bitpacked(int kch, bool fast, bool strips) { return find_stream_part<kDstBitpacked, false, false>(kch, fast, false, strips); }
int mfma_selftest_stream() { return run_mfma_unscaled_selftest<1>(); }
}  // namespace lce
