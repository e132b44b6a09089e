// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_stream.h"

namespace lce {
stream_fn lookup_stream_f32_clamp(int kch, bool fast, bool sign, bool strips) { return find_stream_part<kDstFloat, true, false>(kch, fast, sign, strips); }
}  // namespace lce
