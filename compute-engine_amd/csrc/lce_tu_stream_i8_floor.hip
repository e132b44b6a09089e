// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_stream.h"

namespace lce {
stream_fn lookup_stream_i8_floor(int kch, bool fast, bool sign, bool strips) { return find_stream_part<kDstInt8, false, true>(kch, fast, sign, strips); }
}  // namespace lce
