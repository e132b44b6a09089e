// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_wstream.h"

namespace lce {
wstream_fn lookup_wstream_i8_floor(int kch, int nb, bool sign) { return find_wstream_part<kDstInt8, true>(kch, nb, sign); }
}  // namespace lce
