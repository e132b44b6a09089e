// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_wstream.h"

namespace lce {
wstream_fn lookup_wstream_i8(int kch, int nb, bool sign) { return find_wstream_part<kDstInt8, false>(kch, nb, sign); }
}  // namespace lce
