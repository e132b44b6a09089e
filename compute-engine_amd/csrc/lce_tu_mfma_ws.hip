// One translation unit of the product library (csrc/Makefile): see lce_kernel_types.h.
#include <hip/hip_runtime.h>
#include "lce_dispatch_mfma.h"

namespace lce {
mfma_fn lookup_mfma_workspace(int dst, int bm, int bn, bool zero_pad_correction) {
  return find_mfma_v<false, false>(dst, bm, bn, zero_pad_correction);
}
int launch_expand_fp4(unsigned grid_x, void* stream, const uint32_t* in, void* workspace, const MfmaArgs& G, uint64_t chunks) {
  expand_fp4<<<dim3(grid_x), 256, 0, (hipStream_t)stream>>>(in, (lce_dev::u32x4*)workspace, G, chunks);
  return (int)hipGetLastError();
}
}  // namespace lce
