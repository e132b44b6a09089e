// Known-answer test of the unscaled FP4 MFMA as THIS translation unit's compiler emitted it.
//
// mfma_fp4_32x32x64_unscaled (lce_device_intrinsics.h) asks for the scaled builtin with constant-zero scale operands and
// relies on the compiler selecting the unscaled encoding v_mfma_f32_32x32x64_f8f6f4 (FP4 inputs at scale 1).  A compiler
// without that selection would emit the scaled pair with E8M0 scale 0 = 2^-127: every product vanishes and the streaming
// and pointwise kernels would return K_bt-only results -- silently.  The library therefore runs one MFMA with operands
// +1 x -1 and C = 3 on every device before the first launch of a kernel family that uses the unscaled form; anything but
// 3 - 64 = -61 in all 1024 results makes the run fail (LCE_HIP_ERR_RUNTIME), never a wrong answer.
#pragma once
#include <hip/hip_runtime.h>
#include <lce_device_intrinsics.h>
#include "lce_kernels.h"

namespace lce {

template <int FAMILY>     // (a template: one instance per translation unit that uses it, no duplicate symbols)
LCE_KERNEL void __launch_bounds__(64) mfma_unscaled_selftest(float* __restrict__ out) {
  const u32x4 a = {0x22222222u, 0x22222222u, 0x22222222u, 0x22222222u};     // 32 x (+1)
  const u32x4 b = {0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu};     // 32 x (-1)
  f32x16 c = f32x16_fill(3.0f);
  c = mfma_fp4_32x32x64_unscaled(a, b, c);
#pragma unroll
  for (int i = 0; i < 16; ++i) out[thread_idx_x() * 16 + i] = c[i];
}

// 0: the MFMA computes what the kernels assume; 1: it does not; < 0: -(hipError_t) of the runtime call that failed
template <int FAMILY>
int run_mfma_unscaled_selftest() {
  float* dev = nullptr;
  hipError_t e = hipMalloc((void**)&dev, 1024 * sizeof(float));
  if (e != hipSuccess) return -(int)e;
  float host[1024];
  mfma_unscaled_selftest<FAMILY><<<dim3(1), dim3(64), 0, nullptr>>>(dev);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(host, dev, sizeof host, hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  if (e != hipSuccess) return -(int)e;
  for (float v : host)
    if (v != -61.0f) return 1;
  return 0;
}

}  // namespace lce
