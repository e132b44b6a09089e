// Is v_cvt_rpi_i32_f32 EXACTLY floor(x + 0.5) for every float the int8 epilogue can hand it (|x| <= 129)?  (Round 5: the int8
// output transform rounds half AWAY from zero, output_transform.h:31-44; floor(x + 0.5) differs from that only at exact negative
// ties, which the planner can rule out per plan.)  Every float in [-129, 129] is converted on the GPU -- plain and with the SDWA
// destination byte the epilogues use -- and compared with floor((double)x + 0.5) on the device in double.
// Build: hipcc -O3 --offload-arch=gfx950 -o cvt_rpi cvt_rpi.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>

__device__ int rpi(float x) {
  int r;
  asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ uint32_t rpi_byte1(float x, uint32_t keep) {      // byte 1 of `keep` replaced by the low byte of the conversion
  asm volatile("v_cvt_rpi_i32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\ts_nop 1" : "+v"(keep) : "v"(x));
  return keep;
}

__global__ void check(unsigned long long* bad, uint32_t* first_bad) {
  const uint64_t n = 1ull << 32;
  for (uint64_t bits = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; bits < n; bits += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)bits);
    if (!(fabsf(x) <= 129.0f)) continue;
    const int want = (int)floor((double)x + 0.5);
    const int got = rpi(x);
    const uint32_t packed = rpi_byte1(x, 0xA5A5A5A5u);
    if (got != want || packed != ((0xA5A500A5u) | (((uint32_t)want & 0xffu) << 8))) {
      if (atomicAdd(bad, 1ull) == 0) *first_bad = (uint32_t)bits;
    }
  }
}

int main() {
  unsigned long long* bad; uint32_t* fb;
  hipMalloc(&bad, 8); hipMalloc(&fb, 4); hipMemset(bad, 0, 8); hipMemset(fb, 0, 4);
  check<<<256 * 8, 256>>>(bad, fb);
  unsigned long long h = 0; uint32_t f = 0;
  hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fb, 4, hipMemcpyDeviceToHost);
  printf("{\"probe\": \"v_cvt_rpi_i32_f32 == floor(x + 0.5) for every float |x| <= 129 (plain and SDWA byte form)\", \"mismatches\": %llu, \"first_bad_bits\": \"0x%08x\"}\n", h, f);
  return h != 0;
}
