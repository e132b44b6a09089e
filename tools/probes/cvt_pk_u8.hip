// What does v_cvt_pk_u8_f32 compute?  (Round 5: an int8 epilogue that converts, saturates and packs in ONE instruction per value.)
// Every float with |x| <= 300 (and NaN / inf / huge values) goes through the instruction with each of the four byte selects; the byte
// that comes back is compared with three candidate definitions -- saturate_u8(trunc(x)), saturate_u8(rint(x)) [half to even],
// saturate_u8(floor(x + 0.5)) -- and the untouched bytes with the operand they came from.
// Build: hipcc -O3 --offload-arch=gfx950 -o cvt_pk_u8 cvt_pk_u8.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>

template <int SEL>
__device__ uint32_t pk_u8(float x, uint32_t keep) {
  uint32_t r;
  asm volatile("v_cvt_pk_u8_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(SEL), "v"(keep));
  return r;
}
__device__ uint32_t sat(double v) { return v != v ? 0u : v < 0.0 ? 0u : v > 255.0 ? 255u : (uint32_t)v; }

__global__ void check(unsigned long long* bad, uint32_t* first_bad) {
  const uint64_t n = 1ull << 32;
  for (uint64_t bits = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; bits < n; bits += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)bits);
    const bool special = x != x || fabsf(x) > 1e30f;
    if (!(fabsf(x) <= 300.0f) && !special && (bits & 0xfffff)) continue;     // (a sample of the large finite values)
    const uint32_t keep = 0xA5C3E187u;
    const uint32_t g[4] = {pk_u8<0>(x, keep), pk_u8<1>(x, keep), pk_u8<2>(x, keep), pk_u8<3>(x, keep)};
    const uint32_t byte = g[0] & 0xffu;
    bool shape_ok = true;
    for (int s = 0; s < 4; ++s) shape_ok &= g[s] == ((keep & ~(0xffu << (8 * s))) | (byte << (8 * s)));
    const double d = (double)x;
    const uint32_t c_trunc = sat(trunc(d)), c_rne = sat(rint(d)), c_rpi = sat(floor(d + 0.5));
    if (!shape_ok) { if (atomicAdd(bad + 0, 1ull) == 0) first_bad[0] = (uint32_t)bits; }
    if (byte != c_trunc) { if (atomicAdd(bad + 1, 1ull) == 0) first_bad[1] = (uint32_t)bits; }
    if (byte != c_rne) { if (atomicAdd(bad + 2, 1ull) == 0) first_bad[2] = (uint32_t)bits; }
    if (byte != c_rpi) { if (atomicAdd(bad + 3, 1ull) == 0) first_bad[3] = (uint32_t)bits; }
    atomicAdd(bad + 4, 1ull);
  }
}

int main() {
  unsigned long long* bad; uint32_t* fb;
  hipMalloc(&bad, 40); hipMalloc(&fb, 16); hipMemset(bad, 0, 40); hipMemset(fb, 0, 16);
  check<<<256 * 8, 256>>>(bad, fb);
  unsigned long long h[5]; uint32_t f[4];
  hipMemcpy(h, bad, 40, hipMemcpyDeviceToHost); hipMemcpy(f, fb, 16, hipMemcpyDeviceToHost);
  printf("{\"probe\": \"v_cvt_pk_u8_f32 over every float |x| <= 300, NaN, inf, a sample of the rest\", \"values\": %llu, \"byte_placement_mismatches\": %llu, "
         "\"vs_sat_trunc\": %llu, \"vs_sat_rint_half_even\": %llu, \"vs_sat_floor_x_plus_half\": %llu, \"first_bad_bits\": [\"0x%08x\", \"0x%08x\", \"0x%08x\", \"0x%08x\"]}\n",
         h[4], h[0], h[1], h[2], h[3], f[0], f[1], f[2], f[3]);
  return 0;
}
