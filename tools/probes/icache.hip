// What does straight-line code cost the FIRST time a CU runs it?  A block of 256 threads (one wave per SIMD, as the
// streaming kernel) executes N independent v_add_u32 (4 bytes... v_add_u32 with a literal: 8 bytes each) twice in a loop;
// s_memtime around each pass.  Pass 0 fetches the code (the instruction cache is invalidated at kernel start), pass 1 re-runs
// it from the cache.  Printed: cycles per instruction, cold and warm, for code bodies of 2 / 8 / 24 KiB, with 256 and with
// 1024 blocks (every CU busy / four blocks queued per CU).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/icache tools/probes/icache.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ADD8 "v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x23451, %1\n v_add_u32 %2, 0x34512, %2\n v_add_u32 %3, 0x45123, %3\n" \
             "v_add_u32 %0, 0x51234, %0\n v_add_u32 %1, 0x12346, %1\n v_add_u32 %2, 0x23457, %2\n v_add_u32 %3, 0x34568, %3\n"
#define ADD64 ADD8 ADD8 ADD8 ADD8 ADD8 ADD8 ADD8 ADD8
#define ADD256 ADD64 ADD64 ADD64 ADD64          /* 256 instructions x 8 bytes = 2 KiB */

template <int KB2>   // code body in units of 2 KiB
__global__ void __launch_bounds__(256) body(unsigned long long* out, unsigned* sink) {
  unsigned a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  unsigned long long t[3];
  for (int pass = 0; pass < 2; ++pass) {
    t[pass] = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < KB2; ++k) asm volatile(ADD256 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  }
  t[2] = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2 + 0] = t[1] - t[0];
    out[blockIdx.x * 2 + 1] = t[2] - t[1];
  }
  if (a + b + c + d == 0x7fffffff) *sink = a;
}

template <int KB2>
void run(int blocks) {
  unsigned long long* d;
  unsigned* s;
  hipMalloc(&d, blocks * 16);
  hipMalloc(&s, 4);
  std::vector<unsigned long long> h(blocks * 2);
  for (int rep = 0; rep < 3; ++rep) {
    body<KB2><<<blocks, 256>>>(d, s);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
  double cold = 0, warm = 0;
  for (int b = 0; b < blocks; ++b) { cold += h[b * 2]; warm += h[b * 2 + 1]; }
  const double n = 256.0 * KB2;
  printf("code %2d KiB, %4d blocks: cold %6.0f cycles = %5.2f per instruction | warm %6.0f = %5.2f per instruction\n", 2 * KB2, blocks,
         cold / blocks, cold / blocks / n, warm / blocks, warm / blocks / n);
  hipFree(d);
  hipFree(s);
}

int main() {
  for (int blocks : {256, 1024}) {
    run<1>(blocks);
    run<4>(blocks);
    run<12>(blocks);
  }
  return 0;
}
