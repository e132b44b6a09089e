// What does a launch cost before it does anything?  Kernel durations (rocprofv3 --kernel-trace) of
//   empty            : s_endpgm
//   args             : reads its kernel arguments (one s_load) and exits
//   touch            : every wave reads 16 bytes per lane of a small L2-resident table and stores nothing
//   store            : every lane stores 16 bytes (grid x 4 KiB of output), plain or streaming (nt)
// at the grids of the small 1x1 layers (784 / 1568 / 3136 blocks of 256 threads, 0 or 32 KiB of LDS).
//   hipcc -O3 --offload-arch=gfx950 -o launch_floor launch_floor.hip && rocprofv3 --kernel-trace --stats -- ./launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_empty() {}
__global__ void k_args(const uint32_t* p, int n) {
  if (n == -12345) ((volatile uint32_t*)p)[0] = 1;
}
__global__ void k_touch(const uint4* __restrict__ tab, uint32_t* sink, int n) {
  const uint4 v = tab[(blockIdx.x * 256 + threadIdx.x) & (n - 1)];
  if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[0] = v.z;
}
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void k_store(u4* out) {
  const u4 v = {blockIdx.x, threadIdx.x, 1u, 2u};
  u4* o = out + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (NT) __builtin_nontemporal_store(v, o);
  else *o = v;
}
extern __shared__ char lds[];
__global__ void k_lds(uint32_t* sink, int n) {
  if (n == -12345) sink[0] = lds[threadIdx.x];
}

int main() {
  uint4* tab; uint32_t* sink; u4* out;
  hipMalloc(&tab, 1 << 20); hipMemset(tab, 0, 1 << 20);
  hipMalloc(&sink, 4096);
  hipMalloc(&out, (size_t)8192 * 4096);
  for (int rep = 0; rep < 300; ++rep) {
    for (int grid : {784, 1568, 3136, 6272}) {
      hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0);
      hipLaunchKernelGGL(k_args, dim3(grid), dim3(256), 0, 0, sink, grid);
      hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 32768, 0, sink, grid);
      hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, 0, tab, sink, 65536);
      hipLaunchKernelGGL(k_store<false>, dim3(grid), dim3(256), 0, 0, out);
      hipLaunchKernelGGL(k_store<true>, dim3(grid), dim3(256), 0, 0, out);
    }
  }
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
