// Probe: operand layout and exactness of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1)
// A and B operands on gfx950.  Writes D = A(32x64) * B(64x32) for probe matrices and
// compares with a host reference under candidate layouts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(const uint32_t* a, const uint32_t* b, float* d, int scale_a, int scale_b) {
  const int lane = threadIdx.x;
  v8i va = {0,0,0,0,0,0,0,0}, vb = {0,0,0,0,0,0,0,0};
  for (int i = 0; i < 4; ++i) { va[i] = a[lane * 4 + i]; vb[i] = b[lane * 4 + i]; }
  v16f c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  // cbsz = 4 (A is fp4), blgp = 4 (B is fp4)
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, scale_a, 0, scale_b);
  for (int i = 0; i < 16; ++i) d[lane * 16 + i] = c[i];
}

static uint8_t fp4_of(int v) {  // v in {-1, 0, +1, 2, ...}: E2M1 codes
  switch (v) { case 0: return 0x0; case 1: return 0x2; case -1: return 0xA; case 2: return 0x4; case -2: return 0xC; default: return 0; }
}

int main() {
  // logical A[m][k] (32x64), B[k][n] (64x32) with values in {-1,+1} (pseudo-random, asymmetric)
  static int A[32][64], B[64][32];
  srand(1);
  for (int m = 0; m < 32; ++m) for (int k = 0; k < 64; ++k) A[m][k] = (rand() & 1) ? 1 : -1;
  for (int k = 0; k < 64; ++k) for (int n = 0; n < 32; ++n) B[k][n] = (rand() % 3 == 0) ? 1 : -1;
  // candidate layout: lane l holds row (l & 31), k = 32*(l>>5) + j, j=0..31, nibble j of the 16 bytes
  std::vector<uint32_t> ha(64 * 4, 0), hb(64 * 4, 0);
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; ++j) {
      const int k = 32 * (l >> 5) + j;
      ha[l * 4 + j / 8] |= (uint32_t)fp4_of(A[l & 31][k]) << (4 * (j % 8));
      hb[l * 4 + j / 8] |= (uint32_t)fp4_of(B[k][l & 31]) << (4 * (j % 8));
    }
  uint32_t *da, *db; float* dd;
  hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 64 * 16 * 4);
  hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
  for (int sc : {(int)0x7F7F7F7F, (int)0x80808080, 0}) {
    probe<<<1, 64>>>(da, db, dd, sc, sc);
    std::vector<float> hd(64 * 16);
    hipMemcpy(hd.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost);
    // C/D layout for 32x32 (guide): col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    int bad = 0; double maxabs = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
      const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      int ref = 0; for (int k = 0; k < 64; ++k) ref += A[row][k] * B[k][col];
      if (hd[l * 16 + r] != (float)ref) ++bad;
      if (fabs(hd[l * 16 + r]) > maxabs) maxabs = fabs(hd[l * 16 + r]);
    }
    printf("scale=0x%08x mismatches=%d/1024 max|d|=%g d[0..3]=%g %g %g %g\n", sc, bad, maxabs, hd[0], hd[1], hd[2], hd[3]);
  }
  return 0;
}
