// Matrix-core engine for LceBconv2d on gfx950.
//
// Why it exists: the XOR-popcount formulation is bound by the integer VALU, and on gfx950
// v_bcnt_u32_b32 issues at half rate -- the measured ceiling of the v_xor+v_bcnt pair is
// 8.1e14 binary MAC/s (profiles/r01/valu_peak_microbench.jsonl), 4.4x above the HBM time of
// the BASELINE layer.  The same sum is the +-1 dot product
//       sum_k popcount(a_k ^ w_k) = (K - <a, w>) / 2        (output_transform.h:62-91)
// and the CDNA4 matrix cores evaluate <a, w> EXACTLY for +-1 operands stored as FP4 (E2M1:
// +1 = 0x2, -1 = 0xA, 0 = 0x0) with fp32 accumulation (every partial sum is an integer of
// magnitude < 2^24), at 4.2e15 MAC/s measured (tools/probes/mfma_fp4_peak.hip).  So this
// engine is bit-exact to the xor-popcount one and ~5x higher in ceiling.
//
// Pipeline of one call:
//   1. expand_fp4: bitpacked activations [B,H,W,Cw] -> FP4, spatially PADDED workspace
//      [B,Hp,Wp,Cpad/2 bytes]; border pixels hold +1 (pad_values 1) or 0 (exact SAME-zero
//      padding: an outside tap then contributes 0 to <a,w>, which is what
//      reference.h:100-103 adds as (Cin/G)/2 in popcount units); channels >= Cin hold 0.
//      After this no kernel needs a bounds check.
//   2. bconv2d_mfma: implicit GEMM, M = B*OH*OW pixels, N = Cout, K = KH*KW*Cpad, in K-steps
//      of 64 (one v_mfma_scale_f32_32x32x64_f8f6f4 deep).  A block is WGM x WGN waves, each
//      wave owns WM x WN MFMA tiles of 32x32; A (pixels) and B (weights, pre-expanded and
//      pre-tiled by the planner) tiles go global -> LDS by asynchronous LDS-DMA
//      (buffer_load_dwordx4 ... lds, no VGPR round trip) into two stages, fragments are
//      read with conflict-free ds_read_b128, and the output transform
//      (output_transform.h:93-168) is fused on the fp32 accumulators:
//         2*accum = K_bt - d  ->  clamp -> * mul + bias   (two roundings)
#pragma once
#include <lce_device_intrinsics.h>
#include "lce_kernel_args.h"
#include "lce_kernels.h"

namespace lce {

// ---------------------------------------------------------------------------------
// Step 1: bitpacked -> FP4 padded workspace.  One thread per 16-byte output chunk
// (= 32 channels = one input word).
// ---------------------------------------------------------------------------------
LCE_DEVICE uint32_t spread8_to_nibbles(uint32_t bits8) {
  uint32_t x = bits8 & 0xffu;
  x = (x | (x << 12)) & 0x000f000fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x;  // bit i of the input sits at bit 4*i
}

LCE_KERNEL void __launch_bounds__(256)
expand_fp4(const uint32_t* __restrict__ in, u32x4* __restrict__ out, const MfmaArgs G, uint64_t total) {
  const uint64_t stride = (uint64_t)grid_dim_x() * (uint64_t)block_dim_x();
  for (uint64_t e = (uint64_t)block_idx_x() * (uint64_t)block_dim_x() + (uint64_t)thread_idx_x();
       e < total; e += stride) {
    const uint32_t pix = fastdiv((uint32_t)e, G.div_cpw);   // total < 2^31 (planner chunks the batch)
    const int cc = (int)((uint32_t)e - pix * (uint32_t)G.CPW);
    const uint32_t rowp = fastdiv(pix, G.div_wp);           // b * Hp + yp
    const int xp = (int)(pix - rowp * (uint32_t)G.Wp);
    const uint32_t b = fastdiv(rowp, G.div_hp);
    const int yp = (int)(rowp - b * (uint32_t)G.Hp);
    const int iy = yp - G.PH, ix = xp - G.PW;
    const bool inside = (uint32_t)iy < (uint32_t)G.H && (uint32_t)ix < (uint32_t)G.W;
    uint32_t word = 0;  // outside the image: bit 0 = +1 (pad_values 1)
    if (inside && cc < G.Cw) word = in[(((size_t)b * G.H + iy) * G.W + ix) * (size_t)G.Cw + cc];
    int valid = G.Cin - cc * 32;  // channels of this chunk that really exist
    valid = valid < 0 ? 0 : (valid > 32 ? 32 : valid);
    if (!inside && G.zero_border) valid = 0;
    u32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cnt = valid - 8 * q;
      const uint32_t mask = cnt >= 8 ? 0xffffffffu : (cnt <= 0 ? 0u : ((1u << (4 * cnt)) - 1u));
      v[q] = ((spread8_to_nibbles(word >> (8 * q)) << 3) | 0x22222222u) & mask;
    }
    out[e] = v;
  }
}

// ---------------------------------------------------------------------------------
// Step 2: the GEMM.
//   xp : FP4 workspace, pixel (b, yp, xp) at ((b*Hp + yp)*Wp + xp) * Kc bytes
//   wq : FP4 weights [KS][Npad][32 bytes]  (K-step major, so a block's B tile is one
//        contiguous BN*32-byte run)
//   thrf : bitpacked output: per-channel float t with  bit = (d < t)   (= accum > threshold)
// ---------------------------------------------------------------------------------
template <int DST, int WGM, int WGN, int WM, int WN>
LCE_KERNEL void __launch_bounds__(64 * WGM * WGN)
bconv2d_mfma(const ConvArgs A, const MfmaArgs G, const uint8_t* __restrict__ xp,
             const uint8_t* __restrict__ wq, const float* __restrict__ mul,
             const float* __restrict__ bias, const float* __restrict__ thrf,
             const float* __restrict__ zpc, void* __restrict__ out) {
  // WGM x WGN waves per block, each owning WM x WN MFMA tiles of 32x32
  constexpr int NWAVES = WGM * WGN;
  constexpr int BM = 32 * WM * WGM, BN = 32 * WN * WGN;
  constexpr int A_BYTES = BM * 32, B_BYTES = BN * 32, STAGE = A_BYTES + B_BYTES;
  // LDS image of a stage: A as [k-half][row][16 B], B as [k-half][channel][16 B]; a wave
  // fills it in 1-KiB pieces (64 rows of one half) with one LDS-DMA instruction each.
  constexpr int A_PIECES = BM / 32, PIECES = (BM + BN) / 32;
  constexpr int NP = (PIECES + NWAVES - 1) / NWAVES;  // pieces per wave per K-step
  static_assert(BM % 64 == 0 && BN % 64 == 0, "tiles are filled in 64-row pieces");

  uint8_t* lds = lds_base();
  const int tid = thread_idx_x();
  const int lane = tid & (kWave - 1);
  const int wave = uniform(tid >> 6);
  const int wm = wave % WGM, wn = wave / WGM;
  const int m0 = block_idx_x() * BM, n0 = block_idx_y() * BN;

  const rsrc_t rx = make_rsrc(xp, G.x_bytes);
  const rsrc_t rw = make_rsrc(wq, G.w_bytes);

  // ---- this wave's share of the staging work ------------------------------------------
  uint32_t src[NP];   // byte offset in xp (A piece, at tap (0,0) chunk 0) or in wq (B piece, K-step 0)
  int dst[NP];        // LDS byte offset of the piece inside a stage (wave-uniform)
  bool is_a[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    // surplus slots (PIECES not a multiple of NWAVES) re-copy an earlier piece: harmless
    const int p = (wave + i * NWAVES) % PIECES;
    is_a[i] = p < A_PIECES;
    if (is_a[i]) {
      const int half = p / (BM / 64), blk = p % (BM / 64);
      dst[i] = half * (BM * 16) + blk * 1024;
      int m = m0 + blk * 64 + lane;
      m = m < A.M ? m : A.M - 1;  // tail rows re-read the last pixel; their results are not stored
      const uint32_t rw_ = fastdiv((uint32_t)m, A.div_ow);
      const int ox = m - (int)rw_ * A.OW;
      const uint32_t b = fastdiv(rw_, A.div_oh);
      const int oy = (int)(rw_ - b * (uint32_t)A.OH);
      src[i] = (uint32_t)((((int)b * G.Hp + oy * A.SH) * G.Wp + ox * A.SW) * G.Kc + half * 16);
    } else {
      const int q = p - A_PIECES;
      const int half = q / (BN / 64), blk = q % (BN / 64);
      dst[i] = A_BYTES + half * (BN * 16) + blk * 1024;
      src[i] = (uint32_t)((n0 + blk * 64 + lane) * 32 + half * 16);
    }
  }

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = f32x16_zero();

  const int KS = A.KH * A.KW * G.KCH;

  auto fill = [&](int ks, int stage) {
    const int tap = ks / G.KCH, kc = ks - tap * G.KCH;
    const int fy = tap / A.KW, fx = tap - fy * A.KW;
    const uint32_t a_delta = (uint32_t)((fy * A.DH * G.Wp + fx * A.DW) * G.Kc + kc * 32);
    const uint32_t b_delta = (uint32_t)ks * (uint32_t)(G.Npad * 32);
    uint8_t* base = lds + stage * STAGE;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      buf_load_to_lds16(is_a[i] ? rx : rw, base + dst[i], src[i] + (is_a[i] ? a_delta : b_delta));
  };

  // Two LDS stages: the DMA of K-step ks+1 flies while K-step ks is multiplied; the barrier
  // at the end of the iteration waits for it (vmcnt) and for every wave's fragment reads.
  fill(0, 0);
  block_sync();

  const int half = lane >> 5, l31 = lane & 31;
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) fill(ks + 1, (ks + 1) & 1);
    const uint8_t* base = lds + (ks & 1) * STAGE;
    u32x4 af[WM], bf[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
      af[i] = *(const u32x4*)(base + half * (BM * 16) + ((wm * WM + i) * 32 + l31) * 16);
#pragma unroll
    for (int j = 0; j < WN; ++j)
      bf[j] = *(const u32x4*)(base + A_BYTES + half * (BN * 16) + ((wn * WN + j) * 32 + l31) * 16);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = mfma_fp4_32x32x64(af[i], bf[j], acc[i][j]);
    block_sync();
  }

  // ------------------------------ fused output transform ------------------------------
  const float a_bt = G.a_bt, cminf = G.cmin, cmaxf = G.cmax;
  float mj[WN], bj[WN], tj[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = n0 + (wn * WN + j) * 32 + l31;               // < Npad: the tables are padded
    mj[j] = bj[j] = tj[j] = 0.0f;
    if constexpr (DST == kDstBitpacked) tj[j] = thrf[n];
    else { mj[j] = mul[n]; bj[j] = bias[n]; }
  }
  const bool correct = DST == kDstFloat && A.zero_pad_mode == kZeroPadCorrection;
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int m = m0 + (wm * WM + i) * 32 + row;
      const bool mvalid = m < A.M;
      int zrow = -1;
      if (correct && mvalid) {                                  // optimized_bgemm.h:153-177
        const uint32_t rw_ = fastdiv((uint32_t)m, A.div_ow);
        const int ox = m - (int)rw_ * A.OW;
        const int oy = (int)(rw_ - fastdiv(rw_, A.div_oh) * (uint32_t)A.OH);
        zrow = zero_pad_cache_row(A, oy, ox);
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int nbase = n0 + (wn * WN + j) * 32;
        const int n = nbase + l31;
        const float d = acc[i][j][r];                           // exact +-1 dot product
        if constexpr (DST == kDstBitpacked) {
          // accum > threshold  <=>  d < K_bt - 2*threshold (output_transform.h:160-168)
          const unsigned long long bits = wave_ballot(d < tj[j]);
          if (l31 == 0 && mvalid && nbase < A.N)
            ((uint32_t*)out)[(size_t)m * (size_t)A.Wout + (size_t)(nbase >> 5)] = (uint32_t)(bits >> (32 * half));
        } else {
          const float x = med3(a_bt - d, cminf, cmaxf);         // = float(clamp(accum << 1))
          float y = mul_then_add(x, mj[j], bj[j]);
          if (mvalid && n < A.N) {
            if constexpr (DST == kDstFloat) {
              if (zrow >= 0) y = __fadd_rn(y, zpc[zrow + n]);
              ((float*)out)[(size_t)m * (size_t)A.N + (size_t)n] = y;
            } else {
              float q = round_half_away(y);
              q = fminf(fmaxf(q, -128.0f), 127.0f);
              ((int8_t*)out)[(size_t)m * (size_t)A.N + (size_t)n] = (int8_t)(int)q;
            }
          }
        }
      }
    }
  }
}

}  // namespace lce
