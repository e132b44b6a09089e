// Pointwise (1x1) LceBconv2d on the matrix cores: a streaming kernel.
//
// A 1x1 binary convolution has no spatial structure: every output pixel is the +-1 dot product of
// its own Cin bits with each of the Cout filters (reference.h:62-126 with filter extent 1; no
// padding can occur), so the batch is ONE matrix of M = B*OH*OW pixel rows (a STRIDED 1x1 layer -- the shortcut
// convolutions of ResNet-style binary nets -- only changes which input pixel a row reads).  The general kernel
// (lce_kernels_mfma.h) pays a per-block prologue -- halo rows into LDS, weight ring, barriers --
// that a layer with one or two K-steps cannot amortise: its 1x1 int8 layers sit at 8-24 % of the HBM
// roofline (profiles/r02/layer_kernel_stats.txt).  Here instead:
//   * the whole filter bank of the block's channels lives in REGISTERS for the life of a wave
//     (NC K-steps x NJ channel tiles x 4 VGPRs: <= 64),
//   * a wave walks 32-pixel tiles: one 4-byte load per lane and K-step (lane = pixel l & 31, K-half
//     l >> 5, the operand layout of v_mfma_scale_f32_32x32x64_f8f6f4), bits -> FP4 in registers
//     (fp4_of_full_word, 17 VALU), NC x NJ MFMAs, the fused output transform, one transpose through a
//     wave-private LDS scratch, 16-byte stores of whole rows; the next tile's words are already in flight,
//   * no LDS operand staging, no barriers, no per-block state: waves never wait for each other.
// Same arithmetic as the general kernel -- weights are the planner's FP4 image (pack_for_mfma: negated,
// [K-step][K-half][Npad][16 B]), the accumulators start at K_bt and end as 2 * popcount-accumulator --
// so the results are bit-identical to it and to the oracle.
#pragma once
#include <lce_device_intrinsics.h>
#include "lce_kernel_args.h"
#include "lce_kernels.h"
#include "lce_kernels_mfma.h"

namespace lce {

#ifdef LCE_PW_PHASES
// Profiling aid (never defined in the product build; tools/pw_phases.py): wave 0 of every block stamps the cycle counter
// at entry (0), filter bank + first words resident (1), first tile's stores issued (2), exit (3).
__device__ unsigned long long lce_pw_tl[8192 * 4];
#define LCE_PWPH(slot)                                                                                            \
  do {                                                                                                            \
    const unsigned bid_ = block_idx_y() * grid_dim_x() + block_idx_x();                                           \
    if (thread_idx_x() == 0 && bid_ < 8192) lce_pw_tl[bid_ * 4 + (slot)] = __builtin_readcyclecounter();          \
  } while (0)
#else
#define LCE_PWPH(slot) do {} while (0)
#endif

// DST: kDstFloat / kDstInt8 / kDstBitpacked.  NC = K-steps (64 input channels each; 1, 2, 4 or 8 -- the last with
// NJ <= 2 so that the filter bank stays within 64 registers), NJ = 32-channel tiles per block (grid.y covers
// N / (32*NJ) of them; requires N % 32 == 0).  STRIDED: stride > 1 in either direction.
// <= 256 registers, so the accumulators stay in VGPRs, which the epilogue reads directly (AGPR accumulators cost a
// v_accvgpr_read per value); no tighter cap: a spill reload in the tile loop is a VMEM operation, and waiting for it
// waits for the tile's stores as well
constexpr int pw_min_blocks(int, int) { return 2; }

// I8F (int8 output): the rounding is floor(x + 0.5) in one instruction (lce_kernels.h, pack8_i8_clamped): instances the planner selects only
// when that equals the reference's round-half-away on every value the plan can produce.
template <int DST, int NC, int NJ, bool STRIDED, bool I8F = false>
LCE_KERNEL void __launch_bounds__(256, pw_min_blocks(NC, NJ))
bconv2d_pointwise(const PwArgs P, const uint32_t* __restrict__ in, const uint8_t* __restrict__ wq,
                  const float* __restrict__ mul, const float* __restrict__ bias,
                  const float* __restrict__ thrf, void* __restrict__ out, uint32_t* __restrict__ sign_words) {
  LCE_PWPH(0);
  const int tid = thread_idx_x();
  const int lane = tid & (kWave - 1), wave = uniform(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int n0 = block_idx_y() * (NJ * 32);

  // the filter bank of this block's channels, resident: fragment (K-step c, tile j) of lane (column l31, K-half)
  u32x4 bf[NC][NJ];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      bf[c][j] = *(const u32x4*)(wq + ((size_t)(c * 2 + half) * (size_t)P.Npad + (size_t)(n0 + j * 32 + l31)) * 16);
  // per-channel constants of this lane's NJ channels.  int8: `thrf` holds [lo | hi], the range of the transformed
  // value -- the transform is monotone in the accumulator, so clamping the accumulator to [cmin, cmax] first
  // (output_transform.h:105) equals clamping the result to [f(cmin), f(cmax)] (swapped for negative multipliers), and
  // the planner has already intersected that with int8's [-128, 127]: ONE v_med3 does both clamps
  float mj[NJ], bj[NJ], tj[NJ], uj[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = n0 + j * 32 + l31;
    mj[j] = bj[j] = tj[j] = uj[j] = 0.0f;
    if constexpr (DST == kDstBitpacked) tj[j] = thrf[n];
    else { mj[j] = mul[n]; bj[j] = bias[n]; }
    if constexpr (DST == kDstInt8) { tj[j] = thrf[n]; uj[j] = thrf[P.Npad + n]; }
  }
  // Only the LAST K-step can hold channels that do not exist (a partial last word, or -- odd word count -- an
  // empty upper K-half): their codes are masked to FP4 zero, which contributes nothing.  One mask per lane, computed
  // once (all ones when Cin % 64 == 0), instead of a slow expansion path inside the tile loop.
  u32x4 last_mask;
  {
    int valid = P.Cin - (2 * (NC - 1) + half) * 32;
    valid = valid < 0 ? 0 : (valid > 32 ? 32 : valid);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cnt = valid - 8 * q;
      last_mask[q] = cnt >= 8 ? 0xffffffffu : (cnt <= 0 ? 0u : ((1u << (4 * cnt)) - 1u));
    }
  }

  const rsrc_t rin = make_rsrc(in, P.in_bytes);
  const rsrc_t rout = make_rsrc(out, P.out_bytes);
  constexpr int RW = NJ * 32;                                   // floats per scratch row
  float* scratch = (float*)(lds_base() + wave * (NJ * 4096));   // [32 rows][RW], wave-private
  const int wstride = grid_dim_x() * 4;
  const uint32_t lane_in = (uint32_t)l31 * (uint32_t)P.Cw * 4u + (uint32_t)half * 4u;
  const uint32_t tile_in = 32u * (uint32_t)P.Cw * 4u;

  auto load_words = [&](int t, uint32_t (&w)[NC]) LCE_LAMBDA_INLINE {
    // a lane past the last pixel, or a K-half past the last word, reads out of range or a neighbour's word:
    // zeros from the range check, or masked by `valid` below
    uint32_t base;
    if constexpr (STRIDED) {
      // (a pixel past the launch's last lands in a later image or behind the input: zeros, and nothing stores its row)
      const uint32_t p = (uint32_t)t * 32u + (uint32_t)l31;
      const uint32_t b = fastdiv(p, P.div_ohw), r = p - b * (uint32_t)P.OHW;
      const uint32_t oy = fastdiv(r, P.div_ow), ox = r - oy * (uint32_t)P.OW;
      const uint32_t pix = b * (uint32_t)P.IHW + oy * (uint32_t)(P.SH * P.IW) + ox * (uint32_t)P.SW;
      base = t < P.tiles ? pix * (uint32_t)P.Cw * 4u + (uint32_t)half * 4u : kOobOffset;
    } else {
      base = t < P.tiles ? (uint32_t)t * tile_in + lane_in : kOobOffset;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) w[c] = buf_load(rin, base + (uint32_t)c * 8u, (uint32_t*)nullptr);
  };

  const f32x16 kbt = f32x16_fill(P.a_bt);     // the accumulators' start value, kept in registers (no per-tile fill)
  // int8 / bitpacked: the registers a tile's stores read are NOT reused by the next tile (two sets, alternating, kept
  // alive below).  Otherwise the wave has to wait for the stores' completion -- the full HBM write latency -- before it
  // may overwrite them, i.e. before every tile: compute and store time ADD UP (measured: 56x56x64 int8 12.7 us without
  // the stores, 20.7 with them, 51 MB at the chip's full write rate being 8 us).
  constexpr int HN = DST == kDstInt8 ? 32 / (64 / (NJ * 2)) : 1;
  u32x4 hold[2][HN];
#pragma unroll
  for (int k = 0; k < HN; ++k) hold[0][k] = hold[1][k] = u32x4{0u, 0u, 0u, 0u};

  auto tile = [&](auto sc, int t, const uint32_t (&wcur)[NC]) LCE_LAMBDA_INLINE {
    constexpr int S = decltype(sc)::value;
    f32x16 acc[NJ];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      u32x4 a = fp4_of_full_word(wcur[c]);
      if (c == NC - 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] &= last_mask[q];
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)   // K_bt + <a, -w> = K_bt - <a, w> = 2 * accum: the first step adds to the constant tile
        acc[j] = mfma_fp4_32x32x64_unscaled(a, bf[c][j], c == 0 ? kbt : acc[j]);
    }

    const uint32_t row0 = (uint32_t)t * 32u;                      // first pixel of the tile
    if constexpr (DST == kDstBitpacked) {
      // bit = accum > threshold; register r holds pixel rows q (lanes 0-31) and q + 4 (lanes 32-63),
      // q = (r & 3) + 8 * (r >> 2): one compare + ballot per 32 channels, v_writelane drops the word into
      // the lane that stores it -- afterwards lane p < 32 owns row p (as bit_rows in lce_kernels_mfma.h)
      uint32_t words[NJ];
      gather_tile_bits<NJ, false>(acc, tj, words);          // (lce_kernels_mfma.h: no hazard padding per register)
      const uint32_t m = row0 + (uint32_t)lane;
#pragma unroll
      for (int j = 0; j < NJ; ++j) hold[S][0][j] = words[j];
      if (lane < 32 && m < (uint32_t)P.M) {
        uint32_t* o = (uint32_t*)out + (size_t)m * (size_t)P.Wout + (size_t)(n0 >> 5);
        if constexpr (NJ == 4) {
          if ((P.Wout & 3) == 0) { store_words(o, hold[S][0]); }
          else { store_words(o, hold[S][0][0]); store_words(o + 1, hold[S][0][1]); store_words(o + 2, hold[S][0][2]); store_words(o + 3, hold[S][0][3]); }
        } else if constexpr (NJ == 2) {
          if ((P.Wout & 1) == 0) { u32x2 v = {hold[S][0][0], hold[S][0][1]}; store_words(o, v); }
          else { store_words(o, hold[S][0][0]); store_words(o + 1, hold[S][0][1]); }
        } else {
          store_words(o, hold[S][0][0]);
        }
      }
    } else {
      // the transform on the accumulators (output_transform.h:93-157), then the transpose: a lane holds one
      // channel of 16 pixel rows, the output wants rows of consecutive channels
      // in place: the accumulators become the values the output (or its int8 rounding) is made of
      if constexpr (DST == kDstInt8) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 x = {acc[j][r], acc[j][r + 1]};
            const f32x2 y = mul_then_add2(x, mj[j], bj[j]);
            acc[j][r] = med3(y[0], tj[j], uj[j]);
            acc[j][r + 1] = med3(y[1], tj[j], uj[j]);
          }
      } else if (P.noclamp) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 x = {acc[j][r], acc[j][r + 1]};
            const f32x2 y = mul_then_add2(x, mj[j], bj[j]);
            acc[j][r] = y[0];
            acc[j][r + 1] = y[1];
          }
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 x = {med3(acc[j][r], P.cmin, P.cmax), med3(acc[j][r + 1], P.cmin, P.cmax)};
            const f32x2 y = mul_then_add2(x, mj[j], bj[j]);
            acc[j][r] = y[0];
            acc[j][r + 1] = y[1];
          }
      }
      // optional second output: the LceQuantize of those values (bit = value < bit_thr: 0 for float, the planner's
      // "rounds below the zero point" threshold for int8), gathered like the bitpacked output above
      if (sign_words != nullptr) {
        uint32_t words[NJ];
        float bthr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bthr[j] = P.bit_thr;
        gather_tile_bits<NJ, true>(acc, bthr, words);
        const uint32_t m = row0 + (uint32_t)lane;
        if (lane < 32 && m < (uint32_t)P.M) {
          uint32_t* o = sign_words + (size_t)m * (size_t)P.Wout + (size_t)(n0 >> 5);
#pragma unroll
          for (int j = 0; j < NJ; ++j) store_words(o + j, words[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * half) * RW + j * 32 + l31] = acc[j][r];
      wave_lds_fence();
      if constexpr (DST == kDstFloat) {
        constexpr int LPR = RW / 4, RPI = 64 / LPR, NK = 32 / RPI;   // lanes per row, rows per store, stores per tile
        const int g = lane % LPR, rr = lane / LPR;
        f32x4 y[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) y[k] = *(const f32x4*)(scratch + (rr + k * RPI) * RW + g * 4);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const uint32_t row = row0 + (uint32_t)(rr + k * RPI);
          buf_store_streaming(rout, (row * (uint32_t)P.N + (uint32_t)(n0 + g * 4)) * 4u, y[k]);
        }
      } else {
        // int8: a lane converts 16 consecutive channels of one row into ONE 16-byte store
        constexpr int GPR = NJ * 2, RPI = 64 / GPR, NK = 32 / RPI;   // 16-channel groups per row, rows per store
        const int g = lane % GPR, rr = lane / GPR;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const f32x4* src = (const f32x4*)(scratch + (rr + k * RPI) * RW + g * 16);
#pragma unroll
          for (int q = 0; q < 4; q += 2) {       // (values already inside [-128, 127])
            uint32_t lo, hi;
            pack8_i8_clamped<I8F>(src[q], src[q + 1], lo, hi);
            hold[S][k][q] = lo;
            hold[S][k][q + 1] = hi;
          }
          const uint32_t row = row0 + (uint32_t)(rr + k * RPI);
#ifdef LCE_PW_NOSTORE   // timing ablation (results are wrong)
          if (hold[S][k][0] == 0x12345678u)
#endif
          buf_store(rout, row * (uint32_t)P.N + (uint32_t)(n0 + g * 16), hold[S][k]);
        }
      }
      wave_lds_fence();
    }
    // the OTHER set stays untouched (and its registers allocated) through this tile
#pragma unroll
    for (int k = 0; k < HN; ++k) keep_alive(hold[S ^ 1][k]);
  };

  int t = block_idx_x() * 4 + wave;
  uint32_t w0[NC], w1[NC];
  load_words(t, w0);
  // nothing pending when the loop is entered: the compiler merges the "what may still be in flight" state of the
  // preheader (where this load is the NEWEST memory operation) with the back edge's (where it is followed by two
  // stores) pessimistically and would otherwise wait with vmcnt(0) at the loop head, i.e. for the previous tile's stores
  wait_vmcnt<0>();
  LCE_PWPH(1);
  // pairs of tiles in a branch-free body (the compiler's s_waitcnt placement stays exact: with a conditional second
  // tile it falls back to vmcnt(0) at the loop head, which waits for the stores again), then the odd one
  for (; t + wstride < P.tiles; t += 2 * wstride) {
    load_words(t + wstride, w1);                 // the next tile's words are in flight while this one computes
    tile(IntC<0>{}, t, w0);
    load_words(t + 2 * wstride, w0);
    tile(IntC<1>{}, t + wstride, w1);
  }
  if (t < P.tiles) tile(IntC<0>{}, t, w0);
  LCE_PWPH(2);
#ifdef LCE_PW_PHASES
  wait_vmcnt<0>();      // the stores' acknowledgements
  LCE_PWPH(3);
#endif
}

}  // namespace lce
