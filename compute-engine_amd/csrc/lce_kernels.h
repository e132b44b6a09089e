// Hand-written CDNA4 (gfx950) kernels for Larq Compute Engine's LceBconv2d /
// LceQuantize hot path.  Citations are relative to
// /root/reference/larq_compute_engine/.
//
// Design (see DESIGN.md for the numbers):
//   * bconv2d is an implicit-GEMM with NO im2col buffer and NO indirection buffer
//     (the reference materialises one of those, core/bconv2d/optimized_bgemm.h:19-61 /
//     core/indirect_bgemm/kernel.h:101-174).  A lane owns TM output pixels, a wave owns
//     a tile of TN output channels.  Because all 64 lanes of a wave work on the same
//     channels, the bitpacked WEIGHT words are wave-uniform: they are fetched with
//     scalar loads (s_load_dwordx16 through the scalar cache) and enter the VALU as
//     SGPR operands -- no LDS traffic, no VGPRs for the B operand.  The bitpacked
//     ACTIVATION words are per-lane raw buffer loads; an out-of-image tap is steered
//     to an out-of-range buffer offset, for which the hardware returns 0 = the "+1"
//     padding word of the reference (reference.h:105-106).
//   * The inner product is v_xor_b32 + v_bcnt_u32_b32 (accumulating popcount): two
//     VALU ops per 32 binary MACs, int32 accumulators in VGPRs.  MFMA is not used.
//   * The output transform (output_transform.h:93-168) is fused: shift, clamp,
//     int->float, multiply, add with SEPARATE roundings, then store / round+saturate /
//     threshold+bitpack.
#pragma once
#include <lce_device_intrinsics.h>  // resolved through -I: csrc/ for the product
#include "lce_kernel_args.h"

namespace lce {
using namespace lce_dev;

LCE_DEVICE uint32_t fastdiv(uint32_t n, FastDiv d) {
  return d.magic == 0u ? n : (mulhi_u32(n, d.magic) >> d.shift);
}
// fastdiv without the divisor-1 branch: a divisor of 1 has magic 0, so the multiply-high contributes 0 and the
// masked addend is n itself.
LCE_DEVICE uint32_t fastdiv_nb(uint32_t n, FastDivNB d) {
  return (mulhi_u32(n, d.magic) >> d.shift) + (n & d.pass);
}

// ---------------------------------------------------------------------------------
// Output transform pieces (core/bconv2d/output_transform.h).
// ---------------------------------------------------------------------------------

// :99-106 -- x = accum << 1, clamp in int32, then float(x) * mul + bias as TWO
// roundings (mul_then_add keeps the product out of reach of FMA contraction).
LCE_DEVICE float ot_float(int acc, int cmin, int cmax, float mul, float bias) {
  int x = acc << 1;
  x = x < cmax ? x : cmax;
  x = x > cmin ? x : cmin;
  return mul_then_add((float)x, mul, bias);
}

// saturate<int8>(std::round(y)) (output_transform.h:31-44) in four instructions.  Rounding is
// monotone and the bounds are integers, so clamping first is equivalent; for |y| <= 128,
// trunc(y + copysign(pred(0.5), y)) IS round-half-away: with 0.5 itself the sum for
// y = pred(0.5) would round up to 1.0, with pred(0.5) = 0x1.fffffep-2 every exact tie
// n + 0.5 lands within half an ulp below n + 1 and rounds up to it, and everything below a
// tie stays below.  Checked against roundf for EVERY float by
// tests/test_hostsim_kernels.py::test_round_sat_i8_every_float.
LCE_DEVICE int round_sat_i8(float y) {
  const float c = med3(y, -128.0f, 127.0f);
  return (int)(c + __builtin_copysignf(0x1.fffffep-2f, c));   // float -> int conversion truncates
}

// The same rounding for two values that are already inside [-128, 127] (the clamp happened earlier): the add is
// one v_pk_add_f32.
LCE_DEVICE void round_i8_clamped2(float c0, float c1, int& q0, int& q1) {
  const f32x2 c = {c0, c1};
  const f32x2 h = {__builtin_copysignf(0x1.fffffep-2f, c0), __builtin_copysignf(0x1.fffffep-2f, c1)};
  const f32x2 s = add2(c, h);
  q0 = (int)s[0];
  q1 = (int)s[1];
}

// Eight values that are already inside [-128, 127] -> eight int8 in two dwords: the same rounding, the truncating conversion packs
// as it goes (cvt_pack8_i8, lce_device_intrinsics.h).
LCE_DEVICE void round_pack8_i8_clamped(f32x4 a, f32x4 b, uint32_t& lo, uint32_t& hi) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = a[i] + __builtin_copysignf(0x1.fffffep-2f, a[i]);
    b[i] = b[i] + __builtin_copysignf(0x1.fffffep-2f, b[i]);
  }
  cvt_pack8_i8(a, b, lo, hi);
}

// The two int8 roundings of the streaming kernels' epilogues, on eight values already inside [-128, 127]: I8F = false: round half away
// from zero as the reference writes it (the add + the truncating conversion); I8F = true: floor(x + 0.5) in one instruction per value
// (cvt_rpi_pack8_i8) -- the planner selects those instances only for plans on which the two agree on every value the layer can produce.
template <bool I8F>
LCE_DEVICE void pack8_i8_clamped(f32x4 a, f32x4 b, uint32_t& lo, uint32_t& hi) {
  if constexpr (I8F) cvt_rpi_pack8_i8(a, b, lo, hi);
  else round_pack8_i8_clamped(a, b, lo, hi);
}

// :17-27,31-44,133-143 -- std::round (half away from zero), saturate to int8.
LCE_DEVICE int ot_int8(int acc, int cmin, int cmax, float mul, float bias) {
  return round_sat_i8(ot_float(acc, cmin, cmax, mul, bias));
}

// zero_padding_correction.h:196-275: which cached correction row (if any) applies to
// output pixel (oy, ox).  Returns the float offset of that row in the cache, or -1.
LCE_DEVICE int zero_pad_cache_row(const ConvArgs& A, int oy, int ox) {
  const int o_top = A.top_off - oy * A.SH;
  const int o_bot = -o_top - A.H + A.eKH;
  const int o_left = A.left_off - ox * A.SW;
  const int o_right = -o_left - A.W + A.eKW;
  if (o_left <= 0 && o_right <= 0 && o_top <= 0 && o_bot <= 0) return -1;
  int kase, cx, cy;
  if (o_right <= 0 && o_top > 0 && o_bot < 0) {
    kase = 0; cx = o_left >= 0 ? o_left : 0; cy = o_top;
  } else if (o_left < 0 && o_right > 0 && o_bot <= 0) {
    kase = 1; cx = o_right; cy = o_top >= 0 ? o_top : 0;
  } else if (o_left > 0 && o_right < 0 && o_top <= 0) {
    kase = 2; cx = o_left; cy = o_bot >= 0 ? o_bot : 0;
  } else if (o_left <= 0 && o_top < 0 && o_bot > 0) {
    kase = 3; cx = o_right >= 0 ? o_right : 0; cy = o_bot;
  } else {
    return -1;  // the reference's "cannot happen" branch (:268-271): no correction
  }
  return ((kase * A.eKH + cy) * A.eKW + cx) * A.N;
}

template <int CH> struct WordVec;
template <> struct WordVec<1> { typedef uint32_t type; };
template <> struct WordVec<2> { typedef u32x2 type; };
template <> struct WordVec<4> { typedef u32x4 type; };
LCE_DEVICE uint32_t word_of(uint32_t v, int) { return v; }
LCE_DEVICE uint32_t word_of(u32x2 v, int i) { return v[i]; }
LCE_DEVICE uint32_t word_of(u32x4 v, int i) { return v[i]; }

// ---------------------------------------------------------------------------------
// bconv2d, tiled kernel.
//
//   DST : kDstFloat / kDstInt8 / kDstBitpacked (bitpacked requires TN == 32)
//   TM  : output pixels per lane  (1, 2 or 4);  a wave covers 64*TM pixels
//   TN  : output channels per wave tile (16 or 32); acc registers = TM*TN
//   CH  : activation words per vector load (1, 2 or 4); requires Cwg % CH == 0
//
// Work item = (pixel tile, channel tile); consecutive waves take consecutive channel
// tiles of the SAME pixel tile so that their activation loads hit in the CU's L1.
//
// Packed weights `wp` : [NT][KH*KW][Cwg][TN] words (channels >= N zero-filled); built
// once by the planner (the role of indirect_bgemm::Kernel::PackWeights, kernel.h:54-94).
// `mul`/`bias`/`thr` are padded to NT*TN entries; `oobc` is [NT][KH*KW][TN] int32 with
// (Cin/G)/2 - popcount(filter tap) for kZeroPadExact.
// ---------------------------------------------------------------------------------
template <int DST, int TM, int TN, int CH>
LCE_KERNEL void __launch_bounds__(256)
bconv2d_tiled(const ConvArgs A, const uint32_t* __restrict__ in,
              const uint32_t* __restrict__ wp, const float* __restrict__ mul,
              const float* __restrict__ bias, const int32_t* __restrict__ thr,
              const int32_t* __restrict__ oobc, const float* __restrict__ zpc,
              void* __restrict__ out) {
  static_assert(DST != kDstBitpacked || TN == 32, "one output word per tile");
  typedef typename WordVec<CH>::type vec_t;

  const int lane = thread_idx_x() & (kWave - 1);
  const int wave = uniform(thread_idx_x() >> 6);
  const int task = block_idx_x() * (block_dim_x() >> 6) + wave;  // wave-uniform
  if (task >= A.PT * A.NT) return;
  const int pt = task / A.NT;
  const int nt = task - pt * A.NT;
  const int n0 = nt * TN;
  const int group = n0 / A.Npg;
  const int taps = A.KH * A.KW;

  const rsrc_t rs = make_rsrc(in, A.in_bytes);

  int acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = 0;

  // decode this lane's TM output pixels
  int m_of[TM], y0[TM], x0[TM], base[TM];
  bool valid[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = (pt * TM + i) * kWave + lane;
    valid[i] = m < A.M;
    const uint32_t mm = valid[i] ? (uint32_t)m : 0u;
    const uint32_t row = fastdiv(mm, A.div_ow);       // b * OH + oy
    const int ox = (int)(mm - row * (uint32_t)A.OW);
    const uint32_t b = fastdiv(row, A.div_oh);
    const int oy = (int)(row - b * (uint32_t)A.OH);
    m_of[i] = (int)mm;
    y0[i] = oy * A.SH - A.PH;
    x0[i] = ox * A.SW - A.PW;
    base[i] = (((int)b * A.H + y0[i]) * A.W + x0[i]) * A.Cw + group * A.Cwg;  // in words
  }

  const uint32_t* wrow = wp + (size_t)nt * (size_t)(taps * A.Cwg * TN);
  const int32_t* crow = oobc + (size_t)nt * (size_t)(taps * TN);
  const int nchunks = A.Cwg / CH;

  for (int fy = 0; fy < A.KH; ++fy) {
    for (int fx = 0; fx < A.KW; ++fx) {
      uint32_t off[TM];
      bool oob[TM];
      const int tap_delta = (fy * A.DH * A.W + fx * A.DW) * A.Cw;  // words, wave-uniform
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int iy = y0[i] + fy * A.DH, ix = x0[i] + fx * A.DW;
        const bool inside =
            valid[i] && (uint32_t)iy < (uint32_t)A.H && (uint32_t)ix < (uint32_t)A.W;
        oob[i] = !inside;
        off[i] = inside ? (uint32_t)(base[i] + tap_delta) * 4u : kOobOffset;
      }
      for (int c = 0; c < nchunks; ++c) {
        vec_t a[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = buf_load(rs, off[i] + (uint32_t)(c * CH * 4), (vec_t*)nullptr);
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const uint32_t w = wrow[cc * TN + j];  // scalar load, SGPR operand
            if constexpr (TM == 1) {
              xor_popc_acc(acc[0][j], w, word_of(a[0], cc));
            } else if constexpr (TM == 2) {
              xor_popc_acc(acc[0][j], acc[1][j], w, word_of(a[0], cc), word_of(a[1], cc));
            } else {
              xor_popc_acc(acc[0][j], acc[1][j], acc[2][j], acc[3][j], w, word_of(a[0], cc),
                           word_of(a[1], cc), word_of(a[2], cc), word_of(a[3], cc));
            }
          }
        }
        wrow += CH * TN;
      }
      if (A.zero_pad_mode == kZeroPadExact) {
        // The loads above returned 0 for an outside tap, so popcount(filter tap) was
        // accumulated; replace it by (Cin/G)/2 (reference.h:100-103).
        bool any = false;
#pragma unroll
        for (int i = 0; i < TM; ++i) any = any || (oob[i] && valid[i]);
        if (wave_any(any)) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int cj = crow[j];
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] += oob[i] ? cj : 0;
          }
        }
      }
      crow += TN;
    }
  }

  // ------------------------------ fused output transform ------------------------------
  const bool full_tile = n0 + TN <= A.N;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (!valid[i]) continue;
    const size_t m = (size_t)m_of[i];
    if constexpr (DST == kDstBitpacked) {
      // output_transform.h:160-168: bit = accum > threshold; channel c -> bit c%32 of word
      // c/32 (reference.h:120-136).  thr[] is padded with INT32_MAX so bits >= N stay 0.
      uint32_t word = 0;
#pragma unroll
      for (int j = 0; j < TN; ++j) word |= (acc[i][j] > thr[n0 + j] ? 1u : 0u) << j;
      ((uint32_t*)out)[m * (size_t)A.Wout + (size_t)nt] = word;
    } else {
      int zrow = -1;
      if (DST == kDstFloat && A.zero_pad_mode == kZeroPadCorrection) {
        const uint32_t row = fastdiv((uint32_t)m_of[i], A.div_ow);
        const int ox = m_of[i] - (int)row * A.OW;
        const int oy = (int)(row - fastdiv(row, A.div_oh) * (uint32_t)A.OH);
        zrow = zero_pad_cache_row(A, oy, ox);
      }
      if constexpr (DST == kDstFloat) {
        float* o = (float*)out + m * (size_t)A.N + (size_t)n0;
        float y[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          y[j] = ot_float(acc[i][j], A.clamp_min, A.clamp_max, mul[n0 + j], bias[n0 + j]);
        if (zrow >= 0) {
          // optimized_bgemm.h:153-177: the correction is a float add AFTER the transform
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (n0 + j < A.N) y[j] = __fadd_rn(y[j], zpc[zrow + n0 + j]);
        }
        if (full_tile && (A.N & 3) == 0) {
#pragma unroll
          for (int j = 0; j < TN; j += 4) {
            f32x4 v = {y[j], y[j + 1], y[j + 2], y[j + 3]};
            *(f32x4*)(o + j) = v;
          }
        } else {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (n0 + j < A.N) o[j] = y[j];
        }
      } else {
        int8_t* o = (int8_t*)out + m * (size_t)A.N + (size_t)n0;
        int q[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          q[j] = ot_int8(acc[i][j], A.clamp_min, A.clamp_max, mul[n0 + j], bias[n0 + j]);
        if (full_tile && (A.N & 3) == 0) {
#pragma unroll
          for (int j = 0; j < TN; j += 4) {
            *(uint32_t*)(o + j) = pack4_u8(q[j], q[j + 1], q[j + 2], q[j + 3]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (n0 + j < A.N) o[j] = (int8_t)q[j];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// bconv2d, general kernel: any groups / channel counts, one output channel at a time.
// Lane = one output pixel, blockIdx.y = a chunk of 32 output channels (= one bitpacked
// output word).  Weights are read in their ORIGINAL OHWI layout with scalar loads.
// Used for the shapes the tiled kernel does not take (grouped convolutions whose
// channels-per-group is not a multiple of the tile) and as an independent second
// implementation in the tests.
// ---------------------------------------------------------------------------------
template <int DST>
LCE_KERNEL void __launch_bounds__(256)
bconv2d_general(const ConvArgs A, const uint32_t* __restrict__ in,
                const uint32_t* __restrict__ filt, const float* __restrict__ mul,
                const float* __restrict__ bias, const int32_t* __restrict__ thr,
                const float* __restrict__ zpc, void* __restrict__ out) {
  const int m = block_idx_x() * block_dim_x() + thread_idx_x();
  const bool valid = m < A.M;
  const uint32_t mm = valid ? (uint32_t)m : 0u;
  const uint32_t row = fastdiv(mm, A.div_ow);
  const int ox = (int)(mm - row * (uint32_t)A.OW);
  const uint32_t b = fastdiv(row, A.div_oh);
  const int oy = (int)(row - b * (uint32_t)A.OH);
  const int y0 = oy * A.SH - A.PH, x0 = ox * A.SW - A.PW;
  const int pix_base = (((int)b * A.H + y0) * A.W + x0) * A.Cw;
  const rsrc_t rs = make_rsrc(in, A.in_bytes);
  const int chunk = block_idx_y();
  const int n_begin = chunk * 32;
  const int n_end = n_begin + 32 < A.N ? n_begin + 32 : A.N;
  const int taps = A.KH * A.KW;

  int zrow = -1;
  if (DST == kDstFloat && A.zero_pad_mode == kZeroPadCorrection) zrow = zero_pad_cache_row(A, oy, ox);

  uint32_t word = 0;
  for (int n = n_begin; n < n_end; ++n) {
    const int group = n / A.Npg;
    const uint32_t* w = filt + (size_t)n * (size_t)(taps * A.Cwg);
    int acc = 0;
    for (int fy = 0; fy < A.KH; ++fy) {
      for (int fx = 0; fx < A.KW; ++fx) {
        const int iy = y0 + fy * A.DH, ix = x0 + fx * A.DW;
        const bool inside = valid && (uint32_t)iy < (uint32_t)A.H && (uint32_t)ix < (uint32_t)A.W;
        if (A.zero_pad_mode == kZeroPadExact && !inside) {
          acc += A.bzp;  // reference.h:100-103
        } else {
          const uint32_t off =
              inside ? (uint32_t)(pix_base + (fy * A.DH * A.W + fx * A.DW) * A.Cw + group * A.Cwg) * 4u
                     : kOobOffset;
          for (int c = 0; c < A.Cwg; ++c)
            acc += popc(buf_load(rs, off + (uint32_t)c * 4u, (uint32_t*)nullptr) ^ w[c]);
        }
        w += A.Cwg;
      }
    }
    if (!valid) continue;
    if constexpr (DST == kDstBitpacked) {
      word |= (acc > thr[n] ? 1u : 0u) << (n - n_begin);
    } else if constexpr (DST == kDstFloat) {
      float y = ot_float(acc, A.clamp_min, A.clamp_max, mul[n], bias[n]);
      if (zrow >= 0) y = __fadd_rn(y, zpc[zrow + n]);
      ((float*)out)[(size_t)m * (size_t)A.N + (size_t)n] = y;
    } else {
      ((int8_t*)out)[(size_t)m * (size_t)A.N + (size_t)n] =
          (int8_t)ot_int8(acc, A.clamp_min, A.clamp_max, mul[n], bias[n]);
    }
  }
  if (DST == kDstBitpacked && valid) ((uint32_t*)out)[(size_t)m * (size_t)A.Wout + (size_t)chunk] = word;
}

// ---------------------------------------------------------------------------------
// LceQuantize: sign-bit packing (core/bitpacking/bitpack.h:248-308).
// bit i of a word = (x[i] < zero_point); LSB first; per-row padding bits are 0.
// ---------------------------------------------------------------------------------

// Any element type, any `cols`: one wave packs 64 consecutive columns of one row with a
// single ballot (coalesced 64-element loads, two output words per ballot).
template <typename T>
LCE_KERNEL void __launch_bounds__(256)
bitpack_rows(const T* __restrict__ in, uint32_t* __restrict__ out, uint32_t rows,
             uint32_t cols, uint32_t wpr, int32_t zero_point, FastDiv div_segs,
             uint32_t segs, uint64_t total_tasks) {
  const int lane = thread_idx_x() & (kWave - 1);
  const uint64_t wave0 = (uint64_t)block_idx_x() * (uint64_t)(block_dim_x() >> 6) + (uint64_t)(thread_idx_x() >> 6);
  const uint64_t nwaves = (uint64_t)grid_dim_x() * (uint64_t)(block_dim_x() >> 6);
  for (uint64_t t = wave0; t < total_tasks; t += nwaves) {
    const uint64_t row = t / segs;
    const uint32_t seg = (uint32_t)(t - row * segs);
    const uint32_t col = seg * 64u + (uint32_t)lane;
    bool neg = false;
    if (col < cols) {
      const T v = in[row * (uint64_t)cols + col];
      if constexpr (sizeof(T) == 4) neg = v < (T)0;            // float: zero_point is 0 (bitpack.h:202-203)
      else neg = (int32_t)v < zero_point;                        // int8 / bool-as-uint8
    }
    const unsigned long long bits = wave_ballot(neg);
    const uint32_t w = seg * 2u + (uint32_t)lane;
    if (lane < 2 && w < wpr) out[row * (uint64_t)wpr + w] = (uint32_t)(bits >> (32 * lane));
  }
  (void)rows; (void)div_segs;
}

// Float, cols % 32 == 0 (the tensor is one flat array, bitpack.h:294-298): 16-byte loads.
// Per iteration a wave turns 4 KiB of floats into 32 words: the 8 lanes of a lane group
// cover one 128-B line (one output word) per load, 4 loads in flight per lane; the 8
// nibbles of a word are OR-reduced across the group with 3 xor-shuffles and lane 0 of
// each group stores its 4 words as one 16-byte store.
template <int UNUSED = 0>     // (a template so that only the translation unit that launches it emits it)
LCE_KERNEL void __launch_bounds__(256)
bitpack_f32_flat(const float* __restrict__ in, uint32_t* __restrict__ out, uint64_t nblocks32) {
  const int lane = thread_idx_x() & (kWave - 1);
  const int grp = lane >> 3, sub = lane & 7;
  const uint64_t wave0 = (uint64_t)block_idx_x() * (uint64_t)(block_dim_x() >> 6) + (uint64_t)(thread_idx_x() >> 6);
  const uint64_t nwaves = (uint64_t)grid_dim_x() * (uint64_t)(block_dim_x() >> 6);
  for (uint64_t blk = wave0; blk < nblocks32; blk += nwaves) {  // 32 words = 1024 floats
    const f32x4* src = (const f32x4*)(in + blk * 1024ull) + (grp * 4) * 8 + sub;
    f32x4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = load_streaming(src + c * 8);   // read once: +9 % measured
    u32x4 words;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint32_t nib = (v[c][0] < 0.0f ? 1u : 0u) | (v[c][1] < 0.0f ? 2u : 0u) |
                     (v[c][2] < 0.0f ? 4u : 0u) | (v[c][3] < 0.0f ? 8u : 0u);
      uint32_t w = nib << (4 * sub);
      w |= shfl_xor(w, 1);
      w |= shfl_xor(w, 2);
      w |= shfl_xor(w, 4);
      words[c] = w;
    }
    if (sub == 0) *((u32x4*)(out + blk * 32ull) + grp) = words;
  }
}

// Int8 (or bool viewed as uint8 with zero_point 1), cols % 32 == 0: 16-byte loads, two
// lanes per output word.
template <bool kUnsigned>
LCE_KERNEL void __launch_bounds__(256)
bitpack_b8_flat(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t nwords_pairs,
                int32_t zero_point) {
  // one lane = 16 bytes = half a word; a wave = 32 words per iteration
  const int lane = thread_idx_x() & (kWave - 1);
  const uint64_t wave0 = (uint64_t)block_idx_x() * (uint64_t)(block_dim_x() >> 6) + (uint64_t)(thread_idx_x() >> 6);
  const uint64_t nwaves = (uint64_t)grid_dim_x() * (uint64_t)(block_dim_x() >> 6);
  for (uint64_t blk = wave0; blk < nwords_pairs; blk += nwaves) {  // 32 words = 1024 bytes
    // plain (cacheable) load: an int8 feature map of this size is usually still in the
    // 256 MB memory-side cache from the layer that produced it; the streaming hint lost 12 % here
    const u32x4 v = *((const u32x4*)(in + blk * 1024ull) + lane);
    uint32_t half = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t byte = (v[d] >> (8 * k)) & 0xffu;
        const int32_t x = kUnsigned ? (int32_t)byte : (int32_t)(int8_t)byte;
        half |= (x < zero_point ? 1u : 0u) << (d * 4 + k);
      }
    }
    const uint32_t other = shfl_xor(half, 1);
    if ((lane & 1) == 0) out[blk * 32ull + (uint32_t)(lane >> 1)] = half | (other << 16);
  }
}

// ---------------------------------------------------------------------------------
// LceDequantize: bit 0 -> zero_bit_value, bit 1 -> one_bit_value
// (core/bitpacking/bitpack.h:310-346, tflite/kernels/quantization.cc:116-147).
// ---------------------------------------------------------------------------------
template <typename T>
LCE_KERNEL void __launch_bounds__(256)
unpack_rows(const uint32_t* __restrict__ in, T* __restrict__ out, uint64_t total, uint32_t cols,
            uint32_t wpr, T zero_bit_value, T one_bit_value) {
  const uint64_t stride = (uint64_t)grid_dim_x() * (uint64_t)block_dim_x();
  for (uint64_t e = (uint64_t)block_idx_x() * (uint64_t)block_dim_x() + (uint64_t)thread_idx_x(); e < total; e += stride) {
    const uint64_t row = e / cols;
    const uint32_t c = (uint32_t)(e - row * cols);
    const uint32_t w = in[row * (uint64_t)wpr + (c >> 5)];
    out[e] = ((w >> (c & 31)) & 1u) ? one_bit_value : zero_bit_value;
  }
}

// cols % 32 == 0: the tensor is one flat bit stream (bitpack.h:294-298 in reverse).  A thread
// turns 16 / sizeof(T) consecutive bits into one 16-byte store, so the lanes of a wave write
// consecutive 16-byte pieces (float: 8 lanes share a word and fill one 128-byte line), and
// there is no division anywhere.
template <typename T>
LCE_KERNEL void __launch_bounds__(256)
unpack_flat(const uint32_t* __restrict__ in, u32x4* __restrict__ out, uint64_t chunks, T zero_bit_value,
            T one_bit_value) {
  constexpr int EPT = 16 / (int)sizeof(T);        // elements per thread: 4 floats or 16 bytes
  constexpr int TPW = 32 / EPT;                   // threads per input word
  const uint64_t stride = (uint64_t)grid_dim_x() * (uint64_t)block_dim_x();
  for (uint64_t e = (uint64_t)block_idx_x() * (uint64_t)block_dim_x() + (uint64_t)thread_idx_x(); e < chunks; e += stride) {
    const uint32_t bits = in[e / TPW] >> ((uint32_t)(e % TPW) * EPT);
    union { T v[EPT]; u32x4 q; } u;
#pragma unroll
    for (int k = 0; k < EPT; ++k) u.v[k] = ((bits >> k) & 1u) ? one_bit_value : zero_bit_value;
    store_streaming(out + e, u.q);
  }
}

// ---------------------------------------------------------------------------------
// LceBMaxPool2d: bitwise AND over the (clipped) window (core/bmaxpool.h:24-88).
// One thread per VEC consecutive output words (VEC = 4: 16-byte loads and stores, 8 lanes cover a
// 128-byte line of a 1024-channel pixel... or four 256-channel pixels' worth; VEC = 1: any word count).
// Windows that do not overlap (filter <= stride) read every input word exactly once: streaming loads.
// ---------------------------------------------------------------------------------
template <int VEC>
LCE_KERNEL void __launch_bounds__(256)
bmaxpool_words(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int B, int H, int W,
               int C, int OH, int OW, int FH, int FW, int SH, int SW, int PH, int PW,
               uint64_t total, FastDiv div_c, FastDiv div_ow, FastDiv div_oh) {
  // C, total, div_c count VEC-word groups; the tensors are addressed in words
  const uint64_t stride = (uint64_t)grid_dim_x() * (uint64_t)block_dim_x();
  const bool small = total < (1ull << 31);   // the multiply-shift division is exact below 2^31
  const bool once = FH <= SH && FW <= SW;
  for (uint64_t e = (uint64_t)block_idx_x() * (uint64_t)block_dim_x() + (uint64_t)thread_idx_x(); e < total; e += stride) {
    int c, ox, oy, b;
    if (small) {
      uint32_t p = fastdiv((uint32_t)e, div_c);
      c = (int)((uint32_t)e - p * (uint32_t)C);
      uint32_t q = fastdiv(p, div_ow);
      ox = (int)(p - q * (uint32_t)OW);
      b = (int)fastdiv(q, div_oh);
      oy = (int)(q - (uint32_t)b * (uint32_t)OH);
    } else {
      c = (int)(e % (uint64_t)C);
      uint64_t p = e / (uint64_t)C;
      ox = (int)(p % (uint64_t)OW); p /= (uint64_t)OW;
      oy = (int)(p % (uint64_t)OH);
      b = (int)(p / (uint64_t)OH);
    }
    const int x0 = ox * SW - PW, y0 = oy * SH - PH;
    const int xs = x0 < 0 ? 0 : x0, ys = y0 < 0 ? 0 : y0;
    const int xe = x0 + FW < W ? x0 + FW : W, ye = y0 + FH < H ? y0 + FH : H;
    if constexpr (VEC == 4) {
      u32x4 m = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      for (int y = ys; y < ye; ++y)
        for (int x = xs; x < xe; ++x) {
          const u32x4* src = (const u32x4*)in + (((size_t)b * H + y) * W + x) * (size_t)C + c;
          const u32x4 v = once ? load_streaming(src) : *src;
          m[0] &= v[0]; m[1] &= v[1]; m[2] &= v[2]; m[3] &= v[3];
        }
      store_streaming((u32x4*)out + e, m);
    } else {
      uint32_t m = 0xffffffffu;
      for (int y = ys; y < ye; ++y)
        for (int x = xs; x < xe; ++x) m &= in[(((size_t)b * H + y) * W + x) * (size_t)C + c];
      out[e] = m;
    }
  }
  (void)B;
}

}  // namespace lce
