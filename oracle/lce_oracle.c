/*
 * lce_oracle.c -- CPU restatement (plain C99) of the LceBconv2d / LceQuantize hot
 * path of Larq Compute Engine.  TEST INFRASTRUCTURE ONLY -- see lce_oracle.h.
 *
 * Build: see oracle/Makefile (-O3 -ffp-contract=off: the reference semantics are
 * `float(x) * mul + bias` with TWO roundings, core/bconv2d/output_transform.h:99-106).
 *
 * Citations are relative to /root/reference/larq_compute_engine/.
 */
#include "lce_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define WORD_BITS 32 /* core/types.h:41-43: TBitpacked = int32, bitwidth 32 */

/* core/types.h:45-47 */
static inline int xor_popcount(int32_t a, int32_t b) {
  return __builtin_popcount((uint32_t)(a ^ b));
}

int lce_oracle_bitpacked_size(int n) { return (n + WORD_BITS - 1) / WORD_BITS; }

/* ------------------------------------------------------------------------- */
/* Bitpacking: bit i of a word is 1 iff element i "is negative", LSB first,   */
/* padding bits 0 (core/bitpacking/bitpack.h:72-110,151-191,209-245).         */
/* ------------------------------------------------------------------------- */

void lce_oracle_bitpack_f32(const float* in, size_t rows, size_t cols, int32_t* out) {
  /* float path: zero_point is 0 and the test is `x < 0` (bitpack.h:159-190), so
   * -0.0f and NaN give bit 0.  Rows are padded independently (bitpack.h:299-307);
   * the padding elements are set to zero_point (= 0) and therefore pack as 0. */
  const size_t wpr = (size_t)lce_oracle_bitpacked_size((int)cols);
  for (size_t r = 0; r < rows; ++r) {
    for (size_t w = 0; w < wpr; ++w) {
      uint32_t word = 0;
      for (size_t b = 0; b < WORD_BITS; ++b) {
        const size_t col = w * WORD_BITS + b;
        if (col < cols && in[r * cols + col] < 0) word |= (uint32_t)1 << b;
      }
      out[r * wpr + w] = (int32_t)word;
    }
  }
}

void lce_oracle_bitpack_i8(const int8_t* in, size_t rows, size_t cols, int32_t zero_point,
                           int32_t* out) {
  /* bitpack.h:259-288: a zero point outside the int8 range short-circuits.  Both
   * shortcuts agree with the generic rule `int32(x) < zero_point`, padding 0. */
  const size_t wpr = (size_t)lce_oracle_bitpacked_size((int)cols);
  for (size_t r = 0; r < rows; ++r) {
    for (size_t w = 0; w < wpr; ++w) {
      uint32_t word = 0;
      for (size_t b = 0; b < WORD_BITS; ++b) {
        const size_t col = w * WORD_BITS + b;
        if (col < cols && (int32_t)in[r * cols + col] < zero_point) word |= (uint32_t)1 << b;
      }
      out[r * wpr + w] = (int32_t)word;
    }
  }
}

void lce_oracle_bitpack_bool(const uint8_t* in, size_t rows, size_t cols, int32_t* out) {
  /* quantization.cc:86-108: bool is viewed as uint8 and packed with zero point 1,
   * so false (all-zero byte) -> bit 1, anything else -> bit 0.  Padding elements
   * are filled with the zero point (1) and so pack as 0 (bitpack.h:238-244). */
  const size_t wpr = (size_t)lce_oracle_bitpacked_size((int)cols);
  for (size_t r = 0; r < rows; ++r) {
    for (size_t w = 0; w < wpr; ++w) {
      uint32_t word = 0;
      for (size_t b = 0; b < WORD_BITS; ++b) {
        const size_t col = w * WORD_BITS + b;
        if (col < cols && in[r * cols + col] < 1) word |= (uint32_t)1 << b;
      }
      out[r * wpr + w] = (int32_t)word;
    }
  }
}

/* bitpack.h:310-346: bit 0 -> zero_bit_result (+1), bit 1 -> one_bit_result (-1) */
void lce_oracle_unpack_f32(const int32_t* in, size_t rows, size_t cols, float* out) {
  const size_t wpr = (size_t)lce_oracle_bitpacked_size((int)cols);
  for (size_t r = 0; r < rows; ++r)
    for (size_t c = 0; c < cols; ++c)
      out[r * cols + c] =
          (((uint32_t)in[r * wpr + c / WORD_BITS] >> (c % WORD_BITS)) & 1u) ? -1.0f : 1.0f;
}

void lce_oracle_unpack_i8(const int32_t* in, size_t rows, size_t cols, float scale,
                          int32_t zero_point, int8_t* out) {
  /* quantization.cc:131-138: offset = TfLiteRound(1/scale) (std::round, half
   * away from zero), results clamped to the int8 range with min/max. */
  const int offset = (int)roundf(1.0f / scale);
  int zero_bit = zero_point + offset;
  if (zero_bit > 127) zero_bit = 127;
  int one_bit = zero_point - offset;
  if (one_bit < -128) one_bit = -128;
  const size_t wpr = (size_t)lce_oracle_bitpacked_size((int)cols);
  for (size_t r = 0; r < rows; ++r)
    for (size_t c = 0; c < cols; ++c)
      out[r * cols + c] = (int8_t)(
          (((uint32_t)in[r * wpr + c / WORD_BITS] >> (c % WORD_BITS)) & 1u) ? one_bit : zero_bit);
}

void lce_oracle_unpack_bool(const int32_t* in, size_t rows, size_t cols, uint8_t* out) {
  /* quantization.cc:139-141: bit 0 -> true, bit 1 -> false */
  const size_t wpr = (size_t)lce_oracle_bitpacked_size((int)cols);
  for (size_t r = 0; r < rows; ++r)
    for (size_t c = 0; c < cols; ++c)
      out[r * cols + c] =
          (((uint32_t)in[r * wpr + c / WORD_BITS] >> (c % WORD_BITS)) & 1u) ? 0 : 1;
}

/* ------------------------------------------------------------------------- */
/* Shape inference.  TFLite's ComputePaddingHeightWidth (tensorflow v2.16.1,   */
/* tensorflow/lite/kernels/padding.h -- third-party, absent from the tree)     */
/* restated from its published definition; call site bconv2d.cc:203-210.       */
/* ------------------------------------------------------------------------- */

static int out_size(int padding, int image, int filter, int stride, int dilation) {
  const int eff = (filter - 1) * dilation + 1;
  if (stride == 0) return 0;
  if (padding == LCE_ORACLE_PADDING_SAME) return (image + stride - 1) / stride;
  if (padding == LCE_ORACLE_PADDING_VALID) return (image + stride - eff) / stride;
  return 0;
}

static int pad_before(int stride, int dilation, int in, int filter, int out, int* offset) {
  const int eff = (filter - 1) * dilation + 1;
  int total = (out - 1) * stride + eff - in;
  if (total < 0) total = 0;
  *offset = total % 2;
  return total / 2;
}

int lce_oracle_conv_prepare(lce_oracle_conv* c) {
  if (c->pad_values != 0 && c->pad_values != 1) return 1; /* bconv2d.cc:109-112 */
  if (c->groups < 1 || c->channels_in % c->groups || c->channels_out % c->groups) return 2;
  if (c->groups > 1 && (c->channels_in / c->groups) % WORD_BITS) return 3; /* :180-185 */
  c->out_h = out_size(c->padding, c->in_h, c->filter_h, c->stride_h, c->dilation_h);
  c->out_w = out_size(c->padding, c->in_w, c->filter_w, c->stride_w, c->dilation_w);
  c->pad_h = pad_before(c->stride_h, c->dilation_h, c->in_h, c->filter_h, c->out_h,
                        &c->pad_h_offset);
  c->pad_w = pad_before(c->stride_w, c->dilation_w, c->in_w, c->filter_w, c->out_w,
                        &c->pad_w_offset);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Output-transform folding, tflite/kernels/bconv2d.cc:324-392.               */
/* ------------------------------------------------------------------------- */

void lce_oracle_fold_output_transform(const lce_oracle_conv* c, int dst_type,
                                      const float* post_mul, const float* post_bias,
                                      float out_scale, int32_t out_zero_point,
                                      float* mul_out, float* bias_out,
                                      int32_t* clamp_min, int32_t* clamp_max) {
  const int32_t cig = c->channels_in / c->groups;
  const int32_t a = c->filter_h * c->filter_w * cig; /* backtransform_add, :361-362 */
  const double scale = dst_type == LCE_ORACLE_DST_I8 ? (double)out_scale : 1.0;
  const double zp = dst_type == LCE_ORACLE_DST_I8 ? (double)out_zero_point : 0.0;
  for (int i = 0; i < c->channels_out; ++i) {
    const double m = post_mul[i], b = post_bias[i];
    mul_out[i] = (float)(-1 * m / scale);                 /* :373 */
    bias_out[i] = (float)((b + (double)a * m) / scale + zp); /* :374-377 */
  }
  /* CalculateActivationRange<int32> (TFLite kernel_util.h), then :380-388 */
  int32_t lo = INT32_MIN, hi = INT32_MAX;
  if (c->activation == LCE_ORACLE_ACT_RELU) { lo = 0; hi = INT32_MAX; }
  else if (c->activation == LCE_ORACLE_ACT_RELU6) { lo = 0; hi = 6; }
  else if (c->activation == LCE_ORACLE_ACT_RELU_N1_TO_1) { lo = -1; hi = 1; }
  if (lo < -a) lo = -a;
  if (hi > a) hi = a;
  *clamp_min = -hi + a;
  *clamp_max = -lo + a;
}

/* mlir/transforms/optimize.cc:128-186 (float arithmetic, as in the converter) */
void lce_oracle_thresholds_converter(int32_t backtransform_add, int activation,
                                     const float* post_mul, const float* post_bias,
                                     int n, int32_t* thr) {
  float cmin = -(float)backtransform_add, cmax = (float)backtransform_add; /* :221-229 */
  if (activation == LCE_ORACLE_ACT_RELU) { cmin = 0; cmax = (float)backtransform_add; }
  else if (activation == LCE_ORACLE_ACT_RELU_N1_TO_1) { cmin = -1; cmax = 1; }
  else if (activation == LCE_ORACLE_ACT_RELU6) { cmin = 0; cmax = 6; }
  for (int i = 0; i < n; ++i) {
    const float mult = post_mul[i], bias = post_bias[i];
    if (mult == 0.0f) { thr[i] = bias < 0.0f ? INT32_MIN : INT32_MAX; continue; }
    float emin, emax;
    if (mult > 0.0f) { emin = cmin; emax = cmax; } else { emin = -1 * cmax; emax = -1 * cmin; }
    const float start = emin * fabsf(mult) + bias;
    const float end = emax * fabsf(mult) + bias;
    if (start < 0 && end < 0) { thr[i] = INT32_MIN; continue; }
    if (start >= 0 && end >= 0) { thr[i] = INT32_MAX; continue; }
    /* `0.5 * (float expr)` promotes to double, then std::floor, then the
     * IntegerAttr stores the value as a 32-bit integer (:183-185). */
    thr[i] = (int32_t)floor(0.5 * (bias / fabsf(mult) + (float)backtransform_add));
  }
}

/* tflite/tests/bconv2d_test.cc:327-368 (double arithmetic, truncating cast) */
void lce_oracle_thresholds_optest(int32_t backtransform_add_i, int activation,
                                  const float* post_mul, const float* post_bias,
                                  int n, int32_t* thr) {
  int32_t amin = INT32_MIN, amax = INT32_MAX;
  if (activation == LCE_ORACLE_ACT_RELU) amin = 0;
  const double a = backtransform_add_i;
  for (int i = 0; i < n; ++i) {
    const double m = post_mul[i], b = post_bias[i];
    const double t1 = -b / m;
    const double t2 = 0.5 * (a + b / m);
    thr[i] = (int32_t)t2;
    if (t2 >= 2 * a || t1 <= amin) thr[i] = INT32_MAX;
    else if (t2 <= 0.0 || t1 >= amax) thr[i] = INT32_MIN;
  }
}

/* ------------------------------------------------------------------------- */
/* The convolution: core/bconv2d/reference.h:33-148.                          */
/* ------------------------------------------------------------------------- */

static inline int32_t accum_one(const lce_oracle_conv* c, const int32_t* input,
                                const int32_t* filter, int b, int oy, int ox, int oc,
                                int exact_zero_padding) {
  const int cw_total = lce_oracle_bitpacked_size(c->channels_in);
  const int cwg = lce_oracle_bitpacked_size(c->channels_in / c->groups);
  const int ocg = c->channels_out / c->groups;
  const int group = oc / ocg;
  const int bzp = (c->channels_in / c->groups) / 2; /* reference.h:76-77 */
  const int y0 = oy * c->stride_h - c->pad_h, x0 = ox * c->stride_w - c->pad_w;
  int32_t acc = 0;
  for (int fy = 0; fy < c->filter_h; ++fy) {
    for (int fx = 0; fx < c->filter_w; ++fx) {
      const int iy = y0 + c->dilation_h * fy, ix = x0 + c->dilation_w * fx;
      const int inside = iy >= 0 && iy < c->in_h && ix >= 0 && ix < c->in_w;
      const int32_t* w = filter + (((size_t)oc * c->filter_h + fy) * c->filter_w + fx) * cwg;
      if (!inside) {
        if (exact_zero_padding) { acc += bzp; continue; } /* reference.h:100-103 */
        for (int k = 0; k < cwg; ++k) acc += xor_popcount(0, w[k]); /* word 0 = +1, :105-106 */
        continue;
      }
      const int32_t* a =
          input + (((size_t)b * c->in_h + iy) * c->in_w + ix) * cw_total + (size_t)group * cwg;
      for (int k = 0; k < cwg; ++k) acc += xor_popcount(a[k], w[k]);
    }
  }
  return acc;
}

/* core/bconv2d/output_transform.h:93-107 */
static inline float ot_float(int32_t acc, int32_t cmin, int32_t cmax, float mul, float bias) {
  int32_t x = acc << 1;
  if (x > cmax) x = cmax;
  if (x < cmin) x = cmin;
  const float prod = (float)x * mul; /* separate rounding: built with -ffp-contract=off */
  return prod + bias;
}

/* output_transform.h:17-27,31-44,125-144: std::round (half away), then saturate */
static inline int8_t ot_int8(int32_t acc, int32_t cmin, int32_t cmax, float mul, float bias) {
  const float y = ot_float(acc, cmin, cmax, mul, bias);
  int32_t r = (int32_t)roundf(y);
  if (r > 127) r = 127;
  if (r < -128) r = -128;
  return (int8_t)r;
}

static int exact_zero_pad(const lce_oracle_conv* c) {
  return c->padding == LCE_ORACLE_PADDING_SAME && c->pad_values == 0 &&
         c->semantics == LCE_ORACLE_SEM_REFERENCE;
}

void lce_oracle_bconv2d_accum(const lce_oracle_conv* c, const int32_t* input,
                              const int32_t* filter, int32_t* out, int num_threads) {
  const int ez = exact_zero_pad(c);
  const long rows = (long)c->batch * c->out_h;
#pragma omp parallel for schedule(static) num_threads(num_threads > 1 ? num_threads : 1)
  for (long r = 0; r < rows; ++r) {
    const int b = (int)(r / c->out_h), oy = (int)(r % c->out_h);
    for (int ox = 0; ox < c->out_w; ++ox)
      for (int oc = 0; oc < c->channels_out; ++oc)
        out[(((size_t)b * c->out_h + oy) * c->out_w + ox) * c->channels_out + oc] =
            accum_one(c, input, filter, b, oy, ox, oc, ez);
  }
}

void lce_oracle_bconv2d_f32(const lce_oracle_conv* c, const int32_t* input,
                            const int32_t* filter, const float* mul, const float* bias,
                            int32_t cmin, int32_t cmax, const float* zero_pad_cache,
                            float* out, int num_threads) {
  const int ez = exact_zero_pad(c);
  const long rows = (long)c->batch * c->out_h;
#pragma omp parallel for schedule(static) num_threads(num_threads > 1 ? num_threads : 1)
  for (long r = 0; r < rows; ++r) {
    const int b = (int)(r / c->out_h), oy = (int)(r % c->out_h);
    for (int ox = 0; ox < c->out_w; ++ox)
      for (int oc = 0; oc < c->channels_out; ++oc)
        out[(((size_t)b * c->out_h + oy) * c->out_w + ox) * c->channels_out + oc] =
            ot_float(accum_one(c, input, filter, b, oy, ox, oc, ez), cmin, cmax, mul[oc], bias[oc]);
  }
  /* optimized kernels: one-padded conv first, float correction afterwards
   * (optimized_bgemm.h:153-177, optimized_indirect_bgemm.h:35-61) */
  if (c->padding == LCE_ORACLE_PADDING_SAME && c->pad_values == 0 &&
      c->semantics == LCE_ORACLE_SEM_OPTIMIZED && zero_pad_cache)
    lce_oracle_zero_pad_apply(c, zero_pad_cache, out);
}

void lce_oracle_bconv2d_i8(const lce_oracle_conv* c, const int32_t* input,
                           const int32_t* filter, const float* mul, const float* bias,
                           int32_t cmin, int32_t cmax, int8_t* out, int num_threads) {
  const int ez = exact_zero_pad(c);
  const long rows = (long)c->batch * c->out_h;
#pragma omp parallel for schedule(static) num_threads(num_threads > 1 ? num_threads : 1)
  for (long r = 0; r < rows; ++r) {
    const int b = (int)(r / c->out_h), oy = (int)(r % c->out_h);
    for (int ox = 0; ox < c->out_w; ++ox)
      for (int oc = 0; oc < c->channels_out; ++oc)
        out[(((size_t)b * c->out_h + oy) * c->out_w + ox) * c->channels_out + oc] =
            ot_int8(accum_one(c, input, filter, b, oy, ox, oc, ez), cmin, cmax, mul[oc], bias[oc]);
  }
}

void lce_oracle_bconv2d_bitpacked(const lce_oracle_conv* c, const int32_t* input,
                                  const int32_t* filter, const int32_t* thresholds,
                                  int32_t* out, int num_threads) {
  /* reference.h:120-136: channel oc -> word oc/32, bit oc%32; a word is flushed
   * when full or at the last channel, so padding bits are 0.
   * output_transform.h:160-168: bit = accum > threshold. */
  const int ez = exact_zero_pad(c);
  const int wout = lce_oracle_bitpacked_size(c->channels_out);
  const long rows = (long)c->batch * c->out_h;
#pragma omp parallel for schedule(static) num_threads(num_threads > 1 ? num_threads : 1)
  for (long r = 0; r < rows; ++r) {
    const int b = (int)(r / c->out_h), oy = (int)(r % c->out_h);
    for (int ox = 0; ox < c->out_w; ++ox) {
      int32_t* o = out + (((size_t)b * c->out_h + oy) * c->out_w + ox) * wout;
      for (int w = 0; w < wout; ++w) {
        uint32_t word = 0;
        for (int bit = 0; bit < WORD_BITS && w * WORD_BITS + bit < c->channels_out; ++bit) {
          const int oc = w * WORD_BITS + bit;
          if (accum_one(c, input, filter, b, oy, ox, oc, ez) > thresholds[oc])
            word |= (uint32_t)1 << bit;
        }
        o[w] = (int32_t)word;
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* SAME-zero padding correction, core/bconv2d/zero_padding_correction.h.      */
/* ------------------------------------------------------------------------- */

size_t lce_oracle_zero_pad_cache_size(const lce_oracle_conv* c) {
  const int ew = (c->filter_w - 1) * c->dilation_w + 1; /* :33-36 */
  const int eh = (c->filter_h - 1) * c->dilation_h + 1;
  return (size_t)4 * eh * ew * c->channels_out;
}

void lce_oracle_zero_pad_cache_fill(const lce_oracle_conv* c, const int32_t* filter,
                                    const float* post_mul, float* cache) {
  /* :39-176.  cache[case][y][x][oc] = -post_mul[oc] * sum over the filter taps
   * that stick out for overflow amounts (x, y) of (Cin_g - 2*popcount(tap)). */
  const int ew = (c->filter_w - 1) * c->dilation_w + 1;
  const int eh = (c->filter_h - 1) * c->dilation_h + 1;
  const int cin_g = c->channels_in / c->groups;
  const int cwg = lce_oracle_bitpacked_size(cin_g);
  const int n = c->channels_out;
  for (int y = 0; y < eh; ++y)
    for (int x = 0; x < ew; ++x)
      for (int oc = 0; oc < n; ++oc) {
        float corr[4] = {0, 0, 0, 0};
        for (int fy = 0; fy < c->filter_h; ++fy)
          for (int fx = 0; fx < c->filter_w; ++fx) {
            int pop = 0;
            const int32_t* w =
                filter + (((size_t)oc * c->filter_h + fy) * c->filter_w + fx) * cwg;
            for (int k = 0; k < cwg; ++k) pop += xor_popcount(w[k], 0);
            const float cur = (float)(cin_g - 2 * pop);
            const int efx = c->dilation_w * fx, efy = c->dilation_h * fy;
            if (efy < y || efx < x) corr[0] += cur;                   /* top-left     */
            if (efy < y || (ew - efx) <= x) corr[1] += cur;           /* top-right    */
            if ((eh - efy) <= y || efx < x) corr[2] += cur;           /* bottom-left  */
            if ((eh - efy) <= y || (ew - efx) <= x) corr[3] += cur;   /* bottom-right */
          }
        for (int d = 0; d < 4; ++d) {
          const float m = -1.0f * post_mul[oc];
          cache[(((size_t)d * eh + y) * ew + x) * n + oc] = m * corr[d];
        }
      }
}

void lce_oracle_zero_pad_apply(const lce_oracle_conv* c, const float* cache, float* out) {
  /* :178-297 evaluated per output pixel.  The reference skips the interior of
   * each row with a jump (:215-238); evaluating the same predicate per pixel is
   * equivalent because overflow_left/right are monotone in out_x. */
  const int ew = (c->filter_w - 1) * c->dilation_w + 1;
  const int eh = (c->filter_h - 1) * c->dilation_h + 1;
  const int n = c->channels_out;
  const int left_off = ((c->out_w - 1) * c->stride_w + ew - c->in_w) / 2;
  const int top_off = ((c->out_h - 1) * c->stride_h + eh - c->in_h) / 2;
  for (int b = 0; b < c->batch; ++b)
    for (int oy = 0; oy < c->out_h; ++oy) {
      const int o_top = top_off - oy * c->stride_h;
      const int o_bot = -o_top - c->in_h + eh;
      for (int ox = 0; ox < c->out_w; ++ox) {
        const int o_left = left_off - ox * c->stride_w;
        const int o_right = -o_left - c->in_w + ew;
        if (o_left <= 0 && o_right <= 0 && o_top <= 0 && o_bot <= 0) continue;
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
          continue; /* "This cannot happen" (:268-271) -- kept as in the reference */
        }
        const float* src = cache + (((size_t)kase * eh + cy) * ew + cx) * n;
        float* dst = out + (((size_t)b * c->out_h + oy) * c->out_w + ox) * n;
        for (int oc = 0; oc < n; ++oc) dst[oc] += src[oc];
      }
    }
}

/* ------------------------------------------------------------------------- */
/* LceBMaxPool2d, core/bmaxpool.h:24-88 ("next" row n1 of SURVEY section 8f). */
/* ------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------------------
 * The reference's OTHER portable kernel for the same convolution: the indirect BGEMM
 * (core/indirect_bgemm/kernel.h:16-186 + kernel_4x2_portable.h:22-158).  Restated as the
 * reference structures it -- weights repacked once into blocks of 4 output channels
 * ([group][block of 4][tap][word][4], kernel.h:54-94), an indirection table with one entry
 * per (pair of output pixels, tap, pixel) that points at the pixel's input words or at a
 * shared all-zero row for taps outside the image (= +1 padding, kernel.h:101-174), and a
 * 4-channel x 2-pixel micro-kernel with eight int32 accumulators that walks taps, then
 * words (kernel_4x2_portable.h:84-111) -- so that bench.py can time BOTH portable CPU
 * formulations.  Float / int8 output; VALID or one-padding
 * (SAME-zero is this kernel + lce_oracle_zero_pad_apply in the reference too,
 * optimized_indirect_bgemm.h:35-61).  Results are bit-identical to lce_oracle_bconv2d_*.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t* packed;       /* [G][ceil(Npg/4)][taps][Cwg][4], missing channels of a last block stay 0 */
  int64_t* indirect;     /* [ceil(M/2)][taps][2]: word offset of the pixel's row in `input`, or -1 */
  int64_t pixels;
} indirect_plan;

static void indirect_pack_weights(const lce_oracle_conv* c, const int32_t* filter, int32_t* packed) {
  const int taps = c->filter_h * c->filter_w, g_count = c->groups;
  const int cwg = lce_oracle_bitpacked_size(c->channels_in / g_count), npg = c->channels_out / g_count;
  const int blocks = (npg + 3) / 4;
  memset(packed, 0, sizeof(int32_t) * (size_t)g_count * blocks * taps * cwg * 4);
  for (int g = 0; g < g_count; ++g)
    for (int blk = 0; blk < blocks; ++blk)
      for (int t = 0; t < taps; ++t)
        for (int w = 0; w < cwg; ++w)
          for (int k = 0; k < 4 && blk * 4 + k < npg; ++k) {
            const int oc = g * npg + blk * 4 + k;
            packed[((((size_t)g * blocks + blk) * taps + t) * cwg + w) * 4 + k] =
                filter[((size_t)oc * taps + t) * cwg + w];
          }
}

static void indirect_fill_table(const lce_oracle_conv* c, int64_t* table) {
  const int taps = c->filter_h * c->filter_w, cw = lce_oracle_bitpacked_size(c->channels_in);
  const int64_t m_total = (int64_t)c->batch * c->out_h * c->out_w, pairs = (m_total + 1) / 2;
  for (int64_t pr = 0; pr < pairs; ++pr)
    for (int k = 0; k < 2; ++k) {
      int64_t m = 2 * pr + k;
      if (m >= m_total) m = m_total - 1;                       /* the odd tail re-reads the last pixel (kernel.h:133-134) */
      const int64_t b = m / ((int64_t)c->out_h * c->out_w);
      const int oy = (int)((m / c->out_w) % c->out_h), ox = (int)(m % c->out_w);
      for (int fy = 0; fy < c->filter_h; ++fy)
        for (int fx = 0; fx < c->filter_w; ++fx) {
          const int iy = oy * c->stride_h + fy * c->dilation_h - c->pad_h;
          const int ix = ox * c->stride_w + fx * c->dilation_w - c->pad_w;
          const int inside = iy >= 0 && iy < c->in_h && ix >= 0 && ix < c->in_w;
          table[(pr * taps + fy * c->filter_w + fx) * 2 + k] =
              inside ? ((b * c->in_h + iy) * c->in_w + ix) * cw : -1;
        }
    }
}

static void indirect_run(const lce_oracle_conv* c, const int32_t* input, const int32_t* packed,
                         const int64_t* table, int dst_type, const float* mul, const float* bias,
                         int32_t cmin, int32_t cmax, void* out, int num_threads) {
  const int taps = c->filter_h * c->filter_w, g_count = c->groups;
  const int cw = lce_oracle_bitpacked_size(c->channels_in);
  const int cwg = lce_oracle_bitpacked_size(c->channels_in / g_count), npg = c->channels_out / g_count;
  const int blocks = (npg + 3) / 4, n = c->channels_out;
  const int64_t m_total = (int64_t)c->batch * c->out_h * c->out_w, pairs = (m_total + 1) / 2;
  int32_t* zero_row = (int32_t*)calloc((size_t)cw, sizeof(int32_t));
  (void)num_threads;
#pragma omp parallel for schedule(static) num_threads(num_threads > 1 ? num_threads : 1)
  for (int64_t pr = 0; pr < pairs; ++pr) {
    const int64_t m0 = 2 * pr, m1 = m0 + 1 < m_total ? m0 + 1 : m0;   /* a lone last pixel is computed twice */
    const int64_t* tab = table + pr * taps * 2;
    for (int g = 0; g < g_count; ++g)
      for (int blk = 0; blk < blocks; ++blk) {
        int32_t acc[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        const int32_t* wp = packed + (((size_t)g * blocks + blk) * taps) * cwg * 4;
        for (int t = 0; t < taps; ++t) {
          const int32_t* a0 = (tab[t * 2] < 0 ? zero_row : input + tab[t * 2]) + (size_t)g * cwg;
          const int32_t* a1 = (tab[t * 2 + 1] < 0 ? zero_row : input + tab[t * 2 + 1]) + (size_t)g * cwg;
          for (int w = 0; w < cwg; ++w, wp += 4)
            for (int k = 0; k < 4; ++k) {
              acc[k][0] += xor_popcount(wp[k], a0[w]);
              acc[k][1] += xor_popcount(wp[k], a1[w]);
            }
        }
        for (int k = 0; k < 4 && blk * 4 + k < npg; ++k) {
          const int oc = g * npg + blk * 4 + k;
          if (dst_type == LCE_ORACLE_DST_F32) {
            ((float*)out)[m1 * n + oc] = ot_float(acc[k][1], cmin, cmax, mul[oc], bias[oc]);
            ((float*)out)[m0 * n + oc] = ot_float(acc[k][0], cmin, cmax, mul[oc], bias[oc]);
          } else {
            ((int8_t*)out)[m1 * n + oc] = ot_int8(acc[k][1], cmin, cmax, mul[oc], bias[oc]);
            ((int8_t*)out)[m0 * n + oc] = ot_int8(acc[k][0], cmin, cmax, mul[oc], bias[oc]);
          }
        }
      }
  }
  free(zero_row);
}

int lce_oracle_bconv2d_indirect(const lce_oracle_conv* c, const int32_t* input, const int32_t* filter,
                                int dst_type, const float* mul, const float* bias, int32_t clamp_min,
                                int32_t clamp_max, void* out, int num_threads) {
  if (dst_type != LCE_ORACLE_DST_F32 && dst_type != LCE_ORACLE_DST_I8) return 1;
  if (c->padding == LCE_ORACLE_PADDING_SAME && c->pad_values == 0) return 2;   /* caller adds the correction */
  const int taps = c->filter_h * c->filter_w;
  const int cwg = lce_oracle_bitpacked_size(c->channels_in / c->groups), npg = c->channels_out / c->groups;
  const int64_t m_total = (int64_t)c->batch * c->out_h * c->out_w, pairs = (m_total + 1) / 2;
  if (m_total == 0) return 0;
  int32_t* packed = (int32_t*)malloc(sizeof(int32_t) * (size_t)c->groups * ((npg + 3) / 4) * taps * cwg * 4);
  int64_t* table = (int64_t*)malloc(sizeof(int64_t) * (size_t)pairs * taps * 2);
  if (!packed || !table) { free(packed); free(table); return 3; }
  indirect_pack_weights(c, filter, packed);     /* once per layer in the reference (first Eval) */
  indirect_fill_table(c, table);
  indirect_run(c, input, packed, table, dst_type, mul, bias, clamp_min, clamp_max, out, num_threads);
  free(packed);
  free(table);
  return 0;
}


void lce_oracle_bmaxpool(int32_t batch, int32_t in_h, int32_t in_w, int32_t words,
                         int32_t filter_h, int32_t filter_w, int32_t stride_h,
                         int32_t stride_w, int32_t padding, const int32_t* in,
                         int32_t* out_h_p, int32_t* out_w_p, int32_t* out) {
  int off;
  const int oh = out_size(padding, in_h, filter_h, stride_h, 1);
  const int ow = out_size(padding, in_w, filter_w, stride_w, 1);
  const int ph = pad_before(stride_h, 1, in_h, filter_h, oh, &off);
  const int pw = pad_before(stride_w, 1, in_w, filter_w, ow, &off);
  *out_h_p = oh;
  *out_w_p = ow;
  if (!out) return;
  for (int b = 0; b < batch; ++b)
    for (int oy = 0; oy < oh; ++oy)
      for (int ox = 0; ox < ow; ++ox) {
        const int x0 = ox * stride_w - pw, y0 = oy * stride_h - ph;
        const int fxs = x0 < 0 ? -x0 : 0, fys = y0 < 0 ? -y0 : 0;
        const int ix = x0 + fxs, iy = y0 + fys;
        int fxc = filter_w - fxs; if (in_w - ix < fxc) fxc = in_w - ix;
        int fyc = filter_h - fys; if (in_h - iy < fyc) fyc = in_h - iy;
        for (int ch = 0; ch < words; ++ch) {
          uint32_t m = ~(uint32_t)0;
          for (int y = 0; y < fyc; ++y)
            for (int x = 0; x < fxc; ++x)
              m &= (uint32_t)in[(((size_t)b * in_h + iy + y) * in_w + ix + x) * words + ch];
          out[(((size_t)b * oh + oy) * ow + ox) * words + ch] = (int32_t)m;
        }
      }
}
