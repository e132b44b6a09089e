/*
 * lce_oracle.h -- CPU restatement of Larq Compute Engine's LceBconv2d / LceQuantize
 * hot path in plain C99.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the CPU baseline.  The product path
 * (compute-engine_amd/) never links or calls it.
 *
 * Parity pinning: the reference tree cannot be built in this image (every core
 * header includes TensorFlow Lite / Ruy / flatbuffers headers, which are an empty
 * un-vendored submodule: tensorflow v2.16.1, WORKSPACE:12-24), so this oracle is
 * pinned against the known-answer vectors the reference's own tests hold
 * (tests/golden/reference_kats.json; see DESIGN.md "Oracle pinning") and against
 * the same float-convolution property the reference's op tests use
 * (tflite/tests/bconv2d_test.cc:648-768).
 *
 * All file:line citations are relative to /root/reference/larq_compute_engine/.
 */
#ifndef LCE_ORACLE_H_
#define LCE_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enums follow the TFLite flatbuffer schema values that arrive in the op's
 * flexbuffer options (tflite/kernels/bconv2d.cc:94-124). */
enum { LCE_ORACLE_PADDING_SAME = 0, LCE_ORACLE_PADDING_VALID = 1 };
enum { LCE_ORACLE_ACT_NONE = 0, LCE_ORACLE_ACT_RELU = 1,
       LCE_ORACLE_ACT_RELU_N1_TO_1 = 2, LCE_ORACLE_ACT_RELU6 = 3 };
enum { LCE_ORACLE_DST_F32 = 0, LCE_ORACLE_DST_I8 = 1, LCE_ORACLE_DST_BITPACKED = 2 };
/* which reference kernel's SAME-zero-padding semantics to follow */
enum { LCE_ORACLE_SEM_REFERENCE = 0,   /* core/bconv2d/reference.h:76-103        */
       LCE_ORACLE_SEM_OPTIMIZED = 1 }; /* optimized_*bgemm.h + zero_padding_correction.h */

typedef struct lce_oracle_conv {
  int32_t batch, in_h, in_w, channels_in;
  int32_t filter_h, filter_w, channels_out, groups;
  int32_t stride_h, stride_w, dilation_h, dilation_w;
  int32_t padding;     /* LCE_ORACLE_PADDING_*  */
  int32_t pad_values;  /* 0 or 1                */
  int32_t activation;  /* LCE_ORACLE_ACT_*      */
  int32_t semantics;   /* LCE_ORACLE_SEM_*      */
  /* derived by lce_oracle_conv_prepare() */
  int32_t out_h, out_w, pad_h, pad_w, pad_h_offset, pad_w_offset;
} lce_oracle_conv;

/* core/bitpacking/bitpack.h:24-30 */
int lce_oracle_bitpacked_size(int unpacked_elements);

/* core/bitpacking/bitpack.h:248-308 (bitpack_matrix) for the three input
 * types LceQuantize accepts (tflite/kernels/quantization.cc:76-114). */
void lce_oracle_bitpack_f32(const float* in, size_t rows, size_t cols, int32_t* out);
void lce_oracle_bitpack_i8(const int8_t* in, size_t rows, size_t cols, int32_t zero_point,
                           int32_t* out);
void lce_oracle_bitpack_bool(const uint8_t* in, size_t rows, size_t cols, int32_t* out);

/* core/bitpacking/bitpack.h:310-346 (unpack_matrix) as used by LceDequantize
 * (tflite/kernels/quantization.cc:116-147). */
void lce_oracle_unpack_f32(const int32_t* in, size_t rows, size_t cols, float* out);
void lce_oracle_unpack_i8(const int32_t* in, size_t rows, size_t cols, float scale,
                          int32_t zero_point, int8_t* out);
void lce_oracle_unpack_bool(const int32_t* in, size_t rows, size_t cols, uint8_t* out);

/* Output size + padding: TFLite ComputePaddingHeightWidth as called from
 * tflite/kernels/bconv2d.cc:203-210.  Returns 0 on success. */
int lce_oracle_conv_prepare(lce_oracle_conv* c);

/* tflite/kernels/bconv2d.cc:324-392 (OneTimeSetup): fold the back-transform,
 * the int8 scale/zero-point and the fused activation into per-channel
 * multiplier/bias (computed in double, stored as float) and two int32 clamps. */
void lce_oracle_fold_output_transform(const lce_oracle_conv* c, int dst_type,
                                      const float* post_mul, const float* post_bias,
                                      float out_scale, int32_t out_zero_point,
                                      float* mul_out, float* bias_out,
                                      int32_t* clamp_min, int32_t* clamp_max);

/* Thresholds for bitpacked output as the converter writes them
 * (mlir/transforms/optimize.cc:128-186).  Assumes filters were already
 * multiplied by sign(post_mul). */
void lce_oracle_thresholds_converter(int32_t backtransform_add, int activation,
                                     const float* post_mul, const float* post_bias,
                                     int n, int32_t* thresholds);
/* The op test's own re-derivation (tflite/tests/bconv2d_test.cc:327-368). */
void lce_oracle_thresholds_optest(int32_t backtransform_add, int activation,
                                  const float* post_mul, const float* post_bias,
                                  int n, int32_t* thresholds);

/* The binary convolution itself.  `input` is [B,H,W,ceil(Cin/32)] words,
 * `filter` is [Cout,KH,KW,ceil(Cin/G/32)] words (OHWI).  For float/int8 `mul`,
 * `bias`, `clamp_*` are the FOLDED values; for bitpacked `thresholds` is used.
 * `zero_pad_cache` is only read for SEM_OPTIMIZED + SAME + pad_values 0 + f32.
 * num_threads <= 1 runs single-threaded (the reference kernels are,
 * reference.h:84, indirect_bgemm/kernel.h:180-183); > 1 splits output rows
 * over OpenMP threads for the CPU-baseline leg of bench.py. */
void lce_oracle_bconv2d_f32(const lce_oracle_conv* c, const int32_t* input,
                            const int32_t* filter, const float* mul, const float* bias,
                            int32_t clamp_min, int32_t clamp_max,
                            const float* zero_pad_cache, float* out, int num_threads);
void lce_oracle_bconv2d_i8(const lce_oracle_conv* c, const int32_t* input,
                           const int32_t* filter, const float* mul, const float* bias,
                           int32_t clamp_min, int32_t clamp_max, int8_t* out,
                           int num_threads);
void lce_oracle_bconv2d_bitpacked(const lce_oracle_conv* c, const int32_t* input,
                                  const int32_t* filter, const int32_t* thresholds,
                                  int32_t* out, int num_threads);
/* The same convolution by the reference's indirect-BGEMM formulation (core/indirect_bgemm/kernel.h:16-186,
 * kernel_4x2_portable.h:22-158): packed weight blocks of 4 channels, an indirection table, a 4x2 micro-kernel.
 * dst_type F32 or I8, VALID or one-padding; returns 0 on success.  Bit-identical to the functions above. */
int lce_oracle_bconv2d_indirect(const lce_oracle_conv* c, const int32_t* input, const int32_t* filter,
                                int dst_type, const float* mul, const float* bias, int32_t clamp_min,
                                int32_t clamp_max, void* out, int num_threads);
/* raw accumulators (sum of xor-popcounts, incl. zero-padding terms) for debugging */
void lce_oracle_bconv2d_accum(const lce_oracle_conv* c, const int32_t* input,
                              const int32_t* filter, int32_t* out, int num_threads);

/* core/bconv2d/zero_padding_correction.h:30-37,39-176 */
size_t lce_oracle_zero_pad_cache_size(const lce_oracle_conv* c);
void lce_oracle_zero_pad_cache_fill(const lce_oracle_conv* c, const int32_t* filter,
                                    const float* post_mul, float* cache);
/* zero_padding_correction.h:178-297, applied in place on a float output */
void lce_oracle_zero_pad_apply(const lce_oracle_conv* c, const float* cache, float* out);

/* core/bmaxpool.h:24-88 (bitwise AND over the window; LceBMaxPool2d). */
void lce_oracle_bmaxpool(int32_t batch, int32_t in_h, int32_t in_w, int32_t words,
                         int32_t filter_h, int32_t filter_w, int32_t stride_h,
                         int32_t stride_w, int32_t padding, const int32_t* in,
                         int32_t* out_h, int32_t* out_w, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* LCE_ORACLE_H_ */
