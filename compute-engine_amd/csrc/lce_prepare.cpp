// Converter-side parameter preparation -- host-only C++.
//
// The reference does this work offline in its MLIR converter; a deployment that starts from
// an unconverted Larq model (float +-scale filters in HWIO, batch-norm already folded into
// per-channel mul/add constants by TensorFlow) needs the same arithmetic to arrive at exactly
// the tensors an LCE-converted .tflite would carry into LceBconv2d.  Each function restates
// one rewrite; citations are relative to /root/reference/larq_compute_engine/mlir/.
//
// Nothing here touches activations: it prepares WEIGHTS and per-channel constants once, on
// the host, like lce_hip_bconv2d_plan_set_weights does.
#include "lce_prepare.h"

#include <cmath>
#include <cstdint>
#include <limits>

namespace lce {

// transforms/prepare_tf.cc:66-92 (IsBinaryFilter), :50-64 (GetScaleVector), :40-48
// (GetConstantVector) and transforms/prepare_patterns_common.td:97-126: the Conv2D filter
// (HWIO) must be +-scale[o] within 0.5 %; the op receives filter / scale transposed to OHWI,
// post_activation_multiplier = scale, post_activation_bias = 0.
std::string prepare_binary_filter(const float* hwio, int kh, int kw, int cin, int cout,
                                  float* ohwi, float* mul, float* bias) {
  if (!hwio || !ohwi || kh <= 0 || kw <= 0 || cin <= 0 || cout <= 0) return "prepare_binary_filter: bad argument";
  for (int o = 0; o < cout; ++o) {
    const float scale = hwio[o];  // element {0, 0, 0, o}
    if (std::fabs(scale) <= std::numeric_limits<float>::epsilon())
      return "prepare_binary_filter: not a binary filter (zero scale)";
  }
  for (int h = 0; h < kh; ++h)
    for (int w = 0; w < kw; ++w)
      for (int i = 0; i < cin; ++i)
        for (int o = 0; o < cout; ++o) {
          const float scale = hwio[o];
          const float value = hwio[((size_t)(h * kw + w) * cin + i) * cout + o];
          if (std::fabs(std::fabs(value / scale) - 1.0f) > 0.005f)
            return "prepare_binary_filter: not a binary filter (|value / scale| is not 1)";
          // TF_DivOp(filter, |scale|) then TF_TransposeOp {3, 0, 1, 2}
          ohwi[((size_t)(o * kh + h) * kw + w) * cin + i] = value / std::fabs(scale);
        }
  for (int o = 0; o < cout; ++o) {
    if (mul) mul[o] = std::fabs(hwio[o]);
    if (bias) bias[o] = 0.0f;
  }
  return "";
}

// transforms/optimize_patterns_common.td:39-118: a constant Add/Sub after the convolution
// goes into post_activation_bias; a constant Mul/Div scales BOTH multiplier and bias.
std::string fuse_post_op(int op, const float* value, int value_count, float* mul, float* bias, int n) {
  if (!value || !mul || !bias || n <= 0 || (value_count != 1 && value_count != n))
    return "prepare_fuse_post_op: the constant must be a scalar or have one entry per output channel";
  for (int c = 0; c < n; ++c) {
    const float v = value[value_count == 1 ? 0 : c];
    switch (op) {
      case 0: bias[c] = bias[c] + v; break;
      case 1: bias[c] = bias[c] - v; break;
      case 2: mul[c] = mul[c] * v; bias[c] = bias[c] * v; break;
      case 3: mul[c] = mul[c] / v; bias[c] = bias[c] / v; break;
      default: return "prepare_fuse_post_op: unknown operation";
    }
  }
  return "";
}

// transforms/optimize_patterns_common.td:122-182: Relu/Relu1/Relu6 after the convolution
// become its fused activation only while multiplier == 1 and bias == 0 (nothing else fused
// yet), and only for VALID padding or SAME with pad_values 1.
bool can_fuse_activation(const float* mul, const float* bias, int n, int padding_same, int pad_values) {
  if (padding_same && pad_values != 1) return false;
  for (int c = 0; c < n; ++c)
    if (mul[c] != 1.0f || bias[c] != 0.0f) return false;
  return true;
}

// transforms/optimize.cc:128-186 (ComputeWriteBitpackedOutputThresholds), :188-204
// (GetSignsOfVectorAndBroadcast4D), :206-244 (GetBitpackedOutputThresholds) and
// transforms/bitpack_activations_patterns.td:19-60: LceQuantize(LceBconv2d(x)) becomes a
// convolution that writes bits itself: filter *= sign(multiplier) and per-channel int32
// thresholds with  bit = (accumulator > threshold).
std::string prepare_bitpacked_output(float* ohwi, int kh, int kw, int cin, int cout, int activation,
                                     int padding_same, int pad_values, const float* mul,
                                     const float* bias, int32_t* thresholds) {
  if (!ohwi || !mul || !bias || !thresholds || kh <= 0 || kw <= 0 || cin <= 0 || cout <= 0)
    return "prepare_bitpacked_output: bad argument";
  // the two instantiations of WriteBitpackedActivationsPat: (VALID, pad_values 0), (SAME, 1)
  if ((padding_same && pad_values != 1) || (!padding_same && pad_values != 0))
    return "prepare_bitpacked_output: the converter only rewrites VALID/pad_values=0 and SAME/pad_values=1";
  const int32_t a = kh * kw * cin;  // filter_shape[1] * [2] * [3]
  float cmin = -(float)a, cmax = (float)a;
  switch (activation) {
    case LCE_HIP_ACT_RELU: cmin = 0.0f; cmax = (float)a; break;
    case LCE_HIP_ACT_RELU_N1_TO_1: cmin = -1.0f; cmax = 1.0f; break;
    case LCE_HIP_ACT_RELU6: cmin = 0.0f; cmax = 6.0f; break;
    default: break;
  }
  constexpr int32_t neg_inf = std::numeric_limits<int32_t>::min();
  constexpr int32_t pos_inf = std::numeric_limits<int32_t>::max();
  for (int o = 0; o < cout; ++o) {
    const float m = mul[o], b = bias[o];
    const float sign = m >= 0.0f ? 1.0f : -1.0f;
    float* f = ohwi + (size_t)o * kh * kw * cin;
    for (int k = 0; k < kh * kw * cin; ++k) f[k] = f[k] * sign;
    if (m == 0.0f) {
      thresholds[o] = b < 0.0f ? neg_inf : pos_inf;
      continue;
    }
    const float lo = m > 0.0f ? cmin : -1 * cmax, hi = m > 0.0f ? cmax : -1 * cmin;
    const float start = lo * std::abs(m) + b, end = hi * std::abs(m) + b;
    if (start < 0 && end < 0) { thresholds[o] = neg_inf; continue; }
    if (start >= 0 && end >= 0) { thresholds[o] = pos_inf; continue; }
    // float quotient and sum, then double product and floor, stored as a 32-bit integer
    thresholds[o] = (int32_t)std::floor(0.5 * (b / std::abs(m) + (float)a));
  }
  return "";
}

// transforms/bitpack.cc:19-57 (constant-folded bitpacking of the float filter) ->
// core/bitpacking/bitpack.h:248-308 semantics: bit = (x < 0), LSB first, every
// (o, h, w) row padded with 0 bits to a whole word.
std::string bitpack_filter(const float* ohwi, int kh, int kw, int cin, int cout, int32_t* words) {
  if (!ohwi || !words || kh <= 0 || kw <= 0 || cin <= 0 || cout <= 0) return "prepare_bitpack_filter: bad argument";
  const int cw = (cin + 31) / 32;
  const size_t rows = (size_t)cout * kh * kw;
  for (size_t r = 0; r < rows; ++r)
    for (int w = 0; w < cw; ++w) {
      uint32_t bits = 0;
      for (int j = 0; j < 32 && w * 32 + j < cin; ++j)
        if (ohwi[r * cin + w * 32 + j] < 0.0f) bits |= 1u << j;   // -0.0 and NaN pack as 0
      words[r * cw + w] = (int32_t)bits;
    }
  return "";
}

}  // namespace lce
