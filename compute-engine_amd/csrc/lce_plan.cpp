// Host-side planner for LceBconv2d on MI355X (see lce_plan.h): validation and shape inference, parameter folding, the int8 epilogue's
// proof, weight images, and the kernel selection.  The streaming kernels' launch planners: lce_plan_stream.cpp; the cost estimates the
// selection compares: lce_plan_cost.cpp.
// Citations are relative to /root/reference/larq_compute_engine/.
#include "lce_plan.h"
#include "lce_plan_internal.h"
#include <cstdio>
#include <cstdlib>

#include <limits.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdio>

namespace lce {

// TFLite ComputeOutSize / ComputePaddingWithOffset (tensorflow v2.16.1
// tensorflow/lite/kernels/padding.h; third-party, restated from its published
// definition -- the reference calls it at tflite/kernels/bconv2d.cc:203-210).
static int out_size(int padding, int image, int filter, int stride, int dilation) {
  const int eff = (filter - 1) * dilation + 1;
  if (stride == 0) return 0;
  if (padding == LCE_HIP_PADDING_SAME) return (image + stride - 1) / stride;
  if (padding == LCE_HIP_PADDING_VALID) return (image + stride - eff) / stride;
  return 0;
}
static int pad_before(int stride, int dilation, int in, int filter, int out) {
  const int eff = (filter - 1) * dilation + 1;
  const int total = std::max(0, (out - 1) * stride + eff - in);
  return total / 2;
}

// The A/B and debugging aids of the planner come from the environment ONCE per plan, here, where the plan is created -- the selection
// path (select_kernel, run at every batch-size change) reads fields, never the environment.
static void read_debug_environment(HostPlan& p) {
  const char* dbg = getenv("LCE_PLAN_DEBUG");
  p.dbg_level = dbg ? std::max(1, atoi(dbg)) : 0;
  p.dbg_no_wstream = getenv("LCE_PLAN_NO_WSTREAM") != nullptr;
  p.dbg_int8_exact = getenv("LCE_PLAN_INT8_EXACT") != nullptr;
  p.dbg_int8_full = getenv("LCE_PLAN_INT8_FULL") != nullptr;
  static bool table_printed = false;
  if (p.dbg_level >= 3 && !table_printed) { table_printed = true; dump_cost_table(stderr); }
}

std::string validate_and_infer(HostPlan& p) {
  const lce_hip_bconv2d_desc& d = p.d;
  char buf[256];
  read_debug_environment(p);
  // an EMPTY batch is legal (the reference's loops simply do not run, reference.h:84); every
  // other extent must be positive
  if (d.batch < 0 || d.in_height < 1 || d.in_width < 1 || d.channels_in < 1 ||
      d.filter_height < 1 || d.filter_width < 1 || d.channels_out < 1)
    return "bconv2d: all tensor dimensions must be positive";
  if (d.stride_height < 1 || d.stride_width < 1 || d.dilation_height < 1 || d.dilation_width < 1)
    return "bconv2d: strides and dilations must be positive";
  if (d.padding != LCE_HIP_PADDING_SAME && d.padding != LCE_HIP_PADDING_VALID)
    return "bconv2d: padding must be SAME or VALID";
  if (d.pad_values != 0 && d.pad_values != 1)
    return "Attribute pad_values must be 0 or 1.";  // bconv2d.cc:109-112
  if (d.activation < LCE_HIP_ACT_NONE || d.activation > LCE_HIP_ACT_RELU6)
    return "bconv2d: unsupported fused activation";
  if (d.dst_type != LCE_HIP_F32 && d.dst_type != LCE_HIP_I8 && d.dst_type != LCE_HIP_BITPACKED)
    return "Supported output types are int8, int32, and float32.";  // bconv2d.cc:158-162
  if (d.groups < 1) return "bconv2d: groups must be >= 1";
  if (d.groups > 1) {
    // bconv2d.cc:173-185
    if (d.channels_in % d.groups) return "bconv2d: channels_in must be divisible by groups";
    if ((d.channels_in / d.groups) % 32)
      return "bconv2d: group_size % core::bitpacking_bitwidth was not 0";
    if (d.channels_out % d.groups) return "bconv2d: channels_out % groups was not 0";
  }
  if (d.padding == LCE_HIP_PADDING_SAME && d.pad_values == 0) {
    // bconv2d.cc:188-200
    const bool ok = (d.semantics == LCE_HIP_SEM_REFERENCE && d.channels_in % 2 == 0) ||
                    (d.semantics != LCE_HIP_SEM_REFERENCE && d.dst_type == LCE_HIP_F32 &&
                     d.activation == LCE_HIP_ACT_NONE);
    if (!ok)
      return "Zero-padding is only supported by the reference kernel with an even number of "
             "input channels, or when using float output with no fused activation function.";
  }
  if (d.dst_type == LCE_HIP_I8 && !(d.out_scale > 0.0f))
    return "bconv2d: int8 output needs a positive output scale";

  p.out_h = out_size(d.padding, d.in_height, d.filter_height, d.stride_height, d.dilation_height);
  p.out_w = out_size(d.padding, d.in_width, d.filter_width, d.stride_width, d.dilation_width);
  if (p.out_h < 1 || p.out_w < 1) {
    snprintf(buf, sizeof buf, "bconv2d: empty output (%d x %d)", p.out_h, p.out_w);
    return buf;
  }
  p.pad_h = pad_before(d.stride_height, d.dilation_height, d.in_height, d.filter_height, p.out_h);
  p.pad_w = pad_before(d.stride_width, d.dilation_width, d.in_width, d.filter_width, p.out_w);
  p.cw = ceil_div(d.channels_in, 32);
  p.cwg = ceil_div(d.channels_in / d.groups, 32);
  p.npg = d.channels_out / d.groups;
  p.wout = ceil_div(d.channels_out, 32);
  p.backtransform_add = d.filter_height * d.filter_width * (d.channels_in / d.groups);
  p.zero_pad_mode = kZeroPadNone;
  if (d.padding == LCE_HIP_PADDING_SAME && d.pad_values == 0)
    p.zero_pad_mode = d.semantics == LCE_HIP_SEM_REFERENCE ? kZeroPadExact : kZeroPadCorrection;
  const int64_t per_image_bytes = (int64_t)d.in_height * d.in_width * p.cw * 4;
  if (per_image_bytes >= (1ll << 31)) return "bconv2d: one input image must be smaller than 2 GiB";
  if ((int64_t)p.out_h * p.out_w >= (1ll << 31)) return "bconv2d: output plane too large";
  return "";
}

static int popcount32(uint32_t x) { return __builtin_popcount(x); }

// core/bconv2d/zero_padding_correction.h:39-176
static void fill_zero_pad_cache(HostPlan& p, const float* post_mul) {
  const lce_hip_bconv2d_desc& d = p.d;
  const int ew = (d.filter_width - 1) * d.dilation_width + 1;
  const int eh = (d.filter_height - 1) * d.dilation_height + 1;
  const int cin_g = d.channels_in / d.groups;
  const int n = d.channels_out;
  p.zero_pad_cache.assign((size_t)4 * eh * ew * n, 0.0f);
  // value of every filter tap as a +-1 sum: Cin_g - 2*popcount (:124-125)
  std::vector<float> tap_val((size_t)n * d.filter_height * d.filter_width);
  for (int oc = 0; oc < n; ++oc)
    for (int t = 0; t < d.filter_height * d.filter_width; ++t) {
      int pop = 0;
      const uint32_t* w = &p.filter[((size_t)oc * d.filter_height * d.filter_width + t) * p.cwg];
      for (int k = 0; k < p.cwg; ++k) pop += popcount32(w[k]);
      tap_val[(size_t)oc * d.filter_height * d.filter_width + t] = (float)(cin_g - 2 * pop);
    }
  for (int y = 0; y < eh; ++y)
    for (int x = 0; x < ew; ++x)
      for (int oc = 0; oc < n; ++oc) {
        float corr[4] = {0.f, 0.f, 0.f, 0.f};
        for (int fy = 0; fy < d.filter_height; ++fy)
          for (int fx = 0; fx < d.filter_width; ++fx) {
            const float cur = tap_val[((size_t)oc * d.filter_height + fy) * d.filter_width + fx];
            const int efx = d.dilation_width * fx, efy = d.dilation_height * fy;
            const bool top = efy < y, bot = (eh - efy) <= y;
            const bool left = efx < x, right = (ew - efx) <= x;
            if (top || left) corr[0] += cur;
            if (top || right) corr[1] += cur;
            if (bot || left) corr[2] += cur;
            if (bot || right) corr[3] += cur;
          }
        const float m = -1.0f * post_mul[oc];
        for (int k = 0; k < 4; ++k)
          p.zero_pad_cache[(((size_t)k * eh + y) * ew + x) * n + oc] = m * corr[k];
      }
}

// LceQuantize of an int8 tensor sets bit = (q < zero_point) (quantization.cc:76-114, bitpack.h:72-110).  The kernels
// hold the value BEFORE the int8 rounding, q = round_sat_i8(c) = trunc(c + copysign(pred(0.5), c)) after saturation, which
// is monotone in c, so "q < zero_point" is "c < T" for the smallest float T whose rounding reaches zero_point.  Found by
// bisection over the ordered float bit patterns with the very expression the kernels evaluate (no case analysis of ties).
static int round_sat_i8_host(float c) {
  c = c < -128.0f ? -128.0f : (c > 127.0f ? 127.0f : c);
  volatile float s = c + std::copysign(0x1.fffffep-2f, c);
  return (int)s;
}
float int8_below_threshold(int32_t zero_point) {
  if (zero_point <= -128) return -std::numeric_limits<float>::infinity();   // no int8 is below it
  if (zero_point > 127) return std::numeric_limits<float>::infinity();      // every int8 is
  // monotone key over floats: negative floats descend with their bit pattern
  auto from_key = [](int64_t k) {
    uint32_t u = k >= 0 ? (uint32_t)k : 0x80000000u | (uint32_t)(-k - 1);
    float f;
    memcpy(&f, &u, 4);
    return f;
  };
  auto to_key = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? -(int64_t)(u & 0x7fffffffu) - 1 : (int64_t)u; };
  int64_t lo = to_key(-128.0f), hi = to_key(127.0f);   // round(lo) = -128 < zero_point <= 127 = round(hi)
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (round_sat_i8_host(from_key(mid)) >= zero_point) hi = mid; else lo = mid;
  }
  return from_key(hi);
}

void fold_parameters(HostPlan& p, const int32_t* filter_ohwi, const float* post_mul,
                     const float* post_bias, const int32_t* thresholds) {
  const lce_hip_bconv2d_desc& d = p.d;
  const int n = d.channels_out;
  const size_t fwords = (size_t)n * d.filter_height * d.filter_width * p.cwg;
  p.filter.assign((const uint32_t*)filter_ohwi, (const uint32_t*)filter_ohwi + fwords);
  p.mul.clear();
  p.bias.clear();
  p.thresholds.clear();
  p.zero_pad_cache.clear();
  if (d.dst_type == LCE_HIP_BITPACKED) {
    p.thresholds.assign(thresholds, thresholds + n);
  } else {
    // bconv2d.cc:364-378 -- double arithmetic, stored as float
    const double scale = d.dst_type == LCE_HIP_I8 ? (double)d.out_scale : 1.0;
    const double zp = d.dst_type == LCE_HIP_I8 ? (double)d.out_zero_point : 0.0;
    const double a = (double)p.backtransform_add;
    p.mul.resize(n);
    p.bias.resize(n);
    for (int i = 0; i < n; ++i) {
      const double m = post_mul[i], b = post_bias[i];
      p.mul[i] = (float)(-1 * m / scale);
      p.bias[i] = (float)((b + a * m) / scale + zp);
    }
    // CalculateActivationRange<int32> then bconv2d.cc:380-388
    int32_t lo = INT32_MIN, hi = INT32_MAX;
    switch (d.activation) {
      case LCE_HIP_ACT_RELU: lo = 0; break;
      case LCE_HIP_ACT_RELU6: lo = 0; hi = 6; break;
      case LCE_HIP_ACT_RELU_N1_TO_1: lo = -1; hi = 1; break;
      default: break;
    }
    lo = std::max(lo, -p.backtransform_add);
    hi = std::min(hi, p.backtransform_add);
    p.clamp_min = -hi + p.backtransform_add;
    p.clamp_max = -lo + p.backtransform_add;
    if (p.zero_pad_mode == kZeroPadCorrection) fill_zero_pad_cache(p, post_mul);
    p.bit_thr = d.dst_type == LCE_HIP_I8 ? int8_below_threshold(d.out_zero_point) : 0.0f;
  }
  p.have_weights = true;
  p.packed.clear();  // force a repack on the next select_kernel
  p.wq.clear();
}

bool tiled_supports(const HostPlan& p, int tn) {
  if (p.d.dst_type == LCE_HIP_BITPACKED && tn != 32) return false;
  if (p.d.groups > 1 && p.npg % tn != 0) return false;  // a tile must not straddle groups
  return true;
}

static void pack_for_tile(HostPlan& p) {
  const lce_hip_bconv2d_desc& d = p.d;
  const int tn = p.tile.tn, taps = d.filter_height * d.filter_width, n = d.channels_out;
  p.nt = ceil_div(n, tn);
  p.packed.assign((size_t)p.nt * taps * p.cwg * tn, 0u);
  p.oob_corr.assign((size_t)p.nt * taps * tn, 0);
  const int bzp = (d.channels_in / d.groups) / 2;
  for (int oc = 0; oc < n; ++oc) {
    const int t_idx = oc / tn, j = oc % tn;
    for (int t = 0; t < taps; ++t) {
      int pop = 0;
      for (int c = 0; c < p.cwg; ++c) {
        const uint32_t w = p.filter[((size_t)oc * taps + t) * p.cwg + c];
        p.packed[(((size_t)t_idx * taps + t) * p.cwg + c) * tn + j] = w;
        pop += popcount32(w);
      }
      p.oob_corr[((size_t)t_idx * taps + t) * tn + j] = bzp - pop;
    }
  }
  const size_t padded = (size_t)p.nt * tn;
  p.mul_p.assign(padded, 0.0f);
  p.bias_p.assign(padded, 0.0f);
  p.thr_p.assign(padded, INT32_MAX);  // padded channels never set a bit
  std::copy(p.mul.begin(), p.mul.end(), p.mul_p.begin());
  std::copy(p.bias.begin(), p.bias.end(), p.bias_p.begin());
  std::copy(p.thresholds.begin(), p.thresholds.end(), p.thr_p.begin());
}

// ------------------------------------------------------------------------------------
// matrix-core engine
// ------------------------------------------------------------------------------------
bool mfma_supported(const HostPlan& p) {
  const lce_hip_bconv2d_desc& d = p.d;
  // grouped: a block's channels must lie in one group -- some block width (64) has to divide Cout/G
  if (d.groups != 1 && (d.channels_out / d.groups) % 64 != 0) return false;
  if (p.backtransform_add >= (1 << 23)) return false;    // fp32 accumulation must stay exact
  const int cpad = ceil_div(d.channels_in, 64) * 64;
  const int64_t hp = std::max<int64_t>(p.pad_h + d.in_height,
                                       (int64_t)(p.out_h - 1) * d.stride_height + (d.filter_height - 1) * d.dilation_height + 1);
  const int64_t wp = std::max<int64_t>(p.pad_w + d.in_width,
                                       (int64_t)(p.out_w - 1) * d.stride_width + (d.filter_width - 1) * d.dilation_width + 1);
  if (hp * wp * (cpad / 2) >= (1ll << 31)) return false; // one padded image must fit a buffer resource
  if (hp * wp * (cpad / 32) >= (1ll << 31)) return false;
  return true;
}

// The streaming 1x1 kernel (lce_kernels_pointwise.h): filter extent 1 (then no padding exists -- SAME pads
// (out-1)*stride + 1 - in <= 0 -- and dilation is moot), any stride, one group, whole 32-channel output tiles, and a
// filter bank that fits registers (1, 2, 4 or 8 K-steps of 64 input channels).
// nj = 32-channel tiles per block.  More tiles per block = fewer re-reads of the input words (they come from L2) and
// fewer FP4 conversions per output, but longer waves; measured (profiles/r03/pointwise_widen_*.txt) 64 channels per
// block win on every launch of fewer than ~16k wave-tiles -- whose time is the length of a wave's load -> MFMA ->
// transform -> store chain times the rounds of blocks, not throughput -- and 128 on the large ones; 32 never wins.
bool pointwise_supported(const HostPlan& p, int64_t pixels, int* nc, int* nj) {
  const lce_hip_bconv2d_desc& d = p.d;
  if (!mfma_supported(p)) return false;
  if (d.filter_height != 1 || d.filter_width != 1) return false;
  if (d.groups != 1 || p.pad_h != 0 || p.pad_w != 0) return false;
  if (p.out_h != (d.in_height - 1) / d.stride_height + 1 || p.out_w != (d.in_width - 1) / d.stride_width + 1) return false;
  if (d.channels_out % 32 != 0) return false;
  // (129..192 channels: the four-K-step instances -- the fourth step's activations are masked to code 0 as a partial last step is, and its
  //  weights are the all-zero fourth K-step that pack_for_mfma appends to the three-step image)
  const int c = pointwise_bank_steps(ceil_div(d.channels_in, 64));
  if (c == 0) return false;
  const int t = d.channels_out / 32;
  int j = t % 4 == 0 ? 4 : t % 2 == 0 ? 2 : 1;
  // (float: 16 row stores of a 128-channel tile + the bank do not fit 256 VGPRs)
  if (c == 8 || d.dst_type == LCE_HIP_F32) j = std::min(j, 2);
  if (p.pw_nj_pref > 0) {
    if (t % p.pw_nj_pref != 0) return false;
    j = std::min(j, p.pw_nj_pref);          // (a tuning aid: what the instances cannot do is clamped, not refused)
  } else {
    const int64_t tiles = (pixels + 31) / 32;
    if (j == 4 && tiles * (t / 4) < pw_small_launch_tiles) j = 2;
  }
  *nc = c;
  *nj = j;
  return true;
}

// Auto rule (profiles/r02/pointwise_vs_block_gemm.txt, profiles/r03/pointwise_widen_*.txt): on every 1x1 layer measured --
// 64 ... 512 channels on 56x56 ... 7x7 maps at batch 256, strides 1 and 2, all three output types -- the streaming
// kernel is faster than or equal to the block GEMM (int8 -10...-40 %, bitpacked -5...-55 %, float 0...-25 %) with two
// exceptions that stay on the block GEMM: launches of less than 32 tiles, and float output of a LARGE launch
// (56x56x256 -> 256: 151 vs 144 us; the float epilogue's 64-channel blocks re-read the input words four times).
static bool pointwise_preferred(const HostPlan& p, int64_t pixels) {
  const int64_t tiles = (pixels + 31) / 32;
  if (p.d.dst_type == LCE_HIP_F32 && tiles * (p.d.channels_out / 64) >= 65536) return false;
  return pixels >= 1024;
}

PwArgs make_pw_args(const HostPlan& p, int batch_chunk) {
  const lce_hip_bconv2d_desc& d = p.d;
  PwArgs P{};
  const int64_t m = (int64_t)batch_chunk * p.out_h * p.out_w;
  P.M = (int32_t)m;
  P.N = d.channels_out;
  P.Npad = p.npad;
  P.Cw = p.cw;
  P.Cin = d.channels_in;
  P.Wout = p.wout;
  P.tiles = (int32_t)((m + 31) / 32);
  P.noclamp = (p.clamp_min <= 0 && (int64_t)p.clamp_max >= 2 * (int64_t)p.backtransform_add) ? 1 : 0;
  P.in_bytes = (uint32_t)((int64_t)batch_chunk * d.in_height * d.in_width * p.cw * 4);
  const int64_t row = d.dst_type == LCE_HIP_BITPACKED ? (int64_t)p.wout * 4 : (int64_t)d.channels_out * (d.dst_type == LCE_HIP_I8 ? 1 : 4);
  // (max_batch_per_launch keeps a launch's output below 2 GiB: the kernel's offsets are 32-bit)
  P.out_bytes = (uint32_t)std::min<int64_t>(m * row, (1ll << 31) - 1);
  P.a_bt = (float)p.backtransform_add;
  P.cmin = (float)p.clamp_min;
  P.cmax = (float)p.clamp_max;
  P.bit_thr = p.bit_thr;
  P.OW = p.out_w;
  P.OHW = p.out_h * p.out_w;
  P.IW = d.in_width;
  P.IHW = d.in_height * d.in_width;
  P.SH = d.stride_height;
  P.SW = d.stride_width;
  P.div_ow = make_fastdiv((uint32_t)P.OW);
  P.div_ohw = make_fastdiv((uint32_t)P.OHW);
  return P;
}

static const MfmaCfg kMfmaCfgs[] = {
    {4, 2, 2, 4},  // 256 x 256, 8 waves
    {4, 2, 2, 2},  // 256 x 128, 8 waves
    {8, 1, 2, 2},  // 512 x  64, 8 waves
    {2, 2, 2, 4},  // 128 x 256, 4 waves
    {2, 2, 2, 2},  // 128 x 128, 4 waves
    {4, 1, 2, 2},  // 256 x  64, 4 waves
    {2, 1, 2, 2},  // 128 x  64, 2 waves
};
const MfmaCfg* mfma_cfg_by_tile(int bm, int bn) {
  for (const MfmaCfg& c : kMfmaCfgs)
    if (c.bm() == bm && c.bn() == bn) return &c;
  return nullptr;
}

// 64-channel chunks per tap that the blocks of a grouped convolution run: the chunks group g's channel
// slice [g*Cin_g, (g+1)*Cin_g) touches (Cin_g is a multiple of 32, so a slice may start mid-chunk)
int group_chunks(const lce_hip_bconv2d_desc& d) {
  const int cin_g = d.channels_in / d.groups;
  int most = 0;
  for (int g = 0; g < d.groups; ++g)
    most = std::max(most, ((g + 1) * cin_g + 63) / 64 - (g * cin_g) / 64);
  return most;
}

// int8 plans: the per-channel constants of the matrix-core kernels' epilogues -- mul_q, bias_q, thr_q = [lo | hi] -- and whether the
// selected kernel may run its one-instruction forms (int8_floor_ok).  Runs after EVERY kernel selection (select_kernel): what can be
// proven depends on the arithmetic of the kernel that will run.
//
// thr_q: the per-channel range of the transformed value y over the clamped accumulator x in [clamp_min, clamp_max], intersected with
// int8's range.  y is monotone in x (each rounding is), so med3(y(x), lo, hi) == saturate(y(med3(x, clamp_min, clamp_max))) for every
// x: the kernels spend one clamp instead of two (lce_kernels_pointwise.h).
//
// The one-instruction forms (I8F instances): rounding = floor(y + 0.5) (v_cvt_rpi_i32_f32, exact for every float: tools/probes/cvt_rpi.hip)
// instead of round-half-away (output_transform.h:31-44) -- they differ only where the clamped y is an exact NEGATIVE tie (-k - 0.5) --
// and, in the streaming and the weight-streaming kernel, y = fma(x, mul, bias) (one rounding) instead of fl(fl(x * mul) + bias).  On
// one plan y takes finitely many values: x runs over the values the accumulator can hold inside the clamps -- the EVEN integers
// (x = K_bt - <a, w> = 2 * popcount) and the two clamp ends -- so every one of them is checked, per channel, against the reference's
// arithmetic: equal int8, and no negative tie (so that the second output's y < threshold also means what it means in the reference).
// Layers do hold ties and near-ties (y lives on the grid of its operands' ulps: 2^-17 near |y| = 100 with multipliers well below 1,
// 2^-10 where x * mul reaches 10^4), so a channel that fails is given NEIGHBOURING parameters -- the bias up to four grid steps lower,
// the multiplier up to two ulps either side -- and enumerated again against the reference's arithmetic on the ORIGINAL parameters.  No
// neighbour passes (multipliers like 0.25 with integer biases: positive and negative ties all over the channel): the plan keeps the
// exact instances and the original parameters.
// (Kernels with the reference's own two roundings -- the pointwise kernel, whose launches fall back to the block GEMM on unaligned
//  output pointers, and the block GEMM itself -- get parameters proven under two roundings: with no negative tie left, round-half-away
//  and floor(y + 0.5) agree on them.)
static void prepare_int8_epilogue(HostPlan& p) {
  const lce_hip_bconv2d_desc& d = p.d;
  const int n = d.channels_out, cin_g = d.channels_in / d.groups;
  p.mul_q.assign(p.npad, 0.0f);
  p.bias_q.assign(p.npad, 0.0f);
  std::copy(p.mul.begin(), p.mul.end(), p.mul_q.begin());
  std::copy(p.bias.begin(), p.bias.end(), p.bias_q.begin());
  // Channels past the last one (the tables are padded to the block's width): their value must never count as "below the zero point" in
  // the second output of lce_hip_bconv2d_run_dual -- LceQuantize leaves the padding bits of a row's last word 0 (bitpack.h:238-244).
  // The block GEMM compares every lane's value with one threshold T (lce_kernels_mfma.h, bit_rows), and with mul = bias = 0 a padded
  // lane held 0 < T for every positive zero point: ragged channel counts (Cout % 32 != 0) got ones in those bits (round 6: found by the
  // randomized GPU test).  0 * x + inf = +inf is below no threshold; such lanes store nothing.
  for (int i = n; i < p.npad; ++i) p.bias_q[i] = std::numeric_limits<float>::infinity();
  p.thr_q.assign((size_t)2 * p.npad, 0.0f);
  const bool one_rounding = p.use_stream || p.use_wstream;
  auto two_roundings = [](float mul, float bias, int32_t x) {
    volatile float pr = (float)x * mul;
    volatile float r = pr + bias;
    return (float)r;
  };
  auto proven_form = [&](float mul, float bias, int32_t x) {
    return one_rounding ? std::fmaf((float)x, mul, bias) : two_roundings(mul, bias, x);
  };
  auto set_range = [&](int i, float mul, float bias, bool proven) {
    const float a0 = proven ? proven_form(mul, bias, p.clamp_min) : two_roundings(mul, bias, p.clamp_min);
    const float a1 = proven ? proven_form(mul, bias, p.clamp_max) : two_roundings(mul, bias, p.clamp_max);
    float lo = std::min(a0, a1), hi = std::max(a0, a1);
    if (!(lo == lo) || !(hi == hi)) { lo = -128.0f; hi = 127.0f; }   // NaN parameters: unspecified in the reference
    p.thr_q[i] = std::max(-128.0f, std::min(127.0f, lo));
    p.thr_q[p.npad + i] = std::max(-128.0f, std::min(127.0f, hi));
  };
  for (int i = 0; i < n; ++i) set_range(i, p.mul[i], p.bias[i], false);
  // (LCE_PLAN_INT8_EXACT, read at plan creation: an A/B aid for whole stacks, tools/gpu_r05.sh i8floor)
  p.int8_floor_ok = !p.int8_exact_pref && !p.dbg_int8_exact;
  p.int8_bias_adjusted = 0;
  const int32_t x_lo = std::max<int32_t>(0, p.clamp_min), x_hi = (int32_t)std::min<int64_t>(2 * (int64_t)p.backtransform_add, p.clamp_max);
  const bool even_only = cin_g % 2 == 0;   // (an odd channel count under zero padding: border pixels drop an odd number of terms)
  auto sat8_half_away = [](float y) { const float r = std::round(y); return (int)std::max(-128.0f, std::min(127.0f, r)); };
  // The accumulator values a channel is checked on, in order: x_lo, the even values above it, x_hi.  y is monotone in x (each rounding
  // is), so the values whose y lies strictly inside the clamps are one run of them -- found by bisection -- and every value outside
  // it produces the clamp itself: the run, one value either side of it and the two ends are all that needs checking.
  const int32_t first_even = even_only ? ((x_lo + 2) & ~1) : x_lo + 1;
  const int32_t n_mid = even_only ? (x_hi > first_even ? (x_hi - 1 - first_even) / 2 + 1 : 0) : std::max(0, x_hi - 1 - x_lo);
  const int32_t n_x = x_hi > x_lo ? n_mid + 2 : 1;
  auto x_at = [&](int32_t k) { return k == 0 ? x_lo : k == n_x - 1 ? x_hi : first_even + (even_only ? 2 : 1) * (k - 1); };
  auto inner_run = [&](auto&& y_of, float lo, float hi, int32_t* k0, int32_t* k1) {
    const bool inc = y_of(x_hi) >= y_of(x_lo);
    auto below = [&](int32_t k) { const float y = y_of(x_at(k)); return inc ? y <= lo : y >= hi; };        // before the run
    auto not_past = [&](int32_t k) { const float y = y_of(x_at(k)); return inc ? y < hi : y > lo; };       // before its end
    int32_t a = 0, b = n_x;
    while (a < b) { const int32_t m = a + (b - a) / 2; if (below(m)) a = m + 1; else b = m; }
    *k0 = a;
    b = n_x;
    while (a < b) { const int32_t m = a + (b - a) / 2; if (not_past(m)) a = m + 1; else b = m; }
    *k1 = a;                                                                                                // [k0, k1)
  };
  // (LCE_PLAN_INT8_FULL, read at plan creation; testing aid: every value instead of the run -- tests compare the two)
  const bool check_all = p.dbg_int8_full;
  auto channel_ok = [&](int i, float mul, float bias, int32_t* bad_x) {
    const float a0 = proven_form(mul, bias, p.clamp_min), a1 = proven_form(mul, bias, p.clamp_max);
    if (!(a0 == a0) || !(a1 == a1)) return false;
    const float lo = std::max(-128.0f, std::min(127.0f, std::min(a0, a1))), hi = std::max(-128.0f, std::min(127.0f, std::max(a0, a1)));
    auto value_ok = [&](int32_t x) {
      float yc = proven_form(mul, bias, x);
      yc = yc < lo ? lo : (yc > hi ? hi : yc);
      const bool negative_tie = yc < 0.0f && yc - std::floor(yc) == 0.5f;
      if (!(yc == yc) || negative_tie ||      // (NaN: infinite multipliers -- unspecified in the reference, the exact instances)
          (int)std::floor((double)yc + 0.5) != sat8_half_away(two_roundings(p.mul[i], p.bias[i], x))) {
        if (bad_x) *bad_x = x;
        return false;
      }
      return true;
    };
    // the run inside the candidate's clamps and the run inside int8's range under the reference's parameters, one value either side
    int32_t k0, k1, r0, r1;
    inner_run([&](int32_t x) { return proven_form(mul, bias, x); }, lo, hi, &k0, &k1);
    inner_run([&](int32_t x) { return two_roundings(p.mul[i], p.bias[i], x); }, -128.5f, 127.5f, &r0, &r1);
    k0 = std::max(0, std::min(k0, r0) - 1);
    k1 = std::min(n_x, std::max(k1, r1) + 1);
    if (check_all) { k0 = 0; k1 = n_x; }
    if (!value_ok(x_lo) || !value_ok(x_hi)) return false;
    for (int32_t k = k0; k < k1; ++k)
      if (!value_ok(x_at(k))) return false;
    return true;
  };
  struct Adjusted { int i; float mul, bias; };
  std::vector<Adjusted> adjusted;
  for (int i = 0; i < n && p.int8_floor_ok; ++i) {
    int32_t bad = 0;
    if (channel_ok(i, p.mul[i], p.bias[i], &bad)) continue;
    // the grid step of y around the failing value: the coarser of the product's and the bias's ulp (a smaller step is absorbed by the
    // sum's rounding)
    const float y_bad = proven_form(p.mul[i], p.bias[i], bad), pr_bad = std::fabs((float)bad * p.mul[i]);
    const float big = std::max(std::max(pr_bad, std::fabs(p.bias[i])), std::fabs(y_bad));
    const float step = std::max(std::nextafter(big, INFINITY) - big, std::ldexp(1.0f, -20));
    bool found = false;
    for (int dm = 0; dm <= 4 && !found; ++dm) {            // multiplier: 0, +1, -1, +2, -2 ulps
      float m2 = p.mul[i];
      for (int s = 0; s < (dm + 1) / 2; ++s) m2 = std::nextafter(m2, (dm & 1) ? INFINITY : -INFINITY);
      for (int t = 0; t < 9 && !found; ++t) {
        const int k = t < 5 ? t : 4 - t;                            // 0, 1, 2, 3, 4 steps lower, then 1 .. 4 higher
        if (dm == 0 && k == 0) continue;
        const float b2 = p.bias[i] - (float)k * step;
        if (channel_ok(i, m2, b2, nullptr)) { adjusted.push_back({i, m2, b2}); found = true; }
      }
    }
    if (!found) {
      if (p.dbg_level >= 1)
        fprintf(stderr,
                "[lce plan] int8: channel %d, accumulator %d -> y = %.9g: no neighbouring (multiplier, bias) reproduces the reference "
                "there, round-half-away instances\n",
                i, bad, (double)y_bad);
      p.int8_floor_ok = false;
    }
  }
  if (p.int8_floor_ok) {
    for (const auto& a : adjusted) { p.mul_q[a.i] = a.mul; p.bias_q[a.i] = a.bias; }
    for (int i = 0; i < n; ++i) set_range(i, p.mul_q[i], p.bias_q[i], true);
    p.int8_bias_adjusted = (int)adjusted.size();
    if (p.dbg_level >= 1 && !adjusted.empty())
      fprintf(stderr, "[lce plan] int8: one-instruction forms, %zu channel(s) with adjusted parameters\n", adjusted.size());
  }
}

static void pack_for_mfma(HostPlan& p) {
  const lce_hip_bconv2d_desc& d = p.d;
  const int taps = d.filter_height * d.filter_width, n = d.channels_out;
  const int bn = p.mfma.bn();
  const int cin_g = d.channels_in / d.groups;
  // (the streaming family's instances: 64 / 128 / 256 / 512)
  p.cpad = ((p.use_stream || p.use_wstream) ? stream_chunks(d) : ceil_div(d.channels_in, 64)) * 64;
  p.npad = ceil_div(n, bn) * bn;
  p.kch = d.groups > 1 ? group_chunks(d) : p.cpad / 64;
  const int kch = p.kch, ks_total = taps * kch;
  // 1x1 layers of 129..192 input channels run the pointwise kernel's FOUR-K-step instances (pointwise_supported): the image carries that
  // fourth K-step as zero codes (it used to end after three, and the kernel's bank loads -- plain pointer loads, no range check -- read
  // Npad * 32 bytes past the allocation: the advisor's round-5 finding).  The block GEMM walks p.kch steps and never looks at it.
  const int ks_alloc = (taps == 1 && d.groups == 1 && p.wq_layout == 0) ? pointwise_bank_steps(kch) : ks_total;
  p.wq.assign((size_t)std::max(ks_total, ks_alloc) * p.npad * 32, 0);
  for (int oc = 0; oc < n; ++oc) {
    const int g = oc / p.npg, chunk0 = (g * cin_g) / 64;    // the kernel starts the group's K loop at this chunk
    for (int t = 0; t < taps; ++t)
      for (int ci = 0; ci < cin_g; ++ci) {
        const uint32_t w = p.filter[((size_t)oc * taps + t) * p.cwg + ci / 32];
        // the NEGATED weight: bit 1 (-1) -> +1 = 0x2, bit 0 (+1) -> -1 = 0xA.  The kernel starts its
        // accumulators at K_bt, so they hold K_bt - <a, w> = 2 * popcount-accumulator directly, the
        // value the output transform clamps (output_transform.h:62-91,105) -- no subtraction later.
        // Channels of the chunk that belong to a neighbouring group keep the code 0: they contribute nothing.
        const uint8_t nib = ((w >> (ci % 32)) & 1u) ? 0x2 : 0xA;
        const int c = g * cin_g + ci;                         // position in the pixel's channel vector
        const int ks = t * kch + (c / 64 - chunk0), j = c % 64, half = j / 32, jj = j % 32;
        // K-major (the block GEMM, the pointwise and the weight-stationary kernels): [K-step][K-half][Npad][16 B].  Tile-major (the
        // weight-streaming kernel): [32-channel tile][K-step][K-half][32 x 16 B] -- a wave's fragment is 512 contiguous bytes and
        // the fragments that the waves of a launch fetch at the same moment (one K-step, all tiles) lie KS KiB apart instead of
        // inside two 4-KiB runs, i.e. spread over the L2's channels instead of queueing on two of them.
        const size_t cell = p.wq_layout == 1 ? (((size_t)(oc / 32) * ks_total + ks) * 2 + half) * 32 + oc % 32
                                             : ((size_t)ks * 2 + half) * p.npad + oc;
        uint8_t& byte = p.wq[cell * 16 + jj / 2];
        byte |= (uint8_t)(nib << (4 * (jj & 1)));
      }
  }
  p.mul_q.assign(p.npad, 0.0f);
  p.bias_q.assign(p.npad, 0.0f);
  std::copy(p.mul.begin(), p.mul.end(), p.mul_q.begin());
  std::copy(p.bias.begin(), p.bias.end(), p.bias_q.begin());
  if (d.dst_type == LCE_HIP_I8) { prepare_int8_epilogue(p); return; }
  // bit = (accum > thr)  <=>  2*accum > 2*thr, and the kernel's accumulator IS 2*accum (an integer
  // in [0, 2*K_bt]); clamp so the float is exact, keep the always / never cases
  const int64_t a = p.backtransform_add;
  p.thr_q.assign(p.npad, (float)(2 * a + 2));  // padded channels: never
  for (size_t i = 0; i < p.thresholds.size(); ++i) {
    int64_t t = 2 * (int64_t)p.thresholds[i];
    t = std::max<int64_t>(-1, std::min<int64_t>(2 * a + 2, t));
    p.thr_q[i] = (float)t;
  }
}

MfmaCfg choose_mfma_cfg(const HostPlan& p, int64_t pixels) {
  // Measured on MI355X (profiles/r01/tile_sweep_*.jsonl): 4-wave blocks with two or more
  // blocks resident per CU beat the 8-wave 256x256 block (independent barriers overlap one
  // block's LDS/epilogue phases with another's MFMAs).  Pick BN by the channel count, then
  // halve it while the grid would leave CUs idle.
  const int n = p.d.channels_out;
  int bn = n > 128 ? 256 : n > 64 ? 128 : 64;
  auto blocks = [&](int bm, int bn_) { return ((pixels + bm - 1) / bm) * (int64_t)ceil_div(n, bn_); };
  while (bn > 64 && blocks(128, bn) < 512) bn /= 2;
  while (bn > 64 && p.d.groups > 1 && p.npg % bn) bn /= 2;   // a block's channels lie in one group
  const int bm = bn == 64 ? 256 : 128;
  const MfmaCfg* c = mfma_cfg_by_tile(bm, bn);
  if (bn == 64 && blocks(256, 64) < 512) c = mfma_cfg_by_tile(128, 64);
  return c ? *c : kMfmaCfgs[4];
}

// Direct variant: geometry of a BM-pixel tile and whether its input halo (plus the weight ring)
// fits LDS.  Images of at most BM/2 pixels are tiled IPT whole images at a time, larger ones
// in BM-pixel pieces of ONE image.
bool direct_geometry(const HostPlan& p, const MfmaCfg& c, int* tpi, int* halo_rows, int* ps, int* halo_bytes,
                     int* ipt, int lds_budget, int* tile_tx, int* halo_w) {
  const lce_hip_bconv2d_desc& d = p.d;
  if (!mfma_supported(p)) return false;
  const int cpad = ceil_div(d.channels_in, 64) * 64;
  const int64_t wp = std::max<int64_t>(p.pad_w + d.in_width,
                                       (int64_t)(p.out_w - 1) * d.stride_width + (d.filter_width - 1) * d.dilation_width + 1);
  const int bm = c.bm(), ohow = p.out_h * p.out_w;
  const int64_t ring = (int64_t)MfmaCfg::kDirectStages * c.bn() * 32;
  const int64_t stride = (cpad / 32) * 16 + 16;
  int images = 1;
  int64_t rows;
  if (ohow * 2 <= bm) {
    // whole images: as many as fill the tile, fewer if LDS says so
    rows = (int64_t)(p.out_h - 1) * d.stride_height + (int64_t)(d.filter_height - 1) * d.dilation_height + 1;
    images = (int)std::min<int64_t>(bm / ohow, (lds_budget - ring) / std::max<int64_t>(1, rows * wp * stride));
    if (images < 1) return false;
  } else {
    // a tile of bm consecutive pixels of one image touches at most this many output rows
    const int rows_out = std::min(p.out_h, (bm + p.out_w - 2) / p.out_w + 1);
    rows = (int64_t)(rows_out - 1) * d.stride_height + (int64_t)(d.filter_height - 1) * d.dilation_height + 1;
  }
  int64_t bytes = ((int64_t)images * rows * wp * stride + 1023) / 1024 * 1024;
  int tx = 0;
  int64_t wh = wp;
  // Wide images: a strip of bm consecutive pixels is a sliver of one or two rows whose halo spans their whole width
  // (224-wide, 256 pixels: 4 rows x 226 columns staged for 256 outputs).  2-D tiles of bm/32 rows x 32 columns stage
  // (bm/32 - 1) * SH + eKH rows x 31 * SW + eKW columns instead.  Taken when that is at most 0.7 of the strip's halo
  // and pads the image by no more than 3 % beyond what strips do (56-wide: 64 columns for 56 -- stays with strips).
  if (ohow * 2 > bm && tile_tx != nullptr) {
    const int th = bm / 32, tw = 32;
    const int64_t rows2 = (int64_t)(th - 1) * d.stride_height + (int64_t)(d.filter_height - 1) * d.dilation_height + 1;
    const int64_t wh2 = (int64_t)(tw - 1) * d.stride_width + (int64_t)(d.filter_width - 1) * d.dilation_width + 1;
    const int64_t bytes2 = (rows2 * wh2 * stride + 1023) / 1024 * 1024;
    const int ty_n = ceil_div(p.out_h, th), tx_n = ceil_div(p.out_w, tw);
    const double pad_strip = (double)ceil_div(ohow, bm) * bm / ohow, pad_2d = (double)ty_n * tx_n * bm / ohow;
    const bool strip_fits = bytes + ring <= lds_budget;
    const bool rule = pad_2d <= pad_strip + 0.03 && (!strip_fits || bytes2 * 10 <= bytes * 7);
    if (bytes2 + ring <= lds_budget && p.tile2d_pref != 2 && (rule || p.tile2d_pref == 1)) {
      tx = tx_n;
      rows = rows2;
      wh = wh2;
      bytes = bytes2;
      *tpi = ty_n * tx_n;
    }
  }
  if (bytes + ring > lds_budget) return false;
  if (tx == 0) *tpi = ohow * 2 <= bm ? 1 : ceil_div(ohow, bm);
  *halo_rows = (int)rows;
  *ps = (int)stride;
  *halo_bytes = (int)bytes;
  *ipt = images;
  if (tile_tx) *tile_tx = tx;
  if (halo_w) *halo_w = (int)wh;
  return true;
}

// Block tile of the direct variant.  Tiles never cross an image, so the last tile of every
// image is padded: the rule (from profiles/r01/tile_sweep_direct.jsonl) is 128 channels per
// block when there are that many, 256 pixels unless that pads > 8 % more than 128 pixels
// would, and no direct variant at all when less than 70 % of the tile rows are real (7x7
// images) or the halo does not fit LDS -- the workspace GEMM, whose tiles span images, is
// better there.
bool choose_direct_cfg(const HostPlan& p, MfmaCfg* out) {
  const int ohow = p.out_h * p.out_w;
  // 256 channels per block (4 waves of 2x4 tiles) on long launches (>= 8 rounds of 128-pixel
  // blocks): int8 rows then fill whole 128-byte lines (L0 0.283 -> 0.268 ms) and, with the
  // MFMA-first K-step, float / bitpacked output gains 2-3 % over 256x128 (0.2955 -> 0.288 ms).  On
  // short launches more, smaller blocks win (28x28x256 int8: 128x128 0.082 ms vs 128x256 0.089).
  const bool wide = p.d.channels_out > 128 && (int64_t)p.d.batch * ceil_div(ohow, 128) >= 4096;
  int bn = wide ? 256 : p.d.channels_out > 64 ? 128 : 64;
  while (bn > 64 && p.d.groups > 1 && p.npg % bn) bn /= 2;    // a block's channels lie in one group
  struct Cand { const MfmaCfg* c; double padded; int64_t blocks; };
  Cand cand[2];
  int n = 0;
  for (int bm : {256, 128}) {
    if (bm == 256 && bn == 256) continue;   // the 8-wave 256x256 block loses to two 128x256 blocks per CU on every layer measured
    const MfmaCfg* c = mfma_cfg_by_tile(bm, bn);
    int tpi, rows, ps, bytes, ipt, ttx, hw;
    // two blocks per CU (80 KiB each) is where the variant pays: with one, nothing overlaps a
    // block's halo expansion (measured: 28x28x1024 at 138 KiB loses to the workspace GEMM)
    if (!c || !direct_geometry(p, *c, &tpi, &rows, &ps, &bytes, &ipt, kDirectLdsAuto, &ttx, &hw)) continue;
    // rows of a tile that are real pixels
    const double padded = ohow * 2 <= bm ? (double)bm / ((double)ipt * ohow) : (double)tpi * bm / (double)ohow;
    if (padded > 1.0 / 0.7) continue;
    const int64_t blocks = (int64_t)(ohow * 2 <= bm ? ceil_div(p.d.batch, ipt) : p.d.batch * tpi) * ceil_div(p.d.channels_out, bn);
    cand[n++] = Cand{c, padded, blocks};
  }
  if (n == 0) return false;
  int pick = 0;
  // 256 pixels unless that pads > 8 % more than 128 would, or leaves fewer than 2 blocks per CU
  if (n == 2 && cand[0].c->bm() == 256 && (cand[0].padded - cand[1].padded > 0.08 || cand[0].blocks < 512)) pick = 1;
  *out = *cand[pick].c;
  return true;
}

size_t mfma_workspace_bytes(const HostPlan& p, int batch_chunk) {
  return (size_t)batch_chunk * p.hp * p.wp * (p.cpad / 2);
}

MfmaArgs make_mfma_args(const HostPlan& p, int batch_chunk) {
  const lce_hip_bconv2d_desc& d = p.d;
  MfmaArgs G{};
  G.H = d.in_height; G.W = d.in_width; G.Cw = p.cw; G.Cin = d.channels_in;
  G.Hp = p.hp; G.Wp = p.wp; G.PH = p.pad_h; G.PW = p.pad_w;
  G.NPIX = (uint32_t)((int64_t)batch_chunk * p.hp * p.wp); G.CPW = p.cpad / 32; G.KCH = p.d.groups > 1 ? group_chunks(p.d) : p.cpad / 64;
  G.Npad = p.npad;
  G.zero_border = p.zero_pad_mode == kZeroPadExact ? 1 : 0;
  G.x_bytes = (uint32_t)mfma_workspace_bytes(p, batch_chunk);
  G.w_bytes = (uint32_t)p.wq.size();
  G.div_npix = make_fastdiv(G.NPIX);
  G.div_wp = make_fastdiv((uint32_t)G.Wp);
  G.div_hp = make_fastdiv((uint32_t)G.Hp);
  G.div_npg = make_fastdiv((uint32_t)p.npg);
  G.a_bt = (float)p.backtransform_add;
  G.cmin = (float)p.clamp_min;
  G.cmax = (float)p.clamp_max;
  G.noclamp = (p.clamp_min <= 0 && (int64_t)p.clamp_max >= 2 * (int64_t)p.backtransform_add) ? 1 : 0;
  G.bit_thr = p.bit_thr;
  // the wide int8 epilogue transposes WN tiles at once: only where the block's LDS allocation
  // already covers waves * WN * 4 KiB (it must not cost a resident block)
  {
    const int lds = p.use_direct ? p.mfma.direct_lds_bytes(p.halo_bytes) : p.mfma.lds_bytes();
    const bool room = (p.mfma.threads() / 64) * p.mfma.wn * 4096 <= lds;
    G.i8_wide = (p.d.dst_type == LCE_HIP_I8 && p.d.channels_out % 16 == 0 && room) ? 1 : 0;
    // float: same joint transpose (one LDS fence pair per 32-row block instead of one per tile);
    // not with the SAME-zero correction variant, whose epilogue is per tile
    G.f32_wide = (p.d.dst_type == LCE_HIP_F32 && p.d.channels_out % 4 == 0 && room &&
                  p.zero_pad_mode != kZeroPadCorrection) ? 1 : 0;
    if (p.epilogue_pref == 1) { G.f32_wide = 0; G.i8_wide = 0; }
  }
  if (p.use_direct) {
    G.TPI = p.tpi; G.OHOW = p.out_h * p.out_w; G.halo_rows = p.halo_rows; G.PS = p.ps;
    G.halo_bytes = p.halo_bytes; G.QG = (G.CPW + 3) / 4;
    G.Wh = p.halo_w; G.TX = p.tile_tx;
    G.IPT = p.ipt; G.B = batch_chunk; G.HPIX = p.halo_rows * p.halo_w;
    G.div_wh = make_fastdiv((uint32_t)G.Wh);
    G.div_tx = make_fastdiv((uint32_t)(G.TX > 0 ? G.TX : 1));
    G.div_tpi = make_fastdiv((uint32_t)G.TPI);
    G.div_qg = make_fastdiv((uint32_t)G.QG);
    G.div_ohow = make_fastdiv((uint32_t)G.OHOW);
    G.div_hpix = make_fastdiv((uint32_t)G.HPIX);
  }
  return G;
}

// ------------------------------------------------------------------------------------
// Which kernel runs a layer: a cost estimate per candidate (lce_plan_cost.cpp), not a list of shapes (round 5)
// ------------------------------------------------------------------------------------
// the streaming kernel's candidates: the planner's own segments, then interleaved runs of r-row segments
struct StreamCandidate { int rows, interleave; double us; int occ = 1; };

// plan_wstream succeeded: the plan runs the weight-streaming kernel
static void use_wstream_plan(HostPlan& p) {
  const lce_hip_bconv2d_desc& d = p.d;
  const MfmaCfg want = *mfma_cfg_by_tile(128, 64);
  const bool repack = p.wq.empty() || p.mfma.bn() != want.bn() || p.wq_layout != 1 || p.kch != stream_chunks(d);
  p.wq_layout = 1;                       // tile-major: [32-channel tile][K-step][K-half][32 x 16 B] (pack_for_mfma)
  p.mfma = want;
  p.use_mfma = true;
  p.use_wstream = true;
  p.use_stream = false;
  p.use_tiled = false;
  p.cpad = stream_chunks(d) * 64;
  p.npad = ceil_div(d.channels_out, 64) * 64;
  p.hp = (int)std::max<int64_t>(p.pad_h + d.in_height, (int64_t)(p.out_h - 1) * d.stride_height + d.filter_height);
  if (repack && p.have_weights) pack_for_mfma(p);
  char nm[96];
  snprintf(nm, sizeof nm, "bconv2d_wstream<%s,3x3x%d,images%d,blocks%d>",
           d.dst_type == LCE_HIP_F32 ? "f32" : d.dst_type == LCE_HIP_I8 ? "i8" : "bitpacked", 64 * stream_chunks(d), p.ws_ipb, p.ws_nb);
  p.kernel_name = nm;
}

static std::string select_kernel_impl(HostPlan& p, int64_t pixels);

std::string select_kernel(HostPlan& p, int64_t pixels) {
  const std::string err = select_kernel_impl(p, pixels);
  // int8 plans of the matrix-core kernels: the epilogue's constants follow the kernel that was selected (its arithmetic decides what
  // can be proven); cheap (a bisection per channel), so simply redone at every selection
  if (err.empty() && p.d.dst_type == LCE_HIP_I8 && p.use_mfma && p.have_weights && !p.wq.empty()) prepare_int8_epilogue(p);
  return err;
}

static std::string select_kernel_impl(HostPlan& p, int64_t pixels) {
  const lce_hip_bconv2d_desc& d = p.d;
  const bool bp = d.dst_type == LCE_HIP_BITPACKED;
  p.ch = (p.cwg % 4 == 0) ? 4 : (p.cwg % 2 == 0) ? 2 : 1;

  // ---- engine: matrix cores vs xor-popcount VALU ----
  p.use_mfma = false;
  p.use_direct = false;
  p.use_pointwise = false;
  if (p.engine_pref == 4 && !pointwise_supported(p, pixels, &p.pw_nc, &p.pw_nj))
    return "bconv2d: the pointwise kernel runs 1x1 ungrouped convolutions with 64, 128, 256 or 512 input channels (after padding to 64) "
           "and a multiple of 32 output channels (pointwise_channels must divide them)";
  if (p.engine_pref >= 2 && !mfma_supported(p))
    return "bconv2d: the matrix-core engine cannot run this convolution (channels per group not a multiple of 64, or too deep)";
  p.use_stream = false;
  p.use_wstream = false;
  p.est_us = -1.0;
  if (p.engine_pref == 6) {
    // weight-streaming kernel (activations stationary in LDS): the planner's FP4 weight image with 64-channel granularity
    const int batch_chunk = (int)std::max<int64_t>(1, pixels / std::max<int64_t>(1, (int64_t)p.out_h * p.out_w));
    const std::string err = plan_wstream(p, batch_chunk);
    if (!err.empty()) return err;
    p.est_us = estimate_wstream_us(p, batch_chunk);
    use_wstream_plan(p);
    return "";
  }
  const bool auto_rule = p.engine_pref == 0 && p.kernel_pref == 0 && p.tile_pref.tm == 0;
  bool gemm_by_estimate = false;      // the block GEMM was priced cheapest among the matrix-core kernels: not the xor-popcount engine then
  if ((p.engine_pref == 5 || auto_rule) && stream_supported(p)) {
    // The streaming family: every candidate is planned and priced (estimate_*_us above); the cheapest runs.  engine=stream
    // restricts the choice to the weight-stationary kernel's own variants; stream_rows / stream_interleave pin theirs.
    const int batch_chunk = (int)std::max<int64_t>(1, pixels / std::max<int64_t>(1, (int64_t)p.out_h * p.out_w));
    const int rows_pref = p.stream_rows_pref, il_pref = p.stream_interleave_pref;
    std::vector<StreamCandidate> cands;
    if (il_pref <= 0) cands.push_back(StreamCandidate{rows_pref, 0, 0.0});
    if (il_pref != 0)
      for (int r = p.out_h - 1; r >= 2; --r) {
        if (p.out_h % r != 0 || (rows_pref != 0 && rows_pref != r)) continue;
        // (segments that would leave more than 15 % of their last pixel block empty are not worth pricing: every block of
        //  such a run pays the padded matrix work and the out-of-line stores)
        const int px = r * p.out_w;
        if (rows_pref == 0 && ceil_div(px, 32) * 32 * 100 > px * 115) continue;
        cands.push_back(StreamCandidate{r, 1, 0.0});
      }
    // (nothing to interleave: one segment per image)
    if (il_pref == 1 && cands.empty()) cands.push_back(StreamCandidate{rows_pref, 1, 0.0});
    // every candidate again with two blocks per CU where the instance is compiled for that (a candidate whose two blocks' LDS do not
    // fit a CU is refused by plan_stream and drops out); stream_blocks_per_cu pins the choice
    const int occ_pref = p.stream_occ_pref;
    if (occ_pref == 2 || (occ_pref == 0 && stream_blocks_per_cu_max(p) >= 2)) {
      const size_t n1 = cands.size();
      for (size_t i = 0; i < n1; ++i) { cands.push_back(cands[i]); cands.back().occ = 2; }
      if (occ_pref == 2) cands.erase(cands.begin(), cands.begin() + (long)n1);
    }
    int best = -1;
    std::string first_err;
    for (size_t i = 0; i < cands.size(); ++i) {
      p.stream_rows_pref = cands[i].rows;
      p.stream_interleave_pref = cands[i].interleave;
      p.stream_occ_pref = cands[i].occ;
      const std::string err = plan_stream(p, batch_chunk);
      if (!err.empty()) { if (first_err.empty()) first_err = err; cands[i].us = -1.0; continue; }
      if (cands[i].interleave && p.st_gstr <= 1) { cands[i].us = -1.0; continue; }      // (one segment per block: the same plan as without)
      cands[i].us = estimate_stream_us(p, batch_chunk);
      if (best < 0 || cands[i].us < cands[best].us) best = (int)i;
    }
    if (best < 0 && il_pref == 1) {
      // stream_interleave=1 and nothing to interleave (every interleaved candidate came out as one segment per block, or could not be
      // planned): the consecutive-segment plan is the same launch -- take it instead of refusing engine=stream / dropping the family
      cands.push_back(StreamCandidate{rows_pref, 0, 0.0});
      cands.back().occ = occ_pref == 2 ? 2 : 1;
      p.stream_rows_pref = rows_pref;
      p.stream_interleave_pref = 0;
      p.stream_occ_pref = cands.back().occ;
      const std::string err = plan_stream(p, batch_chunk);
      if (err.empty()) { cands.back().us = estimate_stream_us(p, batch_chunk); best = (int)cands.size() - 1; }
      else { cands.back().us = -1.0; if (first_err.empty()) first_err = err; }
    }
    p.stream_rows_pref = rows_pref;
    p.stream_interleave_pref = il_pref;
    p.stream_occ_pref = occ_pref;
    double best_us = best >= 0 ? cands[best].us : 1e30;
    bool take_wstream = false;
    // (LCE_PLAN_DEBUG, read at plan creation: the estimates of every candidate, on stderr: tools/planner_regret.py)
    const bool debug = p.dbg_level >= 1;
    if (debug)
      for (const StreamCandidate& c : cands)
        fprintf(stderr, "[lce plan] stream rows=%d il=%d blocks/CU=%d: %.2f us\n", c.rows, c.interleave, c.occ, c.us);
    if (auto_rule) {
      // (LCE_PLAN_NO_WSTREAM, read at plan creation: an A/B aid)
      if (wstream_supported(p) && !p.dbg_no_wstream && plan_wstream(p, batch_chunk).empty()) {
        const double us = estimate_wstream_us(p, batch_chunk);
        if (debug) fprintf(stderr, "[lce plan] wstream images=%d blocks=%d: %.2f us\n", p.ws_ipb, p.ws_nb, us);
        if (us < best_us) { best_us = us; take_wstream = true; }
      }
      const double gemm_us = estimate_block_gemm_us(p, pixels);
      if (debug) fprintf(stderr, "[lce plan] block GEMM: %.2f us\n", gemm_us);
      if (gemm_us < best_us) { best = -1; take_wstream = false; best_us = -1.0; gemm_by_estimate = true; }   // the block GEMM, below
    }
    if (take_wstream) {
      p.est_us = best_us;
      use_wstream_plan(p);
      return "";
    }
    if (best >= 0 && best_us >= 0.0) {
      p.est_us = best_us;
      p.stream_rows_pref = cands[best].rows;
      p.stream_interleave_pref = cands[best].interleave;
      p.stream_occ_pref = cands[best].occ;
      const std::string err = plan_stream(p, batch_chunk);
      p.stream_rows_pref = rows_pref;
      p.stream_interleave_pref = il_pref;
      p.stream_occ_pref = occ_pref;
      if (!err.empty()) return err;     // (cannot happen: the same plan succeeded a moment ago)
      const MfmaCfg want = *mfma_cfg_by_tile(128, 64);
      const bool repack = p.wq.empty() || p.mfma.bn() != want.bn() || p.wq_layout != 0 || p.kch != stream_chunks(d);
      p.wq_layout = 0;
      p.mfma = want;
      p.use_mfma = true;
      p.use_stream = true;
      p.use_tiled = false;
      p.cpad = stream_chunks(d) * 64;
      p.npad = ceil_div(d.channels_out, 64) * 64;
      p.hp = (int)std::max<int64_t>(p.pad_h + d.in_height, (int64_t)(p.out_h - 1) * d.stride_height + d.filter_height);
      if (repack && p.have_weights) pack_for_mfma(p);
      char nm[96];
      char ph[40] = "";
      if ((4 >> p.st_pph_log) < std::min(4, ceil_div(d.channels_out, 64))) snprintf(ph, sizeof ph, ",phases%d", 1 << p.st_pph_log);
      if (p.st_nstrip > 1) snprintf(ph + strlen(ph), sizeof ph - strlen(ph), ",strips%d", p.st_wso);
      if (p.st_gstr > 1) snprintf(ph + strlen(ph), sizeof ph - strlen(ph), ",il");
      if (p.st_occ > 1) snprintf(ph + strlen(ph), sizeof ph - strlen(ph), ",x%d", p.st_occ);
      snprintf(nm, sizeof nm, "bconv2d_stream<%s,3x3x%d,rows%d%s>",
               d.dst_type == LCE_HIP_F32 ? "f32" : d.dst_type == LCE_HIP_I8 ? "i8" : "bitpacked", 64 * stream_chunks(d), p.st_rs, ph);
      p.kernel_name = nm;
      return "";
    }
    if (p.engine_pref == 5) return first_err.empty() ? std::string("bconv2d: the streaming kernel cannot run this launch") : first_err;
  } else if (p.engine_pref == 5) {
    return plan_stream(p, 1);      // (the message that says why)
  }
  if (p.engine_pref >= 2 || (p.engine_pref == 0 && p.kernel_pref == 0 && p.tile_pref.tm == 0 &&
                             mfma_supported(p) && (gemm_by_estimate || pixels * d.channels_out >= (1 << 16)))) {
    p.use_mfma = true;
    p.use_tiled = false;
    p.est_us = estimate_block_gemm_us(p, pixels);
    MfmaCfg want = choose_mfma_cfg(p, pixels);
    bool direct = false;
    if (p.engine_pref >= 2 && p.tile_pref.tm != 0) {
      const MfmaCfg* forced = mfma_cfg_by_tile(p.tile_pref.tm, p.tile_pref.tn);
      if (!forced) return "bconv2d: no matrix-core kernel instance for the requested block tile";
      if (d.groups > 1 && p.npg % forced->bn())
        return "bconv2d: the requested block tile would straddle channel groups";
      want = *forced;
      direct = p.engine_pref == 3;
    } else if (p.engine_pref != 2) {
      // auto / engine=direct without a tile: the direct variant when a good tile exists
      MfmaCfg dc;
      // (auto: a launch of at most half a round of blocks with a deep K loop -- >= 27 K-steps: 3x3 over 192+ channels -- runs at the
      //  latency of one block's K loop, and the workspace GEMM's is shorter: its A fragments come from the expanded workspace, no halo
      //  expansion in front of the first MFMA.  profiles/r05/engine_sweep_*.jsonl, batch 1 / 16: direct / workspace = 1.03 ... 1.26 on
      //  every such row, 0.99 at worst; with more blocks the direct variant wins by 10 ... 50 %.)
      const int ks_deep = d.filter_height * d.filter_width * ceil_div(d.channels_in / std::max(1, d.groups), 64);
      auto tiny_deep = [&](const MfmaCfg& c) {
        return p.engine_pref == 0 && ks_deep >= 27 && ((pixels + c.bm() - 1) / c.bm()) * ceil_div(d.channels_out, c.bn()) <= 128;
      };
      if (choose_direct_cfg(p, &dc) && !tiny_deep(dc)) {
        want = dc;
        direct = true;
      } else if (p.engine_pref == 3) {
        // forced: take any tile whose halo fits, whatever the padding waste
        for (int bm : {128, 256}) {
          const MfmaCfg* c = mfma_cfg_by_tile(bm, d.channels_out > 64 && (d.groups == 1 || p.npg % 128 == 0) ? 128 : 64);
          int a, b, c2, e, f, ttx, hw;   // (with the 2-D tile outputs: a wide image's strip may not fit where its 2-D tile does)
          if (c && direct_geometry(p, *c, &a, &b, &c2, &e, &f, kDirectLdsMax, &ttx, &hw)) { want = *c; direct = true; break; }
        }
        if (!direct) direct = true;  // reported below by direct_geometry
      }
    }
    const bool repack = p.wq.empty() || p.mfma.bn() != want.bn() || p.wq_layout != 0 ||
                        p.kch != (d.groups > 1 ? group_chunks(d) : ceil_div(d.channels_in, 64));
    p.wq_layout = 0;
    p.mfma = want;
    p.cpad = ceil_div(d.channels_in, 64) * 64;
    p.npad = ceil_div(d.channels_out, want.bn()) * want.bn();
    p.hp = (int)std::max<int64_t>(p.pad_h + d.in_height,
                                  (int64_t)(p.out_h - 1) * d.stride_height + (d.filter_height - 1) * d.dilation_height + 1);
    p.wp = (int)std::max<int64_t>(p.pad_w + d.in_width,
                                  (int64_t)(p.out_w - 1) * d.stride_width + (d.filter_width - 1) * d.dilation_width + 1);
    if (repack && p.have_weights) pack_for_mfma(p);
    if (direct) {
      if (!direct_geometry(p, want, &p.tpi, &p.halo_rows, &p.ps, &p.halo_bytes, &p.ipt,
                           p.engine_pref == 3 ? kDirectLdsMax : kDirectLdsAuto, &p.tile_tx, &p.halo_w))
        return "bconv2d: the direct matrix-core variant cannot hold this tile's input halo in LDS";
      p.use_direct = true;
    }
    char nm[96];
    if (!direct) { p.tile_tx = 0; p.halo_w = p.wp; }
    snprintf(nm, sizeof nm, "bconv2d_mfma%s<%s,%dx%d>%s", p.use_direct ? "_direct" : "",
             d.dst_type == LCE_HIP_F32 ? "f32" : d.dst_type == LCE_HIP_I8 ? "i8" : "bitpacked", want.bm(), want.bn(),
             p.use_direct && p.tile_tx > 0 ? "/2d" : "");
    p.kernel_name = nm;
    // 1x1 stride-1 layers: the streaming kernel on top of the same weight image (the block GEMM stays the
    // fallback for output pointers that are not 16-byte aligned)
    if ((p.engine_pref == 4 || (p.engine_pref == 0 && p.tile_pref.tm == 0)) && pointwise_supported(p, pixels, &p.pw_nc, &p.pw_nj) &&
        (p.engine_pref == 4 || pointwise_preferred(p, pixels))) {
      p.use_pointwise = true;
      snprintf(nm, sizeof nm, "bconv2d_pointwise<%s,K%dx64,N%dx32%s>",
               d.dst_type == LCE_HIP_F32 ? "f32" : d.dst_type == LCE_HIP_I8 ? "i8" : "bitpacked", p.pw_nc, p.pw_nj,
               d.stride_height != 1 || d.stride_width != 1 ? ",strided" : "");
      p.kernel_name = nm;
    }
    return "";
  }

  TileShape chosen{0, 0};
  if (p.kernel_pref != 2) {
    if (p.tile_pref.tm != 0) {
      if (tiled_supports(p, p.tile_pref.tn)) chosen = p.tile_pref;
      else if (p.kernel_pref == 1) return "bconv2d: the requested tile cannot run this convolution";
    } else {
      // One wave task = 64*TM pixels x TN channels.  Measured on MI355X
      // (profiles/r01/tile_sweep_v8.jsonl): one pixel per lane wins on every BASELINE layer --
      // the bigger accumulator tiles (4x16, 2x32) run out of scalar registers for the weight
      // words and spill -- with 32 channels per task on long launches (L0: 1x32 0.80 ms vs 4x16
      // 0.90) and 16 on short ones (14x14x256: 1x16 0.060 vs 1x32 0.071; 7x7x512: 0.064 vs 0.085).
      const TileShape order_f[] = {{1, 32}, {1, 16}, {2, 16}, {2, 32}, {4, 16}};
      const TileShape order_b[] = {{1, 32}, {2, 32}};
      const TileShape* order = bp ? order_b : order_f;
      const int count = bp ? 2 : 5;
      for (int k = 0; k < count && chosen.tm == 0; ++k) {
        const TileShape t = order[k];
        if (!tiled_supports(p, t.tn)) continue;
        const int64_t tasks = ((pixels + 64 * t.tm - 1) / (64 * t.tm)) * ceil_div(d.channels_out, t.tn);
        if (!bp && k == 0 && tasks < 32768 && tiled_supports(p, 16)) continue;   // short launch: 1x16
        chosen = t;
      }
      if (chosen.tm == 0) {  // nothing large enough: take the smallest supported tile
        for (int k = count - 1; k >= 0 && chosen.tm == 0; --k)
          if (tiled_supports(p, order[k].tn)) chosen = order[k];
      }
    }
  }
  if (chosen.tm == 0 && p.kernel_pref == 1)
    return "bconv2d: the tiled kernel cannot run this convolution (grouped, channels per group "
           "not a multiple of the tile)";

  const bool tiled = chosen.tm != 0;
  char name[96];
  if (tiled) {
    const bool repack = !p.use_tiled || p.tile.tn != chosen.tn || p.packed.empty();
    p.use_tiled = true;
    p.tile = chosen;
    if (repack && p.have_weights) pack_for_tile(p);
    p.nt = ceil_div(d.channels_out, chosen.tn);
    snprintf(name, sizeof name, "bconv2d_tiled<%s,TM=%d,TN=%d,CH=%d>",
             d.dst_type == LCE_HIP_F32 ? "f32" : d.dst_type == LCE_HIP_I8 ? "i8" : "bitpacked",
             chosen.tm, chosen.tn, p.ch);
  } else {
    p.use_tiled = false;
    p.tile = TileShape{0, 0};
    snprintf(name, sizeof name, "bconv2d_general<%s>",
             d.dst_type == LCE_HIP_F32 ? "f32" : d.dst_type == LCE_HIP_I8 ? "i8" : "bitpacked");
  }
  p.kernel_name = name;
  return "";
}

int max_batch_per_launch(const HostPlan& p) {
  const int64_t per_image_bytes = (int64_t)p.d.in_height * p.d.in_width * p.cw * 4;
  const int64_t per_image_pixels = (int64_t)p.out_h * p.out_w;
  int64_t by_bytes = ((1ll << 31) - 1) / per_image_bytes;
  int64_t by_pixels = ((1ll << 31) - 64 * 4 - 1) / per_image_pixels;
  int64_t b = std::min(by_bytes, by_pixels);
  if (p.engine_pref != 1 && mfma_supported(p)) {
    // the FP4 workspace is 4x the bitpacked input (plus the halo) and is indexed in
    // 16-byte chunks by a 32-bit counter
    const int cpad = ceil_div(p.d.channels_in, 64) * 64;
    const int64_t hp = std::max<int64_t>(p.pad_h + p.d.in_height,
                                         (int64_t)(p.out_h - 1) * p.d.stride_height + (p.d.filter_height - 1) * p.d.dilation_height + 1);
    const int64_t wp = std::max<int64_t>(p.pad_w + p.d.in_width,
                                         (int64_t)(p.out_w - 1) * p.d.stride_width + (p.d.filter_width - 1) * p.d.dilation_width + 1);
    b = std::min<int64_t>(b, (int64_t)(((1ll << 31) - 1) / (hp * wp * (cpad / 2))));
    // the pointwise and the streaming kernel address a whole launch's output through ONE buffer resource with 32-bit
    // byte offsets (an out-of-range marker is offset 2^31): a launch's output stays below 2 GiB
    const int64_t out_row = p.d.dst_type == LCE_HIP_BITPACKED ? (int64_t)p.wout * 4
                                                               : (int64_t)p.d.channels_out * (p.d.dst_type == LCE_HIP_I8 ? 1 : 4);
    b = std::min<int64_t>(b, ((1ll << 31) - 1) / std::max<int64_t>(1, per_image_pixels * out_row));
  }
  b = std::max<int64_t>(1, std::min<int64_t>(b, p.d.batch));
  return (int)b;
}

ConvArgs make_conv_args(const HostPlan& p, int batch_chunk) {
  const lce_hip_bconv2d_desc& d = p.d;
  ConvArgs A{};
  A.H = d.in_height; A.W = d.in_width; A.Cw = p.cw; A.Cwg = p.cwg;
  A.OH = p.out_h; A.OW = p.out_w; A.N = d.channels_out; A.Npg = p.npg;
  A.KH = d.filter_height; A.KW = d.filter_width;
  A.SH = d.stride_height; A.SW = d.stride_width;
  A.DH = d.dilation_height; A.DW = d.dilation_width;
  A.PH = p.pad_h; A.PW = p.pad_w;
  A.M = batch_chunk * p.out_h * p.out_w;
  A.NT = p.nt;
  A.PT = p.use_tiled ? (A.M + 64 * p.tile.tm - 1) / (64 * p.tile.tm) : 0;
  A.Wout = p.wout;
  A.in_bytes = (uint32_t)((int64_t)batch_chunk * d.in_height * d.in_width * p.cw * 4);
  A.div_ow = make_fastdiv((uint32_t)p.out_w);
  A.div_oh = make_fastdiv((uint32_t)p.out_h);
  A.clamp_min = p.clamp_min; A.clamp_max = p.clamp_max;
  A.zero_pad_mode = p.zero_pad_mode;
  A.bzp = (d.channels_in / d.groups) / 2;
  A.eKH = (d.filter_height - 1) * d.dilation_height + 1;
  A.eKW = (d.filter_width - 1) * d.dilation_width + 1;
  A.left_off = ((p.out_w - 1) * d.stride_width + A.eKW - d.in_width) / 2;   // zero_padding_correction.h:189-191
  A.top_off = ((p.out_h - 1) * d.stride_height + A.eKH - d.in_height) / 2;  // :192-194
  return A;
}

}  // namespace lce
