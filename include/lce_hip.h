/*
 * lce_hip.h -- C ABI of the MI355X-native LceBconv2d / LceQuantize hot path.
 *
 * This is the drop-in boundary under Larq Compute Engine's TFLite custom-op glue: the
 * C++ op glue (compute-engine_amd/csrc/tflite/, same Register_* entry points as the
 * reference) and any other host language bind THESE functions -- plain pointers,
 * sizes and int status codes, no C++/torch/HIP types in any signature.  Each entry
 * point names the reference interface it replaces (paths relative to
 * /root/reference/larq_compute_engine/).
 *
 * Conventions
 *   - every function returns LCE_HIP_OK (0) or an error code; lce_hip_last_error()
 *     returns a thread-local human-readable message for the last failure;
 *   - `*_dev` pointers are device (HBM) pointers of the current HIP device, `*_host`
 *     pointers are ordinary host memory; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream);
 *   - tensors use the reference's layouts: activations NHWC with channels bitpacked
 *     into int32 words LSB-first (core/types.h:41, core/bitpacking/bitpack.h:72-110),
 *     filters OHWI bitpacked (tflite/kernels/bconv2d.cc:145-152), outputs NHWC;
 *   - there is NO CPU fallback: without a usable GPU the compute entry points fail
 *     with LCE_HIP_ERR_NO_DEVICE.
 */
#ifndef LCE_HIP_H_
#define LCE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCE_HIP_ABI_VERSION 3   /* 3 (round 6): + lce_hip_bconv2d_plan_int8_epilogue; additive */

typedef enum lce_hip_status {
  LCE_HIP_OK = 0,
  LCE_HIP_ERR_INVALID = 1,     /* bad argument / rejected by the same checks as Prepare */
  LCE_HIP_ERR_UNSUPPORTED = 2, /* valid for the reference but outside what this build runs */
  LCE_HIP_ERR_RUNTIME = 3,     /* a HIP call failed */
  LCE_HIP_ERR_NO_DEVICE = 4    /* no gfx950 device / HIP runtime unusable */
} lce_hip_status;

/* element types of the tensors that cross the boundary */
typedef enum lce_hip_dtype {
  LCE_HIP_F32 = 0,       /* kTfLiteFloat32 */
  LCE_HIP_I8 = 1,        /* kTfLiteInt8    */
  LCE_HIP_BITPACKED = 2, /* kTfLiteInt32 holding 32 sign bits (TBitpacked, core/types.h:41) */
  LCE_HIP_BOOL = 3       /* kTfLiteBool (1 byte) */
} lce_hip_dtype;

/* tflite schema enums as they arrive in the op's flexbuffer options
 * (tflite/kernels/bconv2d.cc:94-124, tflite/kernels/utils.h:10-35) */
typedef enum lce_hip_padding { LCE_HIP_PADDING_SAME = 0, LCE_HIP_PADDING_VALID = 1 } lce_hip_padding;
typedef enum lce_hip_activation {
  LCE_HIP_ACT_NONE = 0, LCE_HIP_ACT_RELU = 1, LCE_HIP_ACT_RELU_N1_TO_1 = 2, LCE_HIP_ACT_RELU6 = 3
} lce_hip_activation;

/* Which reference registration's semantics to follow.  They only differ for SAME
 * padding with pad_values == 0 (tflite/kernels/bconv2d.cc:188-200):
 *   REFERENCE : Register_BCONV_2D_REF -- exact integer zero padding
 *               (core/bconv2d/reference.h:76-103); needs an even channels_in.
 *   OPTIMIZED : Register_BCONV_2D_OPT_BGEMM / _OPT_INDIRECT_BGEMM -- one-padded
 *               convolution plus float correction (core/bconv2d/zero_padding_correction.h);
 *               float output and no fused activation only. */
typedef enum lce_hip_semantics { LCE_HIP_SEM_REFERENCE = 0, LCE_HIP_SEM_OPTIMIZED = 1 } lce_hip_semantics;

/* ------------------------------------------------------------------------------------
 * Library / device
 * ---------------------------------------------------------------------------------- */
int lce_hip_abi_version(void);
const char* lce_hip_last_error(void);
/* "product", or "experiment" for a library built with measuring aids compiled in (timing ablations whose results are
 * wrong by construction, time-stamp builds: csrc/lce_experiments.h).  The library csrc/Makefile builds is "product" and
 * cannot be anything else (the switches are compile errors there); a host may refuse to load anything else. */
const char* lce_hip_build_flavor(void);
/* number of usable HIP devices (0 when there is none; never fails) */
int lce_hip_device_count(void);
lce_hip_status lce_hip_set_device(int device);

/* Device-memory plumbing for hosts that have no HIP binding of their own (the TFLite
 * glue stages interpreter tensors through these). */
lce_hip_status lce_hip_malloc(void** dev_ptr, size_t bytes);
lce_hip_status lce_hip_free(void* dev_ptr);
lce_hip_status lce_hip_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
lce_hip_status lce_hip_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
lce_hip_status lce_hip_memset(void* dst_dev, int value, size_t bytes, void* stream);
/* Page-locks [host_ptr, host_ptr + bytes) (hipHostRegister) so that copies from / to it are truly
 * asynchronous and lce_hip_bconv2d_run_host can overlap them with compute.  For buffers the caller keeps
 * alive and reuses (the interpreter's tensor arena); the range must be unregistered before it is freed. */
lce_hip_status lce_hip_host_register(void* host_ptr, size_t bytes);
lce_hip_status lce_hip_host_unregister(void* host_ptr);
lce_hip_status lce_hip_stream_create(void** stream);
lce_hip_status lce_hip_stream_destroy(void* stream);
lce_hip_status lce_hip_stream_synchronize(void* stream);
/* HIP graphs for hosts without a HIP binding: everything launched on `stream` (a stream from lce_hip_stream_create, not the
 * null stream) between begin and end is recorded instead of executed -- lce_hip_bconv2d_run / _run_dual of plans that have
 * run once before, lce_hip_bitpack / _unpack / _bmaxpool -- and comes back as ONE launchable object: a chain of short layers
 * replays without a host call per kernel.  The recorded launches keep the device pointers they were given.  A plan's first run
 * (uploads, a known-answer check) cannot be recorded: run the sequence once eagerly first.  end_capture returns the error of
 * a failed recording and leaves the stream usable. */
lce_hip_status lce_hip_graph_begin_capture(void* stream);
lce_hip_status lce_hip_graph_end_capture(void* stream, void** graph);
lce_hip_status lce_hip_graph_launch(void* graph, void* stream);
lce_hip_status lce_hip_graph_destroy(void* graph);

/* ------------------------------------------------------------------------------------
 * LceQuantize / LceDequantize
 * ---------------------------------------------------------------------------------- */

/* ceil(n / 32): core/bitpacking/bitpack.h:24-26 (GetBitpackedSize) */
int32_t lce_hip_bitpacked_size(int32_t unpacked_elements);

/* Replaces core::bitpacking::bitpack_matrix<T> (core/bitpacking/bitpack.h:248-308) as
 * called by QuantizeEval (tflite/kernels/quantization.cc:76-114): packs each of the
 * `rows` rows of `cols` elements into ceil(cols/32) words; bit = (x < zero_point)
 * (float: zero_point must be 0, test is x < 0; bool: pass zero_point 1), padding bits 0.
 * in_type is F32, I8 or BOOL. */
lce_hip_status lce_hip_bitpack(lce_hip_dtype in_type, const void* in_dev, size_t rows,
                               size_t cols, int32_t zero_point, int32_t* out_dev, void* stream);

/* Replaces core::bitpacking::unpack_matrix<T> (bitpack.h:327-346) as called by
 * DequantizeEval (quantization.cc:116-147).  out_type F32: bit0 -> +1, bit1 -> -1;
 * I8: zero_point +- round(1/scale) clamped to int8; BOOL: bit0 -> true. */
lce_hip_status lce_hip_unpack(lce_hip_dtype out_type, const int32_t* in_dev, size_t rows,
                              size_t cols, float scale, int32_t zero_point, void* out_dev,
                              void* stream);

/* ------------------------------------------------------------------------------------
 * LceBconv2d
 * ---------------------------------------------------------------------------------- */

/* The op attributes + tensor metadata that bconv2d::Init/Prepare collect
 * (tflite/kernels/bconv2d.cc:85-131,137-300; core/bconv2d/params.h:12-32). */
typedef struct lce_hip_bconv2d_desc {
  int32_t batch, in_height, in_width;
  int32_t channels_in;    /* unpacked input channels (attribute "channels_in") */
  int32_t filter_height, filter_width;
  int32_t channels_out;
  int32_t groups;         /* the glue infers it from the filter shape, bconv2d.cc:169-186 */
  int32_t stride_height, stride_width;
  int32_t dilation_height, dilation_width;
  int32_t padding;        /* lce_hip_padding    */
  int32_t pad_values;     /* 0 or 1             */
  int32_t activation;     /* lce_hip_activation */
  int32_t dst_type;       /* LCE_HIP_F32 / LCE_HIP_I8 / LCE_HIP_BITPACKED */
  int32_t semantics;      /* lce_hip_semantics  */
  float out_scale;        /* int8 output: output->params.scale      */
  int32_t out_zero_point; /* int8 output: output->params.zero_point */
} lce_hip_bconv2d_desc;

typedef struct lce_hip_bconv2d_plan lce_hip_bconv2d_plan;

/* Validates the descriptor with the same rules as bconv2d::Prepare (bconv2d.cc:137-300)
 * and infers padding and output shape (TFLite ComputePaddingHeightWidth, :203-210).
 * Host-only: does not touch the GPU, so shape inference works anywhere. */
lce_hip_status lce_hip_bconv2d_plan_create(const lce_hip_bconv2d_desc* desc,
                                           lce_hip_bconv2d_plan** plan);
void lce_hip_bconv2d_plan_destroy(lce_hip_bconv2d_plan* plan);

/* [batch, out_height, out_width, channels_out or ceil(channels_out/32)] (bconv2d.cc:241-248) */
lce_hip_status lce_hip_bconv2d_plan_output_shape(const lce_hip_bconv2d_plan* plan, int32_t dims[4]);
/* padding_values.{height,width} as Prepare stores them (core/bconv2d/params.h:29-31) */
lce_hip_status lce_hip_bconv2d_plan_padding(const lce_hip_bconv2d_plan* plan, int32_t* pad_h,
                                            int32_t* pad_w);

/* Replaces OneTimeSetup (bconv2d.cc:324-392) + indirect_bgemm::Kernel::PackWeights
 * (core/indirect_bgemm/kernel.h:54-94): folds post_activation_{multiplier,bias}, the int8
 * scale/zero-point and the fused activation into the output transform, precomputes the
 * zero-padding correction, repacks the OHWI filter for the kernel.  All pointers are
 * HOST memory and are copied; `thresholds` is required (and the float arrays ignored)
 * iff dst_type is BITPACKED.  Host-only; the upload happens on first run. */
lce_hip_status lce_hip_bconv2d_plan_set_weights(lce_hip_bconv2d_plan* plan,
                                                const int32_t* filter_ohwi_host,
                                                const float* post_activation_multiplier_host,
                                                const float* post_activation_bias_host,
                                                const int32_t* thresholds_host);

/* The folded transform, for inspection: mul/bias have channels_out entries. */
lce_hip_status lce_hip_bconv2d_plan_folded(const lce_hip_bconv2d_plan* plan, float* mul,
                                           float* bias, int32_t* clamp_min, int32_t* clamp_max);

/* Tuning/testing knobs (defaults are all "auto"):
 *   "engine" = "auto" | "valu" (v_xor + v_bcnt popcount kernels) | "mfma" (FP4 matrix cores, FP4
 *              workspace + GEMM whose tiles span images) | "direct" (FP4 matrix cores, each block
 *              expands its own input halo into LDS; what "auto" picks whenever it fits) | "pointwise" (1x1
 *              ungrouped layers of any stride, 64 / 128 / 256 / 512 input channels after padding to 64, a multiple
 *              of 32 output channels: filter bank
 *              in registers, waves stream 32-pixel tiles; what "auto" picks for such layers)
 *              | "stream" (ungrouped 3x3 layers without dilation, up to 512 input channels -- on the 64- / 128- / 256- / 512-channel instance: persistent
 *              blocks, the filter bank resident in registers, input rows expanded once into an LDS ring; its cheapest variant by the
 *              planner's estimate) | "wstream" (the same layers from 65 input channels on: activations stationary in LDS, weights
 *              streamed into registers during the K loop; for launches of a few block steps).  "auto" prices every kernel that can
 *              run the layer -- the streaming kernel's variants, the weight-streaming kernel, the block GEMM -- and takes the
 *              cheapest (csrc/lce_plan.cpp, estimate_*_us; LCE_PLAN_DEBUG=1 in the environment prints the prices);
 *   "stream_rows" = "0" (auto) | rows per segment (a divisor of the output height), "stream_interleave" = "auto" | "0" | "1"
 *              (a block owns segments b, b + grid, ... instead of consecutive ones), "stream_strip" = "-1" (auto) | "0" | a strip width,
 *              "stream_blocks_per_cu" = "auto" | "1" | "2" (two resident blocks per CU: the bitpacked-output instance of the 64-input-channel
 *              bank, where both blocks' LDS fit),
 *              "stream_pixel_phases", "stream_flat", "compute_units", "wstream_blocks" = "0".."4", "wstream_images": tuning / testing
 *              aids of the two streaming kernels;
 *   "int8_rounding" = "auto" | "exact": int8 outputs of the streaming / weight-streaming / pointwise kernels round with floor(y + 0.5)
 *              and (the first two) transform with one fma -- one instruction each -- on plans where the planner proves the bytes equal
 *              to the reference's two roundings + round-half-away for every value the accumulator can take (csrc/lce_plan.cpp,
 *              prepare_int8_epilogue), and run the reference's own sequence otherwise; "exact": always the latter;
 *   "kernel" = "auto" | "tiled" | "general"                        (valu engine);
 *   "tile"   = "auto" | valu lane tile "4x16"|"2x32"|"2x16"|"1x32"|"1x16"
 *                     | matrix-core block tile "256x256"|"256x128"|"512x64"|"128x256"|"128x128"|"256x64"|"128x64"
 *                       (pixels x channels; with engine = mfma or direct);
 *   "phase"  = "all" | "expand" | "gemm"   (engine = mfma, profiling aid: run one of its two kernels);
 *   "epilogue" = "auto" | "tile" | "wide"   (matrix-core float / int8 epilogue: per-tile or joint transpose);
 *   "pointwise_tiles" = "0" (auto) | "1".."8"   (pointwise kernel: 32-pixel tiles per wave);
 *   "pointwise_channels" = "0" (auto) | "32" | "64" | "128"   (pointwise kernel: output channels per block);
 *   "tile2d" = "auto" | "on" | "off"   (direct variant: 2-D tiles of BM/32 rows x 32 columns instead of row-major strips;
 *              auto takes them on wide images, where they stage <= 0.7 of the strip's halo at <= 3 % more padding). */
lce_hip_status lce_hip_bconv2d_plan_set_option(lce_hip_bconv2d_plan* plan, const char* key,
                                               const char* value);
/* Name of the kernel variant the next run will launch (static string owned by the plan). */
const char* lce_hip_bconv2d_plan_kernel_name(lce_hip_bconv2d_plan* plan);
/* The same for lce_hip_bconv2d_run_dual.  Since round 5 the choice does not depend on the kind of call (one selection per plan):
 * kept for callers of round 4's ABI, returns what lce_hip_bconv2d_plan_kernel_name returns. */
const char* lce_hip_bconv2d_plan_kernel_name_dual(lce_hip_bconv2d_plan* plan);

/* What the int8 epilogue of the kernel the next run will launch does (int8 plans with weights set; otherwise both outputs are 0):
 * *one_instruction_forms = 1 when it transforms with one fma and / or rounds with floor(y + 0.5) -- forms the planner has proven
 * byte-identical to the reference's two roundings + round-half-away (core/bconv2d/output_transform.h:31-44,125-144) for every value the
 * accumulator can take on this plan -- and 0 when it runs the reference's own sequence; *adjusted_channels = the number of output
 * channels whose folded multiplier / bias that proof replaced by NEIGHBOURING floats (<= 2 ulps / 4 grid steps; the bytes written are
 * still the reference's on the original parameters).  "int8_rounding" = "exact" forces 0 / 0.  Either pointer may be NULL. */
lce_hip_status lce_hip_bconv2d_plan_int8_epilogue(lce_hip_bconv2d_plan* plan, int32_t* one_instruction_forms,
                                                  int32_t* adjusted_channels);

/* Replaces bconv2d::Eval (bconv2d.cc:550-564) -> BConv2DReference /
 * BConv2DOptimizedBGEMM / BConv2DOptimizedIndirectBGEMM (core/bconv2d/ headers) with
 * device-resident tensors: input int32 [B,H,W,ceil(Cin/32)], output per dst_type.
 * Asynchronous on `stream`.
 *
 * Devices and streams.  A plan's weights, tables, workspace and staging buffers live on ONE HIP device:
 * the one that is current at the plan's first run (lce_hip_bconv2d_plan_device).  Running it while another
 * device is current fails with LCE_HIP_ERR_INVALID -- create one plan per device (the batch-shard mode of
 * SURVEY.md 8(e) runs one process per GPU).  A plan may be run on any stream of that device, one call at a
 * time per plan: calls from several host threads into the SAME plan must be serialised by the caller (the
 * reference's OpData is per node and TFLite invokes a node from one thread, tflite/kernels/bconv2d.cc:44-74);
 * calls on different streams are ordered by the library where they share the plan's workspace.  Different
 * plans are independent. */
lce_hip_status lce_hip_bconv2d_run(lce_hip_bconv2d_plan* plan, const int32_t* input_dev,
                                   void* output_dev, void* stream);
/* The HIP device the plan is bound to, -1 before its first run. */
int lce_hip_bconv2d_plan_device(const lce_hip_bconv2d_plan* plan);

/* LceBconv2d (float or int8 output) and the LceQuantize that follows it in a converted graph
 * (tflite/kernels/quantization.cc:76-114 on the convolution's output), in one pass: `output_dev` gets the
 * tensor [B,OH,OW,Cout] exactly as lce_hip_bconv2d_run writes it, `output_bits_dev` its quantization
 * [B,OH,OW,ceil(Cout/32)] exactly as lce_hip_bitpack(F32, output_dev, ..., 0, ...) -- sign bits -- or, for an int8
 * plan, lce_hip_bitpack(I8, output_dev, ..., out_zero_point, ...) -- bit = (q < zero_point) -- would: from the same
 * epilogue (the value a lane just produced is compared and balloted) where the kernel variant allows, by a second
 * launch on the same stream otherwise.  Saves re-reading the tensor between the binary convolutions of a
 * device-resident chain.  The plan's dst_type must be LCE_HIP_F32 or LCE_HIP_I8. */
lce_hip_status lce_hip_bconv2d_run_dual(lce_hip_bconv2d_plan* plan, const int32_t* input_dev,
                                        void* output_dev, int32_t* output_bits_dev, void* stream);

/* Same as lce_hip_bconv2d_run with host tensors (the interpreter arena), synchronous.  The batch is cut
 * into slices that flow through three streams (H2D | kernel | D2H) so that the copies of neighbouring
 * slices overlap the compute.  The overlap is complete when the two buffers are page-locked
 * (lce_hip_host_register -- an arena is reused for every Invoke, so the glue registers it once); pageable
 * buffers work too, with the runtime staging every copy. */
lce_hip_status lce_hip_bconv2d_run_host(lce_hip_bconv2d_plan* plan, const int32_t* input_host,
                                        void* output_host);

/* ------------------------------------------------------------------------------------
 * LceBMaxPool2d (core/bmaxpool.h:24-88; tflite/kernels/bmaxpool.cc:20-98)
 * ---------------------------------------------------------------------------------- */
lce_hip_status lce_hip_bmaxpool_output_shape(int32_t in_height, int32_t in_width,
                                             int32_t filter_height, int32_t filter_width,
                                             int32_t stride_height, int32_t stride_width,
                                             int32_t padding, int32_t* out_height,
                                             int32_t* out_width);
lce_hip_status lce_hip_bmaxpool(const int32_t* input_dev, int32_t batch, int32_t in_height,
                                int32_t in_width, int32_t words, int32_t filter_height,
                                int32_t filter_width, int32_t stride_height, int32_t stride_width,
                                int32_t padding, int32_t* output_dev, void* stream);

/* ------------------------------------------------------------------------------------
 * Converter-side parameter preparation (host-only).
 * What the reference's MLIR converter does offline to the constants of one binary
 * convolution, so that an unconverted Larq layer (float +-scale HWIO filter, fused
 * mul/add constants) arrives at exactly the tensors LceBconv2d expects.  WEIGHTS only --
 * nothing here processes activations.  Paths relative to larq_compute_engine/mlir/.
 * ---------------------------------------------------------------------------------- */

/* transforms/prepare_patterns_common.td:97-168 + prepare_tf.cc:40-92: checks the filter is
 * binary (+-scale[o] within 0.5 %), writes filter/|scale| transposed HWIO -> OHWI,
 * post_activation_multiplier = |scale|, post_activation_bias = 0 (either may be NULL). */
lce_hip_status lce_hip_prepare_binary_filter(const float* filter_hwio, int32_t filter_height,
                                             int32_t filter_width, int32_t channels_in_per_group,
                                             int32_t channels_out, float* filter_ohwi,
                                             float* post_activation_multiplier,
                                             float* post_activation_bias);

typedef enum lce_hip_post_op {
  LCE_HIP_POST_ADD = 0, LCE_HIP_POST_SUB = 1, LCE_HIP_POST_MUL = 2, LCE_HIP_POST_DIV = 3
} lce_hip_post_op;
/* transforms/optimize_patterns_common.td:39-118: fuse `conv <op> constant` (constant scalar
 * or per-channel) into post_activation_{multiplier,bias}, float arithmetic. */
lce_hip_status lce_hip_prepare_fuse_post_op(lce_hip_post_op op, const float* value,
                                            int32_t value_count, float* post_activation_multiplier,
                                            float* post_activation_bias, int32_t channels_out);
/* transforms/optimize_patterns_common.td:122-182: 1 iff a following Relu/Relu1/Relu6 may
 * become the fused activation (multiplier all 1, bias all 0, VALID or SAME/pad_values 1). */
int lce_hip_prepare_can_fuse_activation(const float* post_activation_multiplier,
                                        const float* post_activation_bias, int32_t channels_out,
                                        int32_t padding, int32_t pad_values);

/* transforms/bitpack_activations_patterns.td:19-60 + optimize.cc:128-244: the rewrite of
 * LceQuantize(LceBconv2d(x)) into a bit-writing convolution: multiplies filter_ohwi IN PLACE
 * by sign(multiplier) and computes the int32 thresholds (bit = accumulator > threshold). */
lce_hip_status lce_hip_prepare_bitpacked_output(float* filter_ohwi, int32_t filter_height,
                                                int32_t filter_width, int32_t channels_in_per_group,
                                                int32_t channels_out, int32_t activation,
                                                int32_t padding, int32_t pad_values,
                                                const float* post_activation_multiplier,
                                                const float* post_activation_bias,
                                                int32_t* thresholds);

/* transforms/bitpack.cc:19-57: float OHWI filter -> int32 [O][H][W][ceil(I/32)]
 * (bit = x < 0, LSB first, rows padded with 0 bits) = input 1 of LceBconv2d. */
lce_hip_status lce_hip_prepare_bitpack_filter(const float* filter_ohwi, int32_t filter_height,
                                              int32_t filter_width, int32_t channels_in_per_group,
                                              int32_t channels_out, int32_t* filter_words);

#ifdef __cplusplus
}
#endif
#endif /* LCE_HIP_H_ */
