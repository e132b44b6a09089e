/* lce_tflite_model.h -- read a converted Larq model (.tflite) and hand its LCE custom ops to
 * the GPU library.  C ABI of liblce_tflite_ops.so.
 *
 * Replaces, for the LCE part of a graph, what the reference's callers obtain from TensorFlow
 * Lite: tflite::FlatBufferModel::BuildFromBuffer + InterpreterBuilder + the interpreter's
 * tensor table (examples/lce_minimal.cc:28-40; tflite/python/interpreter_wrapper_lite.cc:40-58;
 * tflite/python/interpreter_wrapper_utils.h).  Host-only except where a plan is created.
 * The flatbuffer reader restates the published format (see
 * compute-engine_amd/csrc/tflite/tflite_flatbuffer_reader.h); no real converter output exists
 * in the build image, so parity with real files is unpinned.
 */
#ifndef LCE_TFLITE_MODEL_H_
#define LCE_TFLITE_MODEL_H_
#include <stddef.h>
#include <stdint.h>

#include "lce_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lce_tflite_model lce_tflite_model;

/* TensorType values of schema.fbs that LCE graphs use */
enum { LCE_TFLITE_FLOAT32 = 0, LCE_TFLITE_INT32 = 2, LCE_TFLITE_BOOL = 6, LCE_TFLITE_INT8 = 9 };

/* Parses `data` (kept by reference: it must outlive the model).  Returns NULL and writes a
 * message into err (if given) when the buffer is not a well-formed TFL3 flatbuffer. */
lce_tflite_model* lce_tflite_model_open(const void* data, size_t size, char* err, size_t err_len);
void lce_tflite_model_close(lce_tflite_model* model);

int32_t lce_tflite_model_num_tensors(const lce_tflite_model* model);
int32_t lce_tflite_model_num_operators(const lce_tflite_model* model);
/* Subgraph inputs / outputs (tensor indices); returns the count, fills up to `cap`. */
int32_t lce_tflite_model_inputs(const lce_tflite_model* model, int32_t* indices, int32_t cap);
int32_t lce_tflite_model_outputs(const lce_tflite_model* model, int32_t* indices, int32_t cap);

typedef struct lce_tflite_tensor_info {
  int32_t type;            /* schema TensorType */
  int32_t rank;
  int32_t dims[8];
  int32_t quantized;       /* 1 when scale / zero_point are present */
  float scale;
  int32_t zero_point;
  const void* data;        /* constant data inside the model buffer, or NULL */
  size_t bytes;
  const char* name;        /* owned by the model */
} lce_tflite_tensor_info;
lce_hip_status lce_tflite_model_tensor(const lce_tflite_model* model, int32_t index, lce_tflite_tensor_info* info);

typedef struct lce_tflite_operator_info {
  int32_t builtin_code;    /* 32 = CUSTOM */
  const char* custom_code; /* "LceBconv2d", "LceQuantize", "LceDequantize", "LceBMaxPool2d", or "" */
  const int32_t* inputs;   /* tensor indices (-1: optional input absent); owned by the model */
  int32_t num_inputs;
  const int32_t* outputs;
  int32_t num_outputs;
  const uint8_t* custom_options;   /* flexbuffer map, as written by mlir/ir/lce_ops.cc:36-51 */
  size_t custom_options_size;
} lce_tflite_operator_info;
lce_hip_status lce_tflite_model_operator(const lce_tflite_model* model, int32_t index, lce_tflite_operator_info* info);

/* Binary SECTIONS of a mixed graph.  A converted model interleaves builtin float operators (the stem, batch norms, adds,
 * the head) with LCE custom ops; what this library runs are the maximal groups of LCE ops that can execute without a
 * builtin operator in between -- the partition a TFLite delegate would be handed (TensorFlow Lite's
 * PartitionGraphIntoIndependentNodeSubsets, restated): epochs alternate between LCE operators and all others, starting
 * with LCE; in an epoch EVERY operator of the epoch's kind whose inputs are all ready joins -- wherever it stands in the
 * file -- repeatedly, until none is left; the LCE operators of one epoch, sorted by index, are one section.  On a chain
 * that is "cut at every builtin operator"; on a branched graph an LCE op further down the file belongs to an earlier
 * section when nothing it reads depends on a builtin operator in between.  The reference runs whole
 * graphs through one interpreter (tflite/python/interpreter_base.py:74-95, examples/lce_minimal.cc:28-62); a host that
 * keeps TensorFlow Lite for the float operators calls a section between them: feed `inputs`, collect `outputs`.
 *   ops     : operator indices of the section, in execution order
 *   inputs  : non-constant tensors the section reads but does not produce
 *   outputs : tensors it produces that an operator outside the section, or the graph's output list, reads
 * The arrays are owned by the model. */
typedef struct lce_tflite_section_info {
  const int32_t* ops;
  int32_t num_ops;
  const int32_t* inputs;
  int32_t num_inputs;
  const int32_t* outputs;
  int32_t num_outputs;
} lce_tflite_section_info;
int32_t lce_tflite_model_num_sections(const lce_tflite_model* model);
lce_hip_status lce_tflite_model_section(const lce_tflite_model* model, int32_t index, lce_tflite_section_info* info);

/* Runs section `section` on DEVICE tensors at `batch` images per tensor (the file's own leading dimension, which the
 * converter pins to 1, is replaced): the C entry for a host without a TensorFlow Lite interpreter -- what
 * examples/lce_minimal.cc:28-62 / tflite/benchmark/lce_benchmark_main.cc:24-48 do with Interpreter::Invoke, for the
 * binary part of the graph -- and what compute-engine_amd/model_runner.py calls.
 *   inputs_dev[k]  : device pointer of tensor section.inputs[k]  ([batch, H, W, C] in the file's type)
 *   outputs_dev[k] : device buffer for tensor section.outputs[k] (lce_tflite_model_section_tensor_shape says how large)
 * Everything in between stays in device buffers the MODEL owns (grow-only, reused by the next call); plans are made
 * once per (operator, batch, semantics) and cached in the model; an LceBconv2d whose float / int8 output feeds an
 * LceQuantize of the same section writes both tensors from one epilogue (lce_hip_bconv2d_run_dual), the quantize launch
 * disappears.  Asynchronous on `stream` (a hipStream_t, or NULL); calls on one model are serialised by a mutex and must
 * use one stream at a time.  `semantics`: lce_hip_semantics (which registration's SAME-zero behaviour).  Shape inference
 * is the ops' own Prepare (quantization.cc:19-41, bmaxpool.cc:41-77, bconv2d.cc:137-300). */
lce_hip_status lce_tflite_model_run_section(lce_tflite_model* model, int32_t section, int32_t batch, int32_t semantics,
                                            const void* const* inputs_dev, void* const* outputs_dev, void* stream);
/* Shape ([N,H,W,C], C in words for bitpacked tensors) and size in bytes of a tensor section `section` reads or produces,
 * at `batch` images.  Host-only (no device needed). */
lce_hip_status lce_tflite_model_section_tensor_shape(lce_tflite_model* model, int32_t section, int32_t tensor, int32_t batch,
                                                     int32_t semantics, int32_t dims[4], size_t* bytes);

/* What the model holds for lce_tflite_model_run_section: plans cached so far (one per operator x batch size x semantics),
 * LceQuantize launches the LAST run folded into a convolution's epilogue, bytes of intermediate buffers.  Any pointer may
 * be NULL. */
void lce_tflite_model_run_stats(lce_tflite_model* model, int32_t* cached_plans, int32_t* fused_quantize_ops, size_t* scratch_bytes);

/* HIP graphs for lce_tflite_model_run_section (off by default).  A binary section is a chain of short kernels -- QuickNet's
 * last layers take 10-17 us each -- and a host call per kernel leaves gaps between them.  With graphs on, the launches of a
 * section are recorded once per (section, batch, semantics, stream, tensor pointers) and replayed as ONE launch: the first call
 * with such a key runs eagerly (plans, uploads, buffers), the second records and launches, later ones replay.  Needs a stream
 * of its own (not NULL: the null stream cannot be recorded; lce_hip_stream_create).  The recording holds the device pointers it
 * was given: call with the same tensors to replay, with others to get another recording; turning graphs off drops them all.
 * If a section cannot be recorded it simply keeps running eagerly.  graph_stats: recordings made / launches served by one. */
void lce_tflite_model_use_hip_graphs(lce_tflite_model* model, int32_t on);
void lce_tflite_model_graph_stats(lce_tflite_model* model, int32_t* recorded, int32_t* replays);

/* Builds a ready-to-run plan for operator `index`, which must be an LceBconv2d: descriptor
 * from the op's option map + tensor shapes / types / output quantization exactly as
 * bconv2d::Init + Prepare collect them (tflite/kernels/bconv2d.cc:85-131,137-300), weights
 * from the constant tensors (OneTimeSetup, :324-392).  `batch` > 0 overrides the batch
 * dimension recorded in the file (the converter pins it to 1, mlir/tf_tfl_passes.cc:141-144).
 * `semantics` selects which registration's behaviour to follow (lce_hip_semantics). */
lce_hip_status lce_tflite_model_bconv2d_plan(const lce_tflite_model* model, int32_t index, int32_t batch,
                                             int32_t semantics, lce_hip_bconv2d_plan** plan);

/* Option lookup on any LCE op's custom_options (bmaxpool: filter_height, filter_width,
 * stride_height, stride_width, padding).  Returns 0 and writes *value when the key exists. */
int lce_tflite_option_int(const uint8_t* custom_options, size_t size, const char* key, int32_t* value);

/* Message of the last failing lce_tflite_model_* call on this thread (plan creation failures
 * report through lce_hip_last_error()). */
const char* lce_tflite_model_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* LCE_TFLITE_MODEL_H_ */
