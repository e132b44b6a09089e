// TFLite custom-op glue for the MI355X build of Larq Compute Engine's binary ops.
//
// Same entry points, registration contract and error behaviour as the reference's
// tflite/kernels/{bconv2d.cc, quantization.cc, bmaxpool.cc, lce_ops_register.h}
// (paths relative to /root/reference/larq_compute_engine/): C++ functions in namespace
// compute_engine::tflite returning a function-local static TfLiteRegistration with
// {init, free, prepare, invoke}.  Underneath, instead of the CPU kernels, every op calls
// the C ABI in include/lce_hip.h.  Interpreter tensors are host memory; a tensor that is PRODUCED by one
// of these ops and READ only by these ops stays in HBM between them (namespace `resident` below): in the
// reference consecutive LCE ops hand tensors over in the arena at no cost (tflite/kernels/bconv2d.cc:550-564,
// quantization.cc:76-114, bmaxpool.cc:79-91), here a chain LceQuantize -> LceBconv2d -> LceBMaxPool2d ->
// LceBconv2d crosses PCIe once in each direction instead of twice per op.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <utility>

#include "../../../include/lce_hip.h"
#include "flexbuffer_map.h"
#include "lce_ops_register.h"
#include "tflite_abi.h"

namespace compute_engine {
namespace tflite {
namespace {

// --- small helpers in the spirit of tensorflow/lite/kernels/kernel_util.h -----------
inline const TfLiteTensor* GetInput(TfLiteContext* c, const TfLiteNode* n, int i) {
  const int idx = n->inputs->data[i];
  return idx == kTfLiteOptionalTensor ? nullptr : &c->tensors[idx];
}
inline TfLiteTensor* GetOutput(TfLiteContext* c, const TfLiteNode* n, int i) {
  return &c->tensors[n->outputs->data[i]];
}
inline int NumDimensions(const TfLiteTensor* t) { return t->dims->size; }
inline int SizeOfDimension(const TfLiteTensor* t, int d) { return t->dims->data[d]; }
inline int BitpackedSize(int n) { return (n + 31) / 32; }  // core/bitpacking/bitpack.h:24-26

#define LCE_ENSURE(ctx, cond)                                                          \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      (ctx)->ReportError((ctx), "%s:%d %s was not true.", __FILE__, __LINE__, #cond);   \
      return kTfLiteError;                                                             \
    }                                                                                  \
  } while (0)
#define LCE_ENSURE_EQ(ctx, a, b)                                                       \
  do {                                                                                 \
    if ((a) != (b)) {                                                                  \
      (ctx)->ReportError((ctx), "%s:%d %s != %s (%d != %d)", __FILE__, __LINE__, #a, #b, \
                         (int)(a), (int)(b));                                          \
      return kTfLiteError;                                                             \
    }                                                                                  \
  } while (0)
#define LCE_ENSURE_MSG(ctx, cond, msg)                                                 \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      (ctx)->ReportError((ctx), "%s:%d %s", __FILE__, __LINE__, (msg));                 \
      return kTfLiteError;                                                             \
    }                                                                                  \
  } while (0)
#define LCE_ENSURE_HIP(ctx, expr)                                                      \
  do {                                                                                 \
    if ((expr) != LCE_HIP_OK) {                                                        \
      (ctx)->ReportError((ctx), "%s:%d %s", __FILE__, __LINE__, lce_hip_last_error());  \
      return kTfLiteError;                                                             \
    }                                                                                  \
  } while (0)

size_t NumElements(const TfLiteTensor* t) {
  size_t n = 1;
  for (int i = 0; i < t->dims->size; ++i) n *= (size_t)t->dims->data[i];
  return n;
}

// =====================================================================================
// Device residency between LCE ops.
//
// Every op's output lives in a device buffer keyed by (context, tensor index).  Whether the tensor must ALSO travel back
// to the arena depends on who reads it: LCE readers take the device buffer in their invoke (no upload); only when every
// reader is an LCE op, there is at least one, and the tensor is not a graph output may the copy-back be skipped.
//
// An op cannot find that out by itself: TfLiteContext::GetExecutionPlan / GetNodeAndRegistration are delegate-only (in
// kernel context TensorFlow Lite points them at a function that reports an error, Subgraph::SwitchToKernelContext), and
// the context does not say which tensors are graph outputs.  So the HOST declares the graph, once, through the C entry
// points at the end of this file (lce_tflite_ops_declare_graph_begin / _node / _output / _end; lce_ops_register.h wraps
// them for a ::tflite::Interpreter as DeclareGraphForDeviceResidency).  Without a declaration -- the default, and all the
// reference's own callers -- every op leaves its output in the arena exactly as the reference does
// (tflite/kernels/bconv2d.cc:550-564): nothing can go stale, it just costs the copies.  With one, a tensor that is a
// graph output is always copied back, also when LCE ops read it too.  LCE_HIP_TFLITE_RESIDENCY=0 (or
// lce_tflite_ops_set_residency(0)) ignores declarations.  The ops never call the delegate-only callbacks.
// =====================================================================================
namespace resident {

struct Entry {
  void* dev = nullptr;
  size_t bytes = 0;       // allocated
  bool valid = false;     // holds the tensor's current value (set by the producer's invoke)
};
// Who reads a tensor, from the host's declaration.
struct Readers {
  bool known = false;
  int lce = 0, other = 0;
  bool graph_output = false;
  bool keep_on_device() const { return known && lce > 0; }                                     // an LCE reader will pick it up
  bool needs_host() const { return !known || other > 0 || lce == 0 || graph_output; }          // someone reads the arena
};
constexpr int kStagingTensor = -1;       // key of a context's shared staging buffer for inputs no LCE op produced
std::mutex g_mu;
std::map<std::pair<const TfLiteContext*, int>, Entry> g_table;
std::map<const TfLiteContext*, std::map<int, Readers>> g_graphs;     // declared graphs
std::map<const TfLiteContext*, std::map<int, Readers>> g_pending;    // ... being declared
std::map<const TfLiteContext*, int> g_nodes;                         // live LCE nodes per context (Init / Free)
std::atomic<uint64_t> g_version{1};                                  // bumped whenever a declaration changes
std::atomic<uint64_t> g_h2d{0}, g_d2h{0}, g_h2d_bytes{0}, g_d2h_bytes{0};
std::atomic<int> g_enabled{-1};   // -1: read LCE_HIP_TFLITE_RESIDENCY on first use

bool enabled() {
  int e = g_enabled.load();
  if (e < 0) {
    const char* v = getenv("LCE_HIP_TFLITE_RESIDENCY");
    e = (v && (v[0] == '0' || v[0] == 'n' || v[0] == 'N' || v[0] == 'f' || v[0] == 'F')) ? 0 : 1;
    g_enabled.store(e);
  }
  return e != 0;
}

// the device buffer of (context, tensor): created / grown on demand; nullptr on allocation failure
Entry* ensure(const TfLiteContext* c, int tensor, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_mu);
  Entry& e = g_table[std::make_pair(c, tensor)];
  if (e.bytes < bytes) {
    if (e.dev) lce_hip_free(e.dev);
    e.dev = nullptr;
    e.bytes = 0;
    e.valid = false;
    if (lce_hip_malloc(&e.dev, bytes ? bytes : 1) != LCE_HIP_OK) return nullptr;
    e.bytes = bytes;
  }
  return &e;   // (std::map nodes are stable)
}
// the current device copy of (context, tensor), if a producer left one
void* current(const TfLiteContext* c, int tensor, size_t bytes) {
  if (!enabled()) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_table.find(std::make_pair(c, tensor));
  return it != g_table.end() && it->second.valid && it->second.bytes >= bytes ? it->second.dev : nullptr;
}
void invalidate(const TfLiteContext* c, int tensor) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_table.find(std::make_pair(c, tensor));
  if (it != g_table.end()) it->second.valid = false;
}
void drop_locked(const TfLiteContext* c, int tensor) {
  auto it = g_table.find(std::make_pair(c, tensor));
  if (it == g_table.end()) return;
  if (it->second.dev) lce_hip_free(it->second.dev);
  g_table.erase(it);
}
void drop(const TfLiteContext* c, int tensor) {
  std::lock_guard<std::mutex> lock(g_mu);
  drop_locked(c, tensor);
}
// A node of this library was created in / removed from `c`.  With the last one go the context's shared staging buffer
// and its declared graph (an interpreter that is destroyed takes everything it caused with it).
void node_created(const TfLiteContext* c) {
  std::lock_guard<std::mutex> lock(g_mu);
  ++g_nodes[c];
}
void node_freed(const TfLiteContext* c) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_nodes.find(c);
  if (it == g_nodes.end() || --it->second > 0) return;
  g_nodes.erase(it);
  drop_locked(c, kStagingTensor);
  if (g_graphs.erase(c) + g_pending.erase(c)) ++g_version;
}

bool is_lce_invoke(TfLiteStatus (*fn)(TfLiteContext*, TfLiteNode*));   // defined at the end of this file

// Readers of `tensor` as the host declared them; `cache` / `cache_version` belong to the asking node: the table is
// consulted again only after a declaration changed (no lock, no lookup on the steady-state invoke).
const Readers& readers_of(const TfLiteContext* c, int tensor, Readers* cache, uint64_t* cache_version) {
  const uint64_t v = g_version.load(std::memory_order_acquire);
  if (*cache_version == v) return *cache;
  Readers r;
  if (enabled()) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto g = g_graphs.find(c);
    if (g != g_graphs.end()) {
      auto t = g->second.find(tensor);
      r = t != g->second.end() ? t->second : Readers{};
      r.known = true;          // a declared graph in which nobody reads the tensor: lce == 0 -> copied back
    }
  }
  *cache = r;
  *cache_version = v;
  return *cache;
}

// a host-in / host-out call that stages its tensors inside the library (lce_hip_bconv2d_run_host)
void count_host_pass(size_t up_bytes, size_t down_bytes) {
  ++g_h2d;
  g_h2d_bytes += up_bytes;
  ++g_d2h;
  g_d2h_bytes += down_bytes;
}

// ---- what an op's invoke does with its tensors ----
// input: the producer's device copy when there is one, else an upload into the context's ONE staging buffer (grow-only,
// bounded by the largest such tensor; released with the context's last LCE node).  Uploads, kernels and downloads all
// run on the null stream, so the next op's upload into the same buffer is ordered behind this op's kernel.
lce_hip_status input(TfLiteContext* c, int tensor, const void* host, size_t bytes, const void** dev) {
  if (void* cur = current(c, tensor, bytes)) { *dev = cur; return LCE_HIP_OK; }
  Entry* e = ensure(c, kStagingTensor, bytes);
  if (!e) return LCE_HIP_ERR_RUNTIME;
  ++g_h2d;
  g_h2d_bytes += bytes;
  *dev = e->dev;
  return lce_hip_memcpy_h2d(e->dev, host, bytes, nullptr);
}
// output: the device buffer the kernel writes ...
lce_hip_status output(TfLiteContext* c, int tensor, size_t bytes, void** dev) {
  Entry* e = ensure(c, tensor, bytes);
  if (!e) return LCE_HIP_ERR_RUNTIME;
  e->valid = false;
  *dev = e->dev;
  return LCE_HIP_OK;
}
// ... and what happens to it behind the kernel
lce_hip_status publish(TfLiteContext* c, int tensor, const Readers& r, void* host, size_t bytes) {
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_table.find(std::make_pair(static_cast<const TfLiteContext*>(c), tensor));
    if (it != g_table.end()) it->second.valid = r.keep_on_device();
    if (r.needs_host() && it != g_table.end()) {
      ++g_d2h;
      g_d2h_bytes += bytes;
      if (lce_hip_status s = lce_hip_memcpy_d2h(host, it->second.dev, bytes, nullptr)) return s;
    }
  }
  return r.needs_host() ? lce_hip_stream_synchronize(nullptr) : LCE_HIP_OK;
}

// what a node remembers about the tensor it produces
struct OutputRef {
  const TfLiteContext* context = nullptr;   // where the node lives (set by init)
  const TfLiteContext* out_context = nullptr;
  int tensor = -1;
  Readers readers;
  uint64_t readers_version = 0;
  explicit OutputRef(const TfLiteContext* c) : context(c) { node_created(c); }
  ~OutputRef() {
    if (out_context) drop(out_context, tensor);
    node_freed(context);
  }
  void remember(const TfLiteContext* c, int t) {        // Prepare (may run again after a resize)
    if (out_context) invalidate(out_context, tensor);
    out_context = c;
    tensor = t;
    readers_version = 0;
  }
  const Readers& who_reads() { return readers_of(out_context, tensor, &readers, &readers_version); }
};

}  // namespace resident

}  // namespace

// =====================================================================================
// LceBconv2d  (reference: tflite/kernels/bconv2d.cc)
// =====================================================================================
namespace bconv2d {

enum class KernelType { kReference, kOptimizedBGEMM, kOptimizedIndirectBGEMM };  // bconv2d.cc:31-40

struct OpData {  // bconv2d.cc:44-74
  // attributes
  int32_t stride_height = 0, stride_width = 0;
  int32_t dilation_height_factor = 0, dilation_width_factor = 0;
  int32_t padding = 0;  // tflite schema enum: SAME = 0, VALID = 1
  int32_t pad_values = 0;
  int32_t channels_in = 0;
  int32_t fused_activation_function = 0;
  // inferred in Prepare
  int32_t groups = 1;
  lce_hip_bconv2d_plan* plan = nullptr;
  bool successfully_initialized = false;
  bool one_time_setup_complete = false;
  // device residency of the output (namespace resident)
  resident::OutputRef out;
  explicit OpData(const TfLiteContext* c) : out(c) {}
  ~OpData() {
    if (plan) lce_hip_bconv2d_plan_destroy(plan);
  }
};

void* Init(TfLiteContext* context, const char* buffer, size_t length) {  // bconv2d.cc:85-131
  auto* op = new (std::nothrow) OpData(context);
  if (!op) return nullptr;
  const lce_flex::Map m(reinterpret_cast<const uint8_t*>(buffer), length);
  static const char* const kRequired[] = {"stride_height", "stride_width", "dilation_height_factor",
                                          "dilation_width_factor", "padding", "pad_values",
                                          "channels_in", "fused_activation_function"};
  for (const char* key : kRequired) {
    if (!m.valid() || m.IsNull(key)) {  // LCE_ENSURE_PARAM, bconv2d.cc:76-83,96-103
      context->ReportError(context, "%s:%d !m[\"%s\"].IsNull() was not true.", __FILE__, __LINE__, key);
      return op;
    }
  }
  op->stride_height = m.AsInt32("stride_height");
  op->stride_width = m.AsInt32("stride_width");
  op->dilation_height_factor = m.AsInt32("dilation_height_factor");
  op->dilation_width_factor = m.AsInt32("dilation_width_factor");
  op->padding = m.AsInt32("padding");
  op->pad_values = m.AsInt32("pad_values");
  if (op->pad_values != 0 && op->pad_values != 1) {  // :109-112
    context->ReportError(context, "Attribute pad_values must be 0 or 1.");
    return op;
  }
  op->channels_in = m.AsInt32("channels_in");
  // ConvertActivation (tflite/kernels/utils.h:10-25): anything unknown becomes NONE
  const int act = m.AsInt32("fused_activation_function");
  op->fused_activation_function = (act >= LCE_HIP_ACT_NONE && act <= LCE_HIP_ACT_RELU6) ? act : LCE_HIP_ACT_NONE;
  // Init cannot return an error; Prepare checks this flag (:126-129)
  op->successfully_initialized = true;
  return op;
}

void Free(TfLiteContext*, void* buffer) { delete reinterpret_cast<OpData*>(buffer); }  // :133-135

template <KernelType kernel_type>
TfLiteStatus Prepare(TfLiteContext* context, TfLiteNode* node) {  // bconv2d.cc:137-300
  auto* op = reinterpret_cast<OpData*>(node->user_data);
  if (!op || !op->successfully_initialized) return kTfLiteError;

  LCE_ENSURE_EQ(context, node->inputs->size, 5);
  const TfLiteTensor* input = GetInput(context, node, 0);
  const TfLiteTensor* filter = GetInput(context, node, 1);
  const TfLiteTensor* post_activation_multiplier = GetInput(context, node, 2);
  const TfLiteTensor* post_activation_bias = GetInput(context, node, 3);
  const TfLiteTensor* thresholds = GetInput(context, node, 4);
  TfLiteTensor* output = GetOutput(context, node, 0);
  LCE_ENSURE(context, input != nullptr && filter != nullptr);

  LCE_ENSURE_EQ(context, NumDimensions(input), 4);
  LCE_ENSURE_EQ(context, NumDimensions(filter), 4);
  LCE_ENSURE_EQ(context, input->type, kTfLiteInt32);
  LCE_ENSURE_EQ(context, filter->type, kTfLiteInt32);
  LCE_ENSURE_MSG(context,
                 output->type == kTfLiteInt32 || output->type == kTfLiteInt8 || output->type == kTfLiteFloat32,
                 "Supported output types are int8, int32, and float32.");

  const int32_t channels_out = SizeOfDimension(filter, 0);
  LCE_ENSURE(context, op->channels_in > 0);   // (the divisions below must not see a zero group count)
  // groups are inferred from the filter's packed depth (:169-186)
  if (SizeOfDimension(filter, 3) == BitpackedSize(op->channels_in)) {
    op->groups = 1;
  } else {
    LCE_ENSURE_MSG(context, kernel_type != KernelType::kOptimizedBGEMM,
                   "Grouped binary convolutions are not supported with this kernel.");
    LCE_ENSURE(context, SizeOfDimension(filter, 3) > 0);
    LCE_ENSURE_EQ(context, BitpackedSize(op->channels_in) % SizeOfDimension(filter, 3), 0);
    const int32_t groups = BitpackedSize(op->channels_in) / SizeOfDimension(filter, 3);
    LCE_ENSURE(context, groups >= 1);
    const int32_t group_size = op->channels_in / groups;
    LCE_ENSURE_EQ(context, group_size % 32, 0);
    LCE_ENSURE_EQ(context, channels_out % groups, 0);
    op->groups = groups;
  }
  LCE_ENSURE_EQ(context, SizeOfDimension(input, 3), BitpackedSize(op->channels_in));

  const bool is_ref = kernel_type == KernelType::kReference;
  if (op->padding == LCE_HIP_PADDING_SAME && op->pad_values == 0) {  // :188-200
    LCE_ENSURE_MSG(context,
                   (is_ref && op->channels_in % 2 == 0) ||
                       (!is_ref && output->type == kTfLiteFloat32 &&
                        op->fused_activation_function == LCE_HIP_ACT_NONE),
                   "Zero-padding is only supported by the reference kernel with an even "
                   "number of input channels, or when using "
                   "float output with no fused activation function.");
  }

  if (output->type == kTfLiteInt32) {  // :212-227
    LCE_ENSURE(context, thresholds != nullptr);
    LCE_ENSURE_EQ(context, NumDimensions(thresholds), 1);
    LCE_ENSURE_EQ(context, thresholds->type, kTfLiteInt32);
    LCE_ENSURE_EQ(context, SizeOfDimension(thresholds, 0), channels_out);
  } else {
    LCE_ENSURE(context, post_activation_multiplier != nullptr && post_activation_bias != nullptr);
    LCE_ENSURE_EQ(context, post_activation_multiplier->type, kTfLiteFloat32);
    LCE_ENSURE_EQ(context, post_activation_bias->type, kTfLiteFloat32);
    LCE_ENSURE_EQ(context, NumDimensions(post_activation_multiplier), 1);
    LCE_ENSURE_EQ(context, NumDimensions(post_activation_bias), 1);
    LCE_ENSURE_EQ(context, SizeOfDimension(post_activation_multiplier, 0), channels_out);
    LCE_ENSURE_EQ(context, SizeOfDimension(post_activation_bias, 0), channels_out);
  }
  if (output->type == kTfLiteInt8) {
    LCE_ENSURE_EQ(context, output->quantization.type, kTfLiteAffineQuantization);  // :229-232
  }
  if (kernel_type == KernelType::kOptimizedIndirectBGEMM) {
    LCE_ENSURE_MSG(context, input->allocation_type != kTfLiteDynamic,
                   "The input tensor must not have dynamic allocation type");  // :234-238
  }

  // shape inference + the remaining checks live behind the C ABI (host-only, no GPU needed)
  lce_hip_bconv2d_desc d;
  memset(&d, 0, sizeof d);
  d.batch = SizeOfDimension(input, 0);
  d.in_height = SizeOfDimension(input, 1);
  d.in_width = SizeOfDimension(input, 2);
  d.channels_in = op->channels_in;
  d.filter_height = SizeOfDimension(filter, 1);
  d.filter_width = SizeOfDimension(filter, 2);
  d.channels_out = channels_out;
  d.groups = op->groups;
  d.stride_height = op->stride_height;
  d.stride_width = op->stride_width;
  d.dilation_height = op->dilation_height_factor;
  d.dilation_width = op->dilation_width_factor;
  d.padding = op->padding;
  d.pad_values = op->pad_values;
  d.activation = op->fused_activation_function;
  d.dst_type = output->type == kTfLiteFloat32 ? LCE_HIP_F32 : output->type == kTfLiteInt8 ? LCE_HIP_I8 : LCE_HIP_BITPACKED;
  d.semantics = is_ref ? LCE_HIP_SEM_REFERENCE : LCE_HIP_SEM_OPTIMIZED;
  d.out_scale = output->type == kTfLiteInt8 ? output->params.scale : 1.0f;
  d.out_zero_point = output->type == kTfLiteInt8 ? output->params.zero_point : 0;
  if (op->plan) {
    lce_hip_bconv2d_plan_destroy(op->plan);
    op->plan = nullptr;
  }
  LCE_ENSURE_HIP(context, lce_hip_bconv2d_plan_create(&d, &op->plan));

  int32_t dims[4];
  LCE_ENSURE_HIP(context, lce_hip_bconv2d_plan_output_shape(op->plan, dims));
  TfLiteIntArray* output_shape = TfLiteIntArrayCreate(4);  // ResizeTensor takes ownership (:241-248)
  for (int i = 0; i < 4; ++i) output_shape->data[i] = dims[i];
  if (context->ResizeTensor(context, output, output_shape) != kTfLiteOk) return kTfLiteError;

  // No im2col temporary: the GPU kernel is an implicit GEMM (the reference allocates
  // [B,OH,OW,KH*KW*Cw] here, :250-293).
  // Prepare may run again after a resize: redo the one-time setup (:295-297).
  op->one_time_setup_complete = false;
  // who reads the output decides whether it has to travel back to the arena (asked at invoke: the host may declare the
  // graph before or after AllocateTensors)
  op->out.remember(context, node->outputs->data[0]);
  return kTfLiteOk;
}

TfLiteStatus OneTimeSetup(TfLiteContext* context, TfLiteNode* node, OpData* op) {  // :324-392
  const TfLiteTensor* filter = GetInput(context, node, 1);
  const TfLiteTensor* mul = GetInput(context, node, 2);
  const TfLiteTensor* bias = GetInput(context, node, 3);
  const TfLiteTensor* thr = GetInput(context, node, 4);
  LCE_ENSURE_HIP(context, lce_hip_bconv2d_plan_set_weights(op->plan, filter->data.i32,
                                                           mul ? mul->data.f : nullptr,
                                                           bias ? bias->data.f : nullptr,
                                                           thr ? thr->data.i32 : nullptr));
  op->one_time_setup_complete = true;
  return kTfLiteOk;
}

template <KernelType kernel_type>
TfLiteStatus Eval(TfLiteContext* context, TfLiteNode* node) {  // :550-564
  auto* op = reinterpret_cast<OpData*>(node->user_data);
  if (!op || !op->plan) return kTfLiteError;
  if (!op->one_time_setup_complete) {
    if (OneTimeSetup(context, node, op) != kTfLiteOk) return kTfLiteError;
  }
  const TfLiteTensor* input = GetInput(context, node, 0);
  TfLiteTensor* output = GetOutput(context, node, 0);
  if (output->type != kTfLiteFloat32 && output->type != kTfLiteInt8 && output->type != kTfLiteInt32)
    return kTfLiteError;
  const int in_idx = node->inputs->data[0];
  const bool in_resident = resident::current(context, in_idx, input->bytes) != nullptr;
  const resident::Readers& readers = op->out.who_reads();
  if (!in_resident && !readers.keep_on_device()) {
    // host in, host out: the pipelined path (batch slices on three streams); counted as one pass in each direction
    resident::count_host_pass(input->bytes, output->bytes);
    resident::invalidate(context, op->out.tensor);      // (a device copy from an earlier invoke under a declaration is now stale)
    LCE_ENSURE_HIP(context, lce_hip_bconv2d_run_host(op->plan, input->data.i32, output->data.data));
    return kTfLiteOk;
  }
  const void* in_dev = nullptr;
  void* out_dev = nullptr;
  LCE_ENSURE_HIP(context, resident::input(context, in_idx, input->data.data, input->bytes, &in_dev));
  LCE_ENSURE_HIP(context, resident::output(context, op->out.tensor, output->bytes, &out_dev));
  LCE_ENSURE_HIP(context, lce_hip_bconv2d_run(op->plan, (const int32_t*)in_dev, out_dev, nullptr));
  LCE_ENSURE_HIP(context, resident::publish(context, op->out.tensor, readers, output->data.data, output->bytes));
  return kTfLiteOk;
}

}  // namespace bconv2d

TfLiteRegistration* Register_BCONV_2D_REF() {  // bconv2d.cc:568-574
  static TfLiteRegistration r = {bconv2d::Init, bconv2d::Free,
                                 bconv2d::Prepare<bconv2d::KernelType::kReference>,
                                 bconv2d::Eval<bconv2d::KernelType::kReference>};
  return &r;
}
TfLiteRegistration* Register_BCONV_2D_OPT_BGEMM() {  // :576-582
  static TfLiteRegistration r = {bconv2d::Init, bconv2d::Free,
                                 bconv2d::Prepare<bconv2d::KernelType::kOptimizedBGEMM>,
                                 bconv2d::Eval<bconv2d::KernelType::kOptimizedBGEMM>};
  return &r;
}
TfLiteRegistration* Register_BCONV_2D_OPT_INDIRECT_BGEMM() {  // :584-590
  static TfLiteRegistration r = {bconv2d::Init, bconv2d::Free,
                                 bconv2d::Prepare<bconv2d::KernelType::kOptimizedIndirectBGEMM>,
                                 bconv2d::Eval<bconv2d::KernelType::kOptimizedIndirectBGEMM>};
  return &r;
}
// The reference picks OPT_BGEMM whenever TFLITE_WITH_RUY is defined, which its Bazel build
// always does (bconv2d.cc:592-599, .bazelrc:20-21); that is the semantics kept here.
TfLiteRegistration* Register_BCONV_2D() { return Register_BCONV_2D_OPT_BGEMM(); }

// =====================================================================================
// LceQuantize / LceDequantize  (reference: tflite/kernels/quantization.cc)
// =====================================================================================
namespace {

// The reference's LceQuantize / LceDequantize have no per-node state (init = free = nullptr, quantization.cc:149-159);
// here a node remembers which tensor it produces so that the tensor's device buffer is released with the node.
using resident::OutputRef;
void* OutputRefInit(TfLiteContext* context, const char*, size_t) { return new (std::nothrow) OutputRef(context); }
void OutputRefFree(TfLiteContext*, void* buffer) { delete reinterpret_cast<OutputRef*>(buffer); }
void RememberOutput(TfLiteContext* context, TfLiteNode* node) {
  if (auto* r = reinterpret_cast<OutputRef*>(node->user_data)) r->remember(context, node->outputs->data[0]);
}
// (a node without user data -- init failed to allocate -- copies back: the safe default)
const resident::Readers& ReadersOfOutput(TfLiteNode* node) {
  static const resident::Readers unknown;
  auto* r = reinterpret_cast<OutputRef*>(node->user_data);
  return r && r->out_context ? r->who_reads() : unknown;
}

TfLiteStatus QuantizePrepare(TfLiteContext* context, TfLiteNode* node) {  // quantization.cc:19-41
  LCE_ENSURE_EQ(context, node->inputs->size, 1);
  LCE_ENSURE_EQ(context, node->outputs->size, 1);
  const TfLiteTensor* input = GetInput(context, node, 0);
  TfLiteTensor* output = GetOutput(context, node, 0);
  LCE_ENSURE(context, input->type == kTfLiteFloat32 || input->type == kTfLiteInt8 || input->type == kTfLiteBool);
  LCE_ENSURE_EQ(context, output->type, kTfLiteInt32);
  const int num_dims = NumDimensions(input);
  LCE_ENSURE_EQ(context, num_dims, NumDimensions(output));
  TfLiteIntArray* output_dims = TfLiteIntArrayCopy(input->dims);
  output_dims->data[num_dims - 1] = BitpackedSize(SizeOfDimension(input, num_dims - 1));
  RememberOutput(context, node);
  return context->ResizeTensor(context, output, output_dims);
}

TfLiteStatus QuantizeEval(TfLiteContext* context, TfLiteNode* node) {  // quantization.cc:76-114
  const TfLiteTensor* input = GetInput(context, node, 0);
  TfLiteTensor* output = GetOutput(context, node, 0);
  lce_hip_dtype t;
  size_t esz;
  int32_t zp = 0;
  if (input->type == kTfLiteFloat32) { t = LCE_HIP_F32; esz = 4; }
  else if (input->type == kTfLiteInt8) { t = LCE_HIP_I8; esz = 1; zp = input->params.zero_point; }
  else if (input->type == kTfLiteBool) { t = LCE_HIP_BOOL; esz = sizeof(bool); zp = 1; }
  else return kTfLiteError;
  const int nd = NumDimensions(input);
  const size_t cols = (size_t)SizeOfDimension(input, nd - 1);
  const size_t total = NumElements(input);
  if (total == 0) return kTfLiteOk;
  const size_t rows = total / cols;
  const size_t out_bytes = rows * (size_t)BitpackedSize((int)cols) * 4;
  const int out_idx = node->outputs->data[0];
  const resident::Readers& readers = ReadersOfOutput(node);
  const void* in_dev = nullptr;
  void* out_dev = nullptr;
  LCE_ENSURE_HIP(context, resident::input(context, node->inputs->data[0], input->data.data, total * esz, &in_dev));
  LCE_ENSURE_HIP(context, resident::output(context, out_idx, out_bytes, &out_dev));
  LCE_ENSURE_HIP(context, lce_hip_bitpack(t, in_dev, rows, cols, zp, (int32_t*)out_dev, nullptr));
  LCE_ENSURE_HIP(context, resident::publish(context, out_idx, readers, output->data.data, out_bytes));
  return kTfLiteOk;
}

TfLiteStatus DequantizePrepare(TfLiteContext* context, TfLiteNode* node) {  // quantization.cc:43-74
  LCE_ENSURE_EQ(context, node->inputs->size, 1);
  LCE_ENSURE_EQ(context, node->outputs->size, 1);
  const TfLiteTensor* input = GetInput(context, node, 0);
  TfLiteTensor* output = GetOutput(context, node, 0);
  LCE_ENSURE_EQ(context, input->type, kTfLiteInt32);
  LCE_ENSURE(context, output->type == kTfLiteFloat32 || output->type == kTfLiteInt8 || output->type == kTfLiteBool);
  const int num_dims = NumDimensions(input);
  LCE_ENSURE_EQ(context, num_dims, NumDimensions(output));
  for (int i = 0; i < num_dims - 1; ++i) LCE_ENSURE_EQ(context, SizeOfDimension(output, i), SizeOfDimension(input, i));
  LCE_ENSURE_EQ(context, SizeOfDimension(input, num_dims - 1), BitpackedSize(SizeOfDimension(output, num_dims - 1)));
  RememberOutput(context, node);
  return kTfLiteOk;  // no resize: the unpacked channel count cannot be inferred (:69-71)
}

TfLiteStatus DequantizeEval(TfLiteContext* context, TfLiteNode* node) {  // quantization.cc:116-147
  const TfLiteTensor* input = GetInput(context, node, 0);
  TfLiteTensor* output = GetOutput(context, node, 0);
  lce_hip_dtype t;
  size_t esz;
  if (output->type == kTfLiteFloat32) { t = LCE_HIP_F32; esz = 4; }
  else if (output->type == kTfLiteInt8) { t = LCE_HIP_I8; esz = 1; }
  else if (output->type == kTfLiteBool) { t = LCE_HIP_BOOL; esz = sizeof(bool); }
  else return kTfLiteError;
  const int nd = NumDimensions(output);
  const size_t cols = (size_t)SizeOfDimension(output, nd - 1);
  const size_t total = NumElements(output);
  if (total == 0) return kTfLiteOk;
  const size_t rows = total / cols;
  const size_t in_bytes = rows * (size_t)BitpackedSize((int)cols) * 4;
  const int out_idx = node->outputs->data[0];
  const resident::Readers& readers = ReadersOfOutput(node);
  const void* in_dev = nullptr;
  void* out_dev = nullptr;
  LCE_ENSURE_HIP(context, resident::input(context, node->inputs->data[0], input->data.data, in_bytes, &in_dev));
  LCE_ENSURE_HIP(context, resident::output(context, out_idx, total * esz, &out_dev));
  LCE_ENSURE_HIP(context, lce_hip_unpack(t, (const int32_t*)in_dev, rows, cols, output->params.scale,
                                         output->params.zero_point, out_dev, nullptr));
  LCE_ENSURE_HIP(context, resident::publish(context, out_idx, readers, output->data.data, total * esz));
  return kTfLiteOk;
}

}  // namespace

TfLiteRegistration* Register_QUANTIZE() {  // quantization.cc:149-153
  static TfLiteRegistration r = {OutputRefInit, OutputRefFree, QuantizePrepare, QuantizeEval};
  return &r;
}
TfLiteRegistration* Register_DEQUANTIZE() {  // quantization.cc:155-159
  static TfLiteRegistration r = {OutputRefInit, OutputRefFree, DequantizePrepare, DequantizeEval};
  return &r;
}

// =====================================================================================
// LceBMaxPool2d  (reference: tflite/kernels/bmaxpool.cc)
// =====================================================================================
namespace bmaxpool {

struct PoolParams {  // core/bmaxpool.h:15-22
  int32_t filter_height = 0, filter_width = 0, stride_height = 0, stride_width = 0, padding = 0;
  resident::OutputRef out;                      // device residency of the output (namespace resident)
  explicit PoolParams(const TfLiteContext* c) : out(c) {}
};

void* Init(TfLiteContext* context, const char* buffer, size_t length) {  // bmaxpool.cc:20-35
  auto* p = new (std::nothrow) PoolParams(context);
  if (!p) return nullptr;
  const lce_flex::Map m(reinterpret_cast<const uint8_t*>(buffer), length);
  p->filter_height = m.AsInt32("filter_height");
  p->filter_width = m.AsInt32("filter_width");
  p->stride_height = m.AsInt32("stride_height");
  p->stride_width = m.AsInt32("stride_width");
  p->padding = m.AsInt32("padding");
  return p;
}
void Free(TfLiteContext*, void* buffer) { delete reinterpret_cast<PoolParams*>(buffer); }

TfLiteStatus Prepare(TfLiteContext* context, TfLiteNode* node) {  // bmaxpool.cc:41-77
  auto* p = reinterpret_cast<PoolParams*>(node->user_data);
  LCE_ENSURE(context, p != nullptr);
  LCE_ENSURE_EQ(context, node->inputs->size, 1);
  LCE_ENSURE_EQ(context, node->outputs->size, 1);
  const TfLiteTensor* input = GetInput(context, node, 0);
  TfLiteTensor* output = GetOutput(context, node, 0);
  LCE_ENSURE_EQ(context, NumDimensions(input), 4);
  LCE_ENSURE_EQ(context, input->type, kTfLiteInt32);
  LCE_ENSURE_EQ(context, output->type, kTfLiteInt32);
  LCE_ENSURE(context, p->stride_height != 0);
  LCE_ENSURE(context, p->stride_width != 0);
  LCE_ENSURE(context, p->filter_height != 0);
  LCE_ENSURE(context, p->filter_width != 0);
  int32_t oh = 0, ow = 0;
  LCE_ENSURE_HIP(context, lce_hip_bmaxpool_output_shape(SizeOfDimension(input, 1), SizeOfDimension(input, 2),
                                                        p->filter_height, p->filter_width, p->stride_height,
                                                        p->stride_width, p->padding, &oh, &ow));
  TfLiteIntArray* out = TfLiteIntArrayCreate(4);
  out->data[0] = SizeOfDimension(input, 0);
  out->data[1] = oh;
  out->data[2] = ow;
  out->data[3] = SizeOfDimension(input, 3);
  p->out.remember(context, node->outputs->data[0]);
  return context->ResizeTensor(context, output, out);
}

TfLiteStatus Eval(TfLiteContext* context, TfLiteNode* node) {  // bmaxpool.cc:79-91
  auto* p = reinterpret_cast<PoolParams*>(node->user_data);
  const TfLiteTensor* input = GetInput(context, node, 0);
  TfLiteTensor* output = GetOutput(context, node, 0);
  const size_t in_bytes = NumElements(input) * 4, out_bytes = NumElements(output) * 4;
  if (in_bytes == 0 || out_bytes == 0) return kTfLiteOk;
  const int out_idx = node->outputs->data[0];
  const resident::Readers& readers = p->out.who_reads();
  const void* in_dev = nullptr;
  void* out_dev = nullptr;
  LCE_ENSURE_HIP(context, resident::input(context, node->inputs->data[0], input->data.data, in_bytes, &in_dev));
  LCE_ENSURE_HIP(context, resident::output(context, out_idx, out_bytes, &out_dev));
  LCE_ENSURE_HIP(context, lce_hip_bmaxpool((const int32_t*)in_dev, SizeOfDimension(input, 0), SizeOfDimension(input, 1),
                                           SizeOfDimension(input, 2), SizeOfDimension(input, 3), p->filter_height,
                                           p->filter_width, p->stride_height, p->stride_width, p->padding,
                                           (int32_t*)out_dev, nullptr));
  LCE_ENSURE_HIP(context, resident::publish(context, out_idx, readers, output->data.data, out_bytes));
  return kTfLiteOk;
}

}  // namespace bmaxpool

TfLiteRegistration* Register_BMAXPOOL_2D() {  // bmaxpool.cc:95-99
  static TfLiteRegistration r = {bmaxpool::Init, bmaxpool::Free, bmaxpool::Prepare, bmaxpool::Eval};
  return &r;
}

namespace {
namespace resident {
bool is_lce_invoke(TfLiteStatus (*fn)(TfLiteContext*, TfLiteNode*)) {
  return fn == Register_BCONV_2D_REF()->invoke || fn == Register_BCONV_2D_OPT_BGEMM()->invoke ||
         fn == Register_BCONV_2D_OPT_INDIRECT_BGEMM()->invoke || fn == Register_QUANTIZE()->invoke ||
         fn == Register_DEQUANTIZE()->invoke || fn == Register_BMAXPOOL_2D()->invoke;
}
}  // namespace resident
}  // namespace

}  // namespace tflite
}  // namespace compute_engine

// Residency: the host's declaration of its graph, and test / tuning hooks (plain C, not part of the reference's
// interface; declared in lce_ops_register.h).
extern "C" {
// Declaring the graph of `context` (the TfLiteContext the ops of that graph are invoked with):
//   begin; one _node call per node of the execution plan -- its input / output tensor indices (-1 = optional tensor
//   absent) and its registration's invoke function, by which this library recognises its own ops --; one _output call
//   per graph output; end.  May be repeated (a new declaration replaces the old one), before or after AllocateTensors.
void lce_tflite_ops_declare_graph_begin(const TfLiteContext* context) {
  using namespace compute_engine::tflite::resident;
  std::lock_guard<std::mutex> lock(g_mu);
  g_pending[context].clear();
}
void lce_tflite_ops_declare_graph_node(const TfLiteContext* context, const int* inputs, int n_inputs, const int* outputs,
                                       int n_outputs, TfLiteStatus (*invoke)(TfLiteContext*, TfLiteNode*)) {
  using namespace compute_engine::tflite::resident;
  (void)outputs; (void)n_outputs;
  const bool lce = invoke && is_lce_invoke(invoke);
  std::lock_guard<std::mutex> lock(g_mu);
  auto& g = g_pending[context];
  for (int i = 0; i < n_inputs; ++i) {
    if (inputs[i] < 0) continue;
    bool seen = false;                       // a node that lists a tensor twice reads it once
    for (int k = 0; k < i; ++k) seen = seen || inputs[k] == inputs[i];
    if (seen) continue;
    Readers& r = g[inputs[i]];
    if (lce) ++r.lce; else ++r.other;
  }
}
void lce_tflite_ops_declare_graph_output(const TfLiteContext* context, int tensor) {
  using namespace compute_engine::tflite::resident;
  std::lock_guard<std::mutex> lock(g_mu);
  if (tensor >= 0) g_pending[context][tensor].graph_output = true;
}
void lce_tflite_ops_declare_graph_end(const TfLiteContext* context) {
  using namespace compute_engine::tflite::resident;
  std::lock_guard<std::mutex> lock(g_mu);
  g_graphs[context] = std::move(g_pending[context]);
  g_pending.erase(context);
  ++g_version;
}
// forget the declaration: every op of `context` copies its output back again
void lce_tflite_ops_forget_graph(const TfLiteContext* context) {
  using namespace compute_engine::tflite::resident;
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_graphs.erase(context) + g_pending.erase(context)) ++g_version;
}
// device buffers currently held by the residency layer (all contexts): count and bytes
void lce_tflite_ops_device_buffers(uint64_t* count, uint64_t* bytes) {
  using namespace compute_engine::tflite::resident;
  std::lock_guard<std::mutex> lock(g_mu);
  uint64_t n = 0, b = 0;
  for (auto& kv : g_table) if (kv.second.dev) { ++n; b += kv.second.bytes; }
  if (count) *count = n;
  if (bytes) *bytes = b;
}
// 1 / 0: keep LCE-only intermediates on the device / copy every tensor back (also: LCE_HIP_TFLITE_RESIDENCY=0)
void lce_tflite_ops_set_residency(int on) {
  compute_engine::tflite::resident::g_enabled.store(on ? 1 : 0);
  ++compute_engine::tflite::resident::g_version;
}
// host -> device and device -> host tensor copies made by the ops' invokes since the last reset
void lce_tflite_ops_transfer_counts(uint64_t* h2d, uint64_t* d2h, uint64_t* h2d_bytes, uint64_t* d2h_bytes, int reset) {
  using namespace compute_engine::tflite::resident;
  if (h2d) *h2d = g_h2d.load();
  if (d2h) *d2h = g_d2h.load();
  if (h2d_bytes) *h2d_bytes = g_h2d_bytes.load();
  if (d2h_bytes) *d2h_bytes = g_d2h_bytes.load();
  if (reset) { g_h2d = 0; g_d2h = 0; g_h2d_bytes = 0; g_d2h_bytes = 0; }
}
}
