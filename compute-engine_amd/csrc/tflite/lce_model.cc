// C ABI of include/lce_tflite_model.h: model reader + "plan from operator" helper.
// Host-side C++ (the reference's host side is C++); no HIP types here.
#include "../../../include/lce_tflite_model.h"

#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "flexbuffer_map.h"
#include "tflite_flatbuffer_reader.h"

struct lce_tflite_section {
  std::vector<int32_t> ops, inputs, outputs;
};
struct lce_tflite_model {
  lce_tfl::Model m;
  std::vector<lce_tflite_section> sections;   // built by Partition() right after parsing
  void Partition();
  // ---- state of lce_tflite_model_run_section (one run at a time per model) ----
  std::mutex run_mu;
  std::map<std::pair<int32_t, int64_t>, lce_hip_bconv2d_plan*> plans;   // (operator, batch * 2 + semantics) -> ready plan
  struct DevBuf { void* ptr = nullptr; size_t bytes = 0; };
  std::map<int32_t, DevBuf> scratch;                                    // intermediate tensors of a section, grow-only
  int32_t last_run_fused = 0;                                           // LceQuantize launches the last run folded into a convolution
  // ---- HIP graphs (lce_tflite_model_use_hip_graphs): a section's launches recorded once per (section, batch, semantics,
  // stream, tensor pointers) and replayed as one launch.  The first call with a key runs eagerly (plans are made, weights
  // uploaded, intermediate buffers sized), the second records, later ones replay.  Recorded launches hold the model's
  // intermediate buffers: when one of those is reallocated every graph is dropped.
  struct GraphKey {
    int32_t section, batch, semantics;
    void* stream;
    std::vector<const void*> ptrs;
    bool operator<(const GraphKey& o) const {
      return std::tie(section, batch, semantics, stream, ptrs) < std::tie(o.section, o.batch, o.semantics, o.stream, o.ptrs);
    }
  };
  struct GraphEntry { int32_t eager_runs = 0; void* graph = nullptr; bool unrecordable = false; int32_t fused = 0; };
  std::map<GraphKey, GraphEntry> graphs;
  bool use_graphs = false;
  int32_t graph_captures = 0, graph_replays = 0;
  void DropGraphs() {
    for (auto& kv : graphs) if (kv.second.graph) lce_hip_graph_destroy(kv.second.graph);
    graphs.clear();
  }
  ~lce_tflite_model() {
    DropGraphs();
    for (auto& kv : plans) lce_hip_bconv2d_plan_destroy(kv.second);
    for (auto& kv : scratch) if (kv.second.ptr) lce_hip_free(kv.second.ptr);
  }
};

namespace {
bool IsLceOp(const lce_tfl::Operator& o) {
  return o.builtin_code == 32 && (o.custom_code == "LceBconv2d" || o.custom_code == "LceQuantize" ||
                                  o.custom_code == "LceDequantize" || o.custom_code == "LceBMaxPool2d");
}
}  // namespace

// The partition a delegate would get (tensorflow/lite/graph_info.cc, PartitionGraphIntoIndependentNodeSubsets, restated from
// its published description): alternate between epochs of LCE operators and epochs of the others, starting with the kind of
// the first ready operator; in an epoch every operator of the epoch's kind whose inputs are all ready joins, repeatedly, until nothing more
// can; the LCE operators of one epoch, sorted by index, are one section.  (On a chain this is "walk the file, cut at
// every builtin operator"; on a branched graph an LCE op further down the file joins an EARLIER section when nothing it
// reads depends on a builtin operator in between.)  Linear in operators + tensor uses: per-tensor reader lists and a count
// of unready inputs per operator; the model is untrusted input.
void lce_tflite_model::Partition() {
  const int n_ops = (int)m.operators.size(), n_t = (int)m.tensors.size();
  auto valid = [&](int32_t t) { return t >= 0 && t < n_t; };
  std::vector<char> produced(n_t, 0), is_output(n_t, 0), is_lce(n_ops, 0);
  std::vector<std::vector<int32_t>> readers(n_t);
  std::vector<int32_t> unready(n_ops, 0);
  for (int i = 0; i < n_ops; ++i) {
    is_lce[i] = IsLceOp(m.operators[i]) ? 1 : 0;
    for (int32_t t : m.operators[i].outputs)
      if (valid(t)) produced[t] = 1;                         // produced by an operator: not ready until it has run
  }
  for (int i = 0; i < n_ops; ++i)
    for (int32_t t : m.operators[i].inputs)
      if (valid(t)) {
        readers[t].push_back(i);
        if (produced[t]) ++unready[i];
      }
  for (int32_t t : m.outputs)
    if (valid(t)) is_output[t] = 1;
  // ready operators of either kind, waiting for their epoch
  std::vector<int32_t> queue[2];
  for (int i = 0; i < n_ops; ++i)
    if (unready[i] == 0) queue[(int)is_lce[i]].push_back(i);
  std::vector<int32_t> section_of(n_ops, -1);
  std::vector<char> made(n_t, 0), listed(n_t, 0);
  int remaining = n_ops;
  // the first epoch has the kind of the first ready operator in execution order (graph_info.cc takes it from there: usually a
  // builtin stem operator), so an LCE operator that is ready at the start beside a builtin one lands in the same section a
  // delegate would be handed
  int kind = 1;                                              // 1: LCE epoch
  if (!queue[0].empty() && (queue[1].empty() || queue[0].front() < queue[1].front())) kind = 0;
  int idle_epochs = 0;
  while (remaining > 0 && idle_epochs < 2) {                 // (a graph with a cycle or a dangling input never finishes)
    std::vector<int32_t>& q = queue[kind];
    lce_tflite_section sec;
    bool any = false;
    for (size_t head = 0; head < q.size(); ++head) {         // grows while it is walked
      const int32_t i = q[head];
      any = true;
      --remaining;
      if (kind) sec.ops.push_back(i);
      for (int32_t t : m.operators[i].outputs) {
        if (!valid(t) || made[t]) continue;
        made[t] = 1;
        for (int32_t r : readers[t])
          if (--unready[r] == 0) queue[(int)is_lce[r]].push_back(r);
      }
    }
    q.clear();
    if (kind && !sec.ops.empty()) {
      std::sort(sec.ops.begin(), sec.ops.end());
      const int32_t id = (int32_t)sections.size();
      for (int32_t i : sec.ops) section_of[i] = id;
      std::vector<int32_t> ins, outs;
      for (int32_t i : sec.ops) {
        for (int32_t t : m.operators[i].inputs)
          if (valid(t) && !m.tensors[t].data) ins.push_back(t);
        for (int32_t t : m.operators[i].outputs)
          if (valid(t)) outs.push_back(t);
      }
      std::sort(outs.begin(), outs.end());
      outs.erase(std::unique(outs.begin(), outs.end()), outs.end());
      for (int32_t t : ins)                                  // first-use order
        if (!listed[t] && !std::binary_search(outs.begin(), outs.end(), t)) {
          listed[t] = 1;
          sec.inputs.push_back(t);
        }
      for (int32_t t : sec.inputs) listed[t] = 0;
      for (int32_t t : outs) {
        bool outside_reader = is_output[t] != 0;
        for (size_t k = 0; k < readers[t].size() && !outside_reader; ++k) outside_reader = section_of[readers[t][k]] != id;
        if (outside_reader) sec.outputs.push_back(t);
      }
      sections.push_back(sec);
    }
    idle_epochs = any ? 0 : idle_epochs + 1;
    kind ^= 1;
  }
}

namespace {
thread_local std::string g_model_error;
lce_hip_status Fail(lce_hip_status code, const std::string& msg) {
  g_model_error = msg;
  return code;
}
}  // namespace

extern "C" {

lce_tflite_model* lce_tflite_model_open(const void* data, size_t size, char* err, size_t err_len) {
  auto* model = new (std::nothrow) lce_tflite_model{};
  std::string e = "out of memory";
  if (model && data && model->m.Parse(data, size, &e)) {
    model->Partition();
    return model;
  }
  if (!data) e = "null buffer";
  if (err && err_len) snprintf(err, err_len, "%s", e.c_str());
  delete model;
  return nullptr;
}

void lce_tflite_model_close(lce_tflite_model* model) { delete model; }

int32_t lce_tflite_model_num_tensors(const lce_tflite_model* model) { return model ? (int32_t)model->m.tensors.size() : 0; }
int32_t lce_tflite_model_num_operators(const lce_tflite_model* model) { return model ? (int32_t)model->m.operators.size() : 0; }

static int32_t copy_indices(const std::vector<int32_t>& v, int32_t* out, int32_t cap) {
  for (int32_t i = 0; out && i < cap && i < (int32_t)v.size(); ++i) out[i] = v[i];
  return (int32_t)v.size();
}
int32_t lce_tflite_model_inputs(const lce_tflite_model* model, int32_t* indices, int32_t cap) {
  return model ? copy_indices(model->m.inputs, indices, cap) : 0;
}
int32_t lce_tflite_model_outputs(const lce_tflite_model* model, int32_t* indices, int32_t cap) {
  return model ? copy_indices(model->m.outputs, indices, cap) : 0;
}

lce_hip_status lce_tflite_model_tensor(const lce_tflite_model* model, int32_t index, lce_tflite_tensor_info* info) {
  if (!model || !info || index < 0 || index >= (int32_t)model->m.tensors.size())
    return Fail(LCE_HIP_ERR_INVALID, "lce_tflite_model_tensor: bad argument");
  const lce_tfl::Tensor& t = model->m.tensors[index];
  if (t.shape.size() > 8) return Fail(LCE_HIP_ERR_UNSUPPORTED, "lce_tflite_model_tensor: rank > 8");
  memset(info, 0, sizeof *info);
  info->type = t.type;
  info->rank = (int32_t)t.shape.size();
  for (size_t i = 0; i < t.shape.size(); ++i) info->dims[i] = t.shape[i];
  info->quantized = t.quantized ? 1 : 0;
  info->scale = t.scale;
  info->zero_point = (int32_t)t.zero_point;
  info->data = t.data;
  info->bytes = t.bytes;
  info->name = t.name.c_str();
  return LCE_HIP_OK;
}

lce_hip_status lce_tflite_model_operator(const lce_tflite_model* model, int32_t index, lce_tflite_operator_info* info) {
  if (!model || !info || index < 0 || index >= (int32_t)model->m.operators.size())
    return Fail(LCE_HIP_ERR_INVALID, "lce_tflite_model_operator: bad argument");
  const lce_tfl::Operator& o = model->m.operators[index];
  info->builtin_code = o.builtin_code;
  info->custom_code = o.custom_code.c_str();
  info->inputs = o.inputs.data();
  info->num_inputs = (int32_t)o.inputs.size();
  info->outputs = o.outputs.data();
  info->num_outputs = (int32_t)o.outputs.size();
  info->custom_options = o.custom_options;
  info->custom_options_size = o.custom_options_size;
  return LCE_HIP_OK;
}

int32_t lce_tflite_model_num_sections(const lce_tflite_model* model) { return model ? (int32_t)model->sections.size() : 0; }
lce_hip_status lce_tflite_model_section(const lce_tflite_model* model, int32_t index, lce_tflite_section_info* info) {
  if (!model || !info || index < 0 || index >= (int32_t)model->sections.size())
    return Fail(LCE_HIP_ERR_INVALID, "lce_tflite_model_section: bad argument");
  const lce_tflite_section& s = model->sections[index];
  info->ops = s.ops.data();
  info->num_ops = (int32_t)s.ops.size();
  info->inputs = s.inputs.data();
  info->num_inputs = (int32_t)s.inputs.size();
  info->outputs = s.outputs.data();
  info->num_outputs = (int32_t)s.outputs.size();
  return LCE_HIP_OK;
}

int lce_tflite_option_int(const uint8_t* custom_options, size_t size, const char* key, int32_t* value) {
  const lce_flex::Map m(custom_options, size);
  if (!m.valid() || !key || m.IsNull(key)) return 1;
  if (value) *value = m.AsInt32(key);
  return 0;
}

lce_hip_status lce_tflite_model_bconv2d_plan(const lce_tflite_model* model, int32_t index, int32_t batch,
                                             int32_t semantics, lce_hip_bconv2d_plan** plan) {
  g_model_error.clear();   // failures inside the GPU library report through lce_hip_last_error()
  if (!model || !plan || index < 0 || index >= (int32_t)model->m.operators.size())
    return Fail(LCE_HIP_ERR_INVALID, "lce_tflite_model_bconv2d_plan: bad argument");
  const lce_tfl::Model& M = model->m;
  const lce_tfl::Operator& op = M.operators[index];
  if (op.builtin_code != lce_tfl::kBuiltinCustom || op.custom_code != "LceBconv2d")
    return Fail(LCE_HIP_ERR_INVALID, "lce_tflite_model_bconv2d_plan: operator is not an LceBconv2d");
  // bconv2d.cc:145-152: 5 inputs (input, filter, post_activation_multiplier, post_activation_bias,
  // output_threshold), 1 output
  if (op.inputs.size() != 5 || op.outputs.size() != 1)
    return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: expected 5 inputs and 1 output");
  for (int i = 0; i < 2; ++i)
    if (op.inputs[i] < 0) return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: input and filter are required");
  const lce_tfl::Tensor& in = M.tensors[op.inputs[0]];
  const lce_tfl::Tensor& filter = M.tensors[op.inputs[1]];
  const lce_tfl::Tensor& out = M.tensors[op.outputs[0]];
  if (in.shape.size() != 4 || filter.shape.size() != 4 || in.type != lce_tfl::kTensorInt32 ||
      filter.type != lce_tfl::kTensorInt32)
    return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: input and filter must be 4-D int32 (bitpacked)");

  // Init: the option map (bconv2d.cc:85-131)
  const lce_flex::Map m(op.custom_options, op.custom_options_size);
  static const char* const kRequired[] = {"stride_height", "stride_width", "dilation_height_factor",
                                          "dilation_width_factor", "padding", "pad_values",
                                          "channels_in", "fused_activation_function"};
  for (const char* key : kRequired)
    if (!m.valid() || m.IsNull(key)) return Fail(LCE_HIP_ERR_INVALID, std::string("LceBconv2d: option missing: ") + key);

  lce_hip_bconv2d_desc d;
  memset(&d, 0, sizeof d);
  d.batch = batch > 0 ? batch : in.shape[0];
  d.in_height = in.shape[1];
  d.in_width = in.shape[2];
  d.channels_in = m.AsInt32("channels_in");
  d.channels_out = filter.shape[0];
  d.filter_height = filter.shape[1];
  d.filter_width = filter.shape[2];
  if (d.channels_in <= 0) return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: channels_in must be positive");
  // groups from the filter's packed depth (bconv2d.cc:169-186)
  const int32_t cw = (d.channels_in + 31) / 32;
  if (in.shape[3] != cw) return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: input depth does not match channels_in");
  if (filter.shape[3] == cw) {
    d.groups = 1;
  } else {
    if (filter.shape[3] <= 0 || cw % filter.shape[3] != 0)
      return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: filter depth does not divide the input depth");
    d.groups = cw / filter.shape[3];
  }
  d.stride_height = m.AsInt32("stride_height");
  d.stride_width = m.AsInt32("stride_width");
  d.dilation_height = m.AsInt32("dilation_height_factor");
  d.dilation_width = m.AsInt32("dilation_width_factor");
  d.padding = m.AsInt32("padding");
  d.pad_values = m.AsInt32("pad_values");
  const int act = m.AsInt32("fused_activation_function");   // ConvertActivation, tflite/kernels/utils.h:10-25
  d.activation = (act >= LCE_HIP_ACT_NONE && act <= LCE_HIP_ACT_RELU6) ? act : LCE_HIP_ACT_NONE;
  d.semantics = semantics;
  d.out_scale = 1.0f;
  switch (out.type) {   // bconv2d.cc:158-162
    case lce_tfl::kTensorFloat32: d.dst_type = LCE_HIP_F32; break;
    case lce_tfl::kTensorInt8:
      d.dst_type = LCE_HIP_I8;
      if (!out.quantized) return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: int8 output without quantization parameters");
      d.out_scale = out.scale;
      d.out_zero_point = (int32_t)out.zero_point;
      break;
    case lce_tfl::kTensorInt32: d.dst_type = LCE_HIP_BITPACKED; break;
    default: return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: output type must be float32, int8 or int32");
  }

  // OneTimeSetup's sources: constant tensors
  auto constant = [&](int slot, int type, size_t count, const void** p) -> bool {
    *p = nullptr;
    const int32_t ti = op.inputs[slot];
    if (ti < 0) return true;
    const lce_tfl::Tensor& t = M.tensors[ti];
    if (!t.data) return true;                                  // "none" placeholder tensor
    const size_t esz = 4;
    if (t.type != type || t.bytes != count * esz) return false;
    *p = t.data;
    return true;
  };
  const size_t fcount = (size_t)filter.shape[0] * filter.shape[1] * filter.shape[2] * filter.shape[3];
  const void *fw, *mul, *bias, *thr;
  if (!constant(1, lce_tfl::kTensorInt32, fcount, &fw) || !fw)
    return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: the filter must be a constant int32 tensor of the declared shape");
  if (!constant(2, lce_tfl::kTensorFloat32, (size_t)d.channels_out, &mul) ||
      !constant(3, lce_tfl::kTensorFloat32, (size_t)d.channels_out, &bias) ||
      !constant(4, lce_tfl::kTensorInt32, (size_t)d.channels_out, &thr))
    return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: per-channel constants have the wrong type or size");
  if (d.dst_type == LCE_HIP_BITPACKED ? !thr : (!mul || !bias))
    return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: missing thresholds (int32 output) or multiplier/bias (float/int8 output)");

  lce_hip_bconv2d_plan* p = nullptr;
  if (lce_hip_status s = lce_hip_bconv2d_plan_create(&d, &p)) return s;   // message in lce_hip_last_error()
  // flatbuffer vectors are only guaranteed 4-byte aligned; the library copies them
  if (lce_hip_status s = lce_hip_bconv2d_plan_set_weights(p, (const int32_t*)fw, (const float*)mul,
                                                          (const float*)bias, (const int32_t*)thr)) {
    lce_hip_bconv2d_plan_destroy(p);
    return s;
  }
  *plan = p;
  return LCE_HIP_OK;
}

const char* lce_tflite_model_last_error(void) { return g_model_error.c_str(); }

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// Running a binary section on device tensors (the C counterpart of examples/lce_minimal.cc:28-62 for a host without a
// TensorFlow Lite interpreter, and what compute-engine_amd/model_runner.py calls).
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct Shape {
  int32_t dims[4] = {0, 0, 0, 0};
  int type = 0;
  size_t bytes() const {
    const size_t esz = (type == lce_tfl::kTensorInt8 || type == lce_tfl::kTensorBool) ? 1 : 4;
    return (size_t)dims[0] * dims[1] * dims[2] * dims[3] * esz;
  }
};

lce_hip_status PlanFor(lce_tflite_model* model, int32_t op, int32_t batch, int32_t semantics, lce_hip_bconv2d_plan** plan) {
  const auto key = std::make_pair(op, (int64_t)batch * 2 + (semantics ? 1 : 0));
  auto it = model->plans.find(key);
  if (it == model->plans.end()) {
    lce_hip_bconv2d_plan* p = nullptr;
    if (lce_hip_status s = lce_tflite_model_bconv2d_plan(model, op, batch, semantics, &p)) return s;
    it = model->plans.emplace(key, p).first;
  }
  *plan = it->second;
  return LCE_HIP_OK;
}

// The LceQuantize operators of section `sec` that read the float / int8 output of LceBconv2d `conv` (their result is
// the second output of the convolution's epilogue: lce_hip_bconv2d_run_dual).
std::vector<int32_t> QuantizeConsumers(const lce_tflite_model* model, const lce_tflite_section& sec, int32_t conv) {
  std::vector<int32_t> js;
  const lce_tfl::Operator& op = model->m.operators[conv];
  const int out_type = model->m.tensors[op.outputs[0]].type;
  if (out_type != lce_tfl::kTensorFloat32 && out_type != lce_tfl::kTensorInt8) return js;
  for (int32_t j : sec.ops) {
    const lce_tfl::Operator& q = model->m.operators[j];
    if (j > conv && q.custom_code == "LceQuantize" && q.inputs.size() == 1 && q.inputs[0] == op.outputs[0]) js.push_back(j);
  }
  return js;
}

// Walks section `sec` at `batch` images: shapes of every tensor it touches (shape inference exactly as the ops' Prepare
// does it) and, with `run`, the launches.  `ptr` maps tensor -> device pointer (section inputs and outputs on entry;
// intermediates are added from the model's scratch buffers).
lce_hip_status WalkSection(lce_tflite_model* model, const lce_tflite_section& sec, int32_t batch, int32_t semantics,
                           std::map<int32_t, Shape>* shapes, std::map<int32_t, void*>* ptr, bool run, void* stream,
                           bool capturing = false) {
  const lce_tfl::Model& M = model->m;
  for (int32_t t : sec.inputs) {
    const lce_tfl::Tensor& T = M.tensors[t];
    if (T.shape.size() != 4) return Fail(LCE_HIP_ERR_UNSUPPORTED, "run_section: section inputs must be 4-D tensors (NHWC)");
    Shape sh;
    for (int k = 0; k < 4; ++k) sh.dims[k] = T.shape[k];
    sh.dims[0] = batch;
    sh.type = T.type;
    (*shapes)[t] = sh;
  }
  auto buffer_for = [&](int32_t t, size_t bytes, void** out) -> lce_hip_status {
    auto it = ptr->find(t);
    if (it != ptr->end()) { *out = it->second; return LCE_HIP_OK; }
    lce_tflite_model::DevBuf& b = model->scratch[t];
    if (b.bytes < bytes) {
      // while `stream` records a graph nothing may be freed or allocated (a hipFree / hipMalloc inside a capture invalidates it and
      // the pointer would leak): the eager run before the recording sized every buffer, so this only fires if a size changed
      if (capturing) return Fail(LCE_HIP_ERR_INVALID, "run_section: a scratch buffer would have to grow during graph capture");
      // (a buffer a previous run's kernels may still use: the free below is ordered behind them by the runtime)
      if (b.ptr) lce_hip_free(b.ptr);
      model->DropGraphs();          // (recorded launches may hold the old pointer; the free above has drained the device)
      b.ptr = nullptr;
      b.bytes = 0;
      if (lce_hip_status s = lce_hip_malloc(&b.ptr, bytes ? bytes : 1)) return s;
      b.bytes = bytes;
    }
    (*ptr)[t] = b.ptr;
    *out = b.ptr;
    return LCE_HIP_OK;
  };
  std::vector<char> done(M.operators.size(), 0);
  for (int32_t i : sec.ops) {
    if (done[i]) continue;
    const lce_tfl::Operator& op = M.operators[i];
    if (op.inputs.empty() || op.outputs.size() != 1 || op.inputs[0] < 0)
      return Fail(LCE_HIP_ERR_INVALID, "run_section: malformed LCE operator");
    auto in_it = shapes->find(op.inputs[0]);
    if (in_it == shapes->end()) return Fail(LCE_HIP_ERR_INVALID, "run_section: an operator reads a tensor nothing produced");
    const Shape in = in_it->second;
    const int32_t out_t = op.outputs[0];
    const lce_tfl::Tensor& OT = M.tensors[out_t];
    Shape out = in;
    out.type = OT.type;
    const void* in_dev = nullptr;
    if (run) {
      auto p = ptr->find(op.inputs[0]);
      if (p == ptr->end() || !p->second) return Fail(LCE_HIP_ERR_INVALID, "run_section: missing device pointer of an input tensor");
      in_dev = p->second;
    }
    if (op.custom_code == "LceQuantize") {                      // quantization.cc:19-41,76-114
      if (in.type != lce_tfl::kTensorFloat32 && in.type != lce_tfl::kTensorInt8 && in.type != lce_tfl::kTensorBool)
        return Fail(LCE_HIP_ERR_INVALID, "LceQuantize: input must be float32, int8 or bool");
      out.dims[3] = (in.dims[3] + 31) / 32;
      out.type = lce_tfl::kTensorInt32;
      (*shapes)[out_t] = out;
      if (run) {
        void* o = nullptr;
        if (lce_hip_status s = buffer_for(out_t, out.bytes(), &o)) return s;
        const lce_tfl::Tensor& IT = M.tensors[op.inputs[0]];
        const lce_hip_dtype t = in.type == lce_tfl::kTensorFloat32 ? LCE_HIP_F32 : in.type == lce_tfl::kTensorInt8 ? LCE_HIP_I8 : LCE_HIP_BOOL;
        const int32_t zp = in.type == lce_tfl::kTensorInt8 ? (int32_t)IT.zero_point : in.type == lce_tfl::kTensorBool ? 1 : 0;
        if (lce_hip_status s = lce_hip_bitpack(t, in_dev, (size_t)in.dims[0] * in.dims[1] * in.dims[2], (size_t)in.dims[3], zp, (int32_t*)o, stream)) return s;
      }
    } else if (op.custom_code == "LceDequantize") {             // quantization.cc:43-74,116-147
      if (OT.shape.size() != 4) return Fail(LCE_HIP_ERR_UNSUPPORTED, "LceDequantize: the output's channel count comes from the file (4-D)");
      out.dims[3] = OT.shape[3];
      if ((out.dims[3] + 31) / 32 != in.dims[3]) return Fail(LCE_HIP_ERR_INVALID, "LceDequantize: output channels do not match the packed input");
      (*shapes)[out_t] = out;
      if (run) {
        void* o = nullptr;
        if (lce_hip_status s = buffer_for(out_t, out.bytes(), &o)) return s;
        const lce_hip_dtype t = out.type == lce_tfl::kTensorFloat32 ? LCE_HIP_F32 : out.type == lce_tfl::kTensorInt8 ? LCE_HIP_I8 : LCE_HIP_BOOL;
        if (lce_hip_status s = lce_hip_unpack(t, (const int32_t*)in_dev, (size_t)out.dims[0] * out.dims[1] * out.dims[2], (size_t)out.dims[3],
                                              OT.quantized ? OT.scale : 1.0f, OT.quantized ? (int32_t)OT.zero_point : 0, o, stream)) return s;
      }
    } else if (op.custom_code == "LceBMaxPool2d") {             // bmaxpool.cc:20-91
      if (in.type != lce_tfl::kTensorInt32) return Fail(LCE_HIP_ERR_INVALID, "LceBMaxPool2d: input must be bitpacked int32");
      const lce_flex::Map fm(op.custom_options, op.custom_options_size);
      if (!fm.valid()) return Fail(LCE_HIP_ERR_INVALID, "LceBMaxPool2d: unreadable options");
      const int32_t fh = fm.AsInt32("filter_height"), fw = fm.AsInt32("filter_width"), sh = fm.AsInt32("stride_height"),
                    sw = fm.AsInt32("stride_width"), pad = fm.AsInt32("padding");
      if (lce_hip_status s = lce_hip_bmaxpool_output_shape(in.dims[1], in.dims[2], fh, fw, sh, sw, pad, &out.dims[1], &out.dims[2])) return s;
      (*shapes)[out_t] = out;
      if (run) {
        void* o = nullptr;
        if (lce_hip_status s = buffer_for(out_t, out.bytes(), &o)) return s;
        if (lce_hip_status s = lce_hip_bmaxpool((const int32_t*)in_dev, in.dims[0], in.dims[1], in.dims[2], in.dims[3], fh, fw, sh, sw, pad, (int32_t*)o, stream)) return s;
      }
    } else if (op.custom_code == "LceBconv2d") {                // bconv2d.cc:137-300,550-564
      // The plan is built from the FILE's static shape of the input tensor (lce_tflite_model_bconv2d_plan); the buffer it will
      // read was sized from THIS walk's shape inference.  A model whose declared shapes disagree with what its operators
      // produce (dynamic / -1 dimensions, a hand-edited file, an int8 tensor wired into a convolution) must fail here, not read
      // past a scratch buffer on the device: the file is untrusted input.
      {
        const lce_tfl::Tensor& IT = M.tensors[op.inputs[0]];
        if (in.type != lce_tfl::kTensorInt32)
          return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: the tensor feeding the convolution is not bitpacked int32");
        if (IT.shape.size() != 4 || IT.shape[1] != in.dims[1] || IT.shape[2] != in.dims[2] || IT.shape[3] != in.dims[3])
          return Fail(LCE_HIP_ERR_INVALID, "LceBconv2d: the input tensor's declared shape does not match the shape its producer infers");
      }
      lce_hip_bconv2d_plan* plan = nullptr;
      if (lce_hip_status s = PlanFor(model, i, batch, semantics, &plan)) return s;
      if (lce_hip_status s = lce_hip_bconv2d_plan_output_shape(plan, out.dims)) return s;
      (*shapes)[out_t] = out;
      const std::vector<int32_t> fused = QuantizeConsumers(model, sec, i);
      for (int32_t j : fused) {                                 // their shapes; the first one's launch disappears
        Shape q = out;
        q.dims[3] = (out.dims[3] + 31) / 32;
        q.type = lce_tfl::kTensorInt32;
        (*shapes)[M.operators[j].outputs[0]] = q;
      }
      if (!fused.empty()) done[fused[0]] = 1;
      if (run) {
        void* o = nullptr;
        if (lce_hip_status s = buffer_for(out_t, out.bytes(), &o)) return s;
        if (fused.empty()) {
          if (lce_hip_status s = lce_hip_bconv2d_run(plan, (const int32_t*)in_dev, o, stream)) return s;
        } else {
          void* bits = nullptr;
          const int32_t bits_t = M.operators[fused[0]].outputs[0];
          if (lce_hip_status s = buffer_for(bits_t, (*shapes)[bits_t].bytes(), &bits)) return s;
          if (lce_hip_status s = lce_hip_bconv2d_run_dual(plan, (const int32_t*)in_dev, o, (int32_t*)bits, stream)) return s;
          ++model->last_run_fused;
        }
      }
    } else {
      return Fail(LCE_HIP_ERR_UNSUPPORTED, "run_section: operator " + op.custom_code + " is not an LCE op");
    }
  }
  return LCE_HIP_OK;
}
}  // namespace

extern "C" {

lce_hip_status lce_tflite_model_section_tensor_shape(lce_tflite_model* model, int32_t section, int32_t tensor, int32_t batch,
                                                     int32_t semantics, int32_t dims[4], size_t* bytes) {
  g_model_error.clear();
  if (!model || section < 0 || section >= (int32_t)model->sections.size() || batch <= 0)
    return Fail(LCE_HIP_ERR_INVALID, "lce_tflite_model_section_tensor_shape: bad argument");
  std::lock_guard<std::mutex> lock(model->run_mu);
  std::map<int32_t, Shape> shapes;
  std::map<int32_t, void*> ptr;
  if (lce_hip_status s = WalkSection(model, model->sections[section], batch, semantics, &shapes, &ptr, false, nullptr)) return s;
  auto it = shapes.find(tensor);
  if (it == shapes.end()) return Fail(LCE_HIP_ERR_INVALID, "lce_tflite_model_section_tensor_shape: the section does not touch this tensor");
  for (int k = 0; dims && k < 4; ++k) dims[k] = it->second.dims[k];
  if (bytes) *bytes = it->second.bytes();
  return LCE_HIP_OK;
}

lce_hip_status lce_tflite_model_run_section(lce_tflite_model* model, int32_t section, int32_t batch, int32_t semantics,
                                            const void* const* inputs_dev, void* const* outputs_dev, void* stream) {
  g_model_error.clear();
  if (!model || section < 0 || section >= (int32_t)model->sections.size() || batch <= 0 || !inputs_dev || !outputs_dev)
    return Fail(LCE_HIP_ERR_INVALID, "lce_tflite_model_run_section: bad argument");
  const lce_tflite_section& sec = model->sections[section];
  std::lock_guard<std::mutex> lock(model->run_mu);
  std::map<int32_t, Shape> shapes;
  std::map<int32_t, void*> ptr;
  for (size_t k = 0; k < sec.inputs.size(); ++k) {
    if (!inputs_dev[k]) return Fail(LCE_HIP_ERR_INVALID, "lce_tflite_model_run_section: null input pointer");
    ptr[sec.inputs[k]] = const_cast<void*>(inputs_dev[k]);
  }
  for (size_t k = 0; k < sec.outputs.size(); ++k) {
    if (!outputs_dev[k]) return Fail(LCE_HIP_ERR_INVALID, "lce_tflite_model_run_section: null output pointer");
    ptr[sec.outputs[k]] = outputs_dev[k];
  }
  model->last_run_fused = 0;
  if (!model->use_graphs || !stream) return WalkSection(model, sec, batch, semantics, &shapes, &ptr, true, stream);

  lce_tflite_model::GraphKey key{section, batch, semantics, stream, {}};
  for (size_t k = 0; k < sec.inputs.size(); ++k) key.ptrs.push_back(inputs_dev[k]);
  for (size_t k = 0; k < sec.outputs.size(); ++k) key.ptrs.push_back(outputs_dev[k]);
  {
    lce_tflite_model::GraphEntry& e = model->graphs[key];
    if (e.graph) {
      model->last_run_fused = e.fused;
      ++model->graph_replays;
      return lce_hip_graph_launch(e.graph, stream);
    }
    if (e.eager_runs >= 1 && !e.unrecordable) {
      // record: the same walk, on a capturing stream (nothing executes); whatever goes wrong, the section then runs eagerly
      bool recorded = false;
      void* g = nullptr;
      int32_t fused = 0;
      if (lce_hip_graph_begin_capture(stream) == LCE_HIP_OK) {
        std::map<int32_t, Shape> shapes_c;
        std::map<int32_t, void*> ptr_c = ptr;
        const lce_hip_status walked = WalkSection(model, sec, batch, semantics, &shapes_c, &ptr_c, true, stream, /*capturing=*/true);
        const lce_hip_status ended = lce_hip_graph_end_capture(stream, &g);
        recorded = walked == LCE_HIP_OK && ended == LCE_HIP_OK && g != nullptr;
        fused = model->last_run_fused;
        if (!recorded && g) { lce_hip_graph_destroy(g); g = nullptr; }
      }
      g_model_error.clear();
      lce_tflite_model::GraphEntry& e2 = model->graphs[key];     // (the walk may have dropped the table)
      if (recorded) {
        e2.graph = g;
        e2.fused = fused;
        e2.eager_runs = 1;
        ++model->graph_captures;
        ++model->graph_replays;
        return lce_hip_graph_launch(g, stream);
      }
      e2.unrecordable = true;
      model->last_run_fused = 0;
    }
  }
  const lce_hip_status s = WalkSection(model, sec, batch, semantics, &shapes, &ptr, true, stream);
  if (s == LCE_HIP_OK) ++model->graphs[key].eager_runs;
  return s;
}

void lce_tflite_model_use_hip_graphs(lce_tflite_model* model, int32_t on) {
  if (!model) return;
  std::lock_guard<std::mutex> lock(model->run_mu);
  model->use_graphs = on != 0;
  if (!on) model->DropGraphs();
}

void lce_tflite_model_graph_stats(lce_tflite_model* model, int32_t* recorded, int32_t* replays) {
  if (!model) return;
  std::lock_guard<std::mutex> lock(model->run_mu);
  if (recorded) *recorded = model->graph_captures;
  if (replays) *replays = model->graph_replays;
}

void lce_tflite_model_run_stats(lce_tflite_model* model, int32_t* cached_plans, int32_t* fused_quantize_ops, size_t* scratch_bytes) {
  if (!model) return;
  std::lock_guard<std::mutex> lock(model->run_mu);
  if (cached_plans) *cached_plans = (int32_t)model->plans.size();
  if (fused_quantize_ops) *fused_quantize_ops = model->last_run_fused;
  size_t b = 0;
  for (auto& kv : model->scratch) b += kv.second.bytes;
  if (scratch_bytes) *scratch_bytes = b;
}

}  // extern "C"
