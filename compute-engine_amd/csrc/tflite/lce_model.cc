// C ABI of include/lce_tflite_model.h: model reader + "plan from operator" helper.
// Host-side C++ (the reference's host side is C++); no HIP types here.
#include "../../../include/lce_tflite_model.h"

#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
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
};

namespace {
bool IsLceOp(const lce_tfl::Operator& o) {
  return o.builtin_code == 32 && (o.custom_code == "LceBconv2d" || o.custom_code == "LceQuantize" ||
                                  o.custom_code == "LceDequantize" || o.custom_code == "LceBMaxPool2d");
}
}  // namespace

// The partition a delegate would get (tensorflow/lite/graph_info.cc, PartitionGraphIntoIndependentNodeSubsets, restated from
// its published description): alternate between epochs of LCE operators and epochs of the others; in an epoch every
// operator of the epoch's kind whose inputs are all ready joins, repeatedly, until nothing more can; the LCE operators
// of one epoch are one section.
void lce_tflite_model::Partition() {
  const int n_ops = (int)m.operators.size(), n_t = (int)m.tensors.size();
  std::vector<char> ready(n_t, 1), done(n_ops, 0);
  for (const lce_tfl::Operator& o : m.operators)
    for (int32_t t : o.outputs)
      if (t >= 0 && t < n_t) ready[t] = 0;                   // produced by an operator: not ready until it has run
  int remaining = n_ops;
  bool lce_epoch = true;
  int idle_epochs = 0;
  while (remaining > 0 && idle_epochs < 2) {
    lce_tflite_section sec;
    bool progress = true, any = false;
    while (progress) {
      progress = false;
      for (int i = 0; i < n_ops; ++i) {
        const lce_tfl::Operator& o = m.operators[i];
        if (done[i] || IsLceOp(o) != lce_epoch) continue;
        bool ok = true;
        for (int32_t t : o.inputs) ok = ok && (t < 0 || t >= n_t || ready[t]);
        if (!ok) continue;
        done[i] = 1;
        --remaining;
        progress = any = true;
        for (int32_t t : o.outputs)
          if (t >= 0 && t < n_t) ready[t] = 1;
        if (lce_epoch) sec.ops.push_back(i);
      }
    }
    if (lce_epoch && !sec.ops.empty()) {
      std::sort(sec.ops.begin(), sec.ops.end());
      std::vector<char> inside(n_t, 0), in_sec(n_ops, 0);
      for (int32_t i : sec.ops) {
        in_sec[i] = 1;
        for (int32_t t : m.operators[i].outputs)
          if (t >= 0 && t < n_t) inside[t] = 1;
      }
      for (int32_t i : sec.ops)
        for (int32_t t : m.operators[i].inputs)
          if (t >= 0 && t < n_t && !inside[t] && !m.tensors[t].data &&
              std::find(sec.inputs.begin(), sec.inputs.end(), t) == sec.inputs.end())
            sec.inputs.push_back(t);
      for (int32_t t = 0; t < n_t; ++t) {
        if (!inside[t]) continue;
        bool outside_reader = std::find(m.outputs.begin(), m.outputs.end(), t) != m.outputs.end();
        for (int i = 0; i < n_ops && !outside_reader; ++i)
          if (!in_sec[i]) outside_reader = std::find(m.operators[i].inputs.begin(), m.operators[i].inputs.end(), t) != m.operators[i].inputs.end();
        if (outside_reader) sec.outputs.push_back(t);
      }
      sections.push_back(sec);
    }
    idle_epochs = any ? 0 : idle_epochs + 1;                 // (a graph with a cycle or a dangling input would never finish)
    lce_epoch = !lce_epoch;
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
