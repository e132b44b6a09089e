// A tiny stand-in for the part of the TFLite interpreter that drives custom-op nodes (one, as TFLite's SingleOpModel
// does in the reference's op tests, or a short chain of them with an execution plan the ops can inspect):
// resolver lookup -> init(options) -> prepare -> invoke -> free, with a tensor arena,
// ResizeTensor / AddTensors / ReportError callbacks and the TfLiteIntArray helpers.  It
// plays the role TFLite's SingleOpModel plays in the reference's op tests
// (tflite/tests/bconv2d_op_model.h:24-59) so that the Register_* surface can be exercised
// without TensorFlow, and exposes a plain C interface for ctypes.  Test tooling: a real
// deployment links lce_ops.cc into TFLite instead.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "flexbuffer_map.h"
#include "lce_ops_register.h"
#include "tflite_abi.h"

extern "C" {
// tensorflow/lite/core/c/common.cc equivalents (malloc-based, like TFLite's)
TfLiteIntArray* TfLiteIntArrayCreate(int size) {
  TfLiteIntArray* a = (TfLiteIntArray*)malloc(sizeof(TfLiteIntArray) + sizeof(int) * (size_t)(size > 0 ? size : 0));
  if (a) a->size = size;
  return a;
}
TfLiteIntArray* TfLiteIntArrayCopy(const TfLiteIntArray* src) {
  if (!src) return nullptr;
  TfLiteIntArray* a = TfLiteIntArrayCreate(src->size);
  if (a) memcpy(a->data, src->data, sizeof(int) * (size_t)src->size);
  return a;
}
void TfLiteIntArrayFree(TfLiteIntArray* a) { free(a); }
}

namespace {

size_t type_size(TfLiteType t) {
  switch (t) {
    case kTfLiteFloat32: case kTfLiteInt32: return 4;
    case kTfLiteInt8: case kTfLiteUInt8: return 1;
    case kTfLiteBool: return sizeof(bool);
    case kTfLiteInt64: return 8;
    default: return 0;
  }
}

struct Resolver {  // the AddCustom half of ::tflite::MutableOpResolver
  std::map<std::string, const TfLiteRegistration*> ops;
  void AddCustom(const char* name, const TfLiteRegistration* r) { ops[name] = r; }
};

struct Model {
  TfLiteContext ctx{};
  std::vector<TfLiteTensor> tensors;
  std::vector<std::vector<char>> storage;
  std::vector<TfLiteAffineQuantization> quant;
  TfLiteNode node{};                       // node 0 (the single-op interface)
  const TfLiteRegistration* reg = nullptr;
  std::string log;
  bool inited = false;
  // further nodes of a chain (lce_driver_add_node): node i + 1 of the execution plan
  struct Extra {
    TfLiteNode node{};
    const TfLiteRegistration* reg = nullptr;
    std::string options;
  };
  std::vector<Extra*> extra;

  static Model* self(TfLiteContext* c) { return (Model*)c->impl_; }

  static TfLiteStatus Resize(TfLiteContext* c, TfLiteTensor* t, TfLiteIntArray* dims) {
    Model* m = self(c);
    size_t n = type_size(t->type);
    for (int i = 0; i < dims->size; ++i) n *= (size_t)(dims->data[i] > 0 ? dims->data[i] : 0);
    const size_t idx = (size_t)(t - m->tensors.data());
    m->storage[idx].assign(n + 16, 0);   // LCE_EXTRA_BYTES of slack (core/types.h:35-38)
    t->data.raw = m->storage[idx].data();
    t->bytes = n;
    TfLiteIntArrayFree(t->dims);
    t->dims = dims;  // takes ownership, like the interpreter
    return kTfLiteOk;
  }
  static void Report(TfLiteContext* c, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    Model* m = self(c);
    m->log += buf;
    m->log += "\n";
  }
  static TfLiteStatus AddTensors(TfLiteContext* c, int n, int* first) {
    Model* m = self(c);
    if (first) *first = (int)m->tensors.size();
    for (int i = 0; i < n; ++i) m->add(kTfLiteNoType, 0, nullptr, kTfLiteArenaRw);
    return kTfLiteOk;
  }

  // What TensorFlow Lite installs for these two callbacks while kernels run (Subgraph::SwitchToKernelContext ->
  // ForbiddenContextFunction): an error report and kTfLiteError.  The ops must never call them; a test greps the log.
  static TfLiteStatus ForbiddenPlan(TfLiteContext* c, TfLiteIntArray**) {
    self(c)->log += "The function is forbidden if not calling in delegate.\n";
    return kTfLiteError;
  }
  static TfLiteStatus ForbiddenNode(TfLiteContext* c, int, TfLiteNode**, void**) {
    self(c)->log += "The function is forbidden if not calling in delegate.\n";
    return kTfLiteError;
  }

  Model() {
    tensors.reserve(64);
    storage.reserve(64);
    quant.reserve(64);
    ctx.impl_ = this;
    ctx.ResizeTensor = Resize;
    ctx.ReportError = Report;
    ctx.AddTensors = AddTensors;
    ctx.recommended_num_threads = 1;
    ctx.GetExecutionPlan = ForbiddenPlan;
    ctx.GetNodeAndRegistration = ForbiddenNode;
  }
  ~Model() {
    if (reg && reg->free && inited) reg->free(&ctx, node.user_data);
    for (Extra* e : extra) {
      if (e->reg && e->reg->free) e->reg->free(&ctx, e->node.user_data);
      TfLiteIntArrayFree(e->node.inputs);
      TfLiteIntArrayFree(e->node.outputs);
      TfLiteIntArrayFree(e->node.temporaries);
      delete e;
    }
    for (auto& t : tensors) TfLiteIntArrayFree(t.dims);
    TfLiteIntArrayFree(node.inputs);
    TfLiteIntArrayFree(node.outputs);
    TfLiteIntArrayFree(node.temporaries);
    for (auto& q : quant) { free(q.scale); TfLiteIntArrayFree(q.zero_point); }
  }

  int add(TfLiteType type, int rank, const int* dims, TfLiteAllocationType alloc) {
    if (tensors.size() >= 64) return -1;
    TfLiteTensor t;
    memset(&t, 0, sizeof t);
    t.type = type;
    t.dims = TfLiteIntArrayCreate(rank);
    size_t n = type_size(type);
    for (int i = 0; i < rank; ++i) { t.dims->data[i] = dims[i]; n *= (size_t)dims[i]; }
    t.allocation_type = alloc;
    t.bytes = n;
    tensors.push_back(t);
    storage.emplace_back(n + 16, 0);
    tensors.back().data.raw = storage.back().data();
    ctx.tensors = tensors.data();
    ctx.tensors_size = tensors.size();
    return (int)tensors.size() - 1;
  }
};

TfLiteIntArray* make_array(const int* v, int n) {
  TfLiteIntArray* a = TfLiteIntArrayCreate(n);
  for (int i = 0; i < n; ++i) a->data[i] = v[i];
  return a;
}

}  // namespace

// A stand-in for a builtin CPU kernel in chain tests: copies its input tensor to its output tensor in the arena, i.e.
// a reader of host memory that knows nothing about the LCE ops' device buffers.
TfLiteStatus HostCopyPrepare(TfLiteContext* c, TfLiteNode* n) {
  const TfLiteTensor* in = &c->tensors[n->inputs->data[0]];
  return c->ResizeTensor(c, &c->tensors[n->outputs->data[0]], TfLiteIntArrayCopy(in->dims));
}
TfLiteStatus HostCopyInvoke(TfLiteContext* c, TfLiteNode* n) {
  const TfLiteTensor* in = &c->tensors[n->inputs->data[0]];
  TfLiteTensor* out = &c->tensors[n->outputs->data[0]];
  memcpy(out->data.raw, in->data.raw, in->bytes < out->bytes ? in->bytes : out->bytes);
  return kTfLiteOk;
}
const TfLiteRegistration* find_registration(const char* op_name, int variant, int use_resolver) {
  using namespace compute_engine::tflite;
  if (!strcmp(op_name, "HostCopy")) {
    static TfLiteRegistration r = {nullptr, nullptr, HostCopyPrepare, HostCopyInvoke};
    return &r;
  }
  if (use_resolver) {
    Resolver r;
    RegisterLCECustomOps(&r, (variant & 1) != 0, (variant & 2) != 0);
    auto it = r.ops.find(op_name);
    return it == r.ops.end() ? nullptr : it->second;
  }
  if (!strcmp(op_name, "LceBconv2d"))
    return variant == 1 ? Register_BCONV_2D_REF() : variant == 2 ? Register_BCONV_2D_OPT_BGEMM()
           : variant == 3 ? Register_BCONV_2D_OPT_INDIRECT_BGEMM() : Register_BCONV_2D();
  if (!strcmp(op_name, "LceQuantize")) return Register_QUANTIZE();
  if (!strcmp(op_name, "LceDequantize")) return Register_DEQUANTIZE();
  if (!strcmp(op_name, "LceBMaxPool2d")) return Register_BMAXPOOL_2D();
  return nullptr;
}

extern "C" {

// ---- chains: an empty model, nodes added in execution order, an execution plan the ops may inspect ----
void* lce_driver_create_chain(void) { return new Model(); }
// What an application does once with its interpreter (lce_ops_register.h, DeclareGraphForDeviceResidency): tell the ops
// who reads which tensor and which tensors are graph outputs.
void lce_driver_declare_graph(void* h, const int* graph_outputs, int n_outputs) {
  Model* m = (Model*)h;
  lce_tflite_ops_declare_graph_begin(&m->ctx);
  if (m->reg) lce_tflite_ops_declare_graph_node(&m->ctx, m->node.inputs->data, m->node.inputs->size, m->node.outputs->data,
                                                m->node.outputs->size, m->reg->invoke);
  for (Model::Extra* e : m->extra)
    lce_tflite_ops_declare_graph_node(&m->ctx, e->node.inputs->data, e->node.inputs->size, e->node.outputs->data,
                                      e->node.outputs->size, e->reg->invoke);
  for (int i = 0; i < n_outputs; ++i) lce_tflite_ops_declare_graph_output(&m->ctx, graph_outputs[i]);
  lce_tflite_ops_declare_graph_end(&m->ctx);
}
void lce_driver_forget_graph(void* h) { lce_tflite_ops_forget_graph(&((Model*)h)->ctx); }
// returns the node's index in the execution plan, or -1
int lce_driver_add_node(void* h, const char* op_name, int variant, int use_resolver, const int* inputs, int n_in,
                        const int* outputs, int n_out, const char* options, size_t options_len) {
  Model* m = (Model*)h;
  const TfLiteRegistration* reg = find_registration(op_name, variant, use_resolver);
  if (!reg) return -1;
  Model::Extra* e = new Model::Extra();
  e->reg = reg;
  e->options.assign(options ? options : "", options ? options_len : 0);
  e->node.inputs = make_array(inputs, n_in);
  e->node.outputs = make_array(outputs, n_out);
  e->node.temporaries = TfLiteIntArrayCreate(0);
  e->node.custom_initial_data = e->options.data();
  e->node.custom_initial_data_size = (int)e->options.size();
  e->node.user_data = reg->init ? reg->init(&m->ctx, e->options.data(), e->options.size()) : nullptr;
  m->extra.push_back(e);
  return (int)m->extra.size() - 1 + (m->reg ? 1 : 0);
}
int lce_driver_prepare_all(void* h) {
  Model* m = (Model*)h;
  if (m->reg) { const int rc = (int)m->reg->prepare(&m->ctx, &m->node); if (rc) return rc; }
  for (Model::Extra* e : m->extra) { const int rc = (int)e->reg->prepare(&m->ctx, &e->node); if (rc) return rc; }
  return 0;
}
int lce_driver_invoke_all(void* h) {
  Model* m = (Model*)h;
  if (m->reg) { const int rc = (int)m->reg->invoke(&m->ctx, &m->node); if (rc) return rc; }
  for (Model::Extra* e : m->extra) { const int rc = (int)e->reg->invoke(&m->ctx, &e->node); if (rc) return rc; }
  return 0;
}

// variant: for "LceBconv2d" 0 = Register_BCONV_2D (default), 1 = _REF, 2 = _OPT_BGEMM,
// 3 = _OPT_INDIRECT_BGEMM; otherwise ignored.  With use_resolver != 0 the registration is
// looked up through RegisterLCECustomOps(resolver, use_reference_bconv, use_indirect_bgemm)
// where the two flags are bit 0 / bit 1 of `variant`.
void* lce_driver_create(const char* op_name, int variant, int use_resolver) {
  using namespace compute_engine::tflite;
  Model* m = new Model();
  if (use_resolver) {
    Resolver r;
    RegisterLCECustomOps(&r, (variant & 1) != 0, (variant & 2) != 0);
    auto it = r.ops.find(op_name);
    m->reg = it == r.ops.end() ? nullptr : it->second;
  } else if (!strcmp(op_name, "LceBconv2d")) {
    m->reg = variant == 1 ? Register_BCONV_2D_REF() : variant == 2 ? Register_BCONV_2D_OPT_BGEMM()
             : variant == 3 ? Register_BCONV_2D_OPT_INDIRECT_BGEMM() : Register_BCONV_2D();
  } else if (!strcmp(op_name, "LceQuantize")) m->reg = Register_QUANTIZE();
  else if (!strcmp(op_name, "LceDequantize")) m->reg = Register_DEQUANTIZE();
  else if (!strcmp(op_name, "LceBMaxPool2d")) m->reg = Register_BMAXPOOL_2D();
  if (!m->reg) { delete m; return nullptr; }
  return m;
}

void lce_driver_destroy(void* h) { delete (Model*)h; }

// type: TfLiteType value; allocation: TfLiteAllocationType value.  Returns the tensor index.
int lce_driver_add_tensor(void* h, int type, int rank, const int* dims, int allocation, float scale,
                          int zero_point, int affine_quantized) {
  Model* m = (Model*)h;
  const int idx = m->add((TfLiteType)type, rank, dims, (TfLiteAllocationType)allocation);
  if (idx < 0) return idx;
  TfLiteTensor& t = m->tensors[idx];
  t.params.scale = scale;
  t.params.zero_point = zero_point;
  if (affine_quantized) {
    TfLiteAffineQuantization q;
    q.scale = (TfLiteFloatArray*)malloc(sizeof(TfLiteFloatArray) + sizeof(float));
    q.scale->size = 1; q.scale->data[0] = scale;
    q.zero_point = TfLiteIntArrayCreate(1); q.zero_point->data[0] = zero_point;
    q.quantized_dimension = 0;
    m->quant.push_back(q);
    t.quantization.type = kTfLiteAffineQuantization;
    t.quantization.params = &m->quant.back();
  }
  return idx;
}

int lce_driver_set_data(void* h, int tensor, const void* data, size_t bytes) {
  Model* m = (Model*)h;
  if (tensor < 0 || (size_t)tensor >= m->tensors.size() || bytes > m->tensors[tensor].bytes) return 1;
  memcpy(m->tensors[tensor].data.raw, data, bytes);
  return 0;
}

// inputs may contain -1 (kTfLiteOptionalTensor).  Calls init() with the flexbuffer options.
int lce_driver_set_node(void* h, const int* inputs, int n_in, const int* outputs, int n_out,
                        const char* options, size_t options_len) {
  Model* m = (Model*)h;
  m->node.inputs = make_array(inputs, n_in);
  m->node.outputs = make_array(outputs, n_out);
  m->node.temporaries = TfLiteIntArrayCreate(0);
  m->node.custom_initial_data = options;
  m->node.custom_initial_data_size = (int)options_len;
  m->node.user_data = m->reg->init ? m->reg->init(&m->ctx, options, options_len) : nullptr;
  m->inited = true;
  return 0;
}

int lce_driver_prepare(void* h) { Model* m = (Model*)h; return (int)m->reg->prepare(&m->ctx, &m->node); }
int lce_driver_invoke(void* h) { Model* m = (Model*)h; return (int)m->reg->invoke(&m->ctx, &m->node); }

int lce_driver_tensor_rank(void* h, int tensor) { return ((Model*)h)->tensors[tensor].dims->size; }
int lce_driver_tensor_dim(void* h, int tensor, int d) { return ((Model*)h)->tensors[tensor].dims->data[d]; }
size_t lce_driver_tensor_bytes(void* h, int tensor) { return ((Model*)h)->tensors[tensor].bytes; }
int lce_driver_get_data(void* h, int tensor, void* dst, size_t bytes) {
  Model* m = (Model*)h;
  if (bytes > m->tensors[tensor].bytes) return 1;
  memcpy(dst, m->tensors[tensor].data.raw, bytes);
  return 0;
}
const char* lce_driver_log(void* h) { return ((Model*)h)->log.c_str(); }
int lce_driver_num_temporaries(void* h) { return ((Model*)h)->node.temporaries ? ((Model*)h)->node.temporaries->size : 0; }

// flexbuffer KAT hook: parse a custom-options buffer and look one key up
int lce_driver_flex_lookup(const char* buf, size_t len, const char* key, int* is_null, int* value) {
  const lce_flex::Map m((const uint8_t*)buf, len);
  if (!m.valid()) return 1;
  *is_null = m.IsNull(key) ? 1 : 0;
  *value = m.AsInt32(key);
  return 0;
}

}  // extern "C"
