// Drop-in for the reference's tflite/kernels/lce_ops_register.h: forward declarations of
// the LCE custom-op registrations plus RegisterLCECustomOps().  Same names, same
// arguments, same op strings ("LceQuantize", "LceDequantize", "LceBconv2d",
// "LceBMaxPool2d"); the registrations run on the MI355X through include/lce_hip.h.
#ifndef COMPUTE_ENGINE_AMD_TFLITE_LCE_OPS_REGISTER_H_
#define COMPUTE_ENGINE_AMD_TFLITE_LCE_OPS_REGISTER_H_

#include <cstdio>

#include "tflite_abi.h"

namespace compute_engine {
namespace tflite {

TfLiteRegistration* Register_QUANTIZE();
TfLiteRegistration* Register_DEQUANTIZE();
TfLiteRegistration* Register_BCONV_2D();
TfLiteRegistration* Register_BCONV_2D_REF();
TfLiteRegistration* Register_BCONV_2D_OPT_BGEMM();           // declared by the reference's tests (bconv2d_test.cc:191)
TfLiteRegistration* Register_BCONV_2D_OPT_INDIRECT_BGEMM();
TfLiteRegistration* Register_BMAXPOOL_2D();

// lce_ops_register.h:25-53.  Inside a TFLite tree (-DLCE_USE_SYSTEM_TFLITE) this is the reference's exact
// signature -- a plain function taking ::tflite::MutableOpResolver*, so explicit uses of its address or type
// keep compiling; standalone (no TensorFlow headers in this image) the resolver type is a template
// parameter: any type with AddCustom(const char*, const TfLiteRegistration*) works.
#ifdef LCE_USE_SYSTEM_TFLITE
}  // namespace tflite
}  // namespace compute_engine
#include "tensorflow/lite/mutable_op_resolver.h"
namespace compute_engine {
namespace tflite {
inline void RegisterLCECustomOps(::tflite::MutableOpResolver* resolver, const bool use_reference_bconv = false,
                                 const bool use_indirect_bgemm = false) {
#else
template <typename Resolver>
inline void RegisterLCECustomOps(Resolver* resolver, const bool use_reference_bconv = false,
                                 const bool use_indirect_bgemm = false) {
#endif
  if (use_reference_bconv && use_indirect_bgemm) {
    std::fprintf(stderr,
                 "WARNING: 'use_reference_bconv' and `use_indirect_bgemm` are both set to true. "
                 "use_indirect_bgemm==true will have no effect.\n");
  }
  resolver->AddCustom("LceQuantize", Register_QUANTIZE());
  resolver->AddCustom("LceDequantize", Register_DEQUANTIZE());
  if (use_reference_bconv) {
    resolver->AddCustom("LceBconv2d", Register_BCONV_2D_REF());
  } else if (use_indirect_bgemm) {
    resolver->AddCustom("LceBconv2d", Register_BCONV_2D_OPT_INDIRECT_BGEMM());
  } else {
    resolver->AddCustom("LceBconv2d", Register_BCONV_2D());
  }
  resolver->AddCustom("LceBMaxPool2d", Register_BMAXPOOL_2D());
}

}  // namespace tflite
}  // namespace compute_engine

// ---- device residency between LCE ops (an extension; the reference has no counterpart) ----
// By default every op leaves its output in the interpreter's arena, exactly as the reference's kernels do
// (tflite/kernels/bconv2d.cc:550-564).  A host that DECLARES its graph lets tensors that only LCE ops read -- and that
// are not graph outputs -- stay in HBM between them: a binary section then crosses PCIe once in each direction.  The
// ops cannot find this out themselves (TfLiteContext::GetExecutionPlan / GetNodeAndRegistration are delegate-only).
extern "C" {
void lce_tflite_ops_declare_graph_begin(const TfLiteContext* context);
void lce_tflite_ops_declare_graph_node(const TfLiteContext* context, const int* inputs, int n_inputs, const int* outputs,
                                       int n_outputs, TfLiteStatus (*invoke)(TfLiteContext*, TfLiteNode*));
void lce_tflite_ops_declare_graph_output(const TfLiteContext* context, int tensor);
void lce_tflite_ops_declare_graph_end(const TfLiteContext* context);
void lce_tflite_ops_forget_graph(const TfLiteContext* context);
void lce_tflite_ops_set_residency(int on);
}

#ifdef LCE_USE_SYSTEM_TFLITE
#include "tensorflow/lite/interpreter.h"
namespace compute_engine {
namespace tflite {
// Call once after building the interpreter (before or after AllocateTensors; again after ModifyGraphWithDelegate or any
// other change of the execution plan).  Walks the primary subgraph from APPLICATION code, where that is legal.
inline void DeclareGraphForDeviceResidency(::tflite::Interpreter* interpreter) {
  const TfLiteContext* context = interpreter->primary_subgraph().context();
  lce_tflite_ops_declare_graph_begin(context);
  for (int node_index : interpreter->execution_plan()) {
    const auto* nr = interpreter->node_and_registration(node_index);
    if (!nr) continue;
    const TfLiteNode& node = nr->first;
    lce_tflite_ops_declare_graph_node(context, node.inputs->data, node.inputs->size, node.outputs->data,
                                      node.outputs->size, nr->second.invoke);
  }
  for (int t : interpreter->outputs()) lce_tflite_ops_declare_graph_output(context, t);
  lce_tflite_ops_declare_graph_end(context);
}
}  // namespace tflite
}  // namespace compute_engine
#endif

#endif  // COMPUTE_ENGINE_AMD_TFLITE_LCE_OPS_REGISTER_H_
