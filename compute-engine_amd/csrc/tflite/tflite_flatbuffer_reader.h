// Bounds-checked reader for the subset of the TFLite flatbuffer (.tflite) that a converted
// Larq model needs to hand its LCE custom ops to this library: operator codes, the first
// subgraph's tensors / operators / inputs / outputs, quantization scale + zero point, and
// constant buffers.  SURVEY.md section 8(f) row n3.
//
// The reference reads models through TensorFlow Lite itself (tflite::FlatBufferModel +
// InterpreterBuilder, examples/lce_minimal.cc:28-40; tflite/python/interpreter_wrapper_lite.cc:
// 40-58) -- an un-vendored third-party dependency (tensorflow v2.16.1, WORKSPACE:12-24), as is
// the flatbuffers library.  This file restates the published FlatBuffers wire format and the
// field numbering of tensorflow/lite/schema/schema.fbs (file identifier "TFL3"):
//   Model            0 version  1 operator_codes  2 subgraphs  3 description  4 buffers
//   OperatorCode     0 deprecated_builtin_code(int8)  1 custom_code  2 version  3 builtin_code(int32)
//   SubGraph         0 tensors  1 inputs  2 outputs  3 operators  4 name
//   Tensor           0 shape  1 type(int8)  2 buffer(uint32)  3 name  4 quantization
//   QuantizationParameters  0 min  1 max  2 scale[float]  3 zero_point[int64]
//   Operator         0 opcode_index  1 inputs  2 outputs  3/4 builtin_options (union)
//                    5 custom_options[ubyte]  6 custom_options_format
//   Buffer           0 data[ubyte]
// No .tflite file and no flatbuffers library exist in the build image: the only byte-level
// known answers are the reference's flexbuffer option blobs (mlir/tests/legalize-lce.mlir:9,21),
// which custom_options carry unchanged.  Everything else is tested against a writer that
// follows the same specification (tests/tflite_writer.py) -- parity with real converter output
// is UNPINNED until a real model file is available.
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

namespace lce_tfl {

constexpr int32_t kBuiltinCustom = 32;   // BuiltinOperator_CUSTOM
// TensorType values used by LCE graphs
constexpr int kTensorFloat32 = 0, kTensorInt32 = 2, kTensorBool = 6, kTensorInt8 = 9;

struct Tensor {
  std::vector<int32_t> shape;
  int type = 0;
  uint32_t buffer = 0;
  std::string name;
  bool quantized = false;
  float scale = 0.0f;
  int64_t zero_point = 0;
  const uint8_t* data = nullptr;   // constant data (inside the model buffer), or null
  size_t bytes = 0;
};

struct Operator {
  int32_t builtin_code = 0;
  std::string custom_code;         // "LceBconv2d", ... when builtin_code == kBuiltinCustom
  std::vector<int32_t> inputs, outputs;   // tensor indices, -1 = optional input not present
  const uint8_t* custom_options = nullptr;
  size_t custom_options_size = 0;
};

class Model {
 public:
  std::vector<Tensor> tensors;
  std::vector<Operator> operators;
  std::vector<int32_t> inputs, outputs;
  uint32_t version = 0;
  std::string description;

  // `data` must stay alive while the Model is used (constant tensors point into it).
  bool Parse(const void* data, size_t size, std::string* err) {
    b_ = (const uint8_t*)data;
    n_ = size;
    err_.clear();
    tensors.clear(); operators.clear(); inputs.clear(); outputs.clear();
    const bool ok = ParseImpl();
    if (!ok && err) *err = err_.empty() ? "malformed flatbuffer" : err_;
    return ok;
  }

 private:
  const uint8_t* b_ = nullptr;
  size_t n_ = 0;
  std::string err_;

  bool Fail(const char* what) { if (err_.empty()) err_ = what; return false; }
  bool In(size_t pos, size_t len) const { return pos <= n_ && len <= n_ - pos; }
  template <typename T> bool Rd(size_t pos, T* out) const {
    if (!In(pos, sizeof(T))) return false;
    memcpy(out, b_ + pos, sizeof(T));   // little-endian hosts only, like the reference
    return true;
  }
  // Position of field `id` inside table `t`, 0 when absent; false on a malformed table.
  bool Field(size_t t, int id, size_t* pos) const {
    *pos = 0;
    int32_t so;
    if (!Rd(t, &so)) return false;
    const int64_t vt = (int64_t)t - so;
    if (vt < 0 || !In((size_t)vt, 4)) return false;
    uint16_t vsize, fo;
    if (!Rd((size_t)vt, &vsize) || vsize < 4 || !In((size_t)vt, vsize)) return false;
    const size_t slot = 4 + 2 * (size_t)id;
    if (slot + 2 > vsize) return true;               // field not in this (older) vtable
    if (!Rd((size_t)vt + slot, &fo)) return false;
    if (fo == 0) return true;
    if (!In(t + fo, 1)) return false;
    *pos = t + fo;
    return true;
  }
  // Follows the uoffset stored at `pos`.
  bool Indirect(size_t pos, size_t* target) const {
    uint32_t off;
    if (!Rd(pos, &off)) return false;
    *target = pos + off;
    return In(*target, 4);
  }
  // Vector header at `v`: element count and position of element 0.
  bool Vec(size_t v, size_t elem, uint32_t* count, size_t* first) const {
    if (!Rd(v, count)) return false;
    *first = v + 4;
    return In(*first, (size_t)*count * elem);
  }
  template <typename T> bool Scalar(size_t t, int id, T dflt, T* out) const {
    size_t p;
    if (!Field(t, id, &p)) return false;
    *out = dflt;
    return p == 0 || Rd(p, out);
  }
  bool String(size_t t, int id, std::string* out) const {
    size_t p, s, first;
    uint32_t len;
    out->clear();
    if (!Field(t, id, &p)) return false;
    if (p == 0) return true;
    if (!Indirect(p, &s) || !Vec(s, 1, &len, &first)) return false;
    out->assign((const char*)b_ + first, len);
    return true;
  }
  // Vector field of scalars -> (count, first); count 0 when absent.
  bool VecField(size_t t, int id, size_t elem, uint32_t* count, size_t* first) const {
    size_t p, v;
    *count = 0; *first = 0;
    if (!Field(t, id, &p)) return false;
    if (p == 0) return true;
    return Indirect(p, &v) && Vec(v, elem, count, first);
  }
  bool IntVec(size_t t, int id, std::vector<int32_t>* out) const {
    uint32_t c; size_t f;
    if (!VecField(t, id, 4, &c, &f)) return false;
    out->resize(c);
    if (c) memcpy(out->data(), b_ + f, (size_t)c * 4);
    return true;
  }
  // Element i of a vector of tables.
  bool TableAt(size_t first, uint32_t i, size_t* t) const { return Indirect(first + 4 * (size_t)i, t); }

  bool ParseImpl() {
    if (n_ < 8) return Fail("buffer too small for a flatbuffer");
    if (memcmp(b_ + 4, "TFL3", 4) != 0) return Fail("file identifier is not TFL3");
    size_t root;
    if (!Indirect(0, &root)) return Fail("bad root offset");
    if (!Scalar<uint32_t>(root, 0, 0u, &version)) return Fail("bad Model table");
    if (!String(root, 3, &description)) return Fail("bad Model.description");

    // buffers
    uint32_t nbuf; size_t bfirst;
    if (!VecField(root, 4, 4, &nbuf, &bfirst)) return Fail("bad Model.buffers");
    std::vector<std::pair<const uint8_t*, size_t>> buffers(nbuf);
    for (uint32_t i = 0; i < nbuf; ++i) {
      size_t t; uint32_t c; size_t f;
      if (!TableAt(bfirst, i, &t) || !VecField(t, 0, 1, &c, &f)) return Fail("bad Buffer");
      buffers[i] = {c ? b_ + f : nullptr, c};
    }
    // operator codes
    uint32_t ncode; size_t cfirst;
    if (!VecField(root, 1, 4, &ncode, &cfirst)) return Fail("bad Model.operator_codes");
    struct Code { int32_t builtin; std::string custom; };
    std::vector<Code> codes(ncode);
    for (uint32_t i = 0; i < ncode; ++i) {
      size_t t; int8_t dep; int32_t bc;
      if (!TableAt(cfirst, i, &t) || !Scalar<int8_t>(t, 0, 0, &dep) || !Scalar<int32_t>(t, 3, 0, &bc) ||
          !String(t, 1, &codes[i].custom))
        return Fail("bad OperatorCode");
      // schema.fbs: builtin_code supersedes deprecated_builtin_code once it exceeds 127
      codes[i].builtin = bc != 0 ? bc : (int32_t)dep;
    }
    // first subgraph
    uint32_t nsub; size_t sfirst, sg;
    if (!VecField(root, 2, 4, &nsub, &sfirst) || nsub == 0) return Fail("model has no subgraph");
    if (!TableAt(sfirst, 0, &sg)) return Fail("bad SubGraph");
    if (!IntVec(sg, 1, &inputs) || !IntVec(sg, 2, &outputs)) return Fail("bad SubGraph.inputs/outputs");
    uint32_t nt; size_t tfirst;
    if (!VecField(sg, 0, 4, &nt, &tfirst)) return Fail("bad SubGraph.tensors");
    tensors.resize(nt);
    for (uint32_t i = 0; i < nt; ++i) {
      size_t t, qp, q;
      Tensor& T = tensors[i];
      int8_t ty;
      if (!TableAt(tfirst, i, &t) || !IntVec(t, 0, &T.shape) || !Scalar<int8_t>(t, 1, 0, &ty) ||
          !Scalar<uint32_t>(t, 2, 0u, &T.buffer) || !String(t, 3, &T.name) || !Field(t, 4, &qp))
        return Fail("bad Tensor");
      T.type = ty;
      if (T.buffer >= nbuf && nbuf != 0) return Fail("Tensor.buffer out of range");
      if (T.buffer != 0 && T.buffer < nbuf) { T.data = buffers[T.buffer].first; T.bytes = buffers[T.buffer].second; }
      if (qp != 0) {
        uint32_t c; size_t f;
        if (!Indirect(qp, &q)) return Fail("bad Tensor.quantization");
        if (!VecField(q, 2, 4, &c, &f)) return Fail("bad QuantizationParameters.scale");
        if (c >= 1) { memcpy(&T.scale, b_ + f, 4); T.quantized = true; }
        if (!VecField(q, 3, 8, &c, &f)) return Fail("bad QuantizationParameters.zero_point");
        if (c >= 1) memcpy(&T.zero_point, b_ + f, 8);
      }
    }
    uint32_t no; size_t ofirst;
    if (!VecField(sg, 3, 4, &no, &ofirst)) return Fail("bad SubGraph.operators");
    operators.resize(no);
    for (uint32_t i = 0; i < no; ++i) {
      size_t t; uint32_t idx, c; size_t f;
      Operator& O = operators[i];
      if (!TableAt(ofirst, i, &t) || !Scalar<uint32_t>(t, 0, 0u, &idx) || !IntVec(t, 1, &O.inputs) ||
          !IntVec(t, 2, &O.outputs) || !VecField(t, 5, 1, &c, &f))
        return Fail("bad Operator");
      if (idx >= ncode) return Fail("Operator.opcode_index out of range");
      O.builtin_code = codes[idx].builtin;
      O.custom_code = codes[idx].custom;
      O.custom_options = c ? b_ + f : nullptr;
      O.custom_options_size = c;
      for (int32_t x : O.inputs) if (x < -1 || x >= (int32_t)nt) return Fail("Operator input index out of range");
      for (int32_t x : O.outputs) if (x < 0 || x >= (int32_t)nt) return Fail("Operator output index out of range");
    }
    for (int32_t x : inputs) if (x < 0 || x >= (int32_t)nt) return Fail("SubGraph input index out of range");
    for (int32_t x : outputs) if (x < 0 || x >= (int32_t)nt) return Fail("SubGraph output index out of range");
    return true;
  }
};

}  // namespace lce_tfl
