// Just enough of a FlexBuffers reader for the LCE custom-op options: a root MAP whose
// values are small integers (written by mlir/ir/lce_ops.cc:36-51 of the reference with
// flexbuffers::Builder::Map / Int; read at tflite/kernels/bconv2d.cc:90-124 with
// GetRoot(...).AsMap(), m["key"].IsNull(), .AsInt32()).  flatbuffers is an un-vendored
// dependency of the reference; the wire format below follows the published FlexBuffers
// specification.  Known-answer test: tests/golden/reference_kats.json
// (mlir/tests/legalize-lce.mlir:9,21).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace lce_flex {

enum Type : uint8_t { kNull = 0, kInt = 1, kUInt = 2, kFloat = 3, kKey = 4, kString = 5,
                      kIndirectInt = 6, kIndirectUInt = 7, kIndirectFloat = 8, kMap = 9, kBool = 26 };

class Map {
 public:
  Map(const uint8_t* buf, size_t len) { parse(buf, len); }
  bool valid() const { return valid_; }
  size_t size() const { return size_; }

  // Mirrors `m[key].IsNull()`: true when the key is absent (or the value is FBT_NULL).
  bool IsNull(const char* key) const {
    int64_t v;
    return !find(key, &v);
  }
  // Mirrors `m[key].AsInt32()`: 0 when absent, like flexbuffers' null Reference.
  int32_t AsInt32(const char* key) const {
    int64_t v = 0;
    find(key, &v);
    return (int32_t)v;
  }

 private:
  const uint8_t* buf_ = nullptr;
  size_t len_ = 0;
  bool valid_ = false;
  const uint8_t* values_ = nullptr;
  const uint8_t* keys_ = nullptr;
  size_t size_ = 0, bw_ = 1, keys_bw_ = 1;

  bool in(const uint8_t* p, size_t n) const { return p >= buf_ && p + n <= buf_ + len_; }
  static uint64_t read_u(const uint8_t* p, size_t w) {
    uint64_t v = 0;
    memcpy(&v, p, w);  // little-endian hosts only, like the reference (FLATBUFFERS_LITTLEENDIAN)
    return v;
  }
  static int64_t read_i(const uint8_t* p, size_t w) {
    switch (w) {
      case 1: { int8_t v; memcpy(&v, p, 1); return v; }
      case 2: { int16_t v; memcpy(&v, p, 2); return v; }
      case 4: { int32_t v; memcpy(&v, p, 4); return v; }
      default: { int64_t v; memcpy(&v, p, 8); return v; }
    }
  }

  void parse(const uint8_t* buf, size_t len) {
    buf_ = buf;
    len_ = len;
    if (!buf || len < 3) return;
    const size_t root_w = buf[len - 1];
    const uint8_t packed = buf[len - 2];
    if ((packed >> 2) != kMap || (root_w != 1 && root_w != 2 && root_w != 4 && root_w != 8)) return;
    bw_ = (size_t)1 << (packed & 3);
    if (len < 2 + root_w) return;
    const uint8_t* root = buf + len - 2 - root_w;
    const uint64_t off = read_u(root, root_w);
    if (off > (uint64_t)(root - buf)) return;
    values_ = root - off;
    if (!in(values_ - 3 * bw_, 3 * bw_)) return;
    size_ = (size_t)read_u(values_ - bw_, bw_);
    keys_bw_ = (size_t)read_u(values_ - 2 * bw_, bw_);
    const uint8_t* koff = values_ - 3 * bw_;
    const uint64_t kd = read_u(koff, bw_);
    if (kd > (uint64_t)(koff - buf)) return;
    keys_ = koff - kd;
    if (keys_bw_ != 1 && keys_bw_ != 2 && keys_bw_ != 4 && keys_bw_ != 8) return;
    if (!in(values_, size_ * bw_ + size_) || !in(keys_, size_ * keys_bw_)) return;
    valid_ = true;
  }

  bool find(const char* key, int64_t* out) const {
    if (!valid_) return false;
    for (size_t i = 0; i < size_; ++i) {
      const uint8_t* kp = keys_ + i * keys_bw_;
      const uint64_t d = read_u(kp, keys_bw_);
      if (d > (uint64_t)(kp - buf_)) continue;
      const char* ks = (const char*)(kp - d);
      const size_t maxn = (size_t)(buf_ + len_ - (const uint8_t*)ks);
      if (strnlen(ks, maxn) == maxn || strcmp(ks, key) != 0) continue;
      const uint8_t vt = values_[size_ * bw_ + i];
      const uint8_t type = vt >> 2;
      const size_t vw = (size_t)1 << (vt & 3);
      const uint8_t* vp = values_ + i * bw_;
      switch (type) {
        case kInt: *out = read_i(vp, bw_); return true;      // inline scalars use the parent width
        case kUInt: case kBool: *out = (int64_t)read_u(vp, bw_); return true;
        case kIndirectInt: case kIndirectUInt: {
          const uint64_t dd = read_u(vp, bw_);
          if (dd > (uint64_t)(vp - buf_)) return false;
          *out = type == kIndirectInt ? read_i(vp - dd, vw) : (int64_t)read_u(vp - dd, vw);
          return true;
        }
        case kFloat: {
          if (bw_ == 4) { float f; memcpy(&f, vp, 4); *out = (int64_t)f; return true; }
          if (bw_ == 8) { double f; memcpy(&f, vp, 8); *out = (int64_t)f; return true; }
          return false;
        }
        default: return false;  // FBT_NULL and everything else read as "null"
      }
    }
    return false;
  }
};

}  // namespace lce_flex
