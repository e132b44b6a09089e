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
  // All positions are OFFSETS into [0, len_): every value read from the file is checked against what is
  // left of the buffer before it is used, with subtractions that cannot wrap -- no pointer is ever formed
  // outside the buffer (the options blob of a .tflite file is attacker-controlled input).
  const uint8_t* buf_ = nullptr;
  size_t len_ = 0;
  bool valid_ = false;
  size_t values_ = 0, keys_ = 0;   // offsets of the value vector / key vector
  size_t size_ = 0, bw_ = 1, keys_bw_ = 1;

  bool span(size_t pos, size_t n) const { return pos <= len_ && n <= len_ - pos; }
  static bool width_ok(size_t w) { return w == 1 || w == 2 || w == 4 || w == 8; }
  uint64_t read_u(size_t pos, size_t w) const {
    uint64_t v = 0;
    memcpy(&v, buf_ + pos, w);  // little-endian hosts only, like the reference (FLATBUFFERS_LITTLEENDIAN)
    return v;
  }
  int64_t read_i(size_t pos, size_t w) const {
    switch (w) {
      case 1: { int8_t v; memcpy(&v, buf_ + pos, 1); return v; }
      case 2: { int16_t v; memcpy(&v, buf_ + pos, 2); return v; }
      case 4: { int32_t v; memcpy(&v, buf_ + pos, 4); return v; }
      default: { int64_t v; memcpy(&v, buf_ + pos, 8); return v; }
    }
  }
  // `pos` holds a backwards offset of width w: the position it points at, or false
  bool deref(size_t pos, size_t w, size_t* target) const {
    if (!span(pos, w)) return false;
    const uint64_t d = read_u(pos, w);
    if (d > (uint64_t)pos) return false;
    *target = pos - (size_t)d;
    return true;
  }

  void parse(const uint8_t* buf, size_t len) {
    buf_ = buf;
    len_ = len;
    if (!buf || len < 3) return;
    const size_t root_w = buf[len - 1];
    const uint8_t packed = buf[len - 2];
    if ((packed >> 2) != kMap || !width_ok(root_w)) return;
    bw_ = (size_t)1 << (packed & 3);
    if (len < 2 + root_w) return;
    if (!deref(len - 2 - root_w, root_w, &values_)) return;
    // the three words in front of the values: offset to the keys, key width, element count
    if (values_ < 3 * bw_) return;
    const uint64_t count = read_u(values_ - bw_, bw_);
    keys_bw_ = (size_t)read_u(values_ - 2 * bw_, bw_);
    if (!width_ok(keys_bw_) || !deref(values_ - 3 * bw_, bw_, &keys_)) return;
    // count * (bw + 1) bytes of values + type bytes, count * keys_bw of key offsets: divide, never multiply first
    if (count > (uint64_t)(len_ - values_) / (bw_ + 1) || count > (uint64_t)(len_ - keys_) / keys_bw_) return;
    size_ = (size_t)count;
    valid_ = true;
  }

  bool find(const char* key, int64_t* out) const {
    if (!valid_) return false;
    for (size_t i = 0; i < size_; ++i) {
      size_t ks;
      if (!deref(keys_ + i * keys_bw_, keys_bw_, &ks)) continue;
      const size_t maxn = len_ - ks;
      const char* kstr = (const char*)(buf_ + ks);
      if (strnlen(kstr, maxn) == maxn || strcmp(kstr, key) != 0) continue;
      const uint8_t vt = buf_[values_ + size_ * bw_ + i];
      const uint8_t type = vt >> 2;
      const size_t vw = (size_t)1 << (vt & 3);
      const size_t vp = values_ + i * bw_;
      switch (type) {
        case kInt: *out = read_i(vp, bw_); return true;      // inline scalars use the parent width
        case kUInt: case kBool: *out = (int64_t)read_u(vp, bw_); return true;
        case kIndirectInt: case kIndirectUInt: {
          size_t tp;
          if (!deref(vp, bw_, &tp) || !span(tp, vw)) return false;
          *out = type == kIndirectInt ? read_i(tp, vw) : (int64_t)read_u(tp, vw);
          return true;
        }
        case kFloat: {
          if (bw_ == 4) { float f; memcpy(&f, buf_ + vp, 4); *out = (int64_t)f; return true; }
          if (bw_ == 8) { double f; memcpy(&f, buf_ + vp, 8); *out = (int64_t)f; return true; }
          return false;
        }
        default: return false;  // FBT_NULL and everything else read as "null"
      }
    }
    return false;
  }
};

}  // namespace lce_flex
