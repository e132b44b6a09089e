// Converter-side parameter preparation -- see lce_prepare.cpp.  Host-only.
#pragma once
#include <stdint.h>

#include <string>

#include "../../include/lce_hip.h"

namespace lce {
std::string prepare_binary_filter(const float* hwio, int kh, int kw, int cin, int cout, float* ohwi,
                                  float* mul, float* bias);
std::string fuse_post_op(int op, const float* value, int value_count, float* mul, float* bias, int n);
bool can_fuse_activation(const float* mul, const float* bias, int n, int padding_same, int pad_values);
std::string prepare_bitpacked_output(float* ohwi, int kh, int kw, int cin, int cout, int activation,
                                     int padding_same, int pad_values, const float* mul,
                                     const float* bias, int32_t* thresholds);
std::string bitpack_filter(const float* ohwi, int kh, int kw, int cin, int cout, int32_t* words);
}  // namespace lce
