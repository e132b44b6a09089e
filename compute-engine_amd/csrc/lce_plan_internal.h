// Shared by the planner's three translation units (lce_plan.cpp: validation, folding, weight images, kernel selection;
// lce_plan_stream.cpp: the two streaming kernels' launch planners; lce_plan_cost.cpp: the cost estimates).  Not part of lce_plan.h: nothing
// outside the planner needs these.
#pragma once
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "lce_plan.h"

namespace lce {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// bytes of one output pixel's channels in the layer's output type
inline uint32_t stream_row_bytes(const HostPlan& p) {
  return p.d.dst_type == LCE_HIP_BITPACKED ? (uint32_t)p.wout * 4u : (uint32_t)p.d.channels_out * (p.d.dst_type == LCE_HIP_I8 ? 1u : 4u);
}
inline int64_t out_bytes_of(const HostPlan& p, int batch_chunk) {
  return (int64_t)batch_chunk * p.out_h * p.out_w * stream_row_bytes(p);
}

// lce_plan.cpp
int group_chunks(const lce_hip_bconv2d_desc& d);
MfmaCfg choose_mfma_cfg(const HostPlan& p, int64_t pixels);

// lce_plan_cost.cpp: microseconds of one launch of `batch_chunk` images (`pixels` output pixels) on the candidate as it has just been
// planned (st_* / ws_* fields of `p`); the constants and where each was measured: the table in that file (dump_cost_table)
double estimate_stream_us(const HostPlan& p, int batch_chunk);
double estimate_wstream_us(const HostPlan& p, int batch_chunk);
double estimate_block_gemm_us(const HostPlan& p, int64_t pixels);
void dump_cost_table(FILE* f);
#ifdef LCE_COST_TUNABLE            // the re-fit build of tests/hostsim (tools/fit_cost.py): the constants by index, live
int cost_constant_count();
const char* cost_constant_name(int i);
double get_cost_constant(int i);
bool set_cost_constant(int i, double v);
#endif          // every constant of the estimates: name, value, unit, where it was measured (LCE_PLAN_DEBUG=3)

}  // namespace lce
