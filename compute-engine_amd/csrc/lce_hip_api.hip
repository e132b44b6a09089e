// C ABI (include/lce_hip.h) over the gfx950 kernels in lce_kernels.h.
// There is no CPU fallback in this file: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lce_hip.h"
#include "lce_kernels.h"          // the LceQuantize / LceDequantize / LceBMaxPool2d kernels are launched from here
#include "lce_kernel_types.h"    // the convolution kernels live in their own translation units (lce_tu_*.hip)
#ifdef LCE_UNITY
// single-translation-unit build (tools/build_exp.sh): the time-stamp tools read __device__ arrays that must exist once
#include "lce_tu_valu.hip"
#include "lce_tu_mfma_ws.hip"
#include "lce_tu_mfma_direct.hip"
#include "lce_tu_mfma_2d.hip"
#include "lce_tu_pointwise.hip"
#include "lce_tu_stream_f32.hip"
#include "lce_tu_stream_f32_clamp.hip"
#include "lce_tu_stream_i8.hip"
#include "lce_tu_stream_i8_floor.hip"
#include "lce_tu_stream_bitpacked.hip"
#include "lce_tu_wstream_f32.hip"
#include "lce_tu_wstream_i8.hip"
#include "lce_tu_wstream_i8_floor.hip"
#include "lce_tu_wstream_bitpacked.hip"
#endif
#include "lce_plan.h"
#include "lce_prepare.h"

namespace {

thread_local std::string g_last_error;

lce_hip_status fail(lce_hip_status code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define LCE_HIP_TRY(expr)                                                                  \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? LCE_HIP_ERR_NO_DEVICE \
                                                                         : LCE_HIP_ERR_RUNTIME, \
                  "%s failed: %s", #expr, hipGetErrorString(e_));                          \
  } while (0)

// Asked once per process (the answer cannot change while it runs): the run / bitpack entry points are
// called per layer on 10-30 us kernels and must not pay a runtime query each time.
lce_hip_status require_device() {
  static const int count = [] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
      (void)hipGetLastError();
      n = 0;
    }
    return n;
  }();
  if (count <= 0)
    return fail(LCE_HIP_ERR_NO_DEVICE,
                "no usable HIP device (hipGetDeviceCount found %d); this library has no CPU fallback", count);
  return LCE_HIP_OK;
}

unsigned grid_for_stream(uint64_t wave_tasks, int waves_per_block) {
  // memory-bound streams: cap at ~8 blocks per CU and grid-stride the rest
  const uint64_t blocks = (wave_tasks + waves_per_block - 1) / waves_per_block;
  const uint64_t cap = 256ull * 8ull;
  return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

template <typename T>
struct DevBuf {
  T* ptr = nullptr;
  size_t count = 0;
  ~DevBuf() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
  }
  hipError_t upload(const std::vector<T>& v) {
    release();
    if (v.empty()) return hipSuccess;
    // +64 B of slack: the kernels' padded tables are read with wide scalar loads
    hipError_t e = hipMalloc((void**)&ptr, v.size() * sizeof(T) + 64);
    if (e != hipSuccess) return e;
    count = v.size();
    return hipMemcpy(ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  }
};

}  // namespace

struct lce_hip_bconv2d_plan {
  lce::HostPlan host;
  bool device_current = false;
  int64_t selected_for_pixels = -1;
  DevBuf<uint32_t> d_packed, d_filter;
  DevBuf<float> d_mul, d_bias, d_zpc;
  DevBuf<int32_t> d_thr, d_oobc;
  DevBuf<uint8_t> d_wq;
  DevBuf<float> d_thrq;
  DevBuf<uint32_t> d_sched;        // streaming kernel: its production schedule
  bool cus_forced = false;         // num_cus was set through the "compute_units" option
  int device = -1;                 // the HIP device the plan's buffers live on (bound at the first upload)
  void* workspace = nullptr;       // FP4 expanded activations (matrix-core engine, workspace variant)
  void* lds_opt_in = nullptr;      // kernel already granted > 64 KiB of dynamic LDS
  size_t workspace_bytes = 0;
  // The workspace is ONE buffer per plan: a run on another stream must not start expanding into it while
  // the previous run's GEMM still reads it.  ws_done is recorded behind every workspace GEMM; a run on a
  // different stream waits for it first.
  hipEvent_t ws_done = nullptr;
  hipStream_t ws_stream = nullptr;
  bool ws_used = false;
  // run_host: device staging + a three-stream pipeline (H2D | compute | D2H) over batch slices
  void* stage_in = nullptr;
  void* stage_out = nullptr;
  size_t stage_in_bytes = 0, stage_out_bytes = 0;
  hipStream_t s_h2d = nullptr, s_run = nullptr, s_d2h = nullptr;
  std::vector<hipEvent_t> ev_in, ev_run;
  ~lce_hip_bconv2d_plan() {
    for (hipEvent_t e : ev_in) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_run) (void)hipEventDestroy(e);
    if (s_h2d) (void)hipStreamDestroy(s_h2d);
    if (s_run) (void)hipStreamDestroy(s_run);
    if (s_d2h) (void)hipStreamDestroy(s_d2h);
    if (ws_done) (void)hipEventDestroy(ws_done);
    if (stage_in) (void)hipFree(stage_in);
    if (stage_out) (void)hipFree(stage_out);
    if (workspace) (void)hipFree(workspace);
  }
};

namespace {

using lce::ConvArgs;
using lce::tiled_fn;
using lce::general_fn;
using lce::mfma_fn;

size_t out_elem_bytes(int dst) { return dst == LCE_HIP_I8 ? 1 : 4; }

// The unscaled FP4 MFMA's known-answer test (lce_mfma_selftest.h), once per device and kernel family (0: pointwise, 1: stream, 2: wstream)
lce_hip_status mfma_selftest_once(int dev, int family) {
  static std::mutex mu;
  static std::vector<int> state[3];       // -1 unknown, 0 passed, > 0 failed
  std::lock_guard<std::mutex> lock(mu);
  std::vector<int>& st = state[family];
  if (dev < 0) return LCE_HIP_OK;
  if ((size_t)dev >= st.size()) st.resize((size_t)dev + 1, -1);
  if (st[dev] < 0) {
    const int r = family == 2 ? lce::mfma_selftest_wstream() : family ? lce::mfma_selftest_stream() : lce::mfma_selftest_pointwise();
    if (r < 0) return fail(LCE_HIP_ERR_RUNTIME, "the FP4 matrix-core self-test could not run: %s", hipGetErrorString((hipError_t)(-r)));
    st[dev] = r;
  }
  if (st[dev] != 0)
    return fail(LCE_HIP_ERR_RUNTIME, "this build's unscaled FP4 MFMA (v_mfma_f32_32x32x64_f8f6f4 with FP4 operands at scale 1) does not "
                "compute 3 - 64 = -61 for C = 3, A = +1, B = -1 on device %d: the compiler did not select the unscaled encoding "
                "(lce_device_intrinsics.h, mfma_fp4_32x32x64_unscaled); refusing to run the %s kernels", dev, family == 2 ? "weight-streaming" : family ? "streaming" : "pointwise");
  return LCE_HIP_OK;
}

// compute units of HIP device `dev` (0 when unknown), cached per device
int device_compute_units(int dev) {
  static std::mutex mu;
  static std::vector<int> table;
  if (dev < 0) return 0;
  std::lock_guard<std::mutex> lock(mu);
  if ((size_t)dev >= table.size()) table.resize((size_t)dev + 1, -1);
  if (table[dev] < 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
      (void)hipGetLastError();
      n = 0;
    }
    table[dev] = n;
  }
  return table[dev];
}

lce_hip_status ensure_selected(lce_hip_bconv2d_plan* plan, int batch_chunk) {
  lce::HostPlan& h = plan->host;
  const int64_t pixels = (int64_t)batch_chunk * h.out_h * h.out_w;
  if (plan->selected_for_pixels == pixels && !h.kernel_name.empty() &&
      (!h.use_tiled || !h.packed.empty() || !h.have_weights) &&
      (!h.use_mfma || !h.wq.empty() || !h.have_weights))
    return LCE_HIP_OK;
  if (!plan->cus_forced) {
    // the streaming kernel sizes its grid by the compute units of the device the plan runs on: the one it is bound to, or
    // the current one before its first run (asked once per device and process)
    int dev = plan->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
    const int cus = device_compute_units(dev);
    if (cus > 0) h.num_cus = cus;
  }
  const std::string err = lce::select_kernel(h, pixels);
  if (!err.empty()) return fail(LCE_HIP_ERR_UNSUPPORTED, "%s", err.c_str());
  plan->selected_for_pixels = pixels;
  plan->device_current = false;
  return LCE_HIP_OK;
}

// A plan's weights, tables, workspace and staging live on ONE device: the one that is current when it first
// runs.  Running it with another device current would launch on that device with foreign pointers.
lce_hip_status check_device(lce_hip_bconv2d_plan* plan) {
  int cur = -1;
  LCE_HIP_TRY(hipGetDevice(&cur));
  if (plan->device < 0) plan->device = cur;
  if (plan->device != cur)
    return fail(LCE_HIP_ERR_INVALID,
                "bconv2d: this plan is bound to HIP device %d but device %d is current (a plan's buffers live on "
                "the device it first ran on; make that device current or create one plan per device)",
                plan->device, cur);
  return LCE_HIP_OK;
}

lce_hip_status ensure_uploaded(lce_hip_bconv2d_plan* plan) {
  if (plan->device_current) return LCE_HIP_OK;
  lce::HostPlan& h = plan->host;
  if (h.use_mfma) {
    LCE_HIP_TRY(plan->d_wq.upload(h.wq));
    LCE_HIP_TRY(plan->d_mul.upload(h.mul_q));
    LCE_HIP_TRY(plan->d_bias.upload(h.bias_q));
    LCE_HIP_TRY(plan->d_thrq.upload(h.thr_q));
    if (h.use_stream || h.use_wstream) LCE_HIP_TRY(plan->d_sched.upload(h.st_tabs));
    else plan->d_sched.release();
    plan->d_packed.release();
    plan->d_filter.release();
    plan->d_oobc.release();
  } else if (h.use_tiled) {
    LCE_HIP_TRY(plan->d_packed.upload(h.packed));
    LCE_HIP_TRY(plan->d_mul.upload(h.mul_p));
    LCE_HIP_TRY(plan->d_bias.upload(h.bias_p));
    LCE_HIP_TRY(plan->d_thr.upload(h.thr_p));
    LCE_HIP_TRY(plan->d_oobc.upload(h.oob_corr));
    plan->d_filter.release();
  } else {
    LCE_HIP_TRY(plan->d_filter.upload(h.filter));
    LCE_HIP_TRY(plan->d_mul.upload(h.mul));
    LCE_HIP_TRY(plan->d_bias.upload(h.bias));
    LCE_HIP_TRY(plan->d_thr.upload(h.thresholds));
    plan->d_packed.release();
    plan->d_oobc.release();
  }
  LCE_HIP_TRY(plan->d_zpc.upload(h.zero_pad_cache));
  plan->device_current = true;
  return LCE_HIP_OK;
}

}  // namespace

extern "C" {

int lce_hip_abi_version(void) { return LCE_HIP_ABI_VERSION; }
const char* lce_hip_last_error(void) { return g_last_error.c_str(); }
const char* lce_hip_build_flavor(void) { return LCE_BUILD_FLAVOR; }

int lce_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

lce_hip_status lce_hip_set_device(int device) {
  if (lce_hip_status s = require_device()) return s;
  LCE_HIP_TRY(hipSetDevice(device));
  return LCE_HIP_OK;
}

lce_hip_status lce_hip_malloc(void** dev_ptr, size_t bytes) {
  if (!dev_ptr) return fail(LCE_HIP_ERR_INVALID, "lce_hip_malloc: null out pointer");
  if (lce_hip_status s = require_device()) return s;
  LCE_HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 1));
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_free(void* dev_ptr) {
  if (dev_ptr) LCE_HIP_TRY(hipFree(dev_ptr));
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  LCE_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  LCE_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_memset(void* dst, int value, size_t bytes, void* stream) {
  LCE_HIP_TRY(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_host_register(void* host_ptr, size_t bytes) {
  if (!host_ptr || !bytes) return fail(LCE_HIP_ERR_INVALID, "lce_hip_host_register: null or empty range");
  if (lce_hip_status s = require_device()) return s;
  LCE_HIP_TRY(hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_host_unregister(void* host_ptr) {
  if (host_ptr) LCE_HIP_TRY(hipHostUnregister(host_ptr));
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_stream_create(void** stream) {
  if (!stream) return fail(LCE_HIP_ERR_INVALID, "lce_hip_stream_create: null out pointer");
  if (lce_hip_status s = require_device()) return s;
  hipStream_t st;
  LCE_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *stream = (void*)st;
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_stream_destroy(void* stream) {
  if (stream) LCE_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_stream_synchronize(void* stream) {
  LCE_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return LCE_HIP_OK;
}

// ---- HIP graphs behind the C ABI (include/lce_hip.h) ----
lce_hip_status lce_hip_graph_begin_capture(void* stream) {
  if (!stream) return fail(LCE_HIP_ERR_INVALID, "lce_hip_graph_begin_capture: the null stream cannot be captured (lce_hip_stream_create)");
  // thread-local mode: other threads of the host may go on calling the runtime while this one records
  LCE_HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_graph_end_capture(void* stream, void** graph) {
  if (!stream || !graph) return fail(LCE_HIP_ERR_INVALID, "lce_hip_graph_end_capture: null argument");
  *graph = nullptr;
  hipGraph_t g = nullptr;
  LCE_HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &g));
  if (!g) return fail(LCE_HIP_ERR_RUNTIME, "lce_hip_graph_end_capture: the capture was invalidated");
  hipGraphExec_t exec = nullptr;
  const hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return fail(LCE_HIP_ERR_RUNTIME, "hipGraphInstantiate: %s", hipGetErrorString(e));
  *graph = (void*)exec;
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_graph_launch(void* graph, void* stream) {
  if (!graph) return fail(LCE_HIP_ERR_INVALID, "lce_hip_graph_launch: null graph");
  LCE_HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph, (hipStream_t)stream));
  return LCE_HIP_OK;
}
lce_hip_status lce_hip_graph_destroy(void* graph) {
  if (graph) LCE_HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph));
  return LCE_HIP_OK;
}

int32_t lce_hip_bitpacked_size(int32_t n) { return (n + 31) / 32; }

// ------------------------------------------------------------------------------------
// LceQuantize / LceDequantize
// ------------------------------------------------------------------------------------
static lce_hip_status launch_bitpack_rows(lce_hip_dtype in_type, const void* in_dev, uint64_t rows,
                                          uint64_t cols, int32_t zero_point, uint32_t* out_dev,
                                          hipStream_t st) {
  const uint32_t wpr = (uint32_t)((cols + 31) / 32);
  const uint32_t segs = (uint32_t)((cols + 63) / 64);
  const uint64_t tasks = rows * segs;
  const unsigned grid = grid_for_stream(tasks, 4);
  const lce::FastDiv dv = lce::make_fastdiv(segs);
  if (in_type == LCE_HIP_F32)
    lce::bitpack_rows<float><<<grid, 256, 0, st>>>((const float*)in_dev, out_dev, (uint32_t)rows, (uint32_t)cols, wpr, 0, dv, segs, tasks);
  else if (in_type == LCE_HIP_I8)
    lce::bitpack_rows<int8_t><<<grid, 256, 0, st>>>((const int8_t*)in_dev, out_dev, (uint32_t)rows, (uint32_t)cols, wpr, zero_point, dv, segs, tasks);
  else
    lce::bitpack_rows<uint8_t><<<grid, 256, 0, st>>>((const uint8_t*)in_dev, out_dev, (uint32_t)rows, (uint32_t)cols, wpr, zero_point, dv, segs, tasks);
  LCE_HIP_TRY(hipGetLastError());
  return LCE_HIP_OK;
}

lce_hip_status lce_hip_bitpack(lce_hip_dtype in_type, const void* in_dev, size_t rows, size_t cols,
                               int32_t zero_point, int32_t* out_dev, void* stream) {
  if (in_type != LCE_HIP_F32 && in_type != LCE_HIP_I8 && in_type != LCE_HIP_BOOL)
    return fail(LCE_HIP_ERR_INVALID, "lce_hip_bitpack: input type must be float32, int8 or bool");
  if (in_type == LCE_HIP_F32 && zero_point != 0)
    return fail(LCE_HIP_ERR_INVALID, "lce_hip_bitpack: float input requires zero_point 0");
  if (rows == 0 || cols == 0) return LCE_HIP_OK;
  if (!in_dev || !out_dev) return fail(LCE_HIP_ERR_INVALID, "lce_hip_bitpack: null tensor");
  if (cols >= (1ull << 31) || rows >= (1ull << 32))
    return fail(LCE_HIP_ERR_UNSUPPORTED, "lce_hip_bitpack: tensor too large");
  if (lce_hip_status s = require_device()) return s;
  hipStream_t st = (hipStream_t)stream;
  if (in_type == LCE_HIP_BOOL) zero_point = 1;  // quantization.cc:86-108
  const size_t esz = in_type == LCE_HIP_F32 ? 4 : 1;
  const uint64_t wpr = (cols + 31) / 32;
  const uint64_t total_words = (uint64_t)rows * wpr;
  const bool flat = cols % 32 == 0 && ((uintptr_t)in_dev % 16 == 0) && ((uintptr_t)out_dev % 16 == 0);
  if (!flat || total_words < 32)
    return launch_bitpack_rows(in_type, in_dev, rows, cols, zero_point, (uint32_t*)out_dev, st);
  // bitpack.h:294-298: no per-row padding -> the tensor is one flat array; 32 words per wave step
  const uint64_t blocks32 = total_words / 32;
  const unsigned grid = grid_for_stream(blocks32, 4);
  if (in_type == LCE_HIP_F32)
    lce::bitpack_f32_flat<><<<grid, 256, 0, st>>>((const float*)in_dev, (uint32_t*)out_dev, blocks32);
  else if (in_type == LCE_HIP_I8)
    lce::bitpack_b8_flat<false><<<grid, 256, 0, st>>>((const uint8_t*)in_dev, (uint32_t*)out_dev, blocks32, zero_point);
  else
    lce::bitpack_b8_flat<true><<<grid, 256, 0, st>>>((const uint8_t*)in_dev, (uint32_t*)out_dev, blocks32, zero_point);
  LCE_HIP_TRY(hipGetLastError());
  const uint64_t done_words = blocks32 * 32;
  if (done_words == total_words) return LCE_HIP_OK;
  // fewer than 32 words left: one ragged "row" of the flat array
  return launch_bitpack_rows(in_type, (const char*)in_dev + done_words * 32 * esz, 1,
                             (total_words - done_words) * 32, zero_point,
                             (uint32_t*)out_dev + done_words, st);
}

lce_hip_status lce_hip_unpack(lce_hip_dtype out_type, const int32_t* in_dev, size_t rows, size_t cols,
                              float scale, int32_t zero_point, void* out_dev, void* stream) {
  if (out_type != LCE_HIP_F32 && out_type != LCE_HIP_I8 && out_type != LCE_HIP_BOOL)
    return fail(LCE_HIP_ERR_INVALID, "lce_hip_unpack: output type must be float32, int8 or bool");
  if (rows == 0 || cols == 0) return LCE_HIP_OK;
  if (!in_dev || !out_dev) return fail(LCE_HIP_ERR_INVALID, "lce_hip_unpack: null tensor");
  if (cols >= (1ull << 31)) return fail(LCE_HIP_ERR_UNSUPPORTED, "lce_hip_unpack: cols too large");
  if (lce_hip_status s = require_device()) return s;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t wpr = (uint32_t)((cols + 31) / 32);
  const uint64_t total = (uint64_t)rows * cols;
  const unsigned grid = grid_for_stream((total + 63) / 64, 4);
  // whole words only and a 16-byte aligned destination: the flat, division-free kernel
  const bool flat = cols % 32 == 0 && ((uintptr_t)out_dev & 15) == 0;
  const uint64_t chunks_f32 = total / 4, chunks_b8 = total / 16;
  auto dv = (lce_dev::u32x4*)out_dev;
  if (out_type == LCE_HIP_F32) {
    if (flat) lce::unpack_flat<float><<<grid_for_stream((chunks_f32 + 63) / 64, 4), 256, 0, st>>>((const uint32_t*)in_dev, dv, chunks_f32, 1.0f, -1.0f);
    else lce::unpack_rows<float><<<grid, 256, 0, st>>>((const uint32_t*)in_dev, (float*)out_dev, total, (uint32_t)cols, wpr, 1.0f, -1.0f);
  } else if (out_type == LCE_HIP_I8) {
    // quantization.cc:131-138
    if (!(scale > 0.0f)) return fail(LCE_HIP_ERR_INVALID, "lce_hip_unpack: int8 output needs a positive scale");
    const int offset = (int)std::round(1.0f / scale);
    const int zero_bit = std::min(127, zero_point + offset);
    const int one_bit = std::max(-128, zero_point - offset);
    if (flat) lce::unpack_flat<int8_t><<<grid_for_stream((chunks_b8 + 63) / 64, 4), 256, 0, st>>>((const uint32_t*)in_dev, dv, chunks_b8, (int8_t)zero_bit, (int8_t)one_bit);
    else lce::unpack_rows<int8_t><<<grid, 256, 0, st>>>((const uint32_t*)in_dev, (int8_t*)out_dev, total, (uint32_t)cols, wpr, (int8_t)zero_bit, (int8_t)one_bit);
  } else {
    if (flat) lce::unpack_flat<uint8_t><<<grid_for_stream((chunks_b8 + 63) / 64, 4), 256, 0, st>>>((const uint32_t*)in_dev, dv, chunks_b8, (uint8_t)1, (uint8_t)0);
    else lce::unpack_rows<uint8_t><<<grid, 256, 0, st>>>((const uint32_t*)in_dev, (uint8_t*)out_dev, total, (uint32_t)cols, wpr, (uint8_t)1, (uint8_t)0);
  }
  LCE_HIP_TRY(hipGetLastError());
  return LCE_HIP_OK;
}

// ------------------------------------------------------------------------------------
// LceBconv2d
// ------------------------------------------------------------------------------------
lce_hip_status lce_hip_bconv2d_plan_create(const lce_hip_bconv2d_desc* desc, lce_hip_bconv2d_plan** plan) {
  if (!desc || !plan) return fail(LCE_HIP_ERR_INVALID, "lce_hip_bconv2d_plan_create: null argument");
  lce_hip_bconv2d_plan* p = new lce_hip_bconv2d_plan();
  p->host.d = *desc;
  const std::string err = lce::validate_and_infer(p->host);
  if (!err.empty()) {
    delete p;
    *plan = nullptr;
    return fail(LCE_HIP_ERR_INVALID, "%s", err.c_str());
  }
  *plan = p;
  return LCE_HIP_OK;
}

void lce_hip_bconv2d_plan_destroy(lce_hip_bconv2d_plan* plan) { delete plan; }

lce_hip_status lce_hip_bconv2d_plan_output_shape(const lce_hip_bconv2d_plan* plan, int32_t dims[4]) {
  if (!plan || !dims) return fail(LCE_HIP_ERR_INVALID, "plan_output_shape: null argument");
  const lce::HostPlan& h = plan->host;
  dims[0] = h.d.batch;
  dims[1] = h.out_h;
  dims[2] = h.out_w;
  dims[3] = h.d.dst_type == LCE_HIP_BITPACKED ? h.wout : h.d.channels_out;
  return LCE_HIP_OK;
}

lce_hip_status lce_hip_bconv2d_plan_padding(const lce_hip_bconv2d_plan* plan, int32_t* pad_h, int32_t* pad_w) {
  if (!plan || !pad_h || !pad_w) return fail(LCE_HIP_ERR_INVALID, "plan_padding: null argument");
  *pad_h = plan->host.pad_h;
  *pad_w = plan->host.pad_w;
  return LCE_HIP_OK;
}

lce_hip_status lce_hip_bconv2d_plan_set_weights(lce_hip_bconv2d_plan* plan, const int32_t* filter,
                                                const float* post_mul, const float* post_bias,
                                                const int32_t* thresholds) {
  if (!plan || !filter) return fail(LCE_HIP_ERR_INVALID, "plan_set_weights: null plan or filter");
  const bool bp = plan->host.d.dst_type == LCE_HIP_BITPACKED;
  if (bp && !thresholds) return fail(LCE_HIP_ERR_INVALID, "plan_set_weights: bitpacked output needs thresholds");
  if (!bp && (!post_mul || !post_bias))
    return fail(LCE_HIP_ERR_INVALID, "plan_set_weights: float/int8 output needs post_activation_multiplier and _bias");
  lce::fold_parameters(plan->host, filter, post_mul, post_bias, thresholds);
  plan->device_current = false;
  plan->selected_for_pixels = -1;
  return LCE_HIP_OK;
}

lce_hip_status lce_hip_bconv2d_plan_folded(const lce_hip_bconv2d_plan* plan, float* mul, float* bias,
                                           int32_t* clamp_min, int32_t* clamp_max) {
  if (!plan) return fail(LCE_HIP_ERR_INVALID, "plan_folded: null plan");
  const lce::HostPlan& h = plan->host;
  if (!h.have_weights || h.d.dst_type == LCE_HIP_BITPACKED)
    return fail(LCE_HIP_ERR_INVALID, "plan_folded: no folded float transform on this plan");
  if (mul) memcpy(mul, h.mul.data(), h.mul.size() * sizeof(float));
  if (bias) memcpy(bias, h.bias.data(), h.bias.size() * sizeof(float));
  if (clamp_min) *clamp_min = h.clamp_min;
  if (clamp_max) *clamp_max = h.clamp_max;
  return LCE_HIP_OK;
}

lce_hip_status lce_hip_bconv2d_plan_set_option(lce_hip_bconv2d_plan* plan, const char* key, const char* value) {
  if (!plan || !key || !value) return fail(LCE_HIP_ERR_INVALID, "plan_set_option: null argument");
  lce::HostPlan& h = plan->host;
  if (!strcmp(key, "tile2d")) {   // tuning aid for the direct variant: auto | on | off
    if (!strcmp(value, "auto")) h.tile2d_pref = 0;
    else if (!strcmp(value, "on")) h.tile2d_pref = 1;
    else if (!strcmp(value, "off")) h.tile2d_pref = 2;
    else return fail(LCE_HIP_ERR_INVALID, "plan_set_option: tile2d must be auto|on|off");
    plan->selected_for_pixels = -1;
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "pointwise_tiles")) {   // tuning aid for the 1x1 streaming kernel: tiles per wave
    const int v = atoi(value);
    if (v < 0 || v > 8 || (v == 0 && strcmp(value, "0")))
      return fail(LCE_HIP_ERR_INVALID, "plan_set_option: pointwise_tiles must be 0 (auto) .. 8");
    h.pw_tiles_pref = v;
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "pointwise_channels")) {   // tuning aid for the 1x1 streaming kernel: output channels per block
    const int v = atoi(value);
    if (!(v == 32 || v == 64 || v == 128 || (v == 0 && !strcmp(value, "0"))))
      return fail(LCE_HIP_ERR_INVALID, "plan_set_option: pointwise_channels must be 0 (auto), 32, 64 or 128");
    h.pw_nj_pref = v / 32;
    plan->selected_for_pixels = -1;
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "stream_rows")) {       // tuning aid for the streaming kernel: output rows per segment (0 = auto)
    const int v = atoi(value);
    if (v < 0 || (v == 0 && strcmp(value, "0")))
      return fail(LCE_HIP_ERR_INVALID, "plan_set_option: stream_rows must be 0 (auto) or a positive row count");
    h.stream_rows_pref = v;
    plan->selected_for_pixels = -1;
    plan->device_current = false;
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "stream_strip")) {      // testing aid for the streaming kernel: -1 auto, 0 never, else the column strip's width
    const int v = atoi(value);
    if (v < -1 || (v == 0 && strcmp(value, "0"))) return fail(LCE_HIP_ERR_INVALID, "plan_set_option: stream_strip must be -1 (auto), 0 (whole rows) or a strip width");
    h.stream_strip_pref = v;
    plan->selected_for_pixels = -1;
    plan->device_current = false;
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "stream_interleave")) {   // the streaming kernel's segment -> block map: 1 = block b owns segments b, b + grid, ... (a compact write window), 0 = consecutive ones
    if (strcmp(value, "0") && strcmp(value, "1") && strcmp(value, "auto")) return fail(LCE_HIP_ERR_INVALID, "plan_set_option: stream_interleave must be auto, 0 or 1");
    h.stream_interleave_pref = value[0] == 'a' ? -1 : value[0] == '1';
    plan->selected_for_pixels = -1;
    plan->device_current = false;
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "stream_blocks_per_cu")) {   // the streaming kernel's launch: blocks per CU (2: only where the instance is compiled for it and both blocks' LDS fit)
    if (strcmp(value, "1") && strcmp(value, "2") && strcmp(value, "auto")) return fail(LCE_HIP_ERR_INVALID, "plan_set_option: stream_blocks_per_cu must be auto, 1 or 2");
    h.stream_occ_pref = value[0] == 'a' ? 0 : value[0] - '0';
    plan->selected_for_pixels = -1;
    plan->device_current = false;
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "wstream_blocks") || !strcmp(key, "wstream_images")) {   // tuning aids for the weight-streaming kernel: pixel blocks per block (1..4), images per group; 0 = auto
    const int v = atoi(value);
    const bool blocks = key[8] == 'b';
    if (v < 0 || (v == 0 && strcmp(value, "0")) || (blocks && v > 4))
      return fail(LCE_HIP_ERR_INVALID, "plan_set_option: wstream_blocks must be 0 (auto) .. 4, wstream_images 0 (auto) or a positive count");
    (blocks ? h.ws_blocks_pref : h.ws_images_pref) = v;
    plan->selected_for_pixels = -1;
    plan->device_current = false;
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "int8_rounding")) {      // testing aid: "exact" keeps the round-half-away instances even where floor(x + 0.5) is proven equal
    if (strcmp(value, "auto") && strcmp(value, "exact")) return fail(LCE_HIP_ERR_INVALID, "plan_set_option: int8_rounding must be auto or exact");
    h.int8_exact_pref = value[0] == 'e';
    plan->selected_for_pixels = -1;
    plan->device_current = false;
    h.wq.clear();
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "stream_flat")) {       // testing aid for the streaming kernel: 0 = never cut pixel blocks across a block's images
    if (strcmp(value, "0") && strcmp(value, "1")) return fail(LCE_HIP_ERR_INVALID, "plan_set_option: stream_flat must be 0 or 1");
    h.stream_noflat = value[0] == '0';
    plan->selected_for_pixels = -1;
    plan->device_current = false;
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "stream_pixel_phases")) {   // tuning aid for the streaming kernel: pixel phases per block (the other waves take channel slices)
    const int v = atoi(value);
    if (!(v == 1 || v == 2 || v == 4 || (v == 0 && !strcmp(value, "0"))))
      return fail(LCE_HIP_ERR_INVALID, "plan_set_option: stream_pixel_phases must be 0 (auto), 1, 2 or 4");
    h.stream_phases_pref = v;
    plan->selected_for_pixels = -1;
    plan->device_current = false;
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "compute_units")) {     // testing aid: the device's CU count as the streaming kernel's planner sees it
    const int v = atoi(value);
    if (v < 1) return fail(LCE_HIP_ERR_INVALID, "plan_set_option: compute_units must be positive");
    h.num_cus = v;
    plan->cus_forced = true;
    plan->selected_for_pixels = -1;
    plan->device_current = false;
    return LCE_HIP_OK;
  }
  if (!strcmp(key, "engine")) {
    if (!strcmp(value, "auto")) h.engine_pref = 0;
    else if (!strcmp(value, "valu")) h.engine_pref = 1;
    else if (!strcmp(value, "mfma")) h.engine_pref = 2;
    else if (!strcmp(value, "direct")) h.engine_pref = 3;
    else if (!strcmp(value, "pointwise")) h.engine_pref = 4;
    else if (!strcmp(value, "stream")) h.engine_pref = 5;
    else if (!strcmp(value, "wstream")) h.engine_pref = 6;
    else return fail(LCE_HIP_ERR_INVALID, "plan_set_option: engine must be auto|valu|mfma|direct|pointwise|stream|wstream");
  } else if (!strcmp(key, "phase")) {
    // profiling aid for the matrix-core engine: time its two kernels separately
    if (!strcmp(value, "all")) h.phase = 0;
    else if (!strcmp(value, "expand")) h.phase = 1;
    else if (!strcmp(value, "gemm")) h.phase = 2;
    else return fail(LCE_HIP_ERR_INVALID, "plan_set_option: phase must be all|expand|gemm");
    return LCE_HIP_OK;
  } else if (!strcmp(key, "epilogue")) {
    // tuning aid for the matrix-core engine's float / int8 epilogues
    if (!strcmp(value, "auto")) h.epilogue_pref = 0;
    else if (!strcmp(value, "tile")) h.epilogue_pref = 1;
    else if (!strcmp(value, "wide")) h.epilogue_pref = 2;
    else return fail(LCE_HIP_ERR_INVALID, "plan_set_option: epilogue must be auto|tile|wide");
    return LCE_HIP_OK;
  } else if (!strcmp(key, "kernel")) {
    if (!strcmp(value, "auto")) h.kernel_pref = 0;
    else if (!strcmp(value, "tiled")) h.kernel_pref = 1;
    else if (!strcmp(value, "general")) h.kernel_pref = 2;
    else return fail(LCE_HIP_ERR_INVALID, "plan_set_option: kernel must be auto|tiled|general");
  } else if (!strcmp(key, "tile")) {
    int tm = 0, tn = 0;
    if (!strcmp(value, "auto")) { h.tile_pref = lce::TileShape{0, 0}; }
    else if (sscanf(value, "%dx%d", &tm, &tn) == 2 &&
             ((tn <= 32 && lce::lookup_tiled(LCE_HIP_F32, tm, tn, 1)) || lce::mfma_cfg_by_tile(tm, tn))) {
      h.tile_pref = lce::TileShape{tm, tn};
    } else {
      return fail(LCE_HIP_ERR_INVALID, "plan_set_option: tile must be auto, a xor-popcount tile "
                  "(4x16|2x32|2x16|1x32|1x16) or, with engine=mfma, a block tile "
                  "(256x256|256x128|512x64|128x256|128x128|256x64|128x64)");
    }
  } else {
    return fail(LCE_HIP_ERR_INVALID, "plan_set_option: unknown key '%s'", key);
  }
  plan->selected_for_pixels = -1;
  plan->device_current = false;
  h.packed.clear();
  h.wq.clear();
  return LCE_HIP_OK;
}

// (The kernel choice does not depend on whether a call asks for the second output: one plan, one selection, both kinds of call.)
const char* lce_hip_bconv2d_plan_kernel_name(lce_hip_bconv2d_plan* plan);
const char* lce_hip_bconv2d_plan_kernel_name_dual(lce_hip_bconv2d_plan* plan) { return lce_hip_bconv2d_plan_kernel_name(plan); }

const char* lce_hip_bconv2d_plan_kernel_name(lce_hip_bconv2d_plan* plan) {
  if (!plan) return "";
  const int chunk = lce::max_batch_per_launch(plan->host);
  if (ensure_selected(plan, chunk) != LCE_HIP_OK) return "";
  return plan->host.kernel_name.c_str();
}

lce_hip_status lce_hip_bconv2d_plan_int8_epilogue(lce_hip_bconv2d_plan* plan, int32_t* one_instruction_forms, int32_t* adjusted_channels) {
  if (!plan) return fail(LCE_HIP_ERR_INVALID, "plan_int8_epilogue: null plan");
  if (one_instruction_forms) *one_instruction_forms = 0;
  if (adjusted_channels) *adjusted_channels = 0;
  const lce::HostPlan& h = plan->host;
  if (h.d.dst_type != LCE_HIP_I8 || !h.have_weights) return LCE_HIP_OK;
  if (lce_hip_status s = ensure_selected(plan, lce::max_batch_per_launch(h))) return s;
  // (only the streaming / weight-streaming / pointwise kernels have the proven forms; the block GEMM and the xor-popcount engine run
  //  the reference's sequence whatever the proof said)
  const bool forms = h.use_mfma && (h.use_stream || h.use_wstream || h.use_pointwise) && h.int8_floor_ok;
  if (one_instruction_forms) *one_instruction_forms = forms ? 1 : 0;
  if (adjusted_channels) *adjusted_channels = forms ? h.int8_bias_adjusted : 0;
  return LCE_HIP_OK;
}

// Images [first, first + count) of the plan's batch; input_dev / output_dev / sign_dev point at image 0.
// sign_dev (float output only, may be null): the LceQuantize of the output, written by the same epilogue
// when the kernel variant can do it, otherwise by a bitpack launch behind it.
static lce_hip_status run_images(lce_hip_bconv2d_plan* plan, const int32_t* input_dev, void* output_dev,
                                 int32_t* sign_dev, int first, int count, hipStream_t st) {
  lce::HostPlan& h = plan->host;
  const int chunk = lce::max_batch_per_launch(h);
  if (lce_hip_status s = check_device(plan)) return s;          // binds the plan to the current device on its first run ...
  if (!plan->cus_forced && plan->host.num_cus != device_compute_units(plan->device) && device_compute_units(plan->device) > 0)
    plan->selected_for_pixels = -1;                              // ... whose size the streaming kernel's grid follows
  if (lce_hip_status s = ensure_selected(plan, chunk)) return s;
  if (lce_hip_status s = ensure_uploaded(plan)) return s;

  const size_t in_img_words = (size_t)h.d.in_height * h.d.in_width * h.cw;
  const size_t out_row = h.d.dst_type == LCE_HIP_BITPACKED ? (size_t)h.wout : (size_t)h.d.channels_out;
  const size_t out_img_bytes = (size_t)h.out_h * h.out_w * out_row * out_elem_bytes(h.d.dst_type);
  const size_t sign_img_words = (size_t)h.out_h * h.out_w * h.wout;

  for (int b0 = first; b0 < first + count; b0 += chunk) {
    const int nb = std::min(chunk, first + count - b0);
    ConvArgs A = lce::make_conv_args(h, nb);
    const uint32_t* in = (const uint32_t*)input_dev + (size_t)b0 * in_img_words;
    void* out = (char*)output_dev + (size_t)b0 * out_img_bytes;
    uint32_t* sgn = sign_dev ? (uint32_t*)sign_dev + (size_t)b0 * sign_img_words : nullptr;
    bool sign_fused = false;
    if (h.use_mfma && h.use_pointwise && ((uintptr_t)out & 15) == 0) {
      // 1x1 streaming kernel: waves walk 32-pixel tiles of the launch's pixel matrix
      if (lce_hip_status s = mfma_selftest_once(plan->device, 0)) return s;
      lce::pointwise_fn fn = lce::lookup_pointwise(h.d.dst_type, h.pw_nc, h.pw_nj, h.d.stride_height != 1 || h.d.stride_width != 1, h.int8_floor_ok);
      if (!fn) return fail(LCE_HIP_ERR_UNSUPPORTED, "bconv2d_run: no kernel instance for %s", h.kernel_name.c_str());
      const lce::PwArgs P = lce::make_pw_args(h, nb);
      // k tiles per wave: enough blocks (>= 12 per CU when the launch has them) for the dispatcher to even
      // out the CUs, few enough that a wave's register-resident filter bank is loaded once per several tiles
      const int pw_k = h.pw_tiles_pref > 0 ? h.pw_tiles_pref : std::max(1, std::min(8, P.tiles / (4 * 256 * 12)));
      const unsigned gx = (unsigned)(((int64_t)P.tiles + 4 * pw_k - 1) / (4 * pw_k));
      const dim3 grid(gx, (unsigned)(h.d.channels_out / (32 * h.pw_nj)));
      hipLaunchKernelGGL(fn, grid, dim3(256), (size_t)(4 * h.pw_nj * 4096), st, P, in, plan->d_wq.ptr, plan->d_mul.ptr,
                         plan->d_bias.ptr, plan->d_thrq.ptr, out, sgn);
      LCE_HIP_TRY(hipGetLastError());
      sign_fused = true;   // (also when there is none to write)
    } else if (h.use_mfma && h.use_wstream) {
      // weight-streaming kernel: a block per (group of images, part of its pixel blocks), two resident per CU
      if (lce_hip_status s = mfma_selftest_once(plan->device, 2)) return s;
      const lce::WsArgs G = lce::make_ws_args(h, nb);
      const bool with_sign = sgn != nullptr && h.d.dst_type != LCE_HIP_BITPACKED;
      lce::wstream_fn fn = lce::lookup_wstream(h.d.dst_type, lce::stream_chunks(h.d), h.ws_nb, with_sign, h.int8_floor_ok);
      if (!fn) return fail(LCE_HIP_ERR_UNSUPPORTED, "bconv2d_run: no kernel instance for %s", h.kernel_name.c_str());
      const size_t lds = (size_t)lce::wstream_lds_bytes(h);
      if (lds > 64 * 1024 && plan->lds_opt_in != (void*)fn) {
        LCE_HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        plan->lds_opt_in = (void*)fn;
      }
      const dim3 grid((unsigned)(G.GROUPS * G.PARTS), (unsigned)h.ws_ny);
      hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, G, (const uint8_t*)in, plan->d_wq.ptr, plan->d_mul.ptr,
                         plan->d_bias.ptr, plan->d_thrq.ptr, plan->d_sched.ptr, out, with_sign ? sgn : nullptr);
      LCE_HIP_TRY(hipGetLastError());
      sign_fused = true;   // (also when there is none to write)
    } else if (h.use_mfma && h.use_stream) {
      // weight-stationary streaming kernel: one persistent block per CU walks its run of segments
      if (lce_hip_status s = mfma_selftest_once(plan->device, 1)) return s;
      const lce::StreamArgs G = lce::make_stream_args(h, nb);
      const bool with_sign = sgn != nullptr && h.d.dst_type != LCE_HIP_BITPACKED;
      lce::stream_fn fn = lce::lookup_stream(h.d.dst_type, lce::stream_chunks(h.d), lce::stream_fast(G), lce::stream_clamps(G), with_sign, G.NSTRIP > 1, h.int8_floor_ok);
      if (!fn) return fail(LCE_HIP_ERR_UNSUPPORTED, "bconv2d_run: no kernel instance for %s", h.kernel_name.c_str());
      const size_t lds = (size_t)lce::stream_lds_bytes(h);
      if (lds > 64 * 1024 && plan->lds_opt_in != (void*)fn) {
        LCE_HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        plan->lds_opt_in = (void*)fn;
      }
      const dim3 grid((unsigned)G.GX, (unsigned)h.st_ny);
      hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, G, (const uint8_t*)in, plan->d_wq.ptr, plan->d_mul.ptr,
                         plan->d_bias.ptr, plan->d_thrq.ptr, plan->d_sched.ptr, out, with_sign ? sgn : nullptr);
      LCE_HIP_TRY(hipGetLastError());
      sign_fused = true;   // (also when there is none to write)
    } else if (h.use_mfma) {
      mfma_fn fn = lce::lookup_mfma(h.d.dst_type, h.mfma.bm(), h.mfma.bn(), h.zero_pad_mode == lce::kZeroPadCorrection,
                             h.use_direct, h.use_direct && h.tile_tx > 0);
      if (!fn) return fail(LCE_HIP_ERR_UNSUPPORTED, "bconv2d_run: no kernel instance for %s", h.kernel_name.c_str());
      const lce::MfmaArgs G = lce::make_mfma_args(h, nb);
      const int bm = h.mfma.bm(), bn = h.mfma.bn();
      // the joint-transpose float epilogue is the one that can also emit the sign words
      sign_fused = sgn && (h.d.dst_type == LCE_HIP_I8 ? G.i8_wide != 0 : G.f32_wide != 0) &&
                   h.zero_pad_mode != lce::kZeroPadCorrection && ((uintptr_t)out & 15) == 0;
      uint32_t* ksgn = sign_fused ? sgn : nullptr;
      if (h.use_direct) {
        // no workspace: every block expands its own input halo into LDS
        const size_t lds = (size_t)h.mfma.direct_lds_bytes(h.halo_bytes);
        if (lds > 64 * 1024 && plan->lds_opt_in != (void*)fn) {
          LCE_HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          plan->lds_opt_in = (void*)fn;
        }
        const dim3 grid((unsigned)(h.ipt > 1 ? (nb + h.ipt - 1) / h.ipt : (int64_t)nb * h.tpi), (unsigned)(h.npad / bn));
        hipLaunchKernelGGL(fn, grid, dim3(h.mfma.threads()), lds, st, A, G, (const uint8_t*)in, plan->d_wq.ptr,
                           plan->d_mul.ptr, plan->d_bias.ptr, plan->d_thrq.ptr, plan->d_zpc.ptr, out, ksgn);
        LCE_HIP_TRY(hipGetLastError());
      } else {
        const size_t ws = lce::mfma_workspace_bytes(h, nb);
        // While `st` is being captured into a graph the cross-stream ordering is left to the capture's own stream order:
        // an event recorded inside a capture belongs to the graph and must not be waited on or synchronised from outside.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        const bool capturing = cap != hipStreamCaptureStatusNone;
        if (capturing && plan->workspace_bytes < ws)
          return fail(LCE_HIP_ERR_INVALID, "bconv2d_run: the FP4 workspace must exist before a stream capture (run the plan once first)");
        if (!capturing && plan->ws_used && plan->ws_stream != st) LCE_HIP_TRY(hipStreamWaitEvent(st, plan->ws_done, 0));
        if (plan->workspace_bytes < ws) {
          // a larger workspace: the old one may still be read by a run in flight on another stream
          if (plan->ws_used) LCE_HIP_TRY(hipEventSynchronize(plan->ws_done));
          if (plan->workspace) (void)hipFree(plan->workspace);
          plan->workspace = nullptr;
          plan->workspace_bytes = 0;
          LCE_HIP_TRY(hipMalloc(&plan->workspace, ws + 256));
          plan->workspace_bytes = ws;
        }
        const uint64_t chunks = (uint64_t)G.NPIX * (uint64_t)((G.CPW + 3) / 4);  // threads of expand_fp4
        if (h.phase != 2) {
          LCE_HIP_TRY((hipError_t)lce::launch_expand_fp4(grid_for_stream((chunks + 63) / 64, 4), (void*)st, in, plan->workspace, G, chunks));
        }
        if (h.phase != 1) {
          const dim3 grid((unsigned)((A.M + bm - 1) / bm), (unsigned)(h.npad / bn));
          hipLaunchKernelGGL(fn, grid, dim3(h.mfma.threads()), (size_t)h.mfma.lds_bytes(), st, A, G,
                             (const uint8_t*)plan->workspace, plan->d_wq.ptr, plan->d_mul.ptr, plan->d_bias.ptr,
                             plan->d_thrq.ptr, plan->d_zpc.ptr, out, ksgn);
          LCE_HIP_TRY(hipGetLastError());
        } else {
          sign_fused = true;   // profiling mode "expand only": nothing to quantize
        }
        if (!capturing) {
          if (!plan->ws_done) LCE_HIP_TRY(hipEventCreateWithFlags(&plan->ws_done, hipEventDisableTiming));
          LCE_HIP_TRY(hipEventRecord(plan->ws_done, st));
          plan->ws_stream = st;
          plan->ws_used = true;
        }
      }
    } else if (h.use_tiled) {
      tiled_fn fn = lce::lookup_tiled(h.d.dst_type, h.tile.tm, h.tile.tn, h.ch);
      if (!fn) return fail(LCE_HIP_ERR_UNSUPPORTED, "bconv2d_run: no kernel instance for %s", h.kernel_name.c_str());
      const int64_t tasks = (int64_t)A.PT * A.NT;
      const int wpb = 4;
      const unsigned grid = (unsigned)((tasks + wpb - 1) / wpb);
      hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * wpb), 0, st, A, in, plan->d_packed.ptr,
                         plan->d_mul.ptr, plan->d_bias.ptr, plan->d_thr.ptr, plan->d_oobc.ptr,
                         plan->d_zpc.ptr, out);
      LCE_HIP_TRY(hipGetLastError());
    } else {
      general_fn fn = lce::lookup_general(h.d.dst_type);
      const unsigned gx = (unsigned)((A.M + 255) / 256);
      const unsigned gy = (unsigned)((h.d.channels_out + 31) / 32);
      hipLaunchKernelGGL(fn, dim3(gx, gy), dim3(256), 0, st, A, in, plan->d_filter.ptr,
                         plan->d_mul.ptr, plan->d_bias.ptr, plan->d_thr.ptr, plan->d_zpc.ptr, out);
      LCE_HIP_TRY(hipGetLastError());
    }
    if (sgn && !sign_fused) {
      // this kernel variant has no second output: LceQuantize as its own launch on the same stream
      const bool i8 = h.d.dst_type == LCE_HIP_I8;
      if (lce_hip_status s = lce_hip_bitpack(i8 ? LCE_HIP_I8 : LCE_HIP_F32, out, (size_t)nb * h.out_h * h.out_w,
                                             (size_t)h.d.channels_out, i8 ? h.d.out_zero_point : 0, (int32_t*)sgn, (void*)st))
        return s;
    }
  }
  return LCE_HIP_OK;
}

static lce_hip_status run_checks(lce_hip_bconv2d_plan* plan, const void* input, const void* output, const char* who) {
  if (!plan) return fail(LCE_HIP_ERR_INVALID, "%s: null argument", who);
  if (!plan->host.have_weights) return fail(LCE_HIP_ERR_INVALID, "%s: plan_set_weights was not called", who);
  if (plan->host.d.batch == 0) return LCE_HIP_OK;   // empty batch: tensors may legitimately be null
  if (!input || !output) return fail(LCE_HIP_ERR_INVALID, "%s: null argument", who);
  return require_device();
}

lce_hip_status lce_hip_bconv2d_run(lce_hip_bconv2d_plan* plan, const int32_t* input_dev, void* output_dev, void* stream) {
  if (lce_hip_status s = run_checks(plan, input_dev, output_dev, "bconv2d_run")) return s;
  if (plan->host.d.batch == 0) return LCE_HIP_OK;
  return run_images(plan, input_dev, output_dev, nullptr, 0, plan->host.d.batch, (hipStream_t)stream);
}

lce_hip_status lce_hip_bconv2d_run_dual(lce_hip_bconv2d_plan* plan, const int32_t* input_dev, void* output_dev,
                                        int32_t* output_bits_dev, void* stream) {
  if (lce_hip_status s = run_checks(plan, input_dev, output_dev, "bconv2d_run_dual")) return s;
  if (plan->host.d.dst_type == LCE_HIP_BITPACKED)
    return fail(LCE_HIP_ERR_INVALID, "bconv2d_run_dual: the plan's output type must be float32 or int8 (a bitpacked-output "
                "plan already writes bits)");
  if (plan->host.d.batch == 0) return LCE_HIP_OK;
  if (!output_bits_dev) return fail(LCE_HIP_ERR_INVALID, "bconv2d_run_dual: null argument");
  if (lce_hip_status s = check_device(plan)) return s;
  return run_images(plan, input_dev, output_dev, output_bits_dev, 0, plan->host.d.batch, (hipStream_t)stream);
}

int lce_hip_bconv2d_plan_device(const lce_hip_bconv2d_plan* plan) { return plan ? plan->device : -1; }

// Host tensors (the TFLite interpreter's arena).  The batch is cut into slices that flow through three
// streams -- H2D | kernel | D2H -- so the copies of neighbouring slices overlap the compute of the one in
// between.  The copies are truly asynchronous only from page-locked memory: a host that owns long-lived
// buffers (an arena) registers them once with lce_hip_host_register; with pageable memory the runtime
// stages each copy and the slices still pipeline, just less tightly.
lce_hip_status lce_hip_bconv2d_run_host(lce_hip_bconv2d_plan* plan, const int32_t* input_host, void* output_host) {
  if (lce_hip_status s = run_checks(plan, input_host, output_host, "bconv2d_run_host")) return s;
  const lce::HostPlan& h = plan->host;
  if (h.d.batch == 0) return LCE_HIP_OK;   // empty batch
  if (lce_hip_status s = check_device(plan)) return s;
  const size_t in_img = (size_t)h.d.in_height * h.d.in_width * h.cw * 4;
  const size_t out_row = h.d.dst_type == LCE_HIP_BITPACKED ? (size_t)h.wout : (size_t)h.d.channels_out;
  const size_t out_img = (size_t)h.out_h * h.out_w * out_row * out_elem_bytes(h.d.dst_type);
  const size_t in_bytes = in_img * h.d.batch, out_bytes = out_img * h.d.batch;
  if (plan->stage_in_bytes < in_bytes) {
    if (plan->stage_in) (void)hipFree(plan->stage_in);
    plan->stage_in = nullptr;
    plan->stage_in_bytes = 0;
    LCE_HIP_TRY(hipMalloc(&plan->stage_in, in_bytes));
    plan->stage_in_bytes = in_bytes;
  }
  if (plan->stage_out_bytes < out_bytes) {
    if (plan->stage_out) (void)hipFree(plan->stage_out);
    plan->stage_out = nullptr;
    plan->stage_out_bytes = 0;
    LCE_HIP_TRY(hipMalloc(&plan->stage_out, out_bytes));
    plan->stage_out_bytes = out_bytes;
  }
  // slices of >= ~8 MiB of traffic each, at most 8 of them
  const size_t per_image = in_img + out_img;
  int slices = (int)std::min<size_t>(8, std::max<size_t>(1, (per_image * h.d.batch) / (8u << 20)));
  slices = std::min(slices, h.d.batch);
  if (!plan->s_h2d) {
    LCE_HIP_TRY(hipStreamCreateWithFlags(&plan->s_h2d, hipStreamNonBlocking));
    LCE_HIP_TRY(hipStreamCreateWithFlags(&plan->s_run, hipStreamNonBlocking));
    LCE_HIP_TRY(hipStreamCreateWithFlags(&plan->s_d2h, hipStreamNonBlocking));
  }
  while ((int)plan->ev_in.size() < slices) {
    hipEvent_t a, b;
    LCE_HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
    LCE_HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
    plan->ev_in.push_back(a);
    plan->ev_run.push_back(b);
  }
  // on a failure mid-pipeline the copies already queued still read / write the caller's buffers: wait for them before
  // reporting it
  auto drain_and = [&](lce_hip_status s) {
    const std::string keep = g_last_error;
    (void)hipStreamSynchronize(plan->s_h2d);
    (void)hipStreamSynchronize(plan->s_run);
    (void)hipStreamSynchronize(plan->s_d2h);
    (void)hipGetLastError();
    g_last_error = keep;
    return s;
  };
#define LCE_HIP_TRY_DRAIN(expr)                                                                            \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) return drain_and(fail(LCE_HIP_ERR_RUNTIME, "%s failed: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  const int base = h.d.batch / slices, extra = h.d.batch % slices;
  int first = 0;
  for (int k = 0; k < slices; ++k) {
    const int count = base + (k < extra ? 1 : 0);
    LCE_HIP_TRY_DRAIN(hipMemcpyAsync((char*)plan->stage_in + first * in_img, (const char*)input_host + first * in_img,
                               count * in_img, hipMemcpyHostToDevice, plan->s_h2d));
    LCE_HIP_TRY_DRAIN(hipEventRecord(plan->ev_in[k], plan->s_h2d));
    LCE_HIP_TRY_DRAIN(hipStreamWaitEvent(plan->s_run, plan->ev_in[k], 0));
    if (lce_hip_status s = run_images(plan, (const int32_t*)plan->stage_in, plan->stage_out, nullptr, first, count, plan->s_run))
      return drain_and(s);
    LCE_HIP_TRY_DRAIN(hipEventRecord(plan->ev_run[k], plan->s_run));
    LCE_HIP_TRY_DRAIN(hipStreamWaitEvent(plan->s_d2h, plan->ev_run[k], 0));
    LCE_HIP_TRY_DRAIN(hipMemcpyAsync((char*)output_host + first * out_img, (const char*)plan->stage_out + first * out_img,
                               count * out_img, hipMemcpyDeviceToHost, plan->s_d2h));
    first += count;
  }
#undef LCE_HIP_TRY_DRAIN
  LCE_HIP_TRY(hipStreamSynchronize(plan->s_d2h));   // the last slice's copy is the last thing queued anywhere
  LCE_HIP_TRY(hipStreamSynchronize(plan->s_run));
  return LCE_HIP_OK;
}

// ------------------------------------------------------------------------------------
// LceBMaxPool2d
// ------------------------------------------------------------------------------------
static int pool_out(int padding, int in, int filter, int stride) {
  if (stride == 0) return 0;
  return padding == LCE_HIP_PADDING_SAME ? (in + stride - 1) / stride : (in + stride - filter) / stride;
}

lce_hip_status lce_hip_bmaxpool_output_shape(int32_t in_h, int32_t in_w, int32_t fh, int32_t fw, int32_t sh,
                                             int32_t sw, int32_t padding, int32_t* out_h, int32_t* out_w) {
  if (!out_h || !out_w) return fail(LCE_HIP_ERR_INVALID, "bmaxpool_output_shape: null argument");
  if (sh == 0 || sw == 0 || fh == 0 || fw == 0)
    return fail(LCE_HIP_ERR_INVALID, "bmaxpool: strides and filter sizes must be non-zero");  // bmaxpool.cc:52-55
  if (padding != LCE_HIP_PADDING_SAME && padding != LCE_HIP_PADDING_VALID)
    return fail(LCE_HIP_ERR_INVALID, "bmaxpool: padding must be SAME or VALID");
  *out_h = pool_out(padding, in_h, fh, sh);
  *out_w = pool_out(padding, in_w, fw, sw);
  return LCE_HIP_OK;
}

lce_hip_status lce_hip_bmaxpool(const int32_t* input_dev, int32_t batch, int32_t in_h, int32_t in_w,
                                int32_t words, int32_t fh, int32_t fw, int32_t sh, int32_t sw,
                                int32_t padding, int32_t* output_dev, void* stream) {
  int32_t oh = 0, ow = 0;
  if (lce_hip_status s = lce_hip_bmaxpool_output_shape(in_h, in_w, fh, fw, sh, sw, padding, &oh, &ow)) return s;
  if (batch == 0) return LCE_HIP_OK;   // empty batch: nothing to do (core/bmaxpool.h:43 loops zero times)
  if (!input_dev || !output_dev) return fail(LCE_HIP_ERR_INVALID, "bmaxpool: null tensor");
  if (oh < 1 || ow < 1 || batch < 0 || words < 1) return fail(LCE_HIP_ERR_INVALID, "bmaxpool: empty tensor");
  if (lce_hip_status s = require_device()) return s;
  const int ph = std::max(0, (oh - 1) * sh + fh - in_h) / 2;
  const int pw = std::max(0, (ow - 1) * sw + fw - in_w) / 2;
  // 16 bytes per thread when the words of a pixel come in fours and the tensors are 16-byte aligned
  const bool vec = words % 4 == 0 && ((uintptr_t)input_dev & 15) == 0 && ((uintptr_t)output_dev & 15) == 0;
  const int groups = vec ? words / 4 : words;
  const uint64_t total = (uint64_t)batch * oh * ow * groups;
  const unsigned grid = grid_for_stream((total + 63) / 64, 4);
  auto kernel = vec ? lce::bmaxpool_words<4> : lce::bmaxpool_words<1>;
  kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const uint32_t*)input_dev, (uint32_t*)output_dev, batch, in_h, in_w,
                                                  groups, oh, ow, fh, fw, sh, sw, ph, pw, total, lce::make_fastdiv((uint32_t)groups),
                                                  lce::make_fastdiv((uint32_t)ow), lce::make_fastdiv((uint32_t)oh));
  LCE_HIP_TRY(hipGetLastError());
  return LCE_HIP_OK;
}

// ------------------------------------------------------------------------------------
// converter-side parameter preparation (host-only, lce_prepare.cpp)
// ------------------------------------------------------------------------------------
static lce_hip_status prep_status(const std::string& err) {
  return err.empty() ? LCE_HIP_OK : fail(LCE_HIP_ERR_INVALID, "%s", err.c_str());
}

lce_hip_status lce_hip_prepare_binary_filter(const float* filter_hwio, int32_t kh, int32_t kw, int32_t cin,
                                             int32_t cout, float* filter_ohwi, float* mul, float* bias) {
  return prep_status(lce::prepare_binary_filter(filter_hwio, kh, kw, cin, cout, filter_ohwi, mul, bias));
}

lce_hip_status lce_hip_prepare_fuse_post_op(lce_hip_post_op op, const float* value, int32_t value_count,
                                            float* mul, float* bias, int32_t channels_out) {
  return prep_status(lce::fuse_post_op((int)op, value, value_count, mul, bias, channels_out));
}

int lce_hip_prepare_can_fuse_activation(const float* mul, const float* bias, int32_t channels_out,
                                        int32_t padding, int32_t pad_values) {
  if (!mul || !bias || channels_out <= 0) return 0;
  return lce::can_fuse_activation(mul, bias, channels_out, padding == LCE_HIP_PADDING_SAME, pad_values) ? 1 : 0;
}

lce_hip_status lce_hip_prepare_bitpacked_output(float* filter_ohwi, int32_t kh, int32_t kw, int32_t cin,
                                                int32_t cout, int32_t activation, int32_t padding,
                                                int32_t pad_values, const float* mul, const float* bias,
                                                int32_t* thresholds) {
  return prep_status(lce::prepare_bitpacked_output(filter_ohwi, kh, kw, cin, cout, activation,
                                                   padding == LCE_HIP_PADDING_SAME, pad_values, mul, bias, thresholds));
}

lce_hip_status lce_hip_prepare_bitpack_filter(const float* filter_ohwi, int32_t kh, int32_t kw, int32_t cin,
                                              int32_t cout, int32_t* filter_words) {
  return prep_status(lce::bitpack_filter(filter_ohwi, kh, kw, cin, cout, filter_words));
}

#ifdef LCE_TIMELINE
#ifndef LCE_UNITY
#error "the time-stamp builds are single-translation-unit builds (-DLCE_UNITY, tools/build_exp.sh)"
#endif
// profiling aid (tools/timeline.py), not part of the ABI: the K-loop time stamps of the last launch
int lce_hip_debug_read_timeline(void* host, size_t bytes) {
  if (bytes > sizeof(lce::lce_timeline)) bytes = sizeof(lce::lce_timeline);
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(lce::lce_timeline), bytes);
}
#endif

#ifdef LCE_STREAM_PHASES
#ifndef LCE_UNITY
#error "the time-stamp builds are single-translation-unit builds (-DLCE_UNITY, tools/build_exp.sh)"
#endif
// profiling aid (tools/stream_phases.py), not part of the ABI: the per-block tile-step stamps of the last stream launch
int lce_hip_debug_read_stream_tl(void* host, size_t bytes) {
  if (bytes > sizeof(lce::lce_stream_tl)) bytes = sizeof(lce::lce_stream_tl);
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(lce::lce_stream_tl), bytes);
}
#endif

#ifdef LCE_PW_PHASES
#ifndef LCE_UNITY
#error "the time-stamp builds are single-translation-unit builds (-DLCE_UNITY, tools/build_exp.sh)"
#endif
// profiling aid (tools/pw_phases.py), not part of the ABI: the per-block stamps of the last pointwise launch
int lce_hip_debug_read_pw_tl(void* host, size_t bytes) {
  if (bytes > sizeof(lce::lce_pw_tl)) bytes = sizeof(lce::lce_pw_tl);
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(lce::lce_pw_tl), bytes);
}
int lce_hip_debug_clear_pw_tl(void) {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(lce::lce_pw_tl)) != hipSuccess) return 1;
  return (int)hipMemset(p, 0, sizeof(lce::lce_pw_tl));
}
#endif

#ifdef LCE_PHASES
#ifndef LCE_UNITY
#error "the time-stamp builds are single-translation-unit builds (-DLCE_UNITY, tools/build_exp.sh)"
#endif
// profiling aid (tools/phases.py), not part of the ABI: the per-block phase stamps of the last launch
int lce_hip_debug_read_phases(void* host, size_t bytes) {
  if (bytes > sizeof(lce::lce_phase_tl)) bytes = sizeof(lce::lce_phase_tl);
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(lce::lce_phase_tl), bytes);
}
int lce_hip_debug_clear_phases(void) {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(lce::lce_phase_tl)) != hipSuccess) return 1;
  return (int)hipMemset(p, 0, sizeof(lce::lce_phase_tl));
}
#endif

}  // extern "C"
