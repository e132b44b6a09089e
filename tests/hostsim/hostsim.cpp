// Host simulation driver -- TEST ONLY (see tests/hostsim/lce_device_intrinsics.h).
// Runs the real kernel bodies of compute-engine_amd/csrc/lce_kernels.h on the CPU with
// the real planner (lce_plan.cpp), mirroring the launch logic of lce_hip_api.hip.
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "lce_kernel_types.h"   // the kernel instances live in hs_tu_*.cpp (the product build's cut); here only their lookups
#include "lce_kernels.h"
#include "lce_kernels_mfma.h"     // expand_fp4
#include "lce_plan.h"
#include "lce_plan_internal.h"

using namespace lce;

namespace {

// Lanes of a wave executed one after the other (no wave collectives needed).
template <typename F>
void launch_sequential(int grid_x, int grid_y, int block, F&& body) {
  lce_dev::ThreadCtx& c = g_ctx;
  c.bar = nullptr;
  c.xchg = nullptr;
  c.bdim_x = block;
  c.gdim_x = grid_x;
  for (int by = 0; by < grid_y; ++by)
    for (int bx = 0; bx < grid_x; ++bx)
      for (int t = 0; t < block; ++t) {
        c.bid_x = bx; c.bid_y = by; c.tid_x = t;
        body();
      }
}

// The 64 lanes of every wave run in lock step (fibers of this thread: lce_device_intrinsics.h) so ballot / shuffle work; they walk
// the (block, wave) list together.
template <typename F>
void launch_lockstep(int grid_x, int block, F&& body) {
  lce_dev::FiberBarrier bar(64);
  uint32_t xchg[64] = {0};
  lce_dev::run_fibers(64,
    [&](int l, lce_dev::ThreadCtx& c) { c.bar = &bar; c.xchg = xchg; c.bdim_x = block; c.gdim_x = grid_x; (void)l; },
    [&](int l) {
      lce_dev::ThreadCtx& c = g_ctx;
      for (int bx = 0; bx < grid_x; ++bx)
        for (int w = 0; w < block / 64; ++w) {
          c.bid_x = bx; c.bid_y = 0; c.tid_x = w * 64 + l;
          body();
          bar.arrive_and_wait();   // the exchange area is free again before the next wave starts
        }
    });
}

// A block's LDS: exactly the bytes the launch asks for (rounded up to 16), zeroed, ENDING at an inaccessible page -- a kernel body that
// touches LDS past its launch's allocation faults here (on the GPU it would read or overwrite the LDS of a co-resident block: since
// round 6 two blocks of the streaming kernel share a CU, and the bitpacked instances lay out no epilogue scratch any more).
class GuardedLds {
 public:
  explicit GuardedLds(size_t bytes) {
    const size_t page = (size_t)sysconf(_SC_PAGESIZE), need = (bytes + 15) / 16 * 16;
    body_ = (need + page - 1) / page * page;
    if (body_ == 0) body_ = page;
    map_ = (uint8_t*)mmap(nullptr, body_ + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (map_ == MAP_FAILED) { map_ = nullptr; fprintf(stderr, "hostsim: mmap of a block's LDS failed\n"); abort(); }
    mprotect(map_ + body_, page, PROT_NONE);
    data_ = map_ + body_ - need;
    page_ = page;
  }
  ~GuardedLds() { if (map_) munmap(map_, body_ + page_); }
  GuardedLds(const GuardedLds&) = delete;
  GuardedLds& operator=(const GuardedLds&) = delete;
  uint8_t* data() const { return data_; }

 private:
  uint8_t* map_ = nullptr;
  uint8_t* data_ = nullptr;
  size_t body_ = 0, page_ = 0;
};

// A whole thread block in lock step: `block` fibers (64 per wave), a block barrier, shared LDS and a per-wave exchange area for
// the MFMA emulation.  One set of fibers per launch; they run the blocks one after the other, with zeroed LDS for each.
template <typename F>
void launch_block_lockstep(int grid_x, int grid_y, int block, size_t lds_bytes, F&& body) {
  // the blocks of a launch are independent: OS threads take them in turn, each with its own LDS and its own set of fibers
#pragma omp parallel for schedule(dynamic, 1)
  for (int bid = 0; bid < grid_x * grid_y; ++bid) {
    const int bx = bid % grid_x, by = bid / grid_x;
    GuardedLds lds(lds_bytes);
    lce_dev::FiberBarrier block_bar(block);
    std::vector<lce_dev::FiberBarrier> wave_bar(block / 64, lce_dev::FiberBarrier(64));
    std::vector<uint32_t> xchg((block / 64) * 64, 0), mx((block / 64) * 64 * 8, 0);
    lce_dev::run_fibers(block,
      [&](int t, lce_dev::ThreadCtx& c) {
        const int w = t / 64;
        c.bar = &wave_bar[w]; c.xchg = xchg.data() + w * 64; c.mfma_xchg = mx.data() + w * 64 * 8;
        c.block_bar = &block_bar; c.lds = lds.data();
        c.bdim_x = block; c.gdim_x = grid_x; c.tid_x = t; c.bid_x = bx; c.bid_y = by;
      },
      [&](int) { body(); });
  }
}

// An EXACT-size copy of a device upload whose last byte is followed by an inaccessible page: a kernel body that reads past the buffer
// faults here instead of reading whatever the heap holds behind it (the pointwise kernel loads its filter bank with plain pointer
// loads, no range check: round 5 shipped a four-K-step instance on a three-K-step image and the simulation, with 64 bytes of slack
// on a std::vector, read the heap just as the GPU read its neighbour allocation).
class GuardedBytes {
 public:
  explicit GuardedBytes(const std::vector<uint8_t>& v) {
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    body_ = (v.size() + page - 1) / page * page;
    map_ = (uint8_t*)mmap(nullptr, body_ + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (map_ == MAP_FAILED) { map_ = nullptr; return; }
    mprotect(map_ + body_, page, PROT_NONE);
    data_ = map_ + body_ - v.size();          // the copy ENDS at the guard page (16-byte aligned: the images are multiples of 32 bytes)
    if (!v.empty()) memcpy(data_, v.data(), v.size());
    page_ = page;
  }
  ~GuardedBytes() { if (map_) munmap(map_, body_ + page_); }
  GuardedBytes(const GuardedBytes&) = delete;
  GuardedBytes& operator=(const GuardedBytes&) = delete;
  const uint8_t* data() const { return data_; }
 private:
  uint8_t *map_ = nullptr, *data_ = nullptr;
  size_t body_ = 0, page_ = 0;
};

std::string g_err;
void* g_sign_out = nullptr;   // second output of the next hostsim_bconv2d call (float output, matrix-core engine)
int g_num_cus = 256;          // what the streaming kernel's planner takes for the device's CU count
int g_stream_rows = 0;        // its segment size (0 = auto)
int g_stream_phases = 0;      // its pixel phases per block (0 = auto)
int g_stream_strip = -1;      // its column strips (-1 auto, 0 never, else the width)
int g_stream_interleave = 0;  // its segment -> block map (1: interleaved runs)
int g_stream_occ = 0;         // its blocks per CU (0 = the estimate decides, 1, 2)
int g_pw_nj = 0;              // the pointwise kernel's 32-channel tiles per block (0 = auto)
int g_last_int8_adjusted = 0;   // ... and the number of channels whose bias its floor-rounding proof adjusted
int g_last_int8_floor = -1;   // the last convolution's plan: 1 = its int8 rounding ran as floor(x + 0.5), 0 = round-half-away, -1 = not an int8 matrix-core plan

}  // namespace

extern "C" {

const char* hostsim_last_error() { return g_err.c_str(); }
float hostsim_int8_below_threshold(int32_t zero_point) { return int8_below_threshold(zero_point); }
void hostsim_set_sign_output(void* words) { g_sign_out = words; }
void hostsim_set_stream(int num_cus, int rows) { g_num_cus = num_cus; g_stream_rows = rows; }
void hostsim_set_stream_phases(int phases) { g_stream_phases = phases; }
void hostsim_set_stream_strip(int width) { g_stream_strip = width; }
void hostsim_set_stream_interleave(int on) { g_stream_interleave = on; }
void hostsim_set_stream_blocks_per_cu(int n) { g_stream_occ = n; }
void hostsim_set_pointwise(int channel_tiles) { g_pw_nj = channel_tiles; }
int hostsim_last_int8_floor() { return g_last_int8_floor; }
int hostsim_last_int8_adjusted() { return g_last_int8_adjusted; }

// kernel_pref: 0 auto, 1 tiled, 2 general; tm/tn 0 = auto; max_batch 0 = planner's choice
// engine_pref: 0 auto, 1 valu, 2 mfma
int hostsim_bconv2d(const lce_hip_bconv2d_desc* desc, const int32_t* filter, const float* post_mul,
                    const float* post_bias, const int32_t* thresholds, const int32_t* input,
                    void* output, int kernel_pref, int tm, int tn, int max_batch, char* name_out,
                    int name_len, int engine_pref) {
  HostPlan h;
  h.d = *desc;
  std::string err = validate_and_infer(h);
  if (!err.empty()) { g_err = err; return 1; }
  fold_parameters(h, filter, post_mul, post_bias, thresholds);
  h.engine_pref = engine_pref;
  h.num_cus = g_num_cus;
  h.stream_rows_pref = g_stream_rows;
  h.stream_phases_pref = g_stream_phases;
  h.stream_strip_pref = g_stream_strip;
  h.stream_interleave_pref = g_stream_interleave;
  h.stream_occ_pref = g_stream_occ;
  h.pw_nj_pref = g_pw_nj;
  h.kernel_pref = kernel_pref;
  h.tile_pref = TileShape{tm, tn};
  int chunk = max_batch_per_launch(h);
  if (max_batch > 0) chunk = std::min(chunk, max_batch);
  err = select_kernel(h, (int64_t)chunk * h.out_h * h.out_w);
  if (!err.empty()) { g_err = err; return 2; }
  if (name_out && name_len > 0) {
    strncpy(name_out, h.kernel_name.c_str(), name_len - 1);
    name_out[name_len - 1] = 0;
  }
  g_last_int8_adjusted = h.int8_bias_adjusted;
  g_last_int8_floor = h.d.dst_type == LCE_HIP_I8 && h.use_mfma && (h.use_stream || h.use_wstream || h.use_pointwise) ? (h.int8_floor_ok ? 1 : 0) : -1;
  // the padded tables get the same slack as the device uploads
  auto slack_u = [](std::vector<uint32_t> v) { v.resize(v.size() + 16, 0u); return v; };
  const std::vector<uint32_t> packed = slack_u(h.packed), filt = slack_u(h.filter);
  const float* zpc = h.zero_pad_cache.empty() ? nullptr : h.zero_pad_cache.data();

  const size_t in_img_words = (size_t)h.d.in_height * h.d.in_width * h.cw;
  const size_t out_row = h.d.dst_type == LCE_HIP_BITPACKED ? (size_t)h.wout : (size_t)h.d.channels_out;
  const size_t out_img_bytes = (size_t)h.out_h * h.out_w * out_row * (h.d.dst_type == LCE_HIP_I8 ? 1 : 4);
  for (int b0 = 0; b0 < h.d.batch; b0 += chunk) {
    const int nb = std::min(chunk, h.d.batch - b0);
    const ConvArgs A = make_conv_args(h, nb);
    const uint32_t* in = (const uint32_t*)input + (size_t)b0 * in_img_words;
    void* out = (char*)output + (size_t)b0 * out_img_bytes;
    if (h.use_mfma && h.use_wstream) {
      const WsArgs G = make_ws_args(h, nb);
      uint32_t* sgn = g_sign_out && h.d.dst_type != LCE_HIP_BITPACKED ? (uint32_t*)g_sign_out + (size_t)b0 * h.out_h * h.out_w * h.wout : nullptr;
      wstream_fn fn = lookup_wstream(h.d.dst_type, stream_chunks(h.d), h.ws_nb, sgn != nullptr, h.int8_floor_ok);
      if (!fn) { g_err = "no kernel instance for " + h.kernel_name; return 3; }
      std::vector<uint8_t> wq = h.wq;
      wq.resize(wq.size() + 64, 0);
      std::vector<uint32_t> tabs = h.st_tabs;
      tabs.resize(tabs.size() + 16, 0u);
      launch_block_lockstep(G.GROUPS * G.PARTS, h.ws_ny, 256, (size_t)wstream_lds_bytes(h), [&] {
        fn(G, (const uint8_t*)in, wq.data(), h.mul_q.data(), h.bias_q.data(), h.thr_q.data(), tabs.data(), out, sgn);
      });
    } else if (h.use_mfma && h.use_stream) {
      const StreamArgs G = make_stream_args(h, nb);
      uint32_t* sgn = g_sign_out && h.d.dst_type != LCE_HIP_BITPACKED ? (uint32_t*)g_sign_out + (size_t)b0 * h.out_h * h.out_w * h.wout : nullptr;
      stream_fn fn = lookup_stream(h.d.dst_type, stream_chunks(h.d), stream_fast(G), stream_clamps(G), sgn != nullptr, G.NSTRIP > 1, h.int8_floor_ok);
      if (!fn) { g_err = "no kernel instance for " + h.kernel_name; return 3; }
      std::vector<uint8_t> wq = h.wq;
      wq.resize(wq.size() + 64, 0);
      std::vector<uint32_t> sched = h.st_tabs;
      sched.resize(sched.size() + 16, 0u);
      launch_block_lockstep(G.GX, h.st_ny, 256, (size_t)stream_lds_bytes(h), [&] {
        fn(G, (const uint8_t*)in, wq.data(), h.mul_q.data(), h.bias_q.data(), h.thr_q.data(), sched.data(), out, sgn);
      });
    } else if (h.use_mfma && h.use_pointwise) {
      pointwise_fn fn = lookup_pointwise(h.d.dst_type, h.pw_nc, h.pw_nj, h.d.stride_height != 1 || h.d.stride_width != 1, h.int8_floor_ok);
      if (!fn) { g_err = "no kernel instance for " + h.kernel_name; return 3; }
      const PwArgs P = make_pw_args(h, nb);
      const GuardedBytes wq(h.wq);      // exact size, guard page behind it (no slack: the bank loads are unchecked pointer loads)
      if (!wq.data()) { g_err = "hostsim: mmap failed"; return 3; }
      // a small grid: waves loop over several tiles
      launch_block_lockstep(std::min((P.tiles + 3) / 4, 2), h.d.channels_out / (32 * h.pw_nj), 256, (size_t)(4 * h.pw_nj * 4096), [&] {
        fn(P, in, wq.data(), h.mul_q.data(), h.bias_q.data(), h.thr_q.data(), out,
           g_sign_out ? (uint32_t*)g_sign_out + (size_t)b0 * h.out_h * h.out_w * h.wout : nullptr);
      });
    } else if (h.use_mfma) {
      mfma_fn fn = lookup_mfma(h.d.dst_type, h.mfma.bm(), h.mfma.bn(), h.zero_pad_mode == lce::kZeroPadCorrection,
                             h.use_direct, h.use_direct && h.tile_tx > 0);
      if (!fn) { g_err = "no kernel instance for " + h.kernel_name; return 3; }
      const MfmaArgs G = make_mfma_args(h, nb);
      std::vector<uint8_t> wq = h.wq;
      wq.resize(wq.size() + 64, 0);
      const int bm = h.mfma.bm(), bn = h.mfma.bn();
      if (h.use_direct) {
        launch_block_lockstep(h.ipt > 1 ? (nb + h.ipt - 1) / h.ipt : nb * h.tpi, h.npad / bn, h.mfma.threads(), (size_t)h.mfma.direct_lds_bytes(h.halo_bytes), [&] {
          fn(A, G, (const uint8_t*)in, wq.data(), h.mul_q.data(), h.bias_q.data(), h.thr_q.data(), zpc, out, g_sign_out ? (uint32_t*)g_sign_out + (size_t)b0 * h.out_h * h.out_w * h.wout : nullptr);
        });
        continue;
      }
      const size_t ws = mfma_workspace_bytes(h, nb);
      std::vector<lce_dev::u32x4> work(ws / 16 + 16);
      launch_sequential(3, 1, 256, [&] { expand_fp4<>(in, work.data(), G, (uint64_t)G.NPIX * (uint64_t)((G.CPW + 3) / 4)); });
      launch_block_lockstep((A.M + bm - 1) / bm, h.npad / bn, h.mfma.threads(), (size_t)h.mfma.lds_bytes(), [&] {
        fn(A, G, (const uint8_t*)work.data(), wq.data(), h.mul_q.data(), h.bias_q.data(), h.thr_q.data(), zpc, out, g_sign_out ? (uint32_t*)g_sign_out + (size_t)b0 * h.out_h * h.out_w * h.wout : nullptr);
      });
    } else if (h.use_tiled) {
      tiled_fn fn = lookup_tiled(h.d.dst_type, h.tile.tm, h.tile.tn, h.ch);
      if (!fn) { g_err = "no kernel instance for " + h.kernel_name; return 3; }
      const int64_t tasks = (int64_t)A.PT * A.NT;
      const int wpb = 4;
      launch_sequential((int)((tasks + wpb - 1) / wpb), 1, 64 * wpb, [&] {
        fn(A, in, packed.data(), h.mul_p.data(), h.bias_p.data(), h.thr_p.data(), h.oob_corr.data(), zpc, out);
      });
    } else {
      general_fn fn = lookup_general(h.d.dst_type);
      launch_sequential((A.M + 255) / 256, (h.d.channels_out + 31) / 32, 256, [&] {
        fn(A, in, filt.data(), h.mul.data(), h.bias.data(), h.thresholds.data(), zpc, out);
      });
    }
  }
  return 0;
}

// Exhaustive check of lce::round_sat_i8 against saturate(roundf(y)) over every finite float
// (and the infinities).  Returns the number of mismatches; *first_bad = bits of the first one.
uint64_t hostsim_check_round_sat_i8(uint32_t* first_bad) {
  uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (int64_t hi = 0; hi < 65536; ++hi) {
    for (uint32_t lo = 0; lo < 65536; ++lo) {
      const uint32_t bits = ((uint32_t)hi << 16) | lo;
      float y;
      memcpy(&y, &bits, 4);
      if (y != y) continue;   // NaN: the reference's behaviour is unspecified
      float r = roundf(y);
      r = r < -128.0f ? -128.0f : (r > 127.0f ? 127.0f : r);
      if (round_sat_i8(y) != (int)r) {
        if (bad == 0 && first_bad) *first_bad = bits;
        ++bad;
      }
    }
  }
  return bad;
}

// in_type: LCE_HIP_F32 / I8 / BOOL.  force_rows != 0 -> always the ballot kernel.
int hostsim_bitpack(int in_type, const void* in, uint64_t rows, uint64_t cols, int32_t zero_point,
                    uint32_t* out, int force_rows) {
  if (in_type == LCE_HIP_BOOL) zero_point = 1;
  const size_t esz = in_type == LCE_HIP_F32 ? 4 : 1;
  const uint64_t wpr = (cols + 31) / 32, total_words = rows * wpr;
  auto rows_kernel = [&](const void* src, uint64_t r, uint64_t c, uint32_t* dst) {
    const uint32_t w = (uint32_t)((c + 31) / 32), segs = (uint32_t)((c + 63) / 64);
    const uint64_t tasks = r * segs;
    const int grid = (int)std::min<uint64_t>((tasks + 3) / 4, 3);  // small grid: exercises the stride loop
    const FastDiv dv = make_fastdiv(segs);
    launch_lockstep(grid, 256, [&] {
      if (in_type == LCE_HIP_F32) bitpack_rows<float>((const float*)src, dst, (uint32_t)r, (uint32_t)c, w, 0, dv, segs, tasks);
      else if (in_type == LCE_HIP_I8) bitpack_rows<int8_t>((const int8_t*)src, dst, (uint32_t)r, (uint32_t)c, w, zero_point, dv, segs, tasks);
      else bitpack_rows<uint8_t>((const uint8_t*)src, dst, (uint32_t)r, (uint32_t)c, w, zero_point, dv, segs, tasks);
    });
  };
  const bool flat = !force_rows && cols % 32 == 0 && total_words >= 32;
  if (!flat) { rows_kernel(in, rows, cols, out); return 0; }
  const uint64_t blocks32 = total_words / 32;
  const int grid = (int)std::min<uint64_t>((blocks32 + 3) / 4, 2);
  launch_lockstep(grid, 256, [&] {
    if (in_type == LCE_HIP_F32) bitpack_f32_flat((const float*)in, out, blocks32);
    else if (in_type == LCE_HIP_I8) bitpack_b8_flat<false>((const uint8_t*)in, out, blocks32, zero_point);
    else bitpack_b8_flat<true>((const uint8_t*)in, out, blocks32, zero_point);
  });
  const uint64_t done = blocks32 * 32;
  if (done < total_words)
    rows_kernel((const char*)in + done * 32 * esz, 1, (total_words - done) * 32, out + done);
  return 0;
}

int hostsim_unpack(int out_type, const uint32_t* in, uint64_t rows, uint64_t cols, float scale,
                   int32_t zero_point, void* out) {
  const uint32_t wpr = (uint32_t)((cols + 31) / 32);
  const uint64_t total = rows * cols;
  const bool flat = cols % 32 == 0 && ((uintptr_t)out & 15) == 0;   // same rule as lce_hip_unpack
  auto dv = (lce_dev::u32x4*)out;
  launch_sequential(2, 1, 256, [&] {
    if (out_type == LCE_HIP_F32) {
      if (flat) unpack_flat<float>(in, dv, total / 4, 1.0f, -1.0f);
      else unpack_rows<float>(in, (float*)out, total, (uint32_t)cols, wpr, 1.0f, -1.0f);
    } else if (out_type == LCE_HIP_I8) {
      const int offset = (int)roundf(1.0f / scale);
      const int8_t z = (int8_t)std::min(127, zero_point + offset), o = (int8_t)std::max(-128, zero_point - offset);
      if (flat) unpack_flat<int8_t>(in, dv, total / 16, z, o);
      else unpack_rows<int8_t>(in, (int8_t*)out, total, (uint32_t)cols, wpr, z, o);
    } else {
      if (flat) unpack_flat<uint8_t>(in, dv, total / 16, (uint8_t)1, (uint8_t)0);
      else unpack_rows<uint8_t>(in, (uint8_t*)out, total, (uint32_t)cols, wpr, (uint8_t)1, (uint8_t)0);
    }
  });
  return 0;
}

int hostsim_bmaxpool(const uint32_t* in, int b, int h, int w, int c, int fh, int fw, int sh, int sw,
                     int padding, uint32_t* out) {
  const int oh = padding == LCE_HIP_PADDING_SAME ? (h + sh - 1) / sh : (h + sh - fh) / sh;
  const int ow = padding == LCE_HIP_PADDING_SAME ? (w + sw - 1) / sw : (w + sw - fw) / sw;
  const int ph = std::max(0, (oh - 1) * sh + fh - h) / 2, pw = std::max(0, (ow - 1) * sw + fw - w) / 2;
  // as lce_hip_bmaxpool: four words per thread when a pixel's words come in fours (np arrays are 16-byte aligned)
  const bool vec = c % 4 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0;
  const int groups = vec ? c / 4 : c;
  const uint64_t total = (uint64_t)b * oh * ow * groups;
  launch_sequential(2, 1, 256, [&] {
    if (vec) bmaxpool_words<4>(in, out, b, h, w, groups, oh, ow, fh, fw, sh, sw, ph, pw, total, make_fastdiv((uint32_t)groups),
                               make_fastdiv((uint32_t)ow), make_fastdiv((uint32_t)oh));
    else bmaxpool_words<1>(in, out, b, h, w, groups, oh, ow, fh, fw, sh, sw, ph, pw, total, make_fastdiv((uint32_t)groups),
                           make_fastdiv((uint32_t)ow), make_fastdiv((uint32_t)oh));
  });
  return 0;
}

uint32_t hostsim_fastdiv(uint32_t n, uint32_t d) { return fastdiv(n, make_fastdiv(d)); }

// ---- the planner's cost estimate, for tools/fit_cost.py (this library is built with -DLCE_COST_TUNABLE: the constants of
//      csrc/lce_plan_cost.cpp are variables here) ----
int hostsim_cost_count() { return cost_constant_count(); }
const char* hostsim_cost_name(int i) { return cost_constant_name(i); }
double hostsim_cost_get(int i) { return get_cost_constant(i); }
int hostsim_cost_set(int i, double v) { return set_cost_constant(i, v) ? 0 : 1; }
// The estimate (us) and kernel name of the plan that select_kernel makes of `desc` under the given preferences (engine_pref as
// hostsim_bconv2d; stream_rows / interleave: the streaming kernel's segment options); < 0: the configuration cannot be planned.
double hostsim_plan_estimate(const lce_hip_bconv2d_desc* desc, int engine_pref, int stream_rows, int interleave, int num_cus,
                             char* name_out, int name_len) {
  HostPlan h;
  h.d = *desc;
  if (!validate_and_infer(h).empty()) return -2.0;
  h.engine_pref = engine_pref;
  h.num_cus = num_cus;
  h.stream_rows_pref = stream_rows;
  h.stream_interleave_pref = interleave;
  const int chunk = max_batch_per_launch(h);
  if (!select_kernel(h, (int64_t)chunk * h.out_h * h.out_w).empty()) return -3.0;
  if (name_out && name_len > 0) {
    strncpy(name_out, h.kernel_name.c_str(), name_len - 1);
    name_out[name_len - 1] = 0;
  }
  return h.est_us;
}

}  // extern "C"
