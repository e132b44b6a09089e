// Host-side planner for LceBconv2d on MI355X: validation and shape inference (the work of
// bconv2d::Prepare), parameter folding (OneTimeSetup), filter repacking and kernel
// selection.  Pure C++ -- no HIP types -- so it is unit-testable without a GPU.
// Citations are relative to /root/reference/larq_compute_engine/.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/lce_hip.h"
#include "lce_kernel_args.h"

namespace lce {

struct TileShape {
  int tm, tn;
};

// Block shape of the matrix-core kernel: wgm x wgn waves, each wm x wn MFMA tiles of 32x32.
struct MfmaCfg {
  int wgm, wgn, wm, wn;
  int bm() const { return 32 * wm * wgm; }
  int bn() const { return 32 * wn * wgn; }
  int threads() const { return 64 * wgm * wgn; }
  static constexpr int kStages = 4;        // LDS ring depth of bconv2d_mfma (A + B per stage)
  static constexpr int kDirectStages = 3;  // ... of its direct variant (B only; the halo is extra)
  // The joint-transpose epilogues (float / int8 rows of WN*32 channels, and the float kernels' optional
  // sign-bit output) need waves x WN x 4 KiB of scratch.  Every block asks for at least that much: with
  // this kernel's register use no tile loses a resident block to it (e.g. 128x256: 64 KiB, two blocks per
  // CU either way), and every layer gets the lean epilogue.
  int epilogue_scratch_bytes() const { return (threads() / 64) * wn * 4096; }
  int lds_bytes() const { return std::max(kStages * (bm() + bn()) * 32, epilogue_scratch_bytes()); }
  int direct_lds_bytes(int halo_bytes) const {
    return std::max(halo_bytes + kDirectStages * bn() * 32, epilogue_scratch_bytes());
  }
};
// The instantiated shapes, by block tile (pixels x channels).
const MfmaCfg* mfma_cfg_by_tile(int bm, int bn);

struct HostPlan {
  lce_hip_bconv2d_desc d{};
  // inferred by Prepare (tflite/kernels/bconv2d.cc:203-210)
  int out_h = 0, out_w = 0, pad_h = 0, pad_w = 0;
  int cw = 0, cwg = 0, npg = 0, wout = 0;  // words per pixel / per group, Cout per group, out words
  int backtransform_add = 0;               // KH*KW*Cin/G (bconv2d.cc:361-362)
  int zero_pad_mode = kZeroPadNone;

  // OneTimeSetup results (bconv2d.cc:324-392)
  bool have_weights = false;
  // the call asks for the second (LceQuantize) output too: lce_hip_bconv2d_run_dual (the planner's auto rule looks at it)
  bool want_sign = false;
  std::vector<float> mul, bias;            // channels_out entries
  int32_t clamp_min = 0, clamp_max = 0;
  float bit_thr = 0.0f;                    // LceQuantize of the output as a compare: bit = value < bit_thr
  std::vector<int32_t> thresholds;
  std::vector<uint32_t> filter;            // OHWI copy (general kernel + repacking source)
  std::vector<float> zero_pad_cache;       // zero_padding_correction.h:39-176

  // kernel selection
  int kernel_pref = 0;                     // 0 auto, 1 tiled, 2 general
  TileShape tile_pref{0, 0};               // {0,0} = auto
  bool use_tiled = false;
  TileShape tile{0, 0};
  int ch = 1;                              // activation words per vector load
  int nt = 0;                              // channel tiles
  std::string kernel_name;

  // matrix-core engine (lce_kernels_mfma.h)
  int engine_pref = 0;                     // 0 auto, 1 valu (xor-popcount), 2 mfma (FP4 workspace GEMM), 3 direct (LDS halo),
                                           // 4 pointwise (1x1 streaming kernel, lce_kernels_pointwise.h),
                                           // 5 stream (weight-stationary persistent kernel, lce_kernels_stream.h),
                                           // 6 wstream (weight-streaming kernel for short launches, lce_kernels_wstream.h)
  bool use_direct = false;                 // with use_mfma: the LDS-halo variant, no workspace
  int tpi = 0, halo_rows = 0, ps = 0, halo_bytes = 0, ipt = 1;  // direct-variant geometry
  int tile_tx = 0, halo_w = 0;             // ... 2-D tiles: tiles across the image (0 = strip tiles), halo width in pixels
  int tile2d_pref = 0;                     // tuning aid: 0 auto, 1 always when it fits, 2 never
  int phase = 0;                           // profiling aid: 0 all, 1 expand_fp4 only, 2 GEMM only
  int epilogue_pref = 0;                   // float/int8 epilogue: 0 auto, 1 per-tile transpose, 2 joint transpose
  bool use_mfma = false;
  bool use_pointwise = false;              // with use_mfma: the 1x1 streaming kernel runs instead of the block GEMM
  int pw_nc = 0, pw_nj = 0;                // its K-steps and 32-channel tiles per block
  int pw_tiles_pref = 0;                   // tuning aid: 32-pixel tiles per wave (0 = auto)
  int pw_nj_pref = 0;                      // tuning aid: 32-channel tiles per block (0 = auto, 1, 2, 4)
  MfmaCfg mfma{0, 0, 0, 0};                // chosen block shape
  int cpad = 0, hp = 0, wp = 0, npad = 0;  // workspace geometry / padded channel count
  int kch = 0;                             // K-steps (64-channel chunks) per filter tap that a block runs: cpad/64,
                                           // or the chunks one group's channel slice touches
  // int8 plans: floor(y + 0.5) equals the reference's round-half-away on EVERY value this plan can produce (no reachable exact negative
  // tie; pack_for_mfma checks every channel x every accumulator value inside the clamps): the streaming kernels' one-instruction rounding
  bool int8_floor_ok = false;
  // channels that the proof of the one-instruction forms gave neighbouring parameters (lce_plan.cpp, prepare_int8_epilogue)
  int int8_bias_adjusted = 0;
  bool int8_exact_pref = false;            // testing aid (int8_rounding=exact): never take the floor instances
  std::vector<uint8_t> wq;                 // FP4 weights [KS][Npad][32 bytes]
  // 0: K-major [K-step][K-half][Npad][16 B]; 1: tile-major [Npad/32][K-step][K-half][32][16 B] (wstream)
  int wq_layout = 0;
  std::vector<float> mul_q, bias_q, thr_q; // Npad entries

  // weight-stationary streaming kernel (lce_kernels_stream.h); with use_mfma
  bool use_stream = false;
  int num_cus = 256;                       // compute units of the device (the C ABI fills it in; the stream kernel's grid)
  int stream_rows_pref = 0;                // tuning aid: output rows per segment (0 = auto)
  int stream_noflat = 0;                   // testing aid: never cut pixel blocks across a block's segments
  int stream_phases_pref = 0;              // tuning aid: pixel phases per block (0 = auto, else 1, 2 or 4: 4 / phases channel slices)
  int st_rs = 0, st_spi = 0, st_srs = 0, st_pbs = 0, st_pph_log = 0, st_ny = 1, st_qg = 0, st_ipr = 0;
  int st_pitch = 0;                          // bytes per ring row slot
  // column strips of wide images: strips per image, row segments per image, output columns per strip
  int st_nstrip = 1, st_rseg = 0, st_wso = 0;
  // testing aid: -1 auto, 0 never, else the strip width (a multiple of 32 that divides the output width)
  int stream_strip_pref = -1;
  // 1: a block's segments are gx apart (interleaved runs: a compact, moving write window), 0: consecutive, -1: the cost estimate decides
  int stream_interleave_pref = -1;
  int st_gstr = 1;                           // the planned segment stride of a block's run (1: consecutive segments)
  // blocks per CU the launch is planned for: 0 = the cost estimate decides (1, or 2 where the instance is compiled for two and both
  // blocks' LDS fit), 1 / 2 = tuning aid (stream_blocks_per_cu)
  int stream_occ_pref = 0;
  int st_occ = 1;                            // ... as planned
  // 1: 32-pixel blocks are cut from the concatenated pixels of a block's segments (whole small images)
  int st_flat = 0;
  int st_nq = 0;                             // pixel blocks of a full block's stream (rows of the context table)
  int st_spb = 0, st_gx = 0, st_rows = 0, st_ring_bytes = 0, st_batch = 0;   // ... for launches of st_batch images
  std::vector<uint32_t> st_tabs;           // [sched | lim | ctx]: the kernel's tables (lce_kernel_args.h, StreamArgs)
  uint32_t st_tab_lim = 0, st_tab_ctx = 0, st_tab_sgn = 0, st_tab_seg = 0;   // byte offsets of lim / ctx / sgn inside st_tabs

  // weight-streaming kernel (lce_kernels_wstream.h); with use_mfma.  Its tables live in st_tabs as the streaming kernel's do.
  bool use_wstream = false;
  // images per group, blocks per group, pixel blocks / pixels per group, most pixel blocks per block, grid.y
  int ws_ipb = 0, ws_parts = 0, ws_nq = 0, ws_npxg = 0, ws_nb = 0, ws_ny = 1;
  int ws_hp = 0, ws_wp = 0, ws_pitch = 0, ws_img_pitch = 0, ws_qg = 0, ws_lds_images = 0;   // LDS image geometry
  // tuning aids: most pixel blocks per block (0 = 4), images per group (0 = the cost model's choice)
  int ws_blocks_pref = 0, ws_images_pref = 0;
  int ws_occupancy = 1;                      // blocks that fit a CU's LDS side by side (the kernel is built for two)
  uint32_t ws_tab_part = 0, ws_tab_ctx = 0;  // byte offsets inside st_tabs
  int64_t ws_cost = 0;                       // the planner's cycle estimate of a launch (plan_wstream)

  // the cost estimate (us per launch) of the matrix-core kernel the last select_kernel took, as lce_plan_cost.cpp priced it (-1: the
  // selection was not priced -- the xor-popcount engine, the pointwise kernel).  tools/fit_cost.py and tools/planner_regret.py read it.
  double est_us = -1.0;

  // A/B and debugging aids, read from the environment once, when the plan is created (validate_and_infer) -- never on the selection path
  int dbg_level = 0;                         // LCE_PLAN_DEBUG=1|2: every candidate's price (2: and its terms) on stderr
  bool dbg_no_wstream = false;               // LCE_PLAN_NO_WSTREAM: the weight-streaming kernel is not among auto's candidates
  bool dbg_int8_exact = false;               // LCE_PLAN_INT8_EXACT: as the option int8_rounding=exact, for whole stacks
  // LCE_PLAN_INT8_FULL: the int8 proof enumerates every accumulator value (tests compare it with the bisection)
  bool dbg_int8_full = false;

  // tiled-kernel operands (built by pack_for_tile)
  std::vector<uint32_t> packed;            // [NT][KH*KW][Cwg][TN]
  std::vector<float> mul_p, bias_p;        // NT*TN, padded
  std::vector<int32_t> thr_p;              // NT*TN, padded with INT32_MAX
  std::vector<int32_t> oob_corr;           // [NT][KH*KW][TN]: (Cin/G)/2 - popcount(tap)
};

// Returns "" when the descriptor is acceptable, otherwise the message Prepare would log.
std::string validate_and_infer(HostPlan& p);

// OneTimeSetup: fold multiplier/bias/clamps (double arithmetic, float storage), keep
// thresholds, compute the zero-padding correction cache.
void fold_parameters(HostPlan& p, const int32_t* filter_ohwi, const float* post_mul,
                     const float* post_bias, const int32_t* thresholds);

// Smallest float T with round_sat_i8(T) >= zero_point: LceQuantize's "q < zero_point" on the value before rounding.
float int8_below_threshold(int32_t zero_point);

// Can the tiled kernel with channel tile `tn` run this convolution?
bool tiled_supports(const HostPlan& p, int tn);

// Choose kernel + tile for `pixels` output pixels per launch and (re)build the packed
// operands.  Returns "" or an error message (e.g. a forced variant that cannot run).
std::string select_kernel(HostPlan& p, int64_t pixels);

// Largest batch chunk one launch may take (buffer resources bind < 2 GiB).
int max_batch_per_launch(const HostPlan& p);

// Matrix-core engine: can it run this convolution, and its launch constants.
bool mfma_supported(const HostPlan& p);
bool choose_direct_cfg(const HostPlan& p, MfmaCfg* out);
constexpr int kDirectLdsMax = 160 * 1024;   // one block per CU
constexpr int kDirectLdsAuto = 80 * 1024;   // two blocks per CU: what the auto rule requires
bool direct_geometry(const HostPlan& p, const MfmaCfg& c, int* tpi, int* halo_rows, int* ps, int* halo_bytes,
                     int* ipt, int lds_budget, int* tile_tx = nullptr, int* halo_w = nullptr);
MfmaArgs make_mfma_args(const HostPlan& p, int batch_chunk);
// 1x1 streaming kernel: can it run this convolution (fills nc / nj), and its launch constants.
bool pointwise_supported(const HostPlan& p, int64_t pixels, int* nc, int* nj);
// K-steps of the pointwise INSTANCE that runs a 1x1 layer of `chunks` 64-channel chunks: 1, 2, 4 or 8 (3 chunks -- 129..192 channels --
// run the four-step instances: the fourth step's activations are masked to code 0 and pack_for_mfma appends a fourth, all-zero K-step to
// the weight image, so the kernel's register-resident bank is loaded from inside the allocation).  0: no instance.
inline int pointwise_bank_steps(int chunks) { return chunks <= 2 ? chunks : chunks <= 4 ? 4 : chunks == 8 ? 8 : 0; }
// a 1x1 launch with fewer wave-tiles (32 pixels x 128 channels) than this runs 64 channels per block
constexpr int64_t pw_small_launch_tiles = 16384;
PwArgs make_pw_args(const HostPlan& p, int batch_chunk);
size_t mfma_workspace_bytes(const HostPlan& p, int batch_chunk);

ConvArgs make_conv_args(const HostPlan& p, int batch_chunk);

// Streaming kernel: can it run this convolution at all; segment size, grid, ring and production schedule for
// launches of `batch_chunk` images (fills the st_* fields; "" or why not); its launch constants.
bool stream_supported(const HostPlan& p);
std::string plan_stream(HostPlan& p, int batch_chunk);
StreamArgs make_stream_args(const HostPlan& p, int batch_chunk);
// Weight-streaming kernel: can it run this convolution; groups / parts / LDS images for launches of `batch_chunk` images (fills
// the ws_* fields and st_tabs; "" or why not); its launch constants.
bool wstream_supported(const HostPlan& p);
std::string plan_wstream(HostPlan& p, int batch_chunk);
WsArgs make_ws_args(const HostPlan& p, int batch_chunk);
constexpr int kWsLdsExtra = 4 * 8192;      // four waves' epilogue scratch (lce_kernels_wstream.h, kWsScratch)
inline int wstream_lds_bytes(const HostPlan& p) { return p.ws_lds_images + kWsLdsExtra; }
constexpr int kStreamLdsExtra = 4 * 8192 + 4096;   // four waves' epilogue scratch + the dump area of idle producer lanes
// K-split (512 input channels): 4-KiB scratch per wave (it transposes 32 channels, not 64), the dump area, and two 4-KiB inbox
// slots per wave for the pair's partial sums
constexpr int kStreamLdsExtraKsplit = 4 * 4096 + 4096 + 4 * 8192;
// 64-channel chunks per tap of the streaming family's INSTANCE that runs the layer: 1, 2, 4 or 8 (its register-resident filter bank is
// built for those; 129..192 input channels run the 256-channel instance -- the fourth chunk's codes are 0 and contribute nothing --,
// 257..448 the 512-channel one).  0: more than 512 input channels.
inline int stream_chunks(const lce_hip_bconv2d_desc& d) {
  const int c = (d.channels_in + 63) / 64;
  return c <= 2 ? c : c <= 4 ? 4 : c <= 8 ? 8 : 0;
}
inline bool stream_ksplit(const HostPlan& p) { return stream_chunks(p.d) > 4; }
// (+ the strips' segment table)
// (bitpacked output: ballots, no transpose -- its instances lay out no epilogue scratch)
inline int stream_lds_extra(const HostPlan& p) {
  if (stream_ksplit(p)) return kStreamLdsExtraKsplit;
  // (the two-blocks-per-CU instance: a 2-KiB dump area -- its items are 32 bytes -- and no strips, hence no segment table)
  if (p.d.dst_type == LCE_HIP_BITPACKED && stream_chunks(p.d) == 1) return 2048;
  return (p.d.dst_type == LCE_HIP_BITPACKED ? 4096 : kStreamLdsExtra) + 1024;
}
// Blocks of the instance that can be resident on a CU at once (lce_kernels_stream.h, stream_blocks_per_cu: bitpacked output on the
// 64-input-channel bank is compiled for two) -- where both blocks' LDS fit, the planner launches that many per CU.
inline int stream_blocks_per_cu_max(const HostPlan& p) {
  return p.d.dst_type == LCE_HIP_BITPACKED && stream_chunks(p.d) == 1 ? 2 : 1;
}
inline int stream_lds_bytes(const HostPlan& p) { return p.st_ring_bytes + stream_lds_extra(p); }

}  // namespace lce
