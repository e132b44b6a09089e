// The planner's cost estimates (see lce_plan.h, lce_plan_internal.h): microseconds of one launch per candidate kernel.
//
// Rounds 3-4 decided by a list of shape conditions measured at batch 256 on the bench's layers; at other batch sizes the list fell
// through to whatever was left (profiles/r05/engine_sweep_box1.jsonl: batch 1 ... 64, the rule's pick took 1.2 - 7 x the best
// candidate's time on 128 of 192 (layer, batch, output type) rows).  Since round 5 every candidate that can run the layer is PLANNED
// (segments, blocks, block steps per block -- the planner's own simulation) and priced by a small model of where its time goes; the
// cheapest runs (lce_plan.cpp, select_kernel).  Candidates: the weight-stationary streaming kernel with the planner's own segments, or
// with interleaved runs of r-row segments for every divisor r of the output height; the weight-streaming kernel; the block GEMM.
//
// EVERY NUMBER the model uses is in the one table below, with its unit and the measurement it came from (round-5 review, item 6: the
// constants used to sit as literals inside the three functions).  They are fitted quantities: a ROCm, firmware or kernel change ages
// them.  `python tools/planner_regret.py --remeasure OUT.jsonl` (one GPU call) measures every candidate again on the 228-row grid and
// prints the regret of the planner's choice; tests/test_planner_choice.py refuses a sweep that was measured with other kernel sources
// than the tree's (the sweep records their hash).  LCE_PLAN_DEBUG=3 prints the table.
#include <algorithm>
#include <cmath>
#include <cstdio>

#include "lce_plan.h"
#include "lce_plan_internal.h"

namespace lce {
namespace cost {

// Sources.  "sweep": profiles/r05/engine_sweep_box{1,2,3}.jsonl + engine_sweep_i8_box{4,5,6}.jsonl (six boxes, tools/engine_sweep.py),
// re-measured in round 6: profiles/r06/engine_sweep_r06_box{1,2,3}.jsonl.  "re-fit r6": the value tools/fit_cost.py found on those three
// tables (coordinate descent on the estimates' log error + the regret of the pick; the held-out layers of engine_sweep.py never enter
// it).  Probes: tools/probes/*.hip, their outputs under profiles/.
#define LCE_COST_TABLE(X)                                                                                                          \
  /* ---- the chip ---- */                                                                                                         \
  X(kCyclesPerUs, 2100.0, "cycles/us", "the clock short launches sustain: profiles/r03/stream_phases.txt (s_memtime vs events)")   \
  X(kLaunchUs, 1.6, "us", "launch + first instruction of an empty kernel of these grids: "                                         \
    "tools/probes/launch_floor.hip, profiles/r03/pointwise_launch_floor.txt")                                                      \
  X(kMfmaCycles, 33.2, "cycles", "one v_mfma_f32_32x32x64_f8f6f4, FP4 operands, dependent chains: "                                \
    "tools/probes/mfma_gap.hip, profiles/r03/probe_mfma_gap.txt")                                                                  \
  X(kVmemBytesPerClk, 51.0, "B/clk", "a CU's vector memory path, loads into registers: "                                           \
    "profiles/r05/stream_phases_paced.txt (288 KiB in 5.8 k cycles)")                                                              \
  X(kL2BytesPerUs, 37.0e6, "B/us", "the eight L2s when every CU pulls the same filter bank (75 MB in 2.8 us): "                    \
    "profiles/r05/wstream_vs_stream.txt; re-fit r6")                                                                               \
  /* ---- weight-stationary streaming kernel (lce_kernels_stream.h) ---- */                                                        \
  X(kStPrologueUs, 0.9, "us", "a block's prologue without its bank: first rows, ring columns, quota + barrier: "                   \
    "profiles/r03/stream_phases.txt")                                                                                              \
  X(kStPrologueKsplitUs, 0.1, "us", "K-split: inbox setup on top: profiles/r04/stream_phases.txt")                                 \
  X(kStStepF32, 1.30, "x", "block step / its bare MFMA time, float output (woven epilogue + granted clock): sweep, L0 "            \
    "/ 14x14x256 rows")                                                                                                            \
  X(kStStepI8, 1.25, "x", "the same, int8 output: sweep (i8 boxes 4-6)")                                                           \
  X(kStStepBp, 1.05, "x", "the same, bitpacked output: sweep")                                                                     \
  X(kStStepKsplitF32, 1.15, "x", "K-split instances (a wave transforms 32 channels, not 64), float: sweep, 7x7x512 / "             \
    "14x14x512 rows")                                                                                                              \
  X(kStStepKsplitI8, 1.12, "x", "K-split, int8: sweep")                                                                            \
  X(kStStepKsplitBp, 1.05, "x", "K-split, bitpacked: sweep")                                                                       \
  X(kStSustained, 0.2, "x", "extra on launches long enough to reach the sustained (power-limited) state, float / "                 \
    "int8: profiles/r05/bench_box_spread.txt, sweep L0 rows")                                                                      \
  X(kStSustainedBp, 0.1, "x", "the same, bitpacked output")                                                                        \
  X(kStSustainedSteps, 100.0, "steps", "block steps per block at which that state is reached: sweep, 56x56 rows at "               \
    "batch 64 vs 256")                                                                                                             \
  X(kStEpiFloorF32, 1.05, "us", "floor of a block step: the epilogue's own issue time, float: "                                    \
    "profiles/r05/stream_phases_lowk.txt")                                                                                         \
  X(kStEpiFloorI8, 1.0, "us", "the same, int8 (after the one-instruction forms): profiles/r05/int8_floor_rounding.txt")            \
  X(kStEpiFloorBp, 0.8, "us", "the same, bitpacked: profiles/r05/stream_phases_lowk.txt")                                          \
  X(kStStepKsplitUs, 0.08, "us", "K-split: the pair's exchange + barrier per block step: profiles/r04/stream_phases.txt")          \
  X(kStTwoBlocksStepMul, 1.37, "x", "a block step beside a second resident block on the CU (two waves per SIMD; the pair's steps take "  \
    "1.37x, not 2x, one block's): profiles/r06/occ2_potential.txt, 56x56x64 / 112x112x64 bitpacked")                               \
  X(kStQuotaUs, 0.12, "us", "one out-of-line quota of 256 ring items: profiles/r03/stream_ablations.txt")                          \
  X(kStPartialUs, 0.2, "us", "out-of-line stores of a segment's ragged last pixel block: sweep, 14x14 / 7x7 rows with "            \
    "4-row segments")                                                                                                              \
  X(kStPartialKsplitUs, 0.12, "us", "the same, K-split instances")                                                                 \
  X(kStBlockTailUs, 0.5, "us", "drain of a block's last stores: profiles/r03/stream_phases.txt")                                   \
  X(kStoreBytesPerUs, 5.5e6, "B/us", "the chip's write rate, int8 / bitpacked rows: profiles/r04/store_pattern_vs_l0.txt")         \
  X(kStoreF32BytesPerUs, 5.0e6, "B/us", "float rows on launches of >= 205 MB (the power budget is shared with the "                \
    "matrix cores): profiles/r05/store_window.txt")                                                                                \
  X(kStoreF32SmallBonus, 0.65e6, "B/us", "... rising to 5.65 TB/s on <= 51 MB: sweep, 14x14x256 / 28x28x128 float rows")           \
  X(kStoreF32LargeMb, 205.0, "MB", "where the ramp starts")                                                                        \
  X(kStoreF32RampMb, 154.0, "MB", "its width (205 - 51)")                                                                          \
  X(kStoreWindowBonus, 0.42e6, "B/us", "a compact, moving write window (interleaved runs): profiles/r05/interleaved_runs.txt")     \
  X(kStoreWindowFull, 64.0e6, "B", "window size at which the bonus is gone")                                                       \
  X(kStoreWindowRamp, 48.0e6, "B", "... and the width of its ramp")                                                                \
  X(kStPaddedMul, 1.05, "x", "an instance wider than the layer (129..192 channels on the 256-channel bank): "                      \
    "profiles/r05/engine_sweep_padded_channels.jsonl")                                                                             \
  X(kStPaddedUs, 2.0, "us", "... and its general-path expansion")                                                                  \
  X(kStPaddedKsplitMul, 1.2, "x", "257..448 channels on the 512-channel instance: the same file")                                  \
  X(kStPaddedKsplitUs, 4.0, "us", "... and its expansion")                                                                         \
  X(kStPaddedBpMul, 1.13, "x", "... bitpacked output on a padded instance, on top (new in round 6); re-fit r6")                    \
  /* ---- weight-streaming kernel (lce_kernels_wstream.h) ---- */                                                                  \
  X(kWsPrologueCycles, 3100.0, "cycles", "a block's prologue: table loads + one global round trip: "                               \
    "profiles/r05/wstream_vs_stream.txt (phases); re-fit r6")                                                                      \
  X(kWsItemCycles, 660.0, "cycles", "... per 16-byte item per lane of the image expansion; re-fit r6")                             \
  X(kWsCrowdPrologueUs, 0.6, "us", "extra when two blocks share every CU")                                                         \
  X(kWsMfmaExtraCycles, 0.8, "cycles", "per MFMA of the K loop over kMfmaCycles, a block alone on its CU")                         \
  X(kWsMfmaCrowdCycles, 2.0, "cycles", "... more when two blocks share the CU")                                                    \
  X(kWsKstepLatencyUs, 0.05, "us", "floor per K-step with one or two pixel blocks per block: the weight loads' "                   \
    "latency over kWsPrefetch steps; re-fit r6")                                                                                   \
  X(kWsTailF32PerBlockUs, 0.5, "us", "float: transform + stores of one pixel block at the block's end")                            \
  X(kWsTailF32StoreShare, 0.6, "x", "float: share of the launch's output written behind the K loops (K-major); re-fit r6")         \
  X(kWsTailI8Us, 0.5, "us", "int8: transform of one pixel block")                                                                  \
  X(kWsTailI8CrowdUs, 0.45, "us", "... more beside a co-resident block")                                                           \
  X(kWsTailBpUs, 0.15, "us", "bitpacked: ballots + one store per pixel block")                                                     \
  X(kWsOneBlockPerCu, 1.15, "x", "K loops when only ONE block fits a CU's LDS (no co-resident block hides prologue / epilogue)")   \
  /* ---- block GEMM (lce_kernels_mfma.h) ---- */                                                                                  \
  X(kGemmAloneFixedUs, 3.4, "us", "a block alone on its CU: halo expansion + epilogue: profiles/r03/phases_block_gemm.txt")        \
  X(kGemmAloneSetupUs, 0.4, "us", "... ring fill")                                                                                 \
  X(kGemmAloneKstepUs, 0.138, "us", "... per K-step of a 128 x 128 tile (LDS port's rate): the same file")                         \
  X(kGemmAloneBpMul, 0.93, "x", "... bitpacked output")                                                                            \
  X(kGemmRoundFixedUs, 1.6, "us", "a round of blocks at throughput: per-block fixed part")                                         \
  X(kGemmRoundSetupUs, 1.6, "us", "... per unit of tile area (128 x 128)")                                                         \
  X(kGemmRoundKstepUs, 0.14, "us", "... per K-step and unit of area")                                                              \
  X(kGemmI8Mul, 0.94, "x", "rounds with int8 output: sweep")                                                                       \
  X(kGemmBpMul, 0.78, "x", "rounds with bitpacked output: sweep")                                                                  \
  X(kGemmSharedCuMul, 1.33, "x", "two blocks per CU slow each other down (one issue port per SIMD): profiles/r02/phases_l0.txt")   \
  X(kGemmPartialRoundMul, 0.33, "x", "between one and two blocks per CU: linear in the surplus")                                   \
  X(kGemmRoundsOffset, 0.35, "rounds", "fill + drain of a multi-round launch")                                                     \
  X(kGemmStoreFixedUs, 2.4, "us", "time to the first store")                                                                       \
  X(kGemmStoreBytesPerUs, 5.75e6, "B/us", "block tiles written at the chip's rate: profiles/r02/probe_store_overlap.txt")

#ifdef LCE_COST_TUNABLE
// The re-fit build (tests/hostsim only, never the product: csrc/lce_experiments.h): the constants are variables that
// tools/fit_cost.py moves through set_cost_constant() while it compares the estimates with a measured sweep.
#define LCE_COST_DECLARE(name, value, unit, source) double name = value;
#else
#define LCE_COST_DECLARE(name, value, unit, source) constexpr double name = value;
#endif
LCE_COST_TABLE(LCE_COST_DECLARE)
#undef LCE_COST_DECLARE

struct Constant { const char* name; double value; const char* unit; const char* source; };
#define LCE_COST_ROW(name, value, unit, source) {#name, value, unit, source},
constexpr Constant kTable[] = {LCE_COST_TABLE(LCE_COST_ROW)};      // (the values as compiled: what dump_cost_table prints)
#undef LCE_COST_ROW
#ifdef LCE_COST_TUNABLE
#define LCE_COST_PTR(name, value, unit, source) &name,
double* const kLive[] = {LCE_COST_TABLE(LCE_COST_PTR)};
#undef LCE_COST_PTR
#endif

inline double epilogue_floor_us(int dst) { return dst == LCE_HIP_F32 ? kStEpiFloorF32 : dst == LCE_HIP_I8 ? kStEpiFloorI8 : kStEpiFloorBp; }
// what a block step of MFMAs costs over its bare matrix time, by output type (the woven epilogue, and the clock the power manager
// grants: the more the launch writes, the lower), and how much more on launches long enough to reach the sustained state
inline double step_factor(int dst, int64_t usteps, bool ksplit) {
  const double base = ksplit ? (dst == LCE_HIP_F32 ? kStStepKsplitF32 : dst == LCE_HIP_I8 ? kStStepKsplitI8 : kStStepKsplitBp)
                             : (dst == LCE_HIP_F32 ? kStStepF32 : dst == LCE_HIP_I8 ? kStStepI8 : kStStepBp);
  const double sustained = dst == LCE_HIP_BITPACKED ? kStSustainedBp : kStSustained;
  return base * (1.0 + sustained * std::min(1.0, (double)usteps / kStSustainedSteps));
}
inline double ramp(double x) { return std::min(1.0, std::max(0.0, x)); }

}  // namespace cost

#ifdef LCE_COST_TUNABLE
int cost_constant_count() { return (int)(sizeof cost::kTable / sizeof cost::kTable[0]); }
const char* cost_constant_name(int i) { return i >= 0 && i < cost_constant_count() ? cost::kTable[i].name : nullptr; }
double get_cost_constant(int i) { return i >= 0 && i < cost_constant_count() ? *cost::kLive[i] : 0.0; }
bool set_cost_constant(int i, double v) {
  if (i < 0 || i >= cost_constant_count()) return false;
  *cost::kLive[i] = v;
  return true;
}
#endif

void dump_cost_table(FILE* f) {
  for (const cost::Constant& c : cost::kTable) fprintf(f, "[lce plan cost] %-22s %12g %-9s %s\n", c.name, c.value, c.unit, c.source);
}

// The weight-stationary streaming kernel as plan_stream has just planned it (st_* fields) for launches of batch_chunk images.
double estimate_stream_us(const HostPlan& p, int batch_chunk) {
  using namespace cost;
  const int kch = stream_chunks(p.d), dst = p.d.dst_type;
  const bool ksplit = stream_ksplit(p);
  const double bank_kib = ksplit ? 288.0 : 72.0 * kch;                       // 4 waves x K-steps x 2 fragments x 1 KiB
  // (cus: the blocks resident at once -- two per CU where the launch was planned that way)
  const int64_t blocks = (int64_t)p.st_gx * p.st_ny, cus = (int64_t)std::max(1, p.num_cus) * std::max(1, p.st_occ);
  const double bank_us = std::max(bank_kib * 1024.0 / kVmemBytesPerClk / kCyclesPerUs,
                                  (double)std::min(blocks, cus) * bank_kib * 1024.0 / kL2BytesPerUs);
  const double prologue_us = kStPrologueUs + bank_us + (ksplit ? kStPrologueKsplitUs : 0.0);
  const double mfma_us = (ksplit ? 72.0 : 18.0 * kch) * kMfmaCycles / kCyclesPerUs;    // MFMAs per block step and wave
  const int64_t usteps = ((int64_t)p.st_nq + (1 << p.st_pph_log) - 1) >> p.st_pph_log;
  const double step_us = (std::max(step_factor(dst, usteps, ksplit) * mfma_us, epilogue_floor_us(dst)) + (ksplit ? kStStepKsplitUs : 0.0)) *
                         (p.st_occ > 1 && blocks > cus / 2 ? kStTwoBlocksStepMul : 1.0);
  const int64_t rounds = (blocks + cus - 1) / cus;
  // the ring's production: one quota of 256 items rides free per tile step, the rest is handled out of line
  const double quotas = (double)p.st_spb * p.st_srs * p.st_ipr / 256.0, tiles = std::max<double>(1.0, (double)((usteps + 3) / 4));
  const double production_us = std::max(0.0, quotas - tiles) * kStQuotaUs;
  // a segment whose pixels do not fill its last 32-pixel block stores that block out of line, row by row
  const bool ragged = !p.st_flat && (p.st_rs * p.st_wso) % 32 != 0;
  const double partial_us = ragged ? (ksplit ? kStPartialKsplitUs : kStPartialUs) * p.st_spb / (double)(1 << p.st_pph_log) : 0.0;
  const double block_us = prologue_us + usteps * step_us + production_us + partial_us + kStBlockTailUs;
  const double compute_us = kLaunchUs + rounds * block_us;
  // nothing is written before the first block step is over; from then on the chip's write rate for the pattern bounds the launch
  // (interleaved runs: the launch writes gstr consecutive segments at a time -- the more compact that window, the closer to the
  //  rate of one sequential stream; whole images per block: 256 streams megabytes apart)
  const double out_bytes = (double)out_bytes_of(p, batch_chunk);
  double bytes_per_us = kStoreBytesPerUs;
  if (dst == LCE_HIP_F32) {
    bytes_per_us = kStoreF32BytesPerUs + kStoreF32SmallBonus * ramp((kStoreF32LargeMb - out_bytes / 1.0e6) / kStoreF32RampMb);
    // the window the launch's blocks write into at any moment: gstr consecutive segments (interleaved runs), else every block's
    // own run -- the whole output
    const double window = p.st_gstr > 1 ? (double)p.st_gstr * p.st_rs * p.st_wso * stream_row_bytes(p) : out_bytes;
    bytes_per_us += kStoreWindowBonus * ramp((kStoreWindowFull - window) / kStoreWindowRamp);
  }
  const double store_us = kLaunchUs + prologue_us + step_us + out_bytes / bytes_per_us;
  if (p.dbg_level >= 2)
    fprintf(stderr,
            "[lce plan]   rows %d il %d: blocks %lld usteps %lld prologue %.2f step %.2f production %.2f partial %.2f compute %.2f "
            "store %.2f\n",
            p.st_rs, p.st_gstr > 1, (long long)blocks, (long long)usteps, prologue_us, step_us, production_us, partial_us, compute_us,
            store_us);
  // An instance wider than the layer (129..192 channels on the 256-channel bank, 257..448 on the 512-channel one): the expansion
  // takes the general path (word-by-word loads, partial planes) and the K loop multiplies the padding
  const bool padded = p.d.channels_in != 64 * kch && p.d.channels_in > 32 * kch;
  const double us = std::max(compute_us, store_us);
  if (!padded) return us;
  // (bitpacked output: nothing but the K loop and the expansion is left of a block step, so the general expansion path shows in full)
  const double wide = ksplit ? kStPaddedKsplitMul * us + kStPaddedKsplitUs : kStPaddedMul * us + kStPaddedUs;
  return dst == LCE_HIP_BITPACKED ? kStPaddedBpMul * wide : wide;
}

// The weight-streaming kernel as plan_wstream has just planned it (ws_* fields).
double estimate_wstream_us(const HostPlan& p, int batch_chunk) {
  using namespace cost;
  const int kch = stream_chunks(p.d), ks = 9 * kch, cus = std::max(1, p.num_cus);
  const int groups = ceil_div(batch_chunk, p.ws_ipb);
  // blocks are dispatched in index order (part-major), round-robin over the CUs: pixel blocks on the busiest CU
  std::vector<int64_t> load(cus, 0);
  int64_t b = 0, worst = 0;
  int last_nb = 0;
  for (int y = 0; y < p.ws_ny; ++y)
    for (int part = 0; part < p.ws_parts; ++part)
      for (int g = 0; g < groups; ++g, ++b) {
        const int nb = p.ws_nq / p.ws_parts + (part < p.ws_nq % p.ws_parts ? 1 : 0);
        load[b % cus] += nb;
        worst = std::max(worst, load[b % cus]);
        last_nb = nb;
      }
  const int64_t rounds = (b + (int64_t)cus * p.ws_occupancy - 1) / ((int64_t)cus * p.ws_occupancy);
  const double crowd = std::min(1.0, (double)b / (2.0 * cus));                 // 0: blocks alone on their CUs ... 1: two per CU
  const int items_per_lane = ceil_div(p.ws_ipb * p.ws_hp * p.ws_wp * p.ws_qg, 256);
  const double prologue_us = (kWsPrologueCycles + kWsItemCycles * items_per_lane) / kCyclesPerUs + kWsCrowdPrologueUs * crowd;
  // the K loops at the matrix cores' rate -- or at the rate the L2s deliver the launch's weight streams (every block pulls the
  // whole image of its 256 channels: ks x 8 KiB) -- or, with one or two pixel blocks per block, at the latency of the weight loads
  // (kWsPrefetch K-steps in flight)
  const double mfma_rate_us = (double)worst * ks * 2 * (kMfmaCycles + kWsMfmaExtraCycles + kWsMfmaCrowdCycles * crowd) / kCyclesPerUs;
  const double l2_rate_us = (double)b * ks * 8192.0 / kL2BytesPerUs;
  const double kloop_us = std::max(std::max(mfma_rate_us, l2_rate_us), kWsKstepLatencyUs * ks);
  // K-major: a block's outputs all come at its end.  int8 / bitpacked: the transform of its pixel blocks (beside the co-resident
  // block's); float: the stores of (most of) the launch, which the chip writes at its own rate behind the K loops
  double tail_us;
  if (p.d.dst_type == LCE_HIP_F32)
    tail_us = std::max(kWsTailF32PerBlockUs * last_nb, kWsTailF32StoreShare * (double)out_bytes_of(p, batch_chunk) / kStoreF32BytesPerUs);
  else
    tail_us = (p.d.dst_type == LCE_HIP_I8 ? kWsTailI8Us + kWsTailI8CrowdUs * crowd : kWsTailBpUs) * last_nb;
  return kLaunchUs + rounds * prologue_us + (p.ws_occupancy < 2 ? kWsOneBlockPerCu : 1.0) * kloop_us + tail_us;
}

// The block GEMM (direct or workspace variant, whichever select_kernel would take): K-steps at the LDS port's rate, two blocks
// per CU that slow each other down, the output at the chip's write rate for block tiles.
double estimate_block_gemm_us(const HostPlan& p, int64_t pixels) {
  using namespace cost;
  const MfmaCfg c = choose_mfma_cfg(p, pixels);
  const int64_t blocks = ((pixels + c.bm() - 1) / c.bm()) * ceil_div(p.d.channels_out, c.bn()), cus = std::max(1, p.num_cus);
  const int ks = p.d.filter_height * p.d.filter_width * ceil_div(p.d.channels_in / std::max(1, p.d.groups), 64);
  const double dst_f = p.d.dst_type == LCE_HIP_F32 ? 1.0 : p.d.dst_type == LCE_HIP_I8 ? kGemmI8Mul : kGemmBpMul;
  const double area = (double)c.bm() * c.bn() / (128.0 * 128.0);
  // a block alone on its CU runs at the latency of its K-steps (a smaller tile's are no shorter); a launch of many rounds at the
  // matrix cores' / the LDS port's throughput, where the cheaper epilogues show
  const double alone_us = (kGemmAloneFixedUs + (kGemmAloneSetupUs + kGemmAloneKstepUs * ks) * std::max(1.0, area)) *
                          (p.d.dst_type == LCE_HIP_BITPACKED ? kGemmAloneBpMul : 1.0);
  const double round_us = (kGemmRoundFixedUs + (kGemmRoundSetupUs + kGemmRoundKstepUs * ks) * area) * dst_f;
  double compute_us;
  if (blocks <= cus) compute_us = alone_us;
  else if (blocks <= 2 * cus) compute_us = alone_us * (1.0 + kGemmPartialRoundMul * (double)(blocks - cus) / cus);
  else compute_us = std::max(kGemmSharedCuMul * alone_us, kGemmSharedCuMul * round_us * ((double)blocks / (2.0 * cus) + kGemmRoundsOffset));
  const int batch_chunk = (int)std::max<int64_t>(1, pixels / std::max<int64_t>(1, (int64_t)p.out_h * p.out_w));
  const double store_us = kGemmStoreFixedUs + (double)out_bytes_of(p, batch_chunk) / kGemmStoreBytesPerUs;
  return kLaunchUs + std::max(compute_us, store_us);
}

}  // namespace lce
