// Weight-stationary streaming variant of the matrix-core engine ("engine=stream").
//
// Why it exists: the block GEMM of lce_kernels_mfma.h re-streams a tile's weights
// through an LDS ring (a barrier, a counted wait and two LDS-DMA issues per K-step) and runs every tile's
// prologue (halo expansion) and epilogue (transform, transpose, stores) beside ANOTHER block's K loop -- two
// waves per SIMD that share one issue port; the matrix pipe ends up 46 % busy on the BASELINE layer.  Here:
//
//   * ONE wave per SIMD owns 64 output channels and keeps ALL their FP4 weights in registers for its whole
//     life (KH*KW*KCH K-steps x 2 channel tiles x 4 registers: 288 of the 512 for a 3x3x256 filter, most of
//     them in the accumulator half of the register file, which MFMA reads directly).  The K loop has no weight
//     traffic, no barrier and no per-step wait: a K-step is one ds_read_b128 (the A fragment, at an immediate
//     offset from a per-lane tap-row address) and two MFMAs.
//   * The block is PERSISTENT: it walks a run of (image, output-row range) segments as a stream of 32-pixel
//     blocks.  The bitpacked input rows are expanded to FP4 ONCE each into a ring of row slots in LDS (a rolling
//     halo), by the same four waves, as filler between their MFMAs.  Everything that is pure index arithmetic
//     is done by the planner and uploaded as tables (StreamArgs): `sched` says how many 16-byte items must be
//     resident before each tile step (four block steps, one block barrier), `ctx` holds, per pixel block and
//     lane, the three tap-row LDS addresses and the output offset.
//   * The output transform of pixel block u-1 (transform, transpose through a wave-private LDS scratch,
//     16-byte row stores) is woven, unit by unit, between the MFMAs of pixel block u, which accumulate into
//     the other of two accumulator sets.
//
// One wave per SIMD means one instruction issue slot per ~4 cycles for EVERYTHING (scalar instructions and waits
// included), 8 per MFMA: the kernel is written for instruction count -- packed-f32 transform, fragment reads in
// groups of four behind one counted wait, the first MFMA of a block reads a constant K_bt tile instead of
// re-arming accumulators, stores take their row offsets as scalar offsets, no per-block scalar arithmetic.
//
// Same arithmetic as the block GEMM -- the planner's FP4 weight image (pack_for_mfma: negated,
// [K-step][K-half][Npad][16 B]), accumulators start at K_bt and end as 2 * popcount-accumulator
// (output_transform.h:62-91), float transform with two roundings (:99-106), int8 round-half-away + saturate
// (:31-44), bitpacked compare (:160-168) -- so results are bit-identical to it and to the oracle.
#pragma once
#include <lce_device_intrinsics.h>
#include "lce_kernel_args.h"
#include "lce_kernels.h"
#include "lce_kernels_mfma.h"
#include "lce_stream_stamps.h"

namespace lce {

// Dword DD of fp4_of_full_word (lce_kernels_mfma.h): byte DD of the word as eight FP4 codes, 4 VALU.
template <int DD>
LCE_DEVICE uint32_t fp4_full_dword(uint32_t word) {
  constexpr uint32_t sel = DD == 0 ? 0x04000400u : DD == 1 ? 0x05010501u : DD == 2 ? 0x06020602u : 0x07030703u;
  const uint32_t u = pk_lshr_b16<0, 4>(perm_b32(word >> 2, word, sel)) & 0x03030303u;
  return perm_b32(0u, 0xAAA22A22u, u);
}

// Units of an epilogue phase that belong to K-step i of n steps (an even spread of `units`).
constexpr int stream_unit_lo(int units, int i, int n) { return units * i / n; }

// FAST (compile-time): every padded input word exists in full and padding is +1 (Cin == 64 * KCH, no exact
// SAME-zero border) -- the BASELINE layers -- so neither the loads nor the expansion carry per-word conditions.
// CLAMP (float output): the transform's clamp is not the identity (a fused activation).
// SIGN (float / int8 output): the epilogue also writes the LceQuantize of the values it produces (the second output of
// lce_hip_bconv2d_run_dual: bit = value < bit_thr, as the block GEMM's epilogues do) -- the next binary layer of a
// chain reads 1/32 of the bytes and no separate quantize launch runs.
// KSPLIT (round 4; 512 input channels): a 64-channel wave cannot hold 9 x 8 K-steps of weights (576 registers), and
// 32-channel waves would each need their own A fragment per MFMA (4 KiB of LDS reads per 33 cycles per CU: the LDS port's
// whole rate).  Instead the K dimension is split over a PAIR of waves: waves w and w ^ 2 own the same 64 channels and the
// same pixel block, each holds the weights of one HALF of the input channels (KCH / 2 chunks per tap: 288 registers) and
// accumulates a partial sum; before the epilogue the pair swaps halves through LDS -- each wave sends the 32-channel tile
// it does not finalise (4 ds_write_b128 into the partner's inbox), a block barrier, 4 ds_read_b128 + 16 adds -- and then
// transforms and stores ITS 32 channels.  A-fragment traffic stays at one 1-KiB read per two MFMAs.  So that all register
// indices are static, a wave's LOCAL tile 0 is the one it finalises: the K-half-1 wave holds the slice's two channel
// tiles in swapped order.  Block = 2 channel slices x 2 K-halves, one pixel block per block step.
// STRIPS (round 4; wide images): a segment is a run of output rows of ONE column strip of an image (StreamArgs), so that a ring
// row is a strip's width + halo instead of the whole padded row (224 x 144 B x 9-12 slots do not fit LDS).  Only the production's
// address arithmetic and the block's first output pixel differ; a strip is a multiple of 32 columns, so a pixel block never wraps.
// I8F (int8 output): the transform is ONE fma and the rounding floor(x + 0.5), one instruction each per value (lce_kernels.h, pack8_i8_clamped):
// selected by the planner only where that equals the reference's two roundings + round-half-away on every value the plan can produce.
// TWO BLOCKS PER CU (round 6; bitpacked output on the 64-input-channel bank): that instance needs 229 registers and no epilogue scratch, so
// two blocks fit a CU -- two waves per SIMD, each filling the other's stalls (a lone wave's block step takes 2.7x its MFMAs' time there:
// compares, lane writes, waits).  The planner gives such a launch twice the blocks where both blocks' LDS fit (lce_plan_stream.cpp);
// 56x56x64 -> 64 at batch 256: 22.4 -> 17.5 us, profiles/r06/occ2_potential.txt.  (int8 / float would need their 32 KiB of transpose
// scratch out of the way first: the same experiment, 64 -> 256 channels where the ring is small, gained 12 % for int8.)
constexpr int stream_blocks_per_cu(int dst, int kch, bool ksplit, bool strips) {
  return dst == kDstBitpacked && kch == 1 && !ksplit && !strips ? 2 : 1;
}
template <int DST, int KH, int KW, int KCH, bool FAST, bool CLAMP, bool SIGN, bool KSPLIT = false, bool STRIPS = false, bool I8F = false>
LCE_KERNEL void __launch_bounds__(256, stream_blocks_per_cu(DST, KCH, KSPLIT, STRIPS))
bconv2d_stream(const StreamArgs G, const uint8_t* __restrict__ xin, const uint8_t* __restrict__ wq,
               const float* __restrict__ mul, const float* __restrict__ bias, const float* __restrict__ thrf,
               const uint32_t* __restrict__ tabs, void* __restrict__ out, uint32_t* __restrict__ sign_words) {
  static_assert(!SIGN || DST != kDstBitpacked, "a bitpacked-output plan already writes bits");
  constexpr int KCHW = KSPLIT ? KCH / 2 : KCH; // 64-channel chunks per tap that ONE wave multiplies
  // FLATC: instances whose planner may cut pixel blocks across a block's images (StreamArgs::flat).  Only the K-split instances:
  // the handful of per-step instructions it costs (a run that ends in mid-block marks its last block itself) are ~1 % of an L0
  // block step, and the layers that gain from it -- 7x7 images, several per block -- are the 512-channel ones.
  constexpr bool FLATC = KSPLIT;
  constexpr int KS = KH * KW * KCHW;           // K-steps of one pixel block (per wave)
  constexpr int PS = KCH * 32 + 16;            // LDS bytes per ring pixel (the +16 staggers the banks)
  static_assert(!KSPLIT || KCH % 2 == 0, "the K split halves the chunks of a tap");
  constexpr int GA = 4;                        // K-steps per fragment group (one counted wait per group)
  constexpr int NG = (KS + GA - 1) / GA;
  static_assert(KH == 3, "the context table holds three tap-row addresses");
  static_assert(KS >= 8 && KS * 8 <= 288, "the filter bank must fit the register file");

  const int tid = thread_idx_x();
  const int lane = tid & (kWave - 1), wave = uniform(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  // waves -> (64-channel slice, pixel phase): N >= 256: four slices of one pixel block; N = 128: two slices x two
  // pixel blocks; N = 64: four pixel blocks
  const int nslb = KSPLIT ? 2 : 4 >> G.pph_log;
  const int slice = wave & (nslb - 1), pp = KSPLIT ? 0 : wave >> (2 - G.pph_log);
  const int khalf = KSPLIT ? wave >> 1 : 0;                 // which half of the input channels this wave multiplies
  const int n0 = (block_idx_y() * nslb + slice) * 64;
  const bool slice_ok = n0 < G.Npad;
  const int nown = n0 + khalf * 32;                         // KSPLIT: first of the 32 channels this wave finalises

  uint8_t* const lds0 = lds_base();
  constexpr int SCW = KSPLIT ? 32 : 64;                    // floats per scratch row: the channels a wave transposes and stores
  // bytes of a wave's epilogue scratch (bitpacked output: ballots, no transpose -- no scratch; lce_plan.h, stream_lds_extra)
  constexpr int SCRB = DST == kDstBitpacked && !KSPLIT ? 0 : 32 * SCW * 4;
  float* const scratch = (float*)(lds0 + G.ring_bytes) + wave * (32 * SCW);   // [32 pixel rows][SCW channels]
  // where idle lanes' items go (an item of the 64-channel bank is two words: 32 bytes)
  constexpr uint32_t kDumpStride = DST == kDstBitpacked && KCH == 1 ? 32u : 64u;
  const uint32_t dump = (uint32_t)G.ring_bytes + 4u * (uint32_t)SCRB + (uint32_t)lane * kDumpStride;
  // KSPLIT: every wave's inbox for its partner's partial tile, two slots of 4 KiB (block parity), behind the dump area
  uint8_t* const inbox = lds0 + G.ring_bytes + 4 * SCRB + 4096 + wave * 8192;
  uint8_t* const outbox = lds0 + G.ring_bytes + 4 * SCRB + 4096 + (wave ^ 2) * 8192;

  LCE_SPH(0);
  // ---- this block's run of segments ----
  // (segments g0, g0 + GSTR, g0 + 2 GSTR, ... below S, at most SPB of them: consecutive ones with GSTR = 1, G0M = SPB; with
  //  GSTR = the launch's block count and G0M = 1 the blocks' runs interleave, StreamArgs)
  const int g0 = block_idx_x() * G.G0M;
  int nseg = G.S - g0;
  nseg = nseg <= 0 ? 0 : (int)fastdiv_nb((uint32_t)(nseg + G.GSTR - 1), G.div_gstr);
  nseg = nseg > G.SPB ? G.SPB : nseg;
  // pixel blocks of this block's stream: per segment, or (flat) cut from its segments' pixels laid end to end -- a short last
  // block's spare lanes then hold pixels past the launch, whose stores fall outside the output's buffer resource
  const int nblk_all = FLATC && G.flat ? (nseg * G.NPX + 31) >> 5 : nseg * G.PBS;
  const int nblk = slice_ok ? nblk_all : 0;            // ... that produce output
  const int usteps = (nblk_all + (1 << G.pph_log) - 1) >> G.pph_log;   // block steps (2^pph_log pixel blocks each)
  // flat: the pixel block at which this block's run ends in mid-block (-1: none, or the table says so itself), and the last
  // real row of that block -- ONE scalar each for the K loop, which has no scalar registers to spare
  // (kept per lane: the K loop has no scalar registers to spare)
  int dyn_last = FLATC && G.flat && ((nseg * G.NPX) & 31) != 0 ? nblk_all - 1 : -1;
  int dyn_lim = nseg * G.NPX - (nblk_all - 1) * 32 - 1;
  keep_in_vgpr(dyn_last);
  keep_in_vgpr(dyn_lim);
  const int ntile = (usteps + 3) >> 2;                 // tile steps (four block steps + one barrier)
  const uint32_t* const sched = tabs;
  const rsrc_t rtab = make_rsrc(tabs, G.tab_bytes);

  // ---- production: item e of the block's stream = 16 bytes (4 input words) of one input pixel ----
  //   e -> stream row s = e / IPR, pixel x, word group qg; s -> (local segment, row in segment) -> image row iy.
  // Rows outside the image (and images past the batch) are steered to an out-of-range offset and read 0 = the
  // "+1" padding word, exactly as in the direct variant.  Cut into twelve chunks of a handful of instructions each:
  // in the K loop a chunk rides behind one MFMA.
  const rsrc_t rin = make_rsrc(xin, G.in_bytes);
  struct Issue {
    uint32_t s, rem, x, gl, g, img, slot, t, strip;
    int c0, iy, ix, cseg;
    bool on, inside;
  };
  constexpr int kIssueChunks = 12;
  auto issue_chunk = [&](auto cc_, Issue& I, uint32_t e, uint32_t e_end, u32x4& wv, uint32_t& dst, int& meta) LCE_LAMBDA_INLINE {
    constexpr int c = decltype(cc_)::value;
    if constexpr (c == 0) {
      I.on = e < e_end;
      I.s = fastdiv_nb(e, G.div_ipr);
    } else if constexpr (c == 1) {
      I.rem = e - I.s * (uint32_t)G.IPR;
      I.x = fastdiv_nb(I.rem, G.div_qg);
    } else if constexpr (c == 2) {
      I.c0 = (int)(I.rem - I.x * (uint32_t)G.QG) * 4;
      I.gl = fastdiv_nb(I.s, G.div_srs);
    } else if constexpr (c == 3) {
      I.g = (uint32_t)g0 + I.gl * (uint32_t)G.GSTR;
      I.img = fastdiv_nb(I.g, G.div_spi);
    } else if constexpr (c == 4) {
      I.cseg = (int)(I.g - I.img * (uint32_t)G.SPI);
      I.iy = (int)(I.s - I.gl * (uint32_t)G.SRS) - G.PH;
      if constexpr (STRIPS) I.strip = fastdiv_nb((uint32_t)I.cseg, G.div_rseg);      // segment in image = (strip, row segment)
    } else if constexpr (c == 5) {
      if constexpr (STRIPS) {
        I.cseg -= (int)I.strip * G.RSEG;
        I.ix = (int)I.x + (int)I.strip * (G.WSo * G.SW) - G.XS0;                      // the strip's column in the image
        I.iy += I.cseg * (G.RS * G.SH);
        I.inside = I.on && (uint32_t)I.iy < (uint32_t)G.H && (uint32_t)I.ix < (uint32_t)G.W && I.img < (uint32_t)G.B;
      } else {
        I.iy += I.cseg * (G.RS * G.SH);
        I.inside = I.on && (uint32_t)I.iy < (uint32_t)G.H && I.img < (uint32_t)G.B;
      }
    } else if constexpr (c == 6) {
      I.slot = I.s - fastdiv_nb(I.s, G.div_r) * (uint32_t)G.R;
    } else if constexpr (c == 7) {
      I.t = I.slot * (uint32_t)G.pitch + ((uint32_t)G.PW + I.x) * (uint32_t)PS;
    } else if constexpr (c == 8) {
      dst = I.on ? I.t + (uint32_t)I.c0 * 16u : dump;
      meta = I.c0 | (I.inside ? 0x100 : 0);
    } else if constexpr (c == 9) {
      I.t = (uint32_t)((int)I.img * G.H + I.iy) * (uint32_t)G.W;
    } else if constexpr (c == 10) {
      I.t = (I.t + (STRIPS ? (uint32_t)I.ix : I.x)) * (uint32_t)G.Cw * 4u + (uint32_t)I.c0 * 4u;
    } else {
      const uint32_t off = I.t;
      if constexpr (FAST && KCH == 1) {
        const u32x2 v2 = buf_load(rin, I.inside ? off : kOobOffset, (u32x2*)nullptr);
        wv[0] = v2[0]; wv[1] = v2[1]; wv[2] = 0u; wv[3] = 0u;
      } else if (FAST || (G.Cw & 3) == 0) {
        wv = buf_load(rin, I.inside ? off : kOobOffset, (u32x4*)nullptr);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[q] = buf_load(rin, I.inside && I.c0 + q < G.Cw ? off + 4u * q : kOobOffset, (uint32_t*)nullptr);
      }
    }
  };
  auto item_issue = [&](uint32_t e, uint32_t e_end, u32x4& wv, uint32_t& dst, int& meta) LCE_LAMBDA_INLINE {
    Issue I;
    issue_chunk(IntC<0>{}, I, e, e_end, wv, dst, meta); issue_chunk(IntC<1>{}, I, e, e_end, wv, dst, meta);
    issue_chunk(IntC<2>{}, I, e, e_end, wv, dst, meta); issue_chunk(IntC<3>{}, I, e, e_end, wv, dst, meta);
    issue_chunk(IntC<4>{}, I, e, e_end, wv, dst, meta); issue_chunk(IntC<5>{}, I, e, e_end, wv, dst, meta);
    issue_chunk(IntC<6>{}, I, e, e_end, wv, dst, meta); issue_chunk(IntC<7>{}, I, e, e_end, wv, dst, meta);
    issue_chunk(IntC<8>{}, I, e, e_end, wv, dst, meta); issue_chunk(IntC<9>{}, I, e, e_end, wv, dst, meta);
    issue_chunk(IntC<10>{}, I, e, e_end, wv, dst, meta); issue_chunk(IntC<11>{}, I, e, e_end, wv, dst, meta);
    static_assert(kIssueChunks == 12, "calls above");
  };
  // word q of an item -> 16 bytes of FP4 in the ring
  auto item_write_word = [&](auto qc, const u32x4& wv, uint32_t dst, int meta) LCE_LAMBDA_INLINE {
    constexpr int q = decltype(qc)::value;
    if constexpr (q < (KCH * 2 < 4 ? KCH * 2 : 4)) {      // words past the padded channel count do not exist
      u32x4 v;
      if constexpr (FAST) {
        v = fp4_of_full_word(wv[q]);
      } else {
        const int cc = (meta & 0xff) + q;
        int valid = G.Cin - cc * 32;                      // channels of this word that exist
        valid = valid < 0 ? 0 : (valid > 32 ? 32 : valid);
        if (!(meta & 0x100) && G.zero_border) valid = 0;  // exact SAME-zero: 0 contributes 0
        v = valid == 32 ? fp4_of_full_word(wv[q]) : fp4_of_word(wv[q], valid);
        // (a word group that starts past the last plane cannot occur: QG = ceil(planes / 4))
        if (cc >= KCH * 2) return;
      }
      *(u32x4*)(lds0 + dst + q * 16) = v;
    }
  };
  auto item_write = [&](const u32x4& wv, uint32_t dst, int meta) LCE_LAMBDA_INLINE {
    item_write_word(IntC<0>{}, wv, dst, meta);
    item_write_word(IntC<1>{}, wv, dst, meta);
    item_write_word(IntC<2>{}, wv, dst, meta);
    item_write_word(IntC<3>{}, wv, dst, meta);
  };
  // ... and in the K loop dword by dword: chunk c = 4 * word + dword of item A (FAST: one output dword = 4 VALU;
  // otherwise the whole word rides in its last chunk)
  u32x4 pv = {0u, 0u, 0u, 0u};
  auto write_chunk = [&](auto cc_, const u32x4& wv, uint32_t dst, int meta) LCE_LAMBDA_INLINE {
    constexpr int c = decltype(cc_)::value, q = c >> 2, dd = c & 3;
    if constexpr (q < (KCH * 2 < 4 ? KCH * 2 : 4)) {
      if constexpr (FAST) {
        pv[dd] = fp4_full_dword<dd>(wv[q]);
        if constexpr (dd == 3) *(u32x4*)(lds0 + dst + q * 16) = pv;
      } else {
        if constexpr (dd == 3) item_write_word(IntC<q>{}, wv, dst, meta);
      }
    }
  };

  // the filter bank of this wave's 64 channels, resident for the life of the block.  Its 2 * KS 16-byte loads per lane keep
  // the CU's vector memory path busy for ~6 k cycles (295 KB per CU); a quarter goes out at once, the rest in small pieces
  // between the work that does not need them -- the per-channel constants, the ring's padding columns, the expansion of
  // the first rows -- so that work rides in the shadow of the bank's arrival instead of waiting in front of it or behind it.
  // Round 5, K-split instances (kBankPaced): the first rows' loads go out FIRST, the fragments of the first kBankAhead K-steps
  // behind them, and every K-step of the FIRST block step issues the two loads of K-step ks + kBankAhead behind its MFMAs -- the
  // bank arrives during the first block step instead of in front of it.  A lone wave does not overlap its own load issue with
  // its MFMAs (the step runs 3.8 k cycles of loads + 2 k of MFMAs, profiles/r05/stream_phases_paced.txt), so the block's life
  // does not change much: 7x7x512 -2...-3 % (three output types, two boxes), 14x14x256 +1...2 %, L0 +-0 -- adopted where it gains.
  u32x4 W[KS][2];
  const rsrc_t rw = make_rsrc(wq, G.w_bytes);
  // loads [a, b) of the bank's 2 * KS, numbered ks * 2 + j
  auto bank_loads = [&](auto ac, auto bc) LCE_LAMBDA_INLINE {
    constexpr int a = decltype(ac)::value, b = decltype(bc)::value;
#pragma unroll
    for (int i = a; i < b; ++i)
    {
      // (KSPLIT: the wave's K-step ks = (tap, kc) is the layer's K-step tap * KCH + khalf * KCHW + kc, and its local
      //  channel tile j is the slice's tile j ^ khalf)
      const int ksl = i >> 1, ksg = KSPLIT ? (ksl / KCHW) * KCH + khalf * KCHW + ksl % KCHW : ksl;
      const int jt = KSPLIT ? (i & 1) ^ khalf : (i & 1);
#ifdef LCE_ST_NOBANK    // timing ablation (results are wrong): the bank's loads fall off the end of their buffer -- zeros, no transfer
      W[i >> 1][i & 1] = buf_load(rw, slice_ok && ksg < 0 ? (uint32_t)jt : kOobOffset, (u32x4*)nullptr);
#else
      W[i >> 1][i & 1] = buf_load(rw, slice_ok ? (uint32_t)((ksg * 2 + half) * G.Npad + n0 + jt * 32 + l31) * 16u : kOobOffset, (u32x4*)nullptr);
#endif
    }
    sched_fence();
  };
  constexpr bool kBankPaced = KSPLIT;
  constexpr int kBankAhead = 12;                            // paced: K-steps of the bank in flight ahead of the first block step's MFMAs
  constexpr int kBankFirst = kBankPaced ? 2 * kBankAhead : KS / 2;          // the first run (unpaced: a quarter of the bank)
  constexpr int kBankRest = kBankPaced ? 0 : 2 * KS - kBankFirst;           // the rest: sixteen pieces, one behind each expanded word
  static_assert(kBankFirst <= 2 * KS, "the first run is part of the bank");
  auto bank_run = [&](auto rc) LCE_LAMBDA_INLINE { bank_loads(IntC<0>{}, IntC<kBankFirst>{}); };
  auto bank_piece = [&](auto pc) LCE_LAMBDA_INLINE {
    constexpr int p = decltype(pc)::value;
    if constexpr (!kBankPaced) bank_loads(IntC<kBankFirst + kBankRest * p / 16>{}, IntC<kBankFirst + kBankRest * (p + 1) / 16>{});
  };
  sched_fence();
  if constexpr (!kBankPaced) bank_run(IntC<0>{});
  // ---- prologue.  The bank's first run goes out before anything else (it needs nothing but the kernel arguments, and the
  // schedule's first entry -- a dependent scalar load -- is still on its way); the first rows' loads follow, AHEAD of the
  // rest of the bank: the memory counter retires in order, so behind all 72 loads per lane they could not be consumed
  // before the whole bank had arrived. ----
  const uint32_t need0 = kBankPaced ? G.need0 : sched[0];     // (paced: with the kernel arguments, no dependent load in front of the first rows')
  u32x4 wv0[4];
  uint32_t dv0[4];
  int mv0[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) item_issue((uint32_t)(k * 256 + tid), need0, wv0[k], dv0[k], mv0[k]);
  if constexpr (kBankPaced) bank_run(IntC<0>{});
  LCE_SPH(58);

  LCE_SPH(59);
  // per-channel constants of this lane's two channels; multiplier and bias twice each: the transform works on
  // register pairs (v_pk_mul_f32 / v_pk_add_f32, each element rounded twice as output_transform.h:105 does)
  float tj[2], uj[2];
  f32x2 mj[2], bj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = slice_ok ? n0 + (KSPLIT ? j ^ khalf : j) * 32 + l31 : 0;
    tj[j] = uj[j] = 0.0f;
    mj[j] = f32x2{0.0f, 0.0f};
    bj[j] = f32x2{0.0f, 0.0f};
    if constexpr (DST == kDstBitpacked) tj[j] = thrf[n];
    else { mj[j] = f32x2{mul[n], mul[n]}; bj[j] = f32x2{bias[n], bias[n]}; }
    if constexpr (DST == kDstInt8) { tj[j] = thrf[n]; uj[j] = thrf[G.Npad + n]; }
  }
  // the accumulators' start value K_bt as a constant C tile (the first MFMA of every pixel block reads it)
  f32x16 kbt = f32x16_fill(khalf ? 0.0f : G.a_bt);   // (KSPLIT: the pair's two partial sums add up to K_bt - <a, w>)
  pin(kbt);
  float cminv = G.cmin, cmaxv = G.cmax;     // per-lane copies: the scalar registers are for the loop's addressing
  keep_in_vgpr(cminv);
  keep_in_vgpr(cmaxv);

  // The ring's padding COLUMNS (pixels left of PW and right of PW + W in every row slot) are +1 codes, or zeros for
  // exact SAME-zero, for good: production only ever writes pixels PW .. PW + W - 1 of a slot (padding ROWS are
  // produced like any other row, from out-of-range loads).
  if constexpr (!STRIPS) {      // (a strip's ring row has no constant columns: its halo is produced like everything else)
    const uint32_t code = G.zero_border ? 0u : 0x22222222u;
    const u32x4 v = {code, code, code, code};
    const int padc = G.Wp - G.W;                               // padding pixels per slot
    constexpr int CH = KCH * 2;                                // 16-byte chunks per pixel
    for (int e = tid; e < G.R * padc * CH; e += 256) {
      const int c16 = e % CH, t = e / CH, pc = t % padc, slot = t / padc;
      const int px = pc < G.PW ? pc : pc + G.W;
      *(u32x4*)(lds0 + (size_t)slot * G.pitch + (size_t)px * PS + c16 * 16) = v;
    }
  }
  block_barrier_keep_vm();
  LCE_SPH(60);
  // everything tile step 0 needs, with a piece of the bank's remaining loads behind every expanded word
  sched_fence();
  auto item_write_with_bank = [&](auto kc, const u32x4& wv, uint32_t dst, int meta) LCE_LAMBDA_INLINE {
    constexpr int k = decltype(kc)::value;
    item_write_word(IntC<0>{}, wv, dst, meta);
    sched_fence();
    bank_piece(IntC<4 * k + 0>{});
    item_write_word(IntC<1>{}, wv, dst, meta);
    sched_fence();
    bank_piece(IntC<4 * k + 1>{});
    item_write_word(IntC<2>{}, wv, dst, meta);
    sched_fence();
    bank_piece(IntC<4 * k + 2>{});
    item_write_word(IntC<3>{}, wv, dst, meta);
    sched_fence();
    bank_piece(IntC<4 * k + 3>{});
  };
  item_write_with_bank(IntC<0>{}, wv0[0], dv0[0], mv0[0]);
  item_write_with_bank(IntC<1>{}, wv0[1], dv0[1], mv0[1]);
  item_write_with_bank(IntC<2>{}, wv0[2], dv0[2], mv0[2]);
  item_write_with_bank(IntC<3>{}, wv0[3], dv0[3], mv0[3]);
  for (uint32_t e0 = 4u * 256u; e0 < need0; e0 += 4u * 256u) {
#pragma unroll
    for (int k = 0; k < 4; ++k) item_issue(e0 + (uint32_t)(k * 256 + tid), need0, wv0[k], dv0[k], mv0[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) item_write(wv0[k], dv0[k], mv0[k]);
  }
  LCE_SPH(61);
  // The items tile step 0 writes (needed from tile step 1 on) are on their way.  Item A (the first 256 of a tile
  // step's quota) rides between the MFMAs; item B (the rest -- the planner's schedule makes it rare) is handled
  // out of line at the end of a tile step.
  // KSPLIT (a pixel is four items wide): a second woven item, C = the second 256 of a quota, rides in block steps 2 (expand)
  // and 3 (issue) exactly as A does in steps 0 and 1; B is then the third 256.
  constexpr uint32_t kItemB = KSPLIT ? 512u : 256u;        // where item B starts inside a tile step's quota
  u32x4 pwa, pwb = {0u, 0u, 0u, 0u}, pwc = {0u, 0u, 0u, 0u};
  uint32_t pda, pdb = dump, pdc = dump;
  int pma, pmb = 0, pmc = 0;
  bool have_b;
  {
    const uint32_t a = sched[0], b = sched[1];
    item_issue(a + (uint32_t)tid, b, pwa, pda, pma);
    if constexpr (KSPLIT) item_issue(a + (uint32_t)(256 + tid), b, pwc, pdc, pmc);
    have_b = b - a > kItemB;
    if (have_b) item_issue(a + kItemB + (uint32_t)tid, b, pwb, pdb, pmb);
  }
  // The filter bank's home: 256 accumulator registers hold the first 32 K-steps' fragments, the rest stay in VGPRs.
  // The hint is a USE of the loaded value (a counted wait).  Round 4: it no longer sits here, in front of the first MFMA
  // -- the FIRST block step below takes each K-step's two fragments as they arrive (the bank is 295 KB per CU and takes
  // ~6 k cycles to come in; the first pixel block's 72 MFMAs now run inside that time instead of behind it).
  auto bank_home = [&](auto ksc) LCE_LAMBDA_INLINE {
    constexpr int ks = decltype(ksc)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // (a 256-register instance takes no "a" constraint: the compiler would split its budget 128 / 128)
      if (ks < 32 && stream_blocks_per_cu(DST, KCH, KSPLIT, STRIPS) == 1) keep_in_agpr(W[ks][j]);
      else keep_in_vgpr(W[ks][j]);
    }
  };
  block_barrier_keep_vm();
  LCE_SPH(1);

  // ---- per-pixel-block context, from the planner's table: entry (q, lane) = {addr0, addr1, addr2, out offset} ----
  //   addr[fy] : LDS byte address of the lane's pixel in tap row fy (K-half included) at (fx, kc) = (0, 0)
  //   out      : byte offset of the lane's first output row, relative to the block's first output pixel, bit 31 set
  //              when the pixel block is the partial last one of its segment (its stores then go out of line)
  struct Ctx {
    u32x4 t;
    uint32_t s;            // SIGN: byte offset of the lane's pixel row in the sign-word tensor (same marker bit)
    // What the kernel adds to the table's entry -- kept apart and applied where t[3] / s are CONSUMED (a block step later):
    // touching the loaded values right behind the load would make the wave wait for it on the spot (the first build of
    // round 4 did: L0 bitpacked +5.5 %, float +1.2 %, profiles/r04/ab_r03_kernels.txt)
    uint32_t m;            // the partial-block marker of a flat run that ends in mid-block (else 0)
    uint32_t bo, bs;       // STRIPS: the segment's place in the output / in the sign words
  };
  const uint32_t row_bytes = DST == kDstBitpacked ? (uint32_t)G.Wout * 4u : (uint32_t)G.N * (DST == kDstInt8 ? 1u : 4u);
  const rsrc_t rout = make_rsrc(out, G.out_bytes);
  // this lane's share of a stored row, and the block's first output pixel
  // (KSPLIT: a wave stores its OWN 32 channels -- 8 lanes x 16 bytes of floats, 2 lanes x 16 bytes of int8, one word)
  constexpr int LPR = DST == kDstFloat ? (KSPLIT ? 8 : 16) : DST == kDstInt8 ? (KSPLIT ? 2 : 4) : 64;   // lanes per stored pixel row
  constexpr int RPI = 64 / LPR;                                                                          // pixel rows per store instruction
  uint32_t chan_off;
  if constexpr (DST == kDstFloat) {
    const int n = (KSPLIT ? nown : n0) + (lane & (LPR - 1)) * 4;      // 16 lanes x 16 bytes = a pixel's 64 channels
    chan_off = n < G.N ? (uint32_t)n * 4u : kOobOffset;
  } else if constexpr (DST == kDstInt8) {
    const int n = (KSPLIT ? nown : n0) + (lane & (LPR - 1)) * 16;     // 4 lanes x 16 bytes = a pixel's 64 channels
    chan_off = n < G.N ? (uint32_t)n : kOobOffset;
  } else if constexpr (KSPLIT) {
    chan_off = lane < 32 && (nown >> 5) < G.Wout ? (uint32_t)(nown >> 5) * 4u : kOobOffset;   // lane p owns pixel row p: its one word
  } else {
    chan_off = lane < 32 ? (uint32_t)(n0 >> 5) * 4u : kOobOffset;      // lane p owns pixel row p: two words
  }
  // the block's first output pixel: segments are RS whole rows apart.  STRIPS: a run may pass from one strip (or image) to the
  // next, so every local segment's first pixel is worked out here, once, into a small LDS table (segbase: byte offsets into the
  // output and into the sign words); the context table's offsets are then relative to the pixel block's SEGMENT
  const uint32_t first_px = STRIPS ? 0u : (uint32_t)g0 * (uint32_t)(G.RS * G.OW);
  uint32_t* const segbase = (uint32_t*)(lds0 + G.ring_bytes + 4 * SCRB + 4096);      // [SPB][2]
  if constexpr (STRIPS) {
    for (int gl = tid; gl < G.SPB; gl += 256) {
      const uint32_t g = (uint32_t)(g0 + gl * G.GSTR);
      const uint32_t img = fastdiv_nb(g, G.div_spi), rem = g - img * (uint32_t)G.SPI;
      const uint32_t strip = fastdiv_nb(rem, G.div_rseg), rowseg = rem - strip * (uint32_t)G.RSEG;
      const uint32_t px = (img * (uint32_t)G.OH + rowseg * (uint32_t)G.RS) * (uint32_t)G.OW + strip * (uint32_t)G.WSo;
      segbase[2 * gl + 0] = px * row_bytes;
      segbase[2 * gl + 1] = px * (uint32_t)G.Wout * 4u;
    }
    block_barrier_keep_vm();
  }
  chan_off += first_px * row_bytes;      // < 2^31 with the whole output
  const int nq = G.NQ;
  auto load_ctx = [&](int u, Ctx& cx) LCE_LAMBDA_INLINE {
    int q = (u << G.pph_log) + pp;
    q = q < nq ? q : nq - 1;
    cx.t = buf_load(rtab, (uint32_t)G.tab_ctx + (uint32_t)(q * 64 + lane) * 16u, (u32x4*)nullptr);
    if constexpr (SIGN) cx.s = buf_load(rtab, (uint32_t)G.tab_sgn + (uint32_t)(q * 64 + lane) * 4u, (uint32_t*)nullptr);
    else cx.s = 0u;
    // flat: the last block of a run that is SHORTER than the planned one is partial where the table does not say so (its
    // rows past the run are pixels of images beyond the launch; a store's uniform row offset is not range-checked): it
    // gets the table's marker here
    cx.m = FLATC && (u << G.pph_log) + pp == dyn_last ? 0x80000000u : 0u;
    cx.bo = cx.bs = 0u;
    if constexpr (STRIPS) {            // the segment's place in the output (a strip run may cross strips and images)
      const int gl = (int)tabs[G.tab_seg / 4 + q];
      cx.bo = segbase[2 * gl + 0];
      if constexpr (SIGN) cx.bs = segbase[2 * gl + 1];
    }
  };
  // second output: lane p (< 32) owns pixel row p's two sign words of this wave's 64 channels
  const rsrc_t rsgn = make_rsrc(sign_words, SIGN ? G.sign_bytes : 0u);
  const uint32_t sign_chan_off = lane < 32 && (!KSPLIT || (nown >> 5) < G.Wout)
                                     ? (uint32_t)((KSPLIT ? nown : n0) >> 5) * 4u + first_px * (uint32_t)G.Wout * 4u
                                     : kOobOffset;
  auto sign_base = [&](int u, const Ctx& cx) LCE_LAMBDA_INLINE -> uint32_t {
    const int q = (u << G.pph_log) + pp;
    return sat_add_u32((cx.s | cx.m) + cx.bs, q < nblk ? sign_chan_off : kOobOffset);
  };
  // per lane: channels past the last one never set a bit (padding bits of the last word are 0, bitpack.h:248-308)
  float bit_thrv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bit_thrv[j] = n0 + (KSPLIT ? j ^ khalf : j) * 32 + l31 < G.N ? G.bit_thr : -__builtin_inff();
  // the store offset of a context: out of range for pixel blocks past the stream and for partial blocks
  auto out_base = [&](int u, const Ctx& cx) LCE_LAMBDA_INLINE -> uint32_t {
    const int q = (u << G.pph_log) + pp;
    return sat_add_u32((cx.t[3] | cx.m) + cx.bo, q < nblk ? chan_off : kOobOffset);   // saturating: two markers must not wrap around
  };

  // ---- the epilogue of one pixel block, cut into units that ride between the next block's MFMAs ----
  // float / int8, phase A (16 units): two accumulator registers of tile j = t / 8 -> transformed -> scratch
  //   (register r of a tile holds pixel rows (r & 3) + 8 * (r >> 2) + 4 * half of channel column l31)
  // phase B (8 units): ds_read_b128 of the transposed tile; phase C (8 units): [int8: round + pack] + stores
  // float(x) * mul + bias, two roundings, on a register pair.  Packed-f32 instructions share the matrix pipe's data
  // path: beside an MFMA stream each costs a dozen cycles more than its issue slot, so the pair is two scalar
  // multiplies and two adds (LCE_STREAM_PK_F32: the packed form, for comparison).
  auto transform2 = [&](f32x2 x, f32x2 m, f32x2 b) LCE_LAMBDA_INLINE -> f32x2 {
#ifdef LCE_STREAM_PK_F32
    return mul_then_add_pk(x, m, b);
#else
    f32x2 y = {mul_then_add(x[0], m[0], b[0]), mul_then_add(x[1], m[1], b[1])};
    return y;
#endif
  };
  f32x4 yb[8];
  u32x4 pk[2];
  uint32_t bw[2] = {0u, 0u};
  // (KSPLIT: only local tile 0, the wave's own 32 channels -- 8 units for float / int8, 16 single-word units for bits;
  //  the scratch is [32 pixel rows][32 channels])
  constexpr int NUA = (KSPLIT && DST != kDstBitpacked) ? 8 : 16;   // phase A units
  // Ballots -> lanes.  A v_writelane must not read an SGPR a VALU compare wrote less than 4 wait states ago (settle_ballots: an s_nop 4,
  // ~8 cycles of a lone wave, per unit).  The K loop writes a unit's ballots one unit LATER instead (PIPE), no padding: unit t's lane
  // writes of unit t - 1 are held behind unit t's own compares and transform (hold_until) and behind unit t - 1's lane writes of unit
  // t - 2 (the chain through bw[]), so at least six instructions -- with one unit per K-step also two MFMAs -- lie in between.
  // (Float kernels with several units per K-step -- 64 / 128 input channels -- keep the padded form: store-bound, they measured equal
  //  or 3 % slower with it, profiles/r04/ballots_one_kstep_late.txt.)
  constexpr bool kTight = SIGN && !FAST && KCH >= 4;      // instances at the register file's limit: phase B's reads ride in phase C
  constexpr bool kPipeBallots = KSPLIT || KS * 4 / 9 >= 16 || DST != kDstFloat;
  unsigned long long pend[2] = {0ull, 0ull};
  // the lanes of unit T's ballots (pend as unit T left it)
  auto flush_ballots = [&](auto Tc) LCE_LAMBDA_INLINE {
    constexpr int T = decltype(Tc)::value;
    if constexpr (DST == kDstBitpacked) {
      constexpr int q = (T & 3) + 8 * (T >> 2);
#pragma unroll
      for (int j = 0; j < (KSPLIT ? 1 : 2); ++j) {
        bw[j] = write_lane_settled<q>((uint32_t)pend[j], bw[j]);
        bw[j] = write_lane_settled<q + 4>((uint32_t)(pend[j] >> 32), bw[j]);
      }
    } else {
      constexpr int j = KSPLIT ? 0 : T >> 3, r0 = (T & 7) * 2, q0 = (r0 & 3) + 8 * (r0 >> 2);
      bw[j] = write_lane_settled<q0>((uint32_t)pend[0], bw[j]);
      bw[j] = write_lane_settled<q0 + 4>((uint32_t)(pend[0] >> 32), bw[j]);
      bw[j] = write_lane_settled<q0 + 1>((uint32_t)pend[1], bw[j]);
      bw[j] = write_lane_settled<q0 + 5>((uint32_t)(pend[1] >> 32), bw[j]);
    }
  };
  auto epi_a = [&](auto tc, f32x16 (&acc)[2], auto pipe_) LCE_LAMBDA_INLINE {
    constexpr int t = decltype(tc)::value;
    constexpr bool PIPE = decltype(pipe_)::value != 0 && kPipeBallots;
    if constexpr (DST == kDstBitpacked && KSPLIT) {
      constexpr int r = t, q = (r & 3) + 8 * (r >> 2);
      unsigned long long bits[1];
      bits[0] = wave_ballot(acc[0][r] > tj[0]);
      if constexpr (PIPE) {
        if constexpr (t > 0) { hold_until(pend[0], bits[0], bits[0]); flush_ballots(IntC<(t > 0 ? t - 1 : 0)>{}); }
        pend[0] = bits[0];
      } else {
        settle_ballots(bits);
        bw[0] = write_lane_settled<q>((uint32_t)bits[0], bw[0]);
        bw[0] = write_lane_settled<q + 4>((uint32_t)(bits[0] >> 32), bw[0]);
      }
    } else if constexpr (DST == kDstBitpacked) {
      // unit t = register r: the 32 channel bits of rows q and q + 4, dropped into the lanes that will store them
      constexpr int r = t, q = (r & 3) + 8 * (r >> 2);
      unsigned long long bits[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bits[j] = wave_ballot(acc[j][r] > tj[j]);
      if constexpr (PIPE) {
        if constexpr (t > 0) {
          hold_until(pend[0], bits[0], bits[1]);
          hold_until(pend[1], bits[0], bits[1]);
          flush_ballots(IntC<(t > 0 ? t - 1 : 0)>{});
        }
        pend[0] = bits[0];
        pend[1] = bits[1];
      } else {
        settle_ballots(bits);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bw[j] = write_lane_settled<q>((uint32_t)bits[j], bw[j]);
          bw[j] = write_lane_settled<q + 4>((uint32_t)(bits[j] >> 32), bw[j]);
        }
      }
    } else {
      constexpr int j = KSPLIT ? 0 : t >> 3, r0 = (t & 7) * 2;          // registers r0, r0 + 1: pixel rows row0, row0 + 1
      const int row0 = (r0 & 3) + 8 * (r0 >> 2) + 4 * half;
      f32x2 y;
      if constexpr (DST == kDstFloat) {
        f32x2 x = {acc[j][r0], acc[j][r0 + 1]};
        if constexpr (CLAMP) { x[0] = med3(x[0], cminv, cmaxv); x[1] = med3(x[1], cminv, cmaxv); }
        y = transform2(x, mj[j], bj[j]);
      } else {
        const f32x2 x = {acc[j][r0], acc[j][r0 + 1]};
        if constexpr (I8F) { y[0] = fma1(x[0], mj[j][0], bj[j][0]); y[1] = fma1(x[1], mj[j][1], bj[j][1]); }   // one rounding: proven per plan
        else y = transform2(x, mj[j], bj[j]);
        y[0] = med3(y[0], tj[j], uj[j]);                   // one clamp: see lce_kernels_pointwise.h
        y[1] = med3(y[1], tj[j], uj[j]);
      }
      scratch[row0 * SCW + j * 32 + l31] = y[0];
      scratch[(row0 + 1) * SCW + j * 32 + l31] = y[1];
      if constexpr (SIGN) {
        // the 32 channel bits of the four pixel rows these two registers hold, dropped into the lanes that store them
        unsigned long long bits[2];
        bits[0] = wave_ballot(y[0] < bit_thrv[j]);
        bits[1] = wave_ballot(y[1] < bit_thrv[j]);
        if constexpr (PIPE) {
          if constexpr (t > 0) {
            hold_until(pend[0], bits[0], bits[1]);      // behind this unit's compares ...
            hold_until(pend[1], y[0], y[1]);            // ... and its transform
            flush_ballots(IntC<(t > 0 ? t - 1 : 0)>{});
          }
          pend[0] = bits[0];
          pend[1] = bits[1];
        } else {
          settle_ballots(bits);
          constexpr int q0 = (r0 & 3) + 8 * (r0 >> 2);
          bw[j] = write_lane_settled<q0>((uint32_t)bits[0], bw[j]);
          bw[j] = write_lane_settled<q0 + 4>((uint32_t)(bits[0] >> 32), bw[j]);
          bw[j] = write_lane_settled<q0 + 1>((uint32_t)bits[1], bw[j]);
          bw[j] = write_lane_settled<q0 + 5>((uint32_t)(bits[1] >> 32), bw[j]);
        }
      }
    }
  };
  // the second output's store: two words per pixel row (one when the last word does not exist)
  auto sign_store = [&](uint32_t sob) LCE_LAMBDA_INLINE {
    if constexpr (KSPLIT) {
      buf_store1(rsgn, sob, bw[0]);                   // the wave's one word (sign_chan_off is out of range when it does not exist)
    } else if ((G.Wout & 1) == 0) {
      const u32x2 v = {bw[0], bw[1]};
      buf_store2(rsgn, sob, v);
    } else {
      buf_store1(rsgn, sob, bw[0]);
      buf_store1(rsgn, (n0 >> 5) + 1 < G.Wout ? sat_add_u32(sob, 4u) : kOobOffset, bw[1]);
    }
  };
  auto epi_b_read = [&](auto kc) LCE_LAMBDA_INLINE {
    constexpr int k = decltype(kc)::value;
    if constexpr (DST == kDstFloat) {
      yb[k] = *(const f32x4*)(scratch + ((lane / LPR) + RPI * k) * SCW + (lane & (LPR - 1)) * 4);
    } else if constexpr (DST == kDstInt8) {
      // store instruction k / 4 covers rows (lane >> 2) + 16 * (k / 4); a lane's 16 channels = reads 4 (k % 4) .. + 3
      yb[k] = *(const f32x4*)(scratch + ((lane / LPR) + RPI * (k >> 2)) * SCW + (lane & (LPR - 1)) * 16 + (k & 3) * 4);
    }
  };
  auto epi_b = [&](auto kc) LCE_LAMBDA_INLINE {
    if constexpr (!kTight) epi_b_read(kc);
  };
  // `ob` = this lane's byte offset for store instruction 0 (out of range: nothing is stored)
  auto epi_c = [&](auto kc, uint32_t ob) LCE_LAMBDA_INLINE {
    constexpr int k = decltype(kc)::value;
    if constexpr (kTight) epi_b_read(kc);
    if constexpr (DST == kDstFloat) {
      buf_store_streaming_so(rout, ob, (uint32_t)(RPI * k) * row_bytes, yb[k]);
    } else if constexpr (DST == kDstInt8) {
      // round half away from zero on values already clamped to [-128, 127] (lce_kernels.h, round_sat_i8); scalar adds:
      // a packed-f32 add beside the MFMA stream costs a dozen cycles more than its issue slot
      // (the conversion, which truncates, packs as it goes -- two dwords at a time: cvt_pack8_i8)
#pragma unroll
      for (int i = 0; i < 4; ++i) if constexpr (!I8F) yb[k][i] = yb[k][i] + __builtin_copysignf(0x1.fffffep-2f, yb[k][i]);
      if constexpr ((k & 1) == 1) {
        uint32_t lo, hi;
        if constexpr (I8F) cvt_rpi_pack8_i8(yb[k - 1], yb[k], lo, hi);      // floor(x + 0.5): no add in front of the conversion
        else cvt_pack8_i8(yb[k - 1], yb[k], lo, hi);
        pk[k >> 2][(k & 3) - 1] = lo;
        pk[k >> 2][k & 3] = hi;
      }
      if constexpr ((k & 3) == 3) buf_store_so(rout, ob, (uint32_t)(RPI * (k >> 2)) * row_bytes, pk[k >> 2]);
    } else {
      if constexpr (k == 0) {
        // lane p (< 32) owns pixel row p: the two words of this wave's 64 channels
        if constexpr (KSPLIT) {
          buf_store1(rout, ob, bw[0]);
        } else if ((G.Wout & 1) == 0) {
          const u32x2 v = {bw[0], bw[1]};
          buf_store2(rout, ob, v);
        } else {
          buf_store1(rout, ob, bw[0]);
          buf_store1(rout, (n0 >> 5) + 1 < G.Wout ? sat_add_u32(ob, 4u) : kOobOffset, bw[1]);   // (ob may be the saturated marker)
        }
      }
    }
  };
  // A partial pixel block (the last of a segment whose pixel count is not a multiple of 32): the rows past the segment
  // are not stored (their lanes re-read the segment's last pixel -- or, in a flat run that is shorter than the planned
  // one, pixels of images past the launch).  Out of line, after the K loop that carried the block's epilogue.
  auto epi_partial = [&](int u, const Ctx& cx) LCE_LAMBDA_INLINE {
    int q = (u << G.pph_log) + pp;
    if (q >= nblk) return;
    // last real row of the block (flat: the run's last block ends where the block's segments end -- a short last run of a
    // launch ends earlier than the table's)
    const uint32_t lim1 = q == dyn_last ? (uint32_t)dyn_lim : tabs[G.tab_lim / 4 + q];   // (per lane)
    const uint32_t base = (cx.t[3] & 0x7fffffffu) + chan_off;          // the table's offset is for row (lane's first row)
    if constexpr (DST == kDstFloat) {
      const uint32_t rowl = (uint32_t)(lane / LPR);
#pragma unroll
      for (int k = 0; k < 32 / RPI; ++k) {
        const uint32_t row = rowl + (uint32_t)(RPI * k);
        buf_store_streaming(rout, row <= lim1 ? base + (row - rowl) * row_bytes : kOobOffset, yb[k]);
      }
    } else if constexpr (DST == kDstInt8) {
      const uint32_t rowl = (uint32_t)(lane / LPR);
#pragma unroll
      for (int k = 0; k < 32 / RPI; ++k) {
        const uint32_t row = rowl + (uint32_t)(RPI * k);
        buf_store(rout, row <= lim1 ? base + (row - rowl) * row_bytes : kOobOffset, pk[k]);
      }
    } else {
      const uint32_t rowl = (uint32_t)(lane & 31);
      const uint32_t o = rowl <= lim1 ? base : kOobOffset;                 // (base may itself be past the range: chan_off's marker)
      buf_store1(rout, o, bw[0]);
      if constexpr (!KSPLIT) buf_store1(rout, (n0 >> 5) + 1 < G.Wout ? sat_add_u32(o, 4u) : kOobOffset, bw[1]);
    }
    if constexpr (SIGN) {
      const uint32_t rowl = (uint32_t)(lane & 31);
      sign_store(rowl <= lim1 ? sat_add_u32(cx.s & 0x7fffffffu, sign_chan_off) : kOobOffset);
    }
  };
  // which units ride in K-step ks: A in the first 4/9 of the steps, B from the middle on, C at the end
  // B's gaps are the K-steps ks >= SB0 with ks % 4 in {2, 3} (SBG of them); C's the last SBN K-steps
  // KSPLIT: the pair's exchange comes first -- K-steps 0..3 send one accumulator quad each, a block barrier behind K-step 4,
  // the four inbox reads behind K-step 5, four adds behind each of K-steps 6..9 -- then the (halved) phases as above
  constexpr int XS = KSPLIT ? 10 : 0;                     // K-steps of the exchange
  constexpr int NUB = KSPLIT ? 4 : 8;                     // phase B reads / phase C units
  constexpr int SA0 = XS, SA = SA0 + (KSPLIT ? NUA : KS * 4 / 9);   // phase A rides in K-steps [SA0, SA)
  constexpr int SB0 = (SA + GA - 1) / GA * GA, SBN = KSPLIT ? 4 : (KS * 2 + 8) / 9, SC0 = KS - SBN;
  static_assert(SA < KS - 1, "phase layout");
  constexpr int SBG_avail = (SC0 - SB0) / GA * 2 + ((SC0 - SB0) % GA > 2 ? (SC0 - SB0) % GA - 2 : 0);
  constexpr int SBG = SBG_avail >= 4 ? 4 : (SBG_avail >= 2 ? 2 : 1);
  static_assert(SA <= SB0 && SB0 <= SC0 && (SBG_avail >= 1 || DST == kDstBitpacked), "phase layout");

  // KSPLIT: the exchange of partial sums between the two waves of a pair (set = the accumulator set being drained,
  // SLOT = the parity of its block).  A wave's local tile 1 is its partner's local tile 0.
  f32x4 xch[4];
  auto x_send = [&](auto qc, f32x16 (&set)[2], int slot) LCE_LAMBDA_INLINE {
    constexpr int q = decltype(qc)::value;
    const f32x4 v = {set[1][4 * q], set[1][4 * q + 1], set[1][4 * q + 2], set[1][4 * q + 3]};
    *(f32x4*)(outbox + slot * 4096 + q * 1024 + lane * 16) = v;
  };
  auto x_recv = [&](int slot) LCE_LAMBDA_INLINE {
#pragma unroll
    for (int q = 0; q < 4; ++q) xch[q] = *(const f32x4*)(inbox + slot * 4096 + q * 1024 + lane * 16);
  };
  auto x_add = [&](auto qc, f32x16 (&set)[2]) LCE_LAMBDA_INLINE {
    constexpr int q = decltype(qc)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) set[0][4 * q + i] += xch[q][i];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[a][j] = kbt;

  Ctx cur, nxt;             // block being multiplied, the one after it
  load_ctx(0, cur);
  uint32_t epi_ob = kOobOffset;      // the block being drained: nothing yet
  bool epi_part = false;
  int epi_u = 0;
  Ctx epi_cx = cur;
  u32x4 af[2][GA];

  // KSPLIT: this wave's half of a pixel's channels lies KCHW chunks of 32 bytes further on; added to a context's tap-row
  // addresses once it has arrived (the first one here, the others late in the block step that loaded them)
  auto ctx_khalf = [&](Ctx& cx) LCE_LAMBDA_INLINE {
    if constexpr (KSPLIT) {
      const uint32_t koff = (uint32_t)khalf * (uint32_t)(KCHW * 32);
      cx.t[0] += koff; cx.t[1] += koff; cx.t[2] += koff;
    }
  };
  ctx_khalf(cur);
  auto frag_addr = [&](const Ctx& cx, int ks) LCE_LAMBDA_INLINE -> uint32_t {
    const int fy = ks / (KW * KCHW), fx = (ks / KCHW) % KW, kc = ks % KCHW;
    return cx.t[fy] + (uint32_t)(fx * PS + kc * 32);
  };

  // One block step: K loop of pixel block u into accumulator set PAR, with the epilogue of block u - 1
  // (set PAR ^ 1) and this step's share of the production riding between the MFMAs.  A wave issues in order and an
  // MFMA waits for the matrix pipe, so fillers hide only DIRECTLY behind an MFMA, a handful (<= 5) at a time: a
  // K-step is  MFMA . gap 0 . MFMA . gap 1  and every piece of work below is cut to fit a gap:
  //   gap 0: the next fragment group's four reads (every fourth K-step); phase B reads
  //   gap 1: phase A units, then the next block's context load, then production chunks, with phase C's stores
  constexpr int NPG = KS - SA - 1;                          // gap-1 slots behind phase A and the context load
  uint32_t next_ob = kOobOffset, next_sob = kOobOffset, epi_sob = kOobOffset;
  bool next_part = false;
  Issue isa, isc;
  // FIRST (compile-time): the block's very first block step -- each K-step's bank fragments are waited for where they are
  // first used, and there is no previous block to drain (no epilogue units ride along)
  auto step = [&](auto kc_, auto first_, int u, uint32_t sch1, uint32_t sch2) LCE_LAMBDA_INLINE {
    constexpr int k = decltype(kc_)::value, PAR = k & 1;
    constexpr bool FIRST = decltype(first_)::value != 0;
    static_assert(!FIRST || k == 0, "the first block step is step 0 of tile step 0");
    if constexpr (k == 0) {
      // first block of a tile step: its fragments may live in rows the previous tile step wrote
#pragma unroll
      for (int d = GA - 1; d >= 0; --d) af[0][d] = *(const u32x4*)(lds0 + frag_addr(cur, d));
    }
    auto kstep = [&](auto ksc) LCE_LAMBDA_INLINE {
      constexpr int ks = decltype(ksc)::value;
      if constexpr (ks < KS) {
        constexpr int g = ks / GA;
        constexpr int fs = (g + k * NG) & 1;     // the fragment set this block's group g lives in (set 0 after a barrier)
        if constexpr (FIRST) bank_home(IntC<ks>{});     // this K-step's weights have arrived (a counted wait) and are at home
        // ---------------- MFMA 0 ----------------
        if constexpr (ks == 0) acc[PAR][0] = mfma_fp4_32x32x64_unscaled(af[fs][0], W[0][0], kbt);    // K_bt - <a, w> = 2 * accum
        else acc[PAR][0] = mfma_fp4_32x32x64_unscaled(af[fs][ks % GA], W[ks][0], acc[PAR][0]);
        pin(acc[PAR][0]);    // (an MFMA has no side effect the sched_fence could hold: the pin keeps it in place)
        sched_fence();       // ... alone in its scheduling region: no filler may slip in FRONT of it
        // ---------------- gap 0 ----------------
#ifndef LCE_ST_NOFRAG   // timing ablation (results are wrong): no fragment reads in the K loop
        // The fragments of the NEXT group (the next block's first group rides at the end, except across the barrier),
        // two reads behind each of the group's first two MFMA 0s (more than two LDS reads per gap queue up behind the
        // other waves'), issued last-used first: the counted wait for the group's first fragment covers the group.
        if constexpr (ks % GA < 2) {
          constexpr int d1 = GA - 1 - 2 * (ks % GA), d0 = d1 - 1;      // fragments 3, 2 then 1, 0
          if constexpr (g + 1 < NG) {
            if constexpr ((g + 1) * GA + d1 < KS) af[fs ^ 1][d1] = *(const u32x4*)(lds0 + frag_addr(cur, (g + 1) * GA + d1));
            if constexpr ((g + 1) * GA + d0 < KS) af[fs ^ 1][d0] = *(const u32x4*)(lds0 + frag_addr(cur, (g + 1) * GA + d0));
          } else if constexpr (k < 3) {
            af[fs ^ 1][d1] = *(const u32x4*)(lds0 + frag_addr(nxt, d1));
            af[fs ^ 1][d0] = *(const u32x4*)(lds0 + frag_addr(nxt, d0));
            if constexpr (KS - (NG - 1) * GA < 2) {     // a last group of one K-step: all four reads behind it
              af[fs ^ 1][1] = *(const u32x4*)(lds0 + frag_addr(nxt, 1));
              af[fs ^ 1][0] = *(const u32x4*)(lds0 + frag_addr(nxt, 0));
            }
          }
        }
#endif
#ifndef LCE_ST_NOEPI    // timing ablation (results are wrong): no epilogue units
        // phase B: two reads per gap, in the gaps the fragment reads leave free
        if constexpr (!FIRST && ks >= SB0 && ks % GA >= 2 && (ks - SB0) / GA * 2 + (ks % GA - 2) < SBG) {
          constexpr int slot = (ks - SB0) / GA * 2 + (ks % GA - 2);
          constexpr int lo = stream_unit_lo(NUB, slot, SBG), hi = stream_unit_lo(NUB, slot + 1, SBG);
          if constexpr (slot == 0) wave_lds_scratch_fence();   // phase A's writes before phase B's reads
#ifndef LCE_ST_NOEPI_B
          if constexpr (lo + 0 < hi) epi_b(IntC<lo + 0>{});
          if constexpr (lo + 1 < hi) epi_b(IntC<lo + 1>{});
          if constexpr (lo + 2 < hi) epi_b(IntC<lo + 2>{});
          if constexpr (lo + 3 < hi) epi_b(IntC<lo + 3>{});
          if constexpr (lo + 4 < hi) epi_b(IntC<lo + 4>{});
          if constexpr (lo + 5 < hi) epi_b(IntC<lo + 5>{});
          if constexpr (lo + 6 < hi) epi_b(IntC<lo + 6>{});
          if constexpr (lo + 7 < hi) epi_b(IntC<lo + 7>{});
#endif
        }
#endif
        if constexpr (ks == 1) {       // what the NEXT step's drain of this block needs
          next_ob = out_base(u, cur);
          if constexpr (SIGN) next_sob = sign_base(u, cur);
          next_part = (int)(uniform(cur.t[3] | cur.m) >> 31) != 0;
        }
        sched_fence();
        // ---------------- MFMA 1 ----------------
        if constexpr (ks == 0) acc[PAR][1] = mfma_fp4_32x32x64_unscaled(af[fs][0], W[0][1], kbt);
        else acc[PAR][1] = mfma_fp4_32x32x64_unscaled(af[fs][ks % GA], W[ks][1], acc[PAR][1]);
        pin(acc[PAR][1]);
        sched_fence();
        // ---------------- gap 1 ----------------
#ifndef LCE_ST_NOEPI
        if constexpr (KSPLIT && !FIRST && ks < XS) {
          constexpr int slot = (k + 1) & 1;            // parity of the block being drained (u - 1; u = 4 T + k)
          if constexpr (ks < 4) x_send(IntC<ks>{}, acc[PAR ^ 1], slot);
          else if constexpr (ks == 4) block_barrier_keep_vm();     // both halves are in the inboxes
          else if constexpr (ks == 5) x_recv(slot);
          else x_add(IntC<ks - 6>{}, acc[PAR ^ 1]);
        }
        if constexpr (!FIRST && ks >= SA0 && ks < SA) {
          constexpr int lo = stream_unit_lo(NUA, ks - SA0, SA - SA0), hi = stream_unit_lo(NUA, ks - SA0 + 1, SA - SA0);
          if constexpr (lo == 0) wave_lds_order();     // the scratch's previous readers are done (in-order LDS)
#ifndef LCE_ST_NOEPI_A
          if constexpr (lo + 0 < hi) epi_a(IntC<lo + 0>{}, acc[PAR ^ 1], IntC<1>{});
          if constexpr (lo + 1 < hi) epi_a(IntC<lo + 1>{}, acc[PAR ^ 1], IntC<1>{});
          if constexpr (lo + 2 < hi) epi_a(IntC<lo + 2>{}, acc[PAR ^ 1], IntC<1>{});
          if constexpr (lo + 3 < hi) epi_a(IntC<lo + 3>{}, acc[PAR ^ 1], IntC<1>{});
#endif
          static_assert(hi - lo <= 4, "units per gap");
        }
        if constexpr (!FIRST && ks >= SC0) {
          constexpr int lo = stream_unit_lo(NUB, ks - SC0, SBN), hi = stream_unit_lo(NUB, ks - SC0 + 1, SBN);
#ifndef LCE_ST_NOEPI_C
          if constexpr (lo + 0 < hi) epi_c(IntC<lo + 0>{}, epi_ob);
          if constexpr (lo + 1 < hi) epi_c(IntC<lo + 1>{}, epi_ob);
          if constexpr (lo + 2 < hi) epi_c(IntC<lo + 2>{}, epi_ob);
          if constexpr (lo + 3 < hi) epi_c(IntC<lo + 3>{}, epi_ob);
#endif
        }
#endif
        if constexpr (kBankPaced && FIRST && ks + kBankAhead < KS) bank_loads(IntC<2 * (ks + kBankAhead)>{}, IntC<2 * (ks + kBankAhead) + 2>{});
        if constexpr (kPipeBallots && (SIGN || DST == kDstBitpacked) && !FIRST && ks == SA) flush_ballots(IntC<NUA - 1>{});   // the last unit's
        if constexpr (SIGN && !FIRST && ks == SA) sign_store(epi_sob);      // phase A is complete: the drained block's sign words
        if constexpr (ks == SA) load_ctx(u + 1, nxt);
        if constexpr (KSPLIT && ks == (NG - 1) * GA - 1) ctx_khalf(nxt);     // (in front of the last group, whose reads use it)
#ifndef LCE_ST_NOPROD   // timing ablation (results are wrong): no production between the MFMAs
        // production: block step 0 expands item A (16 chunks), block step 1 issues the next one (12 chunks; its load
        // has three block steps to arrive)
        if constexpr (ks > SA) {
          if constexpr (k == 0) {
            constexpr int lo = stream_unit_lo(16, ks - SA - 1, NPG), hi = stream_unit_lo(16, ks - SA, NPG);
            if constexpr (lo + 0 < hi) write_chunk(IntC<lo + 0>{}, pwa, pda, pma);
            if constexpr (lo + 1 < hi) write_chunk(IntC<lo + 1>{}, pwa, pda, pma);
            if constexpr (lo + 2 < hi) write_chunk(IntC<lo + 2>{}, pwa, pda, pma);
            if constexpr (lo + 3 < hi) write_chunk(IntC<lo + 3>{}, pwa, pda, pma);
            static_assert(hi - lo <= 4, "chunks per gap");
          } else if constexpr (k == 1) {
            constexpr int lo = stream_unit_lo(kIssueChunks, ks - SA - 1, NPG), hi = stream_unit_lo(kIssueChunks, ks - SA, NPG);
            if constexpr (lo + 0 < hi) issue_chunk(IntC<lo + 0>{}, isa, sch1 + (uint32_t)tid, sch2, pwa, pda, pma);
            if constexpr (lo + 1 < hi) issue_chunk(IntC<lo + 1>{}, isa, sch1 + (uint32_t)tid, sch2, pwa, pda, pma);
            if constexpr (lo + 2 < hi) issue_chunk(IntC<lo + 2>{}, isa, sch1 + (uint32_t)tid, sch2, pwa, pda, pma);
            static_assert(hi - lo <= 3, "chunks per gap");
          } else if constexpr (KSPLIT && k == 2) {
            constexpr int lo = stream_unit_lo(16, ks - SA - 1, NPG), hi = stream_unit_lo(16, ks - SA, NPG);
            if constexpr (lo + 0 < hi) write_chunk(IntC<lo + 0>{}, pwc, pdc, pmc);
            if constexpr (lo + 1 < hi) write_chunk(IntC<lo + 1>{}, pwc, pdc, pmc);
            if constexpr (lo + 2 < hi) write_chunk(IntC<lo + 2>{}, pwc, pdc, pmc);
            if constexpr (lo + 3 < hi) write_chunk(IntC<lo + 3>{}, pwc, pdc, pmc);
          } else if constexpr (KSPLIT && k == 3) {
            constexpr int lo = stream_unit_lo(kIssueChunks, ks - SA - 1, NPG), hi = stream_unit_lo(kIssueChunks, ks - SA, NPG);
            if constexpr (lo + 0 < hi) issue_chunk(IntC<lo + 0>{}, isc, sch1 + (uint32_t)(256 + tid), sch2, pwc, pdc, pmc);
            if constexpr (lo + 1 < hi) issue_chunk(IntC<lo + 1>{}, isc, sch1 + (uint32_t)(256 + tid), sch2, pwc, pdc, pmc);
            if constexpr (lo + 2 < hi) issue_chunk(IntC<lo + 2>{}, isc, sch1 + (uint32_t)(256 + tid), sch2, pwc, pdc, pmc);
          }
        }
#endif
        sched_fence();
      }
    };
    kstep(IntC<0>{}); kstep(IntC<1>{}); kstep(IntC<2>{}); kstep(IntC<3>{}); kstep(IntC<4>{}); kstep(IntC<5>{});
    kstep(IntC<6>{}); kstep(IntC<7>{}); kstep(IntC<8>{}); kstep(IntC<9>{}); kstep(IntC<10>{}); kstep(IntC<11>{});
    kstep(IntC<12>{}); kstep(IntC<13>{}); kstep(IntC<14>{}); kstep(IntC<15>{}); kstep(IntC<16>{}); kstep(IntC<17>{});
    kstep(IntC<18>{}); kstep(IntC<19>{}); kstep(IntC<20>{}); kstep(IntC<21>{}); kstep(IntC<22>{}); kstep(IntC<23>{});
    kstep(IntC<24>{}); kstep(IntC<25>{}); kstep(IntC<26>{}); kstep(IntC<27>{}); kstep(IntC<28>{}); kstep(IntC<29>{});
    kstep(IntC<30>{}); kstep(IntC<31>{}); kstep(IntC<32>{}); kstep(IntC<33>{}); kstep(IntC<34>{}); kstep(IntC<35>{});
    static_assert(KS <= 36, "kstep calls above");
    // the drained block was a partial one: its stores, redirected, out of line
    if constexpr (!FIRST) {
      if (epi_part) epi_partial(epi_u, epi_cx);
    }
    // this block becomes the one being drained
    epi_ob = next_ob;
    epi_sob = next_sob;
    epi_part = next_part;
    epi_u = u;
    epi_cx = cur;
    cur = nxt;
  };

  using TagFirst = IntC<1>;
  using TagSteady = IntC<0>;
  // (The first tile step is its own copy of all four block steps, not just of the peeled one: with ONE copy of steps 1-3 behind
  // a branch the float kernel measured +1.5-2 % on L0 -- register allocation --, profiles/r04/ab_r03_kernels.txt.)
  auto tile_step = [&](auto first_, int T) LCE_LAMBDA_INLINE {
    const uint32_t sch1 = sched[T + 1], sch2 = sched[T + 2];
    // (the last tile step may be short: block steps past the stream are skipped, not computed and masked)
    step(IntC<0>{}, first_, 4 * T + 0, sch1, sch2);
    if (4 * T + 1 < usteps) step(IntC<1>{}, TagSteady{}, 4 * T + 1, sch1, sch2);
    if (4 * T + 2 < usteps) step(IntC<2>{}, TagSteady{}, 4 * T + 2, sch1, sch2);
    if (4 * T + 3 < usteps) step(IntC<3>{}, TagSteady{}, 4 * T + 3, sch1, sch2);
    // the rare last item of a tile step's quota (beyond what rides between the MFMAs)
    if (have_b) item_write(pwb, pdb, pmb);
    have_b = sch2 - sch1 > kItemB;
    if (have_b) item_issue(sch1 + kItemB + (uint32_t)tid, sch2, pwb, pdb, pmb);
    block_barrier_keep_vm();     // this tile step's rows are visible; the rows it read may be overwritten
    LCE_SPH(2 + T);
  };
  if (ntile > 0) tile_step(TagFirst{}, 0);
  for (int T = 1; T < ntile; ++T) tile_step(TagSteady{}, T);

  // drain: the last block step's accumulators (its set by the parity of the step count)
  auto drain = [&](f32x16 (&last)[2], int slot) LCE_LAMBDA_INLINE {
    if constexpr (KSPLIT) {
      x_send(IntC<0>{}, last, slot); x_send(IntC<1>{}, last, slot); x_send(IntC<2>{}, last, slot); x_send(IntC<3>{}, last, slot);
      block_barrier_keep_vm();
      x_recv(slot);
      x_add(IntC<0>{}, last); x_add(IntC<1>{}, last); x_add(IntC<2>{}, last); x_add(IntC<3>{}, last);
    }
    wave_lds_order();
    auto drain_a = [&](auto tc) LCE_LAMBDA_INLINE {
      if constexpr (decltype(tc)::value < NUA) epi_a(tc, last, IntC<0>{});
    };
    drain_a(IntC<0>{}); drain_a(IntC<1>{}); drain_a(IntC<2>{}); drain_a(IntC<3>{});
    drain_a(IntC<4>{}); drain_a(IntC<5>{}); drain_a(IntC<6>{}); drain_a(IntC<7>{});
    drain_a(IntC<8>{}); drain_a(IntC<9>{}); drain_a(IntC<10>{}); drain_a(IntC<11>{});
    drain_a(IntC<12>{}); drain_a(IntC<13>{}); drain_a(IntC<14>{}); drain_a(IntC<15>{});
    if constexpr (SIGN) sign_store(epi_sob);
    wave_lds_fence();
    auto drain_bc = [&](auto kc) LCE_LAMBDA_INLINE {
      if constexpr (decltype(kc)::value < NUB) epi_b(kc);
    };
    drain_bc(IntC<0>{}); drain_bc(IntC<1>{}); drain_bc(IntC<2>{}); drain_bc(IntC<3>{});
    drain_bc(IntC<4>{}); drain_bc(IntC<5>{}); drain_bc(IntC<6>{}); drain_bc(IntC<7>{});
    auto drain_c = [&](auto kc) LCE_LAMBDA_INLINE {
      if constexpr (decltype(kc)::value < NUB) epi_c(kc, epi_ob);
    };
    drain_c(IntC<0>{}); drain_c(IntC<1>{}); drain_c(IntC<2>{}); drain_c(IntC<3>{});
    drain_c(IntC<4>{}); drain_c(IntC<5>{}); drain_c(IntC<6>{}); drain_c(IntC<7>{});
    if (epi_part) epi_partial(epi_u, epi_cx);
  };
  // (a block without work -- past the launch's segments -- still takes part in the barriers of the pair exchange: every
  //  wave of a block runs the same steps)
  if (usteps & 1) drain(acc[0], 0);     // the last block step u = usteps - 1 is even: set 0, slot 0
  else drain(acc[1], 1);
  LCE_SPH(63);
}

}  // namespace lce
