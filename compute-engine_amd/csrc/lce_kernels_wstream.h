// Weight-STREAMING variant of the matrix-core engine ("engine=wstream", round 5): activations stationary in LDS,
// weights streamed from L2 straight into registers while the matrix cores run.
//
// Why it exists.  The weight-stationary kernel (lce_kernels_stream.h) loads a block's whole filter
// bank -- 295 KB per CU for a 3x3x256 layer, ~5 k cycles of the CU's 64 B/clk vector-memory path however the loads are
// ordered -- before its first useful MFMA.  On L0 (98 pixel blocks per block) that is 3 % of a block's life; on the layers
// whose launch is ONE image per CU (14x14x256: 7 pixel blocks; 7x7x512: 4 images, 7 pixel blocks) it is 28-30 %, with the
// matrix pipe 25 % busy over the launch (profiles/r04/stream_phases.txt, pmc_summary_{14x256,7x512}_f32.json).  For such
// launches every weight fragment is used only a handful of times anyway, so keeping it is worth nothing.  Turned round:
//
//   * a block expands the whole images of its GROUP (a few consecutive images: <= 48 KB of FP4 pixels) into LDS ONCE, in
//     the padded pixel layout of the other matrix-core kernels ([image][row][x][KCH*32+16 B]), padding rows and columns
//     included (out-of-range loads return the "+1" word; exact SAME-zero writes zero codes);
//   * the group's output pixels (NHWC: contiguous in memory) are cut into 32-pixel blocks laid end to end; a block owns up
//     to NB <= 4 of them, wave w the 64-channel slice w of the block's 256 channels (grid.y covers the rest);
//   * the K loop is K-major: per K-step a wave loads its two 32-channel weight fragments (2 KiB) from the planner's FP4
//     image DIRECTLY into registers, kWsPrefetch K-steps ahead of their use, reads one A fragment per pixel block from
//     LDS (shared by the four waves) and issues 2 * NB MFMAs.  No weight ever touches LDS, no barrier sits in the loop,
//     the weight stream (<= 31 B/clk per CU) overlaps the matrix work instead of preceding it;
//   * <= 256 registers: TWO blocks are resident per CU, so one block's prologue (expansion) and epilogue (transform,
//     transpose through a wave-private LDS scratch, row stores) run beside the other block's K loop.
//
// Same arithmetic as the other matrix-core kernels -- the planner's FP4 weight image (pack_for_mfma: negated,
// tile-major: [32-channel tile][K-step][K-half][32 x 16 B]), accumulators start at K_bt and end as 2 * popcount-accumulator (output_transform.h:62-91),
// float transform with two roundings (:99-106), int8 round-half-away + saturate (:31-44), bitpacked compare (:160-168)
// -- so results are bit-identical to them and to the oracle.  Replaces core/indirect_bgemm/kernel_4x2_portable.h:84-111
// (one GEMM for every shape) for the shapes the weight-stationary kernel serves badly.
#pragma once
#include <lce_device_intrinsics.h>
#include "lce_kernel_args.h"
#include "lce_kernels.h"
#include "lce_kernels_mfma.h"
#include "lce_stream_stamps.h"

namespace lce {

constexpr int kWsPrefetch = 3;        // K-steps of weight fragments in flight ahead of the one being multiplied
constexpr int kWsScratch = 8192;      // bytes of a wave's epilogue scratch: [32 pixel rows][64 channels] floats

// DST: kDstFloat / kDstInt8 / kDstBitpacked.  KCH: 64-channel chunks per tap (3x3 filters: 9 * KCH K-steps).  NB: the most
// pixel blocks a block of this launch owns (the planner's parts differ by at most one: a block runs NB or NB - 1).
// SIGN (float / int8): the epilogue also writes the LceQuantize of the values it produces (lce_hip_bconv2d_run_dual).
// I8F (int8 output): transform with one fma, round with floor(x + 0.5) (lce_kernels.h, pack8_i8_clamped; selected only where that is exact for the plan).
template <int DST, int KCH, int NB, bool SIGN, bool I8F = false>
LCE_KERNEL void __launch_bounds__(256, 2)
bconv2d_wstream(const WsArgs G, const uint8_t* __restrict__ xin, const uint8_t* __restrict__ wq,
                const float* __restrict__ mul, const float* __restrict__ bias, const float* __restrict__ thrf,
                const uint32_t* __restrict__ tabs, void* __restrict__ out, uint32_t* __restrict__ sign_words) {
  static_assert(!SIGN || DST != kDstBitpacked, "a bitpacked-output plan already writes bits");
  static_assert(NB >= 1 && NB <= 4, "two blocks per CU: the accumulators of NB pixel blocks x 2 channel tiles must leave room");
  constexpr int KS = 9 * KCH;                  // K-steps (3x3 taps x 64-channel chunks)
  constexpr int PS = KCH * 32 + 16;            // LDS bytes per pixel (the +16 staggers the banks)
  constexpr int D = kWsPrefetch, NW = D + 1;

  const int tid = thread_idx_x();
  const int lane = tid & (kWave - 1), wave = uniform(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  // blocks are numbered part-major: the blocks a CU holds together are then a long and a short part of different groups
  const uint32_t bx = (uint32_t)block_idx_x();
  const uint32_t part = uniform(fastdiv_nb(bx, G.div_groups)), group = bx - part * (uint32_t)G.GROUPS;
  const int n0 = (block_idx_y() * 4 + wave) * 64;
  const bool slice_ok = n0 < G.Npad;
  uint8_t* const lds0 = lds_base();
  LCE_SPH(0);

  // ---- the weight stream: fragment (K-step ks, channel tile j) of lane (column l31, K-half) ----
  const rsrc_t rw = make_rsrc(wq, G.w_bytes);
  // (tile-major image, pack_for_mfma: [32-channel tile][K-step][K-half][32 x 16 B])
  const uint32_t wlane0 = slice_ok ? ((uint32_t)(n0 >> 5) * (uint32_t)(KS * 2) + (uint32_t)half) * 512u + (uint32_t)l31 * 16u : kOobOffset;
  const uint32_t wlane1 = sat_add_u32(wlane0, (uint32_t)(KS * 1024));   // the slice's second 32 channels
  constexpr uint32_t wstep = 1024u;                                     // bytes per K-step of one tile: 2 K-halves x 512
  u32x4 Wr[NW][2];
  auto w_load = [&](int ks) LCE_LAMBDA_INLINE {
    Wr[ks % NW][0] = buf_load_so(rw, wlane0, (uint32_t)ks * wstep, (u32x4*)nullptr);
    Wr[ks % NW][1] = buf_load_so(rw, wlane1, (uint32_t)ks * wstep, (u32x4*)nullptr);
  };
#pragma unroll
  for (int ks = 0; ks < D; ++ks) w_load(ks);

  // ---- the group's images -> FP4 pixels in LDS.  Item e = 16 bytes (4 words) of padded pixel (image, row, x) ----
  const rsrc_t rin = make_rsrc(xin, G.in_bytes);
  const uint32_t img0 = group * (uint32_t)G.IPB;
  for (uint32_t e0 = 0; e0 < (uint32_t)G.items; e0 += 4u * 256u) {
    u32x4 wv[4];
    uint32_t dst[4];
    int meta[4];            // first word of the item | 0x100: inside the image | 0x200: a live item
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t e = e0 + (uint32_t)(k * 256 + tid);
      const bool on = e < (uint32_t)G.items;
      const uint32_t t = fastdiv_nb(e, G.div_qg), g = e - t * (uint32_t)G.QG;
      const uint32_t t2 = fastdiv_nb(t, G.div_wp), x = t - t2 * (uint32_t)G.Wp;
      const uint32_t img = fastdiv_nb(t2, G.div_hp), r = t2 - img * (uint32_t)G.Hp;
      const int iy = (int)r - G.PH, ix = (int)x - G.PW;
      const uint32_t gi = img0 + img;
      const bool inside = on && (uint32_t)iy < (uint32_t)G.H && (uint32_t)ix < (uint32_t)G.W && gi < (uint32_t)G.B;
      const int c0 = (int)g * 4;
      const uint32_t off = (((gi * (uint32_t)G.H + (uint32_t)iy) * (uint32_t)G.W + (uint32_t)ix) * (uint32_t)G.Cw + (uint32_t)c0) * 4u;
      dst[k] = img * (uint32_t)G.img_pitch + r * (uint32_t)G.pitch + x * (uint32_t)PS + g * 64u;
      meta[k] = c0 | (inside ? 0x100 : 0) | (on ? 0x200 : 0);
      if ((G.Cw & 3) == 0) {
        wv[k] = buf_load(rin, inside ? off : kOobOffset, (u32x4*)nullptr);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[k][q] = buf_load(rin, inside && c0 + q < G.Cw ? off + 4u * q : kOobOffset, (uint32_t*)nullptr);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!(meta[k] & 0x200)) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cc = (meta[k] & 0xff) + q;
        if (cc >= KCH * 2) continue;                        // words past the padded channel count do not exist
        int valid = G.Cin - cc * 32;                        // channels of this word that exist
        valid = valid < 0 ? 0 : (valid > 32 ? 32 : valid);
        if (!(meta[k] & 0x100) && G.zero_border) valid = 0; // exact SAME-zero: an outside tap contributes 0
        const u32x4 v = valid == 32 ? fp4_of_full_word(wv[k][q]) : fp4_of_word(wv[k][q], valid);
        *(u32x4*)(lds0 + dst[k] + q * 16) = v;
      }
    }
  }

  // per-channel constants of this lane's two channels
  float mj[2], bj[2], tj[2], uj[2], sthr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = slice_ok ? n0 + j * 32 + l31 : 0;
    mj[j] = bj[j] = tj[j] = uj[j] = 0.0f;
    if constexpr (DST == kDstBitpacked) tj[j] = thrf[n];
    else { mj[j] = mul[n]; bj[j] = bias[n]; }
    if constexpr (DST == kDstInt8) { tj[j] = thrf[n]; uj[j] = thrf[G.Npad + n]; }
    // channels past the last one never set a bit (padding bits of the last word are 0, bitpack.h:248-308)
    sthr[j] = n0 + j * 32 + l31 < G.N ? G.bit_thr : -__builtin_inff();
  }

  // ---- this block's pixel blocks, and per lane the LDS address of its pixel in the three tap rows ----
  const uint32_t q0 = uniform(tabs[G.tab_part / 4 + 2 * part]), nbq = uniform(tabs[G.tab_part / 4 + 2 * part + 1]);
  const rsrc_t rtab = make_rsrc(tabs, G.tab_bytes);
  uint32_t ta[NB][3];
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    uint32_t qq = q0 + (uint32_t)q;
    qq = qq < (uint32_t)G.NQ ? qq : (uint32_t)G.NQ - 1u;
    const u32x4 c = buf_load(rtab, G.tab_ctx + (qq * 64u + (uint32_t)lane) * 16u, (u32x4*)nullptr);
    ta[q][0] = c[0]; ta[q][1] = c[1]; ta[q][2] = c[2];
  }
  block_barrier_keep_vm();        // the images are in LDS
  LCE_SPH(1);

  // ---- K loop, K-major: weights of K-step ks + D on their way, one A fragment per pixel block, 2 MFMAs each ----
  f32x16 acc[NB][2];
  const f32x16 kbt = f32x16_fill(G.a_bt);     // the accumulators' start value: K_bt - <a, w> = 2 * accum
  auto kloop = [&](auto nbc) LCE_LAMBDA_INLINE {
    constexpr int NBR = decltype(nbc)::value;
    if constexpr (NBR >= 1) {
      u32x4 af[2][NBR];
#pragma unroll
      for (int q = 0; q < NBR; ++q) af[0][q] = *(const u32x4*)(lds0 + ta[q][0]);
      sched_fence();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // The scheduling fences pin what the source order says: left alone, the compiler sinks every load to just in front of
        // its use to save registers (the first build: weight loads one K-step ahead instead of D, A fragments read right in
        // front of their MFMAs), which exposes the L2 and LDS latencies the prefetch distances are there to hide.
        if (ks + D < KS) w_load(ks + D);
        sched_fence();
#pragma unroll
        for (int q = 0; q < NBR; ++q) {
          if (ks + 1 < KS) {      // the next K-step's fragment of this pixel block, one K-step ahead of its MFMAs
            const int k1 = ks + 1, fy = k1 / (3 * KCH), fx = (k1 / KCH) % 3, kc = k1 % KCH;
            af[k1 & 1][q] = *(const u32x4*)(lds0 + ta[q][fy] + (uint32_t)(fx * PS + kc * 32));
          }
          acc[q][0] = mfma_fp4_32x32x64_unscaled(af[ks & 1][q], Wr[ks % NW][0], ks == 0 ? kbt : acc[q][0]);
          acc[q][1] = mfma_fp4_32x32x64_unscaled(af[ks & 1][q], Wr[ks % NW][1], ks == 0 ? kbt : acc[q][1]);
          pin(acc[q][0]);          // (an MFMA has no side effect a scheduling fence could hold: the pins keep the pair in place)
          pin(acc[q][1]);
          sched_fence();
        }
      }
    }
  };
  if (slice_ok) {
    if (nbq >= (uint32_t)NB) kloop(IntC<NB>{});
    else kloop(IntC<NB - 1>{});
  }
  LCE_SPH(2);

  // ---- epilogue, one pixel block at a time: transform in place, [sign words], transpose through the wave's scratch, row stores ----
  const rsrc_t rout = make_rsrc(out, G.out_bytes);
  const rsrc_t rsgn = make_rsrc(sign_words, SIGN ? G.sign_bytes : 0u);
  float* const scratch = (float*)(lds0 + G.lds_images + (uint32_t)wave * (uint32_t)kWsScratch);   // [32 rows][64 channels]
  const uint32_t gpx0 = group * (uint32_t)G.NPXG;           // the group's first output pixel
  // 32 channel bits of every pixel row of accumulator tile j -> the lanes that store them (lane p < 32 owns row p):
  // register r holds rows q (lanes 0-31) and q + 4 (lanes 32-63), q = (r & 3) + 8 * (r >> 2)
  auto gather_bits = [&](const f32x16 (&a)[2], const float (&thr)[2], auto below, uint32_t (&words)[2]) LCE_LAMBDA_INLINE {
    gather_tile_bits<2, decltype(below)::value != 0>(a, thr, words);       // (lce_kernels_mfma.h: no hazard padding per register)
  };
  // lane p < 32 stores pixel row p's two words of this wave's 64 channels (the second one only where it exists)
  auto store_words2 = [&](rsrc_t r, uint32_t px_row, bool row_ok, const uint32_t (&words)[2]) LCE_LAMBDA_INLINE {
    const int w0 = n0 >> 5;
    const uint32_t o = lane < 32 && row_ok && w0 < G.Wout ? ((gpx0 + px_row) * (uint32_t)G.Wout + (uint32_t)w0) * 4u : kOobOffset;
    if ((G.Wout & 1) == 0) {
      const u32x2 v = {words[0], words[1]};
      buf_store2(r, o, v);
    } else {
      buf_store1(r, o, words[0]);
      buf_store1(r, w0 + 1 < G.Wout ? sat_add_u32(o, 4u) : kOobOffset, words[1]);
    }
  };
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    if ((uint32_t)q >= nbq || !slice_ok) continue;          // (a short part; a slice past the last channel)
    const uint32_t row0 = (q0 + (uint32_t)q) * 32u;         // first pixel of the block, counted inside the group
    if constexpr (DST == kDstBitpacked) {
      uint32_t words[2];
      gather_bits(acc[q], tj, IntC<0>{}, words);
      const uint32_t pr = row0 + (uint32_t)(lane & 31);
      store_words2(rout, pr, pr < (uint32_t)G.NPXG, words);
    } else {
      // the transform on the accumulators, in place (output_transform.h:93-157)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2 x = {acc[q][j][r], acc[q][j][r + 1]};
          if constexpr (DST == kDstFloat) {
            if (!G.noclamp) { x[0] = med3(x[0], G.cmin, G.cmax); x[1] = med3(x[1], G.cmin, G.cmax); }
          }
          const f32x2 y = I8F ? fma2(x, mj[j], bj[j]) : mul_then_add2(x, mj[j], bj[j]);      // (I8F: one rounding, proven per plan)
          if constexpr (DST == kDstInt8) {   // one clamp: the transformed range of the clamped accumulator, inside int8's (lce_kernels_pointwise.h)
            acc[q][j][r] = med3(y[0], tj[j], uj[j]);
            acc[q][j][r + 1] = med3(y[1], tj[j], uj[j]);
          } else {
            acc[q][j][r] = y[0];
            acc[q][j][r + 1] = y[1];
          }
        }
      if constexpr (SIGN) {
        uint32_t words[2];
        gather_bits(acc[q], sthr, IntC<1>{}, words);
        const uint32_t pr = row0 + (uint32_t)(lane & 31);
        store_words2(rsgn, pr, pr < (uint32_t)G.NPXG, words);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + j * 32 + l31] = acc[q][j][r];
      wave_lds_fence();
      if constexpr (DST == kDstFloat) {
        // 16 lanes x 16 bytes = a pixel row's 64 channels; 4 rows per store instruction
        const int g = lane & 15, rr = lane >> 4;
        const int n = n0 + g * 4;
        f32x4 y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = *(const f32x4*)(scratch + (rr + 4 * k) * 64 + g * 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t pr = row0 + (uint32_t)(rr + 4 * k);
          buf_store_streaming(rout, pr < (uint32_t)G.NPXG && n < G.N ? ((gpx0 + pr) * (uint32_t)G.N + (uint32_t)n) * 4u : kOobOffset, y[k]);
        }
      } else {
        // int8: a lane converts 16 consecutive channels of one row into ONE 16-byte store; 16 rows per store instruction
        const int g = lane & 3, rr = lane >> 2;
        const int n = n0 + g * 16;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const f32x4* src = (const f32x4*)(scratch + (rr + 16 * k) * 64 + g * 16);
          u32x4 pk;
#pragma unroll
          for (int h = 0; h < 4; h += 2) {       // (values already inside [-128, 127])
            uint32_t lo, hi;
            pack8_i8_clamped<I8F>(src[h], src[h + 1], lo, hi);
            pk[h] = lo;
            pk[h + 1] = hi;
          }
          const uint32_t pr = row0 + (uint32_t)(rr + 16 * k);
          buf_store(rout, pr < (uint32_t)G.NPXG && n < G.N ? (gpx0 + pr) * (uint32_t)G.N + (uint32_t)n : kOobOffset, pk);
        }
      }
      wave_lds_fence();
    }
  }
  LCE_SPH(63);
}

}  // namespace lce
