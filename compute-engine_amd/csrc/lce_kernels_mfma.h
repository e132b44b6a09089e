// Matrix-core engine for LceBconv2d on gfx950.
//
// Why it exists: the XOR-popcount formulation is bound by the integer VALU, and on gfx950
// v_bcnt_u32_b32 issues at half rate -- the measured ceiling of the v_xor+v_bcnt pair is
// 8.1e14 binary MAC/s (profiles/r01/valu_peak_microbench.jsonl), 4.4x above the HBM time of
// the BASELINE layer.  The same sum is the +-1 dot product
//       sum_k popcount(a_k ^ w_k) = (K - <a, w>) / 2        (output_transform.h:62-91)
// and the CDNA4 matrix cores evaluate <a, w> EXACTLY for +-1 operands stored as FP4 (E2M1:
// +1 = 0x2, -1 = 0xA, 0 = 0x0) with fp32 accumulation (every partial sum is an integer of
// magnitude < 2^24), at 4.2e15 MAC/s measured (tools/probes/mfma_fp4_peak.hip).  So this
// engine is bit-exact to the xor-popcount one and ~5x higher in ceiling.
//
// Two variants of one kernel template (bconv2d_mfma<..., DIRECT>):
//
//   DIRECT (the default; "direct variant" below): a block's tile is BM consecutive output
//      pixels of one image (or several whole small images).  The block reads the bitpacked
//      input rows the tile needs once, expands them to FP4 in registers and keeps them in
//      LDS for the whole K loop; only the weights stream through an LDS-DMA ring.  One launch
//      per call, no workspace.
//
//   workspace GEMM (engine=mfma; chosen when the LDS halo would not leave room for two blocks
//      per CU, or too little of a tile would be real pixels):
//      1. expand_fp4: bitpacked activations [B,H,W,Cw] -> FP4, spatially PADDED workspace
//         laid out as word planes [Cpad/32][B,Hp,Wp][16 bytes] (one 16-byte element = the 32
//         channels of one input word), so that 64 consecutive pixels of one K-half are one
//         contiguous KiB; border pixels hold +1 (pad_values 1) or 0 (exact SAME-zero padding:
//         an outside tap then contributes 0 to <a,w>, which is what reference.h:100-103 adds
//         as (Cin/G)/2 in popcount units); channels >= Cin hold 0.  After this no kernel
//         needs a bounds check.
//      2. bconv2d_mfma: implicit GEMM over M = B*OH*OW pixels; A and B tiles both go
//         global -> LDS through the ring.
//
// Common: N = Cout, K = KH*KW*Cpad in K-steps of 64 (one v_mfma_scale_f32_32x32x64_f8f6f4
// deep).  A block is WGM x WGN waves, each wave owns WM x WN MFMA tiles of 32x32; operands
// arrive by asynchronous LDS-DMA (buffer_load_dwordx4 ... lds, no VGPR round trip) into a
// 3- or 4-stage ring, fragments are read with ds_read_b128, and the output transform
// (output_transform.h:93-168) is fused on the fp32 accumulators:
//         2*accum (= K_bt - <a,w>, accumulated directly: weights negated, start value K_bt)
//                 ->  clamp -> * mul + bias   (two roundings)
#pragma once
#include <lce_device_intrinsics.h>
#include "lce_kernel_args.h"
#include "lce_kernels.h"

namespace lce {

// ---------------------------------------------------------------------------------
// Step 1: bitpacked -> FP4 padded workspace.  One thread per 16-byte output chunk
// (= 32 channels = one input word).
// ---------------------------------------------------------------------------------
LCE_DEVICE uint32_t spread8_to_nibbles(uint32_t bits8) {
  uint32_t x = bits8 & 0xffu;
  x = (x | (x << 12)) & 0x000f000fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x;  // bit i of the input sits at bit 4*i
}

LCE_DEVICE u32x4 fp4_of_word(uint32_t word, int valid) {
  u32x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cnt = valid - 8 * q;
    const uint32_t mask = cnt >= 8 ? 0xffffffffu : (cnt <= 0 ? 0u : ((1u << (4 * cnt)) - 1u));
    v[q] = ((spread8_to_nibbles(word >> (8 * q)) << 3) | 0x22222222u) & mask;
  }
  return v;
}

// All 32 channels present: 17 VALU instructions per word instead of ~40.  Byte q of the word
// becomes one dword of 8 nibbles: its four bit PAIRS are moved to the low 2 bits of the four
// bytes of a selector (perm + packed 16-bit shifts), and v_perm_b32 then looks each pair up
// in the 4-entry table {++, -+, +-, --} = {0x22, 0x2A, 0xA2, 0xAA}.
LCE_DEVICE u32x4 fp4_of_full_word(uint32_t word) {
  const uint32_t w2 = word >> 2;
  u32x4 v;
  auto one = [&](uint32_t sel) LCE_LAMBDA_INLINE {
    // bytes: [b, b >> 2, b, b >> 2] -> halves shifted by (0, 4) -> pairs 0..3 in the low bits
    const uint32_t u = pk_lshr_b16<0, 4>(perm_b32(w2, word, sel)) & 0x03030303u;
    return perm_b32(0u, 0xAAA22A22u, u);
  };
  v[0] = one(0x04000400u); v[1] = one(0x05010501u); v[2] = one(0x06020602u); v[3] = one(0x07030703u);
  return v;
}

// One thread per (workspace pixel, group of 4 input words): a 16-byte read of the pixel's
// words (when the row allows it) feeds four 16-byte writes into four consecutive word
// planes; consecutive threads are consecutive pixels, so every plane write is coalesced.
template <int UNUSED = 0>     // (a template so that only the translation unit that launches it emits it)
LCE_KERNEL void __launch_bounds__(256)
expand_fp4(const uint32_t* __restrict__ in, u32x4* __restrict__ out, const MfmaArgs G, uint64_t total) {
  const uint64_t stride = (uint64_t)grid_dim_x() * (uint64_t)block_dim_x();
  const bool vec = (G.Cw & 3) == 0;
  for (uint64_t e = (uint64_t)block_idx_x() * (uint64_t)block_dim_x() + (uint64_t)thread_idx_x();
       e < total; e += stride) {
    const int qg = (int)fastdiv((uint32_t)e, G.div_npix);    // word-plane group; total < 2^31
    const uint32_t pix = (uint32_t)e - (uint32_t)qg * G.NPIX;
    const uint32_t rowp = fastdiv(pix, G.div_wp);            // b * Hp + yp
    const int xp = (int)(pix - rowp * (uint32_t)G.Wp);
    const uint32_t b = fastdiv(rowp, G.div_hp);
    const int yp = (int)(rowp - b * (uint32_t)G.Hp);
    const int iy = yp - G.PH, ix = xp - G.PW;
    const bool inside = (uint32_t)iy < (uint32_t)G.H && (uint32_t)ix < (uint32_t)G.W;
    const int c0 = qg * 4;
    uint32_t w[4] = {0u, 0u, 0u, 0u};                        // outside the image: bit 0 = +1 (pad_values 1)
    if (inside) {
      const uint32_t* src = in + (((size_t)b * G.H + iy) * G.W + ix) * (size_t)G.Cw + c0;
      if (vec && c0 + 4 <= G.Cw) {
        const u32x4 v = *(const u32x4*)src;
        w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c0 + k < G.Cw) w[k] = src[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cc = c0 + k;
      if (cc >= G.CPW) break;
      int valid = G.Cin - cc * 32;                             // channels of this word that exist
      valid = valid < 0 ? 0 : (valid > 32 ? 32 : valid);
      if (!inside && G.zero_border) valid = 0;
      out[(size_t)cc * G.NPIX + pix] = valid == 32 ? fp4_of_full_word(w[k]) : fp4_of_word(w[k], valid);
    }
  }
}

// ---------------------------------------------------------------------------------
// Step 2: the GEMM.
//   xp : FP4 workspace, 16-byte element (plane cc, pixel (b, yp, xp)) at
//        (cc * NPIX + (b*Hp + yp)*Wp + xp) * 16 bytes; K-step kc uses planes 2kc, 2kc+1
//   wq : FP4 weights [KS][2 halves][Npad][16 bytes]
//   wq : ... holding the NEGATED weights: the accumulators start at K_bt and end as 2 * accum
//   thrf : bitpacked output: per-channel float t = 2 * threshold with  bit = (2 * accum > t)
// ---------------------------------------------------------------------------------
#ifdef LCE_TIMELINE
// Profiling aid (tools/timeline.py builds the library with -DLCE_TIMELINE; never defined in the
// product build): s_memtime stamps of the K loop of two blocks from the middle of the grid,
// [block][wave][K-step][4 stamps].  A stamp waits for lgkmcnt, so a segment that issues
// ds_reads includes their latency.
__device__ unsigned long long lce_timeline[2 * 8 * 80 * 4];
#define LCE_TL(slot)                                                                          \
  do {                                                                                        \
    if (tl_on && ks < 80 && lane == 0)                                                        \
      lce_timeline[((tl_blk * 8 + wave) * 80 + ks) * 4 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define LCE_TL(slot) do {} while (0)
#endif
#ifdef LCE_PHASES
// Profiling aid (tools/phases.py builds the library with -DLCE_PHASES; never defined in the product
// build): six s_memtime stamps per BLOCK (wave 0, lane 0) -- entry, halo in LDS, K loop entered, K loop
// done, epilogue issued -- plus the hardware id of the CU it ran on, for all blocks of the launch.
__device__ unsigned long long lce_phase_tl[16384 * 16];
#define LCE_PH(slot)                                                                              \
  do {                                                                                            \
    if (thread_idx_x() == 0 && ph_lin < 16384u)                                                   \
      lce_phase_tl[ph_lin * 16 + (slot)] = (slot) == 7                                             \
          ? (((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32) |            \
                (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4)                     \
          : __builtin_readcyclecounter();                                                         \
  } while (0)
#else
#define LCE_PH(slot) do {} while (0)
#endif
#ifndef LCE_STORE_PACE
#define LCE_STORE_PACE 2   // s_sleep argument behind every float row store of the joint-transpose epilogue
#endif
template <int V> struct IntC { static constexpr int value = V; };

// The 32 channel bits of every pixel row of NJ accumulator tiles -> the lanes that store them: register r of a tile holds pixel rows q
// (lanes 0-31) and q + 4 (lanes 32-63), q = (r & 3) + 8 * (r >> 2); one compare + ballot per register and tile, v_writelane drops each
// half into lane q / q + 4 of words[j] -- afterwards lane p < 32 owns row p.  BELOW: bit = value < thr (the second output: LceQuantize of
// the value), else bit = value > thr (bitpacked output: accumulator > threshold, output_transform.h:160-168).
// A v_writelane must not read an SGPR that a VALU compare wrote less than 4 wait states ago (stale lanes on gfx950, round 2).  Until
// round 6 the pointwise and weight-streaming kernels padded every register's compares with an `s_nop 4` -- 16 per tile, 47 of the 159
// instructions of the pointwise kernel's second-output path counting the compiler's own.  Here the ballots of register r reach their
// lanes TWO registers later (the idiom of lce_kernels_stream.h, `hold_until`): a register's lane writes are held behind the compares of
// the register two further on and -- through the chain of words[j] -- behind the lane writes of the two registers before it: in source
// order eight or more instructions from its own compares, at worst (the compiler may hoist the independent compares) four lane writes +
// the one wait state its first lane write carries.  One fully padded point remains, for the last register.
template <int NJ, bool BELOW>
LCE_DEVICE void gather_tile_bits(const f32x16 (&a)[NJ], const float (&thr)[NJ], uint32_t (&words)[NJ]) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) words[j] = 0u;
  unsigned long long p1[NJ], p2[NJ];        // the ballots of registers r - 1 and r - 2
#pragma unroll
  for (int j = 0; j < NJ; ++j) p1[j] = p2[j] = 0ull;
  auto flush = [&](auto rc, unsigned long long (&pend)[NJ]) LCE_LAMBDA_INLINE {
    constexpr int r = decltype(rc)::value, q = (r & 3) + 8 * (r >> 2);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      // (the first lane write of a word carries one wait state: whatever order the compiler gives the independent compares, these
      //  ballots' compares lie behind at least the four lane writes of registers r - 2 and r - 1 plus this one)
      words[j] = write_lane_settled_pad1<q>((uint32_t)pend[j], words[j]);
      words[j] = write_lane_settled<q + 4>((uint32_t)(pend[j] >> 32), words[j]);
    }
  };
  auto unit = [&](auto rc) LCE_LAMBDA_INLINE {
    constexpr int r = decltype(rc)::value;
    unsigned long long bits[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bits[j] = wave_ballot(BELOW ? a[j][r] < thr[j] : a[j][r] > thr[j]);
    if constexpr (r >= 2) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) hold_until(p2[j], bits[0], bits[NJ - 1]);   // register r - 2's lane writes: behind THESE compares
      flush(IntC<(r >= 2 ? r - 2 : 0)>{}, p2);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { p2[j] = p1[j]; p1[j] = bits[j]; }
  };
  unit(IntC<0>{}); unit(IntC<1>{}); unit(IntC<2>{}); unit(IntC<3>{}); unit(IntC<4>{}); unit(IntC<5>{}); unit(IntC<6>{}); unit(IntC<7>{});
  unit(IntC<8>{}); unit(IntC<9>{}); unit(IntC<10>{}); unit(IntC<11>{}); unit(IntC<12>{}); unit(IntC<13>{}); unit(IntC<14>{}); unit(IntC<15>{});
  settle_ballots(p1);                       // (register 15's compares are only 2 * NJ lane writes away: the one padded point)
  flush(IntC<14>{}, p2);
  flush(IntC<15>{}, p1);
}
struct StepSteady { static constexpr bool value = true; };
struct StepTail { static constexpr bool value = false; };

template <int DST, int WGM, int WGN, int WM, int WN, bool CORR = false, bool DIRECT = false,
          int STAGES = DIRECT ? 3 : 4, bool TILE2D = false>
LCE_KERNEL void __launch_bounds__(64 * WGM * WGN, 2)
bconv2d_mfma(const ConvArgs A, const MfmaArgs G, const uint8_t* __restrict__ xp,
             const uint8_t* __restrict__ wq, const float* __restrict__ mul,
             const float* __restrict__ bias, const float* __restrict__ thrf,
             const float* __restrict__ zpc, void* __restrict__ out, uint32_t* __restrict__ sign_words) {
  // WGM x WGN waves per block, each owning WM x WN MFMA tiles of 32x32
  constexpr int NWAVES = WGM * WGN;
  constexpr int BM = 32 * WM * WGM, BN = 32 * WN * WGN;
  // DIRECT: the A operand is not staged per K-step at all (see "direct variant" below)
  constexpr int A_BYTES = DIRECT ? 0 : BM * 32, B_BYTES = BN * 32, STAGE = A_BYTES + B_BYTES;
  // LDS image of a stage: A as [k-half][row][16 B], B as [k-half][channel][16 B]; a wave
  // fills it in 1-KiB pieces (64 rows of one half) with one LDS-DMA instruction each.
  constexpr int A_PIECES = BM / 32;
  static_assert(BM % 64 == 0 && BN % 64 == 0, "tiles are filled in 64-row pieces");

#ifdef LCE_TIMELINE
  const uint32_t tl_lin = (uint32_t)block_idx_y() * (uint32_t)grid_dim_x() + (uint32_t)block_idx_x();
  const uint32_t tl_first = ((uint32_t)grid_dim_x() * (uint32_t)gridDim.y) / 2u;
  const bool tl_on = tl_lin - tl_first < 2u;
  const int tl_blk = (int)(tl_lin - tl_first);
#endif
#ifdef LCE_PHASES
  const uint32_t ph_lin = (uint32_t)block_idx_y() * (uint32_t)grid_dim_x() + (uint32_t)block_idx_x();
#endif
  LCE_PH(0);
  LCE_PH(7);
  uint8_t* const lds0 = lds_base();
  uint8_t* const lds = lds0 + (DIRECT ? G.halo_bytes : 0);   // the K-step ring
  const int tid = thread_idx_x();
  const int lane = tid & (kWave - 1);
  const int wave = uniform(tid >> 6);
  const int wm = wave % WGM, wn = wave / WGM;
  // Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed, not
  // promised).  Remapping so that an XCD works on a CONTIGUOUS run of pixel tiles lets
  // neighbouring tiles share their halo rows in that XCD's L2.  Pure speed: any placement
  // computes the same thing.
  int bx = block_idx_x();
  {
    const int nb = grid_dim_x(), per = nb >> 3;
    if (bx < per * 8) bx = (bx & 7) * per + (bx >> 3);   // the last nb % 8 blocks keep their index
  }
  // GEMM variant: tile = BM consecutive pixels of the whole batch.  Direct variant: tile =
  // BM consecutive pixels of ONE image (the last tile of an image is partial).
  int m0 = bx * BM, m_end = A.M, p0 = 0, img = 0;
  int tile_oy0 = 0, tile_ox0 = 0;          // 2-D tiles (TILE2D; G.TX > 0): first output row / column of the tile
  constexpr bool tile2d = TILE2D;
  static_assert(!TILE2D || DIRECT, "2-D tiles belong to the direct variant");
  if constexpr (DIRECT) {
    if (G.IPT > 1) {                       // small images: IPT whole images per tile
      img = bx * G.IPT;
      m0 = img * G.OHOW;
      m_end = (img + G.IPT < G.B ? img + G.IPT : G.B) * G.OHOW;
    } else {
      img = (int)fastdiv((uint32_t)bx, G.div_tpi);
      const int t = bx - img * G.TPI;
      if constexpr (TILE2D) {              // wide images: BM/32 output rows x 32 output columns
        const int ty = (int)fastdiv((uint32_t)t, G.div_tx);
        tile_oy0 = ty * (BM / 32);
        tile_ox0 = (t - ty * G.TX) * 32;
        p0 = tile_oy0 * A.OW;              // (only its row, oy0 below, is used)
      } else {
        p0 = t * BM;
      }
      m0 = img * G.OHOW + p0;
      m_end = (img + 1) * G.OHOW;
    }
  }
  // The tile's 32-row blocks (one per MFMA row tile of this wave): first output pixel and how many of the 32 are
  // real.  Strip tiles: consecutive pixels up to the end of the image(s); 2-D tiles: one image-row segment each.
  auto blk_m = [&](int i) LCE_LAMBDA_INLINE -> int {
    if constexpr (TILE2D) return img * G.OHOW + (tile_oy0 + wm * WM + i) * A.OW + tile_ox0;
    else return m0 + (wm * WM + i) * 32;
  };
  auto blk_valid = [&](int i) LCE_LAMBDA_INLINE -> int {
    if constexpr (TILE2D) {
      const int left = A.OW - tile_ox0;
      return tile_oy0 + wm * WM + i < A.OH ? (left < 32 ? left : 32) : 0;
    } else {
      const int left = m_end - (m0 + (wm * WM + i) * 32);
      return left < 0 ? 0 : (left < 32 ? left : 32);
    }
  };
  const int n0 = block_idx_y() * BN;
  // grouped convolution: first 64-channel chunk of this block's group (0 when groups == 1: Npg == N > n0)
  const int chunk0 = ((int)fastdiv((uint32_t)n0, G.div_npg) * A.Cwg) >> 1;

  const rsrc_t rx = make_rsrc(xp, G.x_bytes);
  const rsrc_t rw = make_rsrc(wq, G.w_bytes);

  // ---- this wave's share of the staging work ------------------------------------------
  // Slot i of every wave is statically an A piece (i < NPA) or a B piece, so the inner
  // loop needs no per-piece descriptor select.  When the piece count is not a multiple of
  // the wave count the surplus slots re-copy an earlier piece (same bytes, same address).
  constexpr int B_PIECES = BN / 32;
  constexpr int NPA = DIRECT ? 0 : (A_PIECES + NWAVES - 1) / NWAVES, NPB = (B_PIECES + NWAVES - 1) / NWAVES;
  constexpr int NP = NPA + NPB;  // LDS-DMA instructions per wave per K-step
  uint32_t a_src[NPA + 1], b_src[NPB];   // per-lane byte offsets at K-step 0
  int a_dst[NPA + 1], b_dst[NPB];        // wave-uniform LDS byte offsets inside a stage
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int p = (wave + i * NWAVES) % A_PIECES;
    const int half = p / (BM / 64), blk = p % (BM / 64);
    a_dst[i] = half * (BM * 16) + blk * 1024;
    int m = m0 + blk * 64 + lane;
    m = m < m_end ? m : m_end - 1;  // tail rows re-read the last pixel; their results are not stored
    const uint32_t rw_ = fastdiv((uint32_t)m, A.div_ow);
    const int ox = m - (int)rw_ * A.OW;
    const uint32_t b = fastdiv(rw_, A.div_oh);
    const int oy = (int)(rw_ - b * (uint32_t)A.OH);
    a_src[i] = ((uint32_t)half * G.NPIX + (uint32_t)(((int)b * G.Hp + oy * A.SH) * G.Wp + ox * A.SW)) * 16u;
  }
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int q = (NPB == 2 && B_PIECES == 2 * NWAVES) ? 2 * wave + i : (wave + i * NWAVES) % B_PIECES;
    const int half = q / (BN / 64), blk = q % (BN / 64);
    b_dst[i] = A_BYTES + half * (BN * 16) + blk * 1024;
    b_src[i] = (uint32_t)(half * G.Npad + n0 + blk * 64 + lane) * 16u;
  }

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = f32x16_fill(G.a_bt);   // + <a, -w> = K_bt - <a, w> = 2 * accum

  const int KS = A.KH * A.KW * G.KCH;

  // K-step ks = (tap (fy, fx), channel chunk kc).  fill() is always called with
  // consecutive ks, so the operand offsets advance incrementally (a handful of scalar ops
  // per K-step instead of divisions).
  uint32_t a_off = 0, b_off = 0;    // byte offsets of the NEXT K-step to be filled
  int f_kc = 0, f_fx = 0;
  const uint32_t a_step_kc = 2u * G.NPIX * 16u;
  a_off = (uint32_t)chunk0 * a_step_kc;
  const uint32_t a_step_fx = (uint32_t)A.DW * 16u - (uint32_t)G.KCH * a_step_kc;
  const uint32_t a_step_fy = (uint32_t)(A.DH * G.Wp - A.KW * A.DW) * 16u;
  const uint32_t b_step = (uint32_t)G.Npad * 32u;
  // B pieces of a wave: when every wave has exactly two (BN = 256 with four waves), they are NEIGHBOURS
  // (pieces 2w and 2w+1: 1 KiB apart in the weights and in the stage), so the second copy is the first
  // one's instruction with an immediate offset -- same address VGPR, same M0.  The K-step's offset rides
  // in the instruction's scalar offset: no VALU.
  constexpr bool PAIRED_B = NPB == 2 && B_PIECES == 2 * NWAVES;
  auto fill = [&](int stage) {
    uint8_t* base = lds + stage * STAGE;
#ifndef LCE_ABL_NODMA   // timing ablation (results are wrong): the K loop without its LDS-DMA instructions
#pragma unroll
    for (int i = 0; i < NPA; ++i) buf_load_to_lds16<0>(rx, base + a_dst[i], a_src[i], a_off);
    if constexpr (PAIRED_B) {
      buf_load_to_lds16<0>(rw, base + b_dst[0], b_src[0], b_off);
      buf_load_to_lds16<1024>(rw, base + b_dst[0], b_src[0], b_off);
    } else {
      // A 1-KiB copy occupies the CU's load path for ~25-30 cycles and blocks the wave that issued it meanwhile
      // (tools/probes/dma_cost.hip): in the direct variant a block with fewer pieces than waves (256x128: four
      // pieces, eight waves) does not issue the surplus -- there a wave copies either its one piece or nothing,
      // so the counted waits stay exact (a wave that copied nothing has nothing to wait for).  In the workspace
      // variant the surplus slots re-copy an earlier piece: a wave that dropped only its B slot would still
      // wait as if it had issued NP.
#pragma unroll
      for (int i = 0; i < NPB; ++i)
        if (!(DIRECT && NPB == 1) || wave < B_PIECES) buf_load_to_lds16<0>(rw, base + b_dst[i], b_src[i], b_off);
    }
#endif
    // cursor to the next K-step, branch-free (selects, not jumps: a branch here would split the
    // K-step's basic block and with it the scheduling region the MFMA interleave needs)
    b_off += b_step;
    ++f_kc;
    const bool wrap_kc = f_kc == G.KCH;
    f_kc = wrap_kc ? 0 : f_kc;
    f_fx += wrap_kc ? 1 : 0;
    const bool wrap_fx = f_fx == A.KW;
    f_fx = wrap_fx ? 0 : f_fx;
    a_off += a_step_kc + (wrap_kc ? a_step_fx : 0u) + (wrap_fx ? a_step_fy : 0u);
  };

  // STAGES-deep LDS ring + double-buffered fragment registers:
  //   * the LDS-DMA of K-step ks+STAGES is issued during K-step ks, so ~3 K-steps of MFMA
  //     work cover the DMA latency (~1 us under load);
  //   * the fragments of K-step ks+1 are read from LDS while the MFMAs of K-step ks run,
  //     so the matrix pipe restarts right after each barrier instead of waiting for LDS;
  //   * the barrier in front of each K-step does not drain vmcnt: a wave waits (counted
  //     vmcnt) only until ITS pieces of step ks+1 have landed, then the barrier makes all
  //     waves' pieces visible and proves that stage ks % STAGES (whose fragments every wave
  //     has finished reading) can be refilled.
  const int half = lane >> 5, l31 = lane & 31;

  // start the pipeline: the first STAGES K-steps are on their way before anything else happens
  // (direct variant: right behind the halo's global loads, see below)
  if constexpr (!DIRECT) {
#pragma unroll
    for (int d = 0; d < STAGES; ++d)
      if (d < KS) fill(d);
  }

  // ---- direct variant: the tile's input halo, expanded to FP4 once, stays in LDS ---------
  // The GEMM variant re-fetches every pixel of the tile once per filter tap (KH*KW times,
  // 128 bytes each) through LDS-DMA, and needs a separate pass over the whole tensor to
  // produce that FP4 image.  Here a block reads the bitpacked rows its tile needs (tile rows
  // + filter extent) ONCE, expands them in registers and keeps the result in LDS as
  //     halo[slot row][padded x][PS bytes],  PS = 16 bytes per input word + 16;
  // the A fragment of pixel (oy, ox) for tap (fy, fx), chunk kc is then the 16 bytes at
  //     ((oy - oy0)*SH + fy*DH) * Wp + ox*SW + fx*DW  pixels, + kc*32 + k-half*16 bytes:
  // a per-lane base plus a wave-uniform cursor.  The +16 in PS makes 16 consecutive pixels
  // hit 16 different bank groups (PS/4 = 4 mod 8 dwords).  No FP4 workspace, no A traffic in
  // the K loop; only the weights stream through the ring.
  uint32_t a_base[WM];
  uint32_t a_cur = (uint32_t)chunk0 * 32u;   // byte offset of the K-step whose fragments are read next
  int c_kc = 0, c_fx = 0;
  if constexpr (DIRECT) {
    const int oy0 = tile2d ? tile_oy0 : (int)fastdiv((uint32_t)p0, A.div_ow);
    const int iy_first = oy0 * A.SH - G.PH, ix_first = tile_ox0 * A.SW - G.PW;   // input pixel of halo slot (0, 0)
    const int halo_w = TILE2D ? G.Wh : G.Wp;                                      // halo pixels per slot row
    const uint32_t img_bytes = (uint32_t)G.H * (uint32_t)G.W * (uint32_t)G.Cw * 4u;
    // the rows come through a buffer resource over [first image of the tile, end of the launch's
    // input): a tap outside the image (or an image past the batch) is steered to an out-of-range
    // offset and reads 0 = the "+1" padding word, so the loads need no branches
    const rsrc_t rin = make_rsrc(xp + (size_t)img * img_bytes, (uint32_t)(G.B - img) * img_bytes);
    const int items = G.IPT * G.HPIX * G.QG;
    const bool vec = (G.Cw & 3) == 0;
    // Loads first, arithmetic second: a thread's (up to) PRE 16-byte loads are all in flight
    // before the first one is consumed, so the block pays one memory latency for its halo
    // instead of one per item.  The weight ring's first fills are issued BEHIND these loads:
    // the memory counter retires in order, so waiting for the halo rows does not wait for them.
    constexpr int PRE = 4, NT = 64 * NWAVES;
    LCE_PH(8);
    // FAST (compile-time): every item is four full words (Cin % 128 == 0, no zero border) -- the
    // BASELINE layers -- so neither the loads nor the expansion carry per-word conditions.
    // MODE 2 = FAST as above; MODE 1 (round 3) = every EXISTING word is a full word (Cin % 32 == 0, no zero border) but a
    // pixel has fewer than four of them -- QuickNet's 64-channel layers, two words per pixel: no per-word channel counts,
    // and the two words come in ONE 8-byte load instead of four conditional 4-byte loads; MODE 0 = the general path.
    auto halo_pass = [&](auto mode_c) LCE_LAMBDA_INLINE {
      constexpr int MODE = decltype(mode_c)::value;
      constexpr bool FAST = MODE == 2;
      const bool two_words = MODE == 1 && G.Cw == 2;
      for (int e0 = 0; e0 < items; e0 += PRE * NT) {
        u32x4 wv[PRE];
        int pixv[PRE], c0v[PRE];
        bool inv[PRE];
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
          pixv[k] = -1;
#ifndef LCE_NO_HALO_SKIP   // (A/B aid)
          if (e0 + k * NT >= items) continue;      // (block-uniform) a whole round of threads past the halo: 56x56x64 fills 406 of 1024
#endif
          const int e = e0 + k * NT + tid;
          const int pix = (int)fastdiv((uint32_t)e, G.div_qg);     // (image * halo_rows + slot) * Wp + x
          const int c0 = (e - pix * G.QG) * 4;
          const int li = (int)fastdiv((uint32_t)pix, G.div_hpix);  // image of the tile (0 unless IPT > 1)
          const int ipix = pix - li * G.HPIX;
          const int slot = (int)fastdiv((uint32_t)ipix, TILE2D ? G.div_wh : G.div_wp);
          const int iy = iy_first + slot, ix = ipix - slot * halo_w + ix_first;
          const bool inside = e < items && (uint32_t)iy < (uint32_t)G.H && (uint32_t)ix < (uint32_t)G.W && img + li < G.B;
          pixv[k] = e < items ? pix : -1; c0v[k] = c0; inv[k] = inside;
          const uint32_t off = (uint32_t)li * img_bytes + (uint32_t)((iy * G.W + ix) * G.Cw + c0) * 4u;
          if (FAST || vec) {
            wv[k] = buf_load(rin, inside ? off : kOobOffset, (u32x4*)nullptr);
          } else if (two_words) {
            const u32x2 t = buf_load(rin, inside ? off : kOobOffset, (u32x2*)nullptr);
            wv[k] = u32x4{t[0], t[1], 0u, 0u};
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              wv[k][q] = buf_load(rin, inside && c0 + q < G.Cw ? off + 4u * q : kOobOffset, (uint32_t*)nullptr);
          }
        }
        if (e0 == 0) {
          LCE_PH(9);
#pragma unroll
          for (int d = 0; d < STAGES; ++d)
            if (d < KS) fill(d);
          LCE_PH(10);
        }
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
#ifdef LCE_PHASES
          if (e0 == 0 && k == 1) LCE_PH(11);
#endif
          if (pixv[k] < 0) continue;
          uint8_t* dst = lds0 + (size_t)pixv[k] * G.PS + c0v[k] * 16;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if constexpr (FAST) {
              *(u32x4*)(dst + q * 16) = fp4_of_full_word(wv[k][q]);
            } else if constexpr (MODE == 1) {
              const int cc = c0v[k] + q;       // (a word of the 64-channel padding that no real channel fills: FP4 zeros)
              if (cc < G.CPW) *(u32x4*)(dst + q * 16) = cc < G.Cw ? fp4_of_full_word(wv[k][q]) : u32x4{0u, 0u, 0u, 0u};
            } else {
              const int cc = c0v[k] + q;
              if (cc < G.CPW) {
                int valid = G.Cin - cc * 32;                         // channels of this word that exist
                valid = valid < 0 ? 0 : (valid > 32 ? 32 : valid);
                if (!inv[k] && G.zero_border) valid = 0;             // exact SAME-zero: 0 contributes 0
                *(u32x4*)(dst + q * 16) = valid == 32 ? fp4_of_full_word(wv[k][q]) : fp4_of_word(wv[k][q], valid);
              }
            }
          }
        }
      }
    };
    if ((G.Cin & 127) == 0 && !G.zero_border) halo_pass(IntC<2>{});
#ifndef LCE_NO_HALO_FULLWORDS   // (A/B aid)
    else if ((G.Cin & 31) == 0 && !G.zero_border) halo_pass(IntC<1>{});
#endif
    else halo_pass(IntC<0>{});
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      int p = p0 + (wm * WM + i) * 32 + l31;   // pixel of the tile's first image, or beyond it
      const int lim = G.IPT * G.OHOW;
      p = p < lim ? p : lim - 1;               // rows past the tile re-read its last pixel; never stored
      const int li = (int)fastdiv((uint32_t)p, G.div_ohow);
      p -= li * G.OHOW;
      const int oy = (int)fastdiv((uint32_t)p, A.div_ow);
      const int ox = p - oy * A.OW;
      a_base[i] = (uint32_t)((li * G.HPIX + (oy - oy0) * A.SH * G.Wp + ox * A.SW) * G.PS + half * 16);
      if constexpr (TILE2D)   // row block rb of the tile = tile row rb, this lane's pixel = column l31 (past the image: unused, in range)
        a_base[i] = (uint32_t)(((wm * WM + i) * A.SH * G.Wh + l31 * A.SW) * G.PS + half * 16);
    }
  }
  LCE_PH(1);
  constexpr uint32_t kAStep = 32u;               // bytes of halo per K-step and pixel: 64 FP4 codes
  const uint32_t c_step_fx = (uint32_t)(A.DW * G.PS) - (uint32_t)G.KCH * kAStep;
  const uint32_t c_step_fy = (uint32_t)((A.DH * (TILE2D ? G.Wh : G.Wp) - A.KW * A.DW) * G.PS);

  // A-operand cursor of the direct variants: advances by one K-step (branch-free, as in fill())
  auto advance_a = [&]() LCE_LAMBDA_INLINE {
    ++c_kc;
    const bool wrap_kc = c_kc == G.KCH;
    c_kc = wrap_kc ? 0 : c_kc;
    c_fx += wrap_kc ? 1 : 0;
    const bool wrap_fx = c_fx == A.KW;
    c_fx = wrap_fx ? 0 : c_fx;
    a_cur += kAStep + (wrap_kc ? c_step_fx : 0u) + (wrap_fx ? c_step_fy : 0u);
  };
  // load_a / load_b are always called for consecutive K-steps (0, 1, 2, ...), once each;
  // `stage` = ks % STAGES
  auto load_a = [&](int stage, u32x4 (&af)[WM]) LCE_LAMBDA_INLINE {
    if constexpr (DIRECT) {
#pragma unroll
      for (int i = 0; i < WM; ++i) af[i] = *(const u32x4*)(lds0 + (a_base[i] + a_cur));
      advance_a();
    } else {
      const uint8_t* base = lds + stage * STAGE;
#pragma unroll
      for (int i = 0; i < WM; ++i)
        af[i] = *(const u32x4*)(base + half * (BM * 16) + ((wm * WM + i) * 32 + l31) * 16);
    }
  };
  auto load_b = [&](int stage, u32x4 (&bf)[WN]) LCE_LAMBDA_INLINE {
    const uint8_t* base = lds + stage * STAGE;
#pragma unroll
    for (int j = 0; j < WN; ++j)
      bf[j] = *(const u32x4*)(base + A_BYTES + half * (BN * 16) + ((wn * WN + j) * 32 + l31) * 16);
  };
  // One K-step.  `steady` (compile-time) = the ring is full: a refill is due and exactly
  // STAGES-1 younger fills are in flight, so the wait count is exact and nothing branches.
  // fs = ks % STAGES (the stage this step vacates and refills), rs = (ks + 1) % STAGES (the stage
  // the next step's fragments are read from): compile-time constants in the unrolled steady loop
  auto step = [&](auto steady, int ks, int fs, int rs, u32x4 (&af)[WM], u32x4 (&bf)[WN], u32x4 (&af_next)[WM],
                  u32x4 (&bf_next)[WN]) LCE_LAMBDA_INLINE {
    if constexpr (decltype(steady)::value) {
      LCE_TL(0);
      wait_vmcnt<NP * (STAGES - 2)>();          // own pieces of step ks+1 have landed
#ifndef LCE_ABL_NOBAR   // timing ablation (results are wrong): no barrier in the steady K-step
      block_barrier_keep_vm();
#endif
      LCE_TL(1);
      LCE_TL(2);
    } else {
      wait_vmcnt<0>();                           // tail: fewer fills in flight than the exact count
      block_barrier_keep_vm();
      if (ks + STAGES < KS) fill(fs);
      if (ks + 1 < KS) {
        load_a(rs, af_next);
        load_b(rs, bf_next);
      }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = mfma_fp4_32x32x64(af[i], bf[j], acc[i][j]);
    if constexpr (decltype(steady)::value) {
      // everything that prepares LATER steps comes after the MFMAs in program order and is woven
      // between them: the wave's matrix work starts right behind the barrier, and the expensive
      // issues (LDS-DMA, ds_read) overlap the MFMAs' execution instead of preceding it
      fill(fs);                                  // step ks+STAGES into the stage just vacated
#ifndef LCE_ABL_NOFRAG   // timing ablation (results are wrong): no fragment reads in the steady K-step
      load_a(rs, af_next);
      load_b(rs, bf_next);
#endif
      interleave_step<WM * WN, WM + WN, NP>();
    } else {
      interleave_mfma_ldsread<WM + WN>();        // tail: only fragment reads ride between the MFMAs
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) pin(acc[i][j]);  // ... and the MFMAs stay in front of the next barrier
    if constexpr (decltype(steady)::value) LCE_TL(3);
  };

  if (STAGES <= KS) wait_vmcnt<NP * (STAGES - 1)>();
  else wait_vmcnt<0>();
  block_barrier_keep_vm();
  u32x4 af0[WM], bf0[WN], af1[WM], bf1[WN];
  load_a(0, af0);              // stage 0 = K-step 0
  load_b(0, bf0);
  LCE_PH(2);
  // Steady loop, unrolled over one full cycle of (fragment-set parity) x (ring position) so that
  // every stage index is a literal -- no `% STAGES` arithmetic in the K-step.
  constexpr int UNROLL = (STAGES % 2 == 0) ? STAGES : 2 * STAGES;
  int ks = 0;
  auto steady_step = [&](auto uc) LCE_LAMBDA_INLINE {
    constexpr int u = decltype(uc)::value;
    if constexpr (u < UNROLL) {
      if constexpr (u % 2 == 0) step(StepSteady{}, ks + u, u % STAGES, (u + 1) % STAGES, af0, bf0, af1, bf1);
      else step(StepSteady{}, ks + u, u % STAGES, (u + 1) % STAGES, af1, bf1, af0, bf0);
    }
  };
  for (; ks + UNROLL - 1 + STAGES < KS; ks += UNROLL) {   // all UNROLL steps are steady
    steady_step(IntC<0>{}); steady_step(IntC<1>{}); steady_step(IntC<2>{}); steady_step(IntC<3>{});
    steady_step(IntC<4>{}); steady_step(IntC<5>{}); steady_step(IntC<6>{}); steady_step(IntC<7>{});
  }
  static_assert(UNROLL <= 8, "steady_step calls above");
  for (; ks < KS; ks += 2) {                              // ks stays even: fragment set 0 first
    step(StepTail{}, ks, ks % STAGES, (ks + 1) % STAGES, af0, bf0, af1, bf1);
    if (ks + 1 < KS) step(StepTail{}, ks + 1, (ks + 1) % STAGES, (ks + 2) % STAGES, af1, bf1, af0, bf0);
  }

  LCE_PH(3);
  // ------------------------------ fused output transform ------------------------------
  // An accumulator tile is held "one channel per lane" (column = lane & 31, 16 pixel rows in
  // 16 registers), which would make every global store a 4-byte-per-lane affair.  Instead:
  //   * float / int8: the transformed 32x32 tile is transposed through a wave-private 4-KiB
  //     LDS scratch (the pipeline ring is idle by now) so that a lane owns 4 consecutive
  //     channels of one pixel and 8 lanes write a full 128-byte line with 16-byte stores;
  //   * bitpacked: the 32 channel bits of a pixel are one v_cmp + ballot; the words are
  //     gathered so that lane p owns pixel row p and stores its WN words at once.
  const float cminf = G.cmin, cmaxf = G.cmax;
  float mj[WN], bj[WN], tj[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = n0 + (wn * WN + j) * 32 + l31;               // < Npad: the tables are padded
    mj[j] = bj[j] = tj[j] = 0.0f;
    if constexpr (DST == kDstBitpacked) tj[j] = thrf[n];
    else { mj[j] = mul[n]; bj[j] = bias[n]; }
  }

  // Bit rows of one 32-pixel row block: register r of a tile holds pixel rows q (lanes 0-31) and q + 4
  // (lanes 32-63), q = (r & 3) + 8 * (r >> 2); the 32 channel bits of a row are one compare + ballot, and
  // v_writelane drops each word into the lane that will store it -- afterwards lane p (< 32) owns row p.
  // BITPACKED output: bit = acc > thr (accum > threshold <=> 2*accum > 2*threshold, output_transform.h:160-168);
  // second output of a float / int8 layer: the LceQuantize of the value the lane just produced, bit = y < bit_thr --
  // float: bit_thr = 0 (core/bitpacking/bitpack.h:72-110; -0.0 and NaN give 0 there and here); int8: the planner's
  // threshold for "rounds to an int8 below the zero point" (int8_below_threshold, lce_plan.cpp).
  auto bit_rows = [&](int i, auto below_zero, uint32_t* dst_words) LCE_LAMBDA_INLINE {
    constexpr bool SIGN = decltype(below_zero)::value;
    uint32_t words[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) words[j] = 0u;
    auto gather = [&](auto rc) LCE_LAMBDA_INLINE {
      constexpr int r = decltype(rc)::value, q = (r & 3) + 8 * (r >> 2);
      unsigned long long bits[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j) bits[j] = wave_ballot(SIGN ? acc[i][j][r] < G.bit_thr : acc[i][j][r] > tj[j]);
      settle_ballots(bits);                    // ONE hazard pad for the WN compares, not one per v_writelane
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        words[j] = write_lane_settled<q>((uint32_t)bits[j], words[j]);              // lane q     <- row q
        words[j] = write_lane_settled<q + 4>((uint32_t)(bits[j] >> 32), words[j]);  // lane q + 4 <- row q + 4
      }
    };
    gather(IntC<0>{}); gather(IntC<1>{}); gather(IntC<2>{}); gather(IntC<3>{});
    gather(IntC<4>{}); gather(IntC<5>{}); gather(IntC<6>{}); gather(IntC<7>{});
    gather(IntC<8>{}); gather(IntC<9>{}); gather(IntC<10>{}); gather(IntC<11>{});
    gather(IntC<12>{}); gather(IntC<13>{}); gather(IntC<14>{}); gather(IntC<15>{});
    // lane p (< 32) now owns pixel row p of the tile: WN consecutive output words
    const int m = blk_m(i) + lane;
    const int w0 = (n0 + wn * WN * 32) >> 5;
    if (TILE2D ? lane < blk_valid(i) : (lane < 32 && m < m_end)) {
      uint32_t* o = dst_words + (size_t)m * (size_t)A.Wout + (size_t)w0;
      if (WN == 4 && w0 + 4 <= A.Wout && (A.Wout & 3) == 0) {
        u32x4 v = {words[0], words[WN > 1 ? 1 : 0], words[WN > 2 ? 2 : 0], words[WN > 3 ? 3 : 0]};
        store_words(o, v);
      } else if (WN >= 2 && w0 + WN <= A.Wout && (A.Wout & 1) == 0) {
#pragma unroll
        for (int j = 0; j < WN; j += 2) {
          u32x2 v = {words[j], words[j + 1 < WN ? j + 1 : j]};
          store_words(o + j, v);
        }
      } else {
#pragma unroll
        for (int j = 0; j < WN; ++j)
          if (w0 + j < A.Wout) store_words(o + j, words[j]);
      }
    }
  };

  if constexpr (DST == kDstBitpacked) {
#pragma unroll
    for (int i = 0; i < WM; ++i) bit_rows(i, StepTail{}, (uint32_t*)out);
  } else if (DST == kDstFloat && !CORR && G.f32_wide && (((size_t)out) & 15) == 0) {
    // float, wide path: the WN tiles of a 32-row block are transposed together
    // ([32 rows][WN*32] floats of scratch), so there is one LDS fence pair per row block instead of
    // one per tile, and a store instruction covers whole row segments of WN*128 bytes.
    // Everything outside the K loop runs beside the co-resident block's MFMA stream at ~8-10 cycles per
    // instruction (tools/probes/coissue.hip), so the epilogue is written for instruction COUNT: no clamp
    // when the clamp is the identity (activation NONE), no per-store predicates -- the stores go through
    // a buffer resource over the tile's real rows, rows past it fall off its end and lanes past the last
    // channel carry an out-of-range offset -- and the LDS reads of a batch are issued together.
    constexpr int RW = WN * 32;                                  // floats per scratch row
    constexpr int LPR = RW / 4, RPI = 64 / LPR;                  // lanes per row (16 bytes each), rows per store instruction
    constexpr int NK = 32 / RPI, KB = NK < 8 ? NK : 8;           // store instructions per row block, in batches of KB
    float* scratch = (float*)(lds0 + wave * (WN * 4096));
    const uint32_t row_bytes = (uint32_t)A.N * 4u;
    // strip tiles: ONE resource over the tile's real rows (rows past it fall off its end); 2-D tiles: one per 32-row
    // block (an image-row segment each), made inside the loop
    const int tile_rows = m_end - m0 < BM ? m_end - m0 : BM;
    const rsrc_t ro_tile = make_rsrc((float*)out + (size_t)m0 * (size_t)A.N, (uint32_t)tile_rows * row_bytes);
    const int g = lane % LPR;
    const int n = n0 + wn * RW + g * 4;
    const uint32_t lane_off = n < A.N ? (uint32_t)(lane / LPR) * row_bytes + (uint32_t)n * 4u : kOobOffset;   // N % 4 == 0
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const rsrc_t ro = TILE2D ? make_rsrc((float*)out + (size_t)blk_m(i) * (size_t)A.N, (uint32_t)blk_valid(i) * row_bytes) : ro_tile;
      const uint32_t blk_off = TILE2D ? 0u : (uint32_t)((wm * WM + i) * 32) * row_bytes;
      // the transform in place (the accumulators become the outputs) ...
      if (G.noclamp) {
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = mul_then_add(acc[i][j][r], mj[j], bj[j]);
      } else {
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = mul_then_add(med3(acc[i][j][r], cminf, cmaxf), mj[j], bj[j]);
      }
      // ... the optional second output: their sign bits, packed like LceQuantize would pack them ...
      if (sign_words != nullptr) bit_rows(i, StepSteady{}, sign_words);
      // ... and the transpose through LDS
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          scratch[((r & 3) + 8 * (r >> 2) + 4 * half) * RW + j * 32 + l31] = acc[i][j][r];
      wave_lds_fence();
#pragma unroll
      for (int k0 = 0; k0 < NK; k0 += KB) {
        f32x4 y[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) y[k] = *(const f32x4*)(scratch + (lane / LPR + (k0 + k) * RPI) * RW + g * 4);
#pragma unroll
        for (int k = 0; k < KB; ++k) {
#ifdef LCE_ABL_NOSTORE   // timing ablation (results are wrong): the float epilogue without its global stores
          if (y[k][0] == 1234.5678f)
#endif
          buf_store_through(ro, lane_off + blk_off + (uint32_t)((k0 + k) * RPI) * row_bytes, y[k]);
          // pace the burst: 128 KiB per block pushed out back to back fills the CU's memory pipeline and
          // the co-resident block's weight DMAs queue behind it (its K loop 18.7k -> 16.4k cycles with the
          // pause, L0 float -2.5 %, tools/phases.py); the sleeping wave also leaves its issue slots to it
#ifndef LCE_ABL_NOSLEEP
          yield_issue_slots<LCE_STORE_PACE>();
#endif
        }
      }
      wave_lds_fence();
    }
  } else if (DST == kDstInt8 && G.i8_wide && (((size_t)out) & 15) == 0) {
    // int8, wide path: a pixel's WN*32 channels of this wave are contiguous bytes, so the WN
    // tiles of a row block are transposed together ([32 rows][WN*32] floats of scratch) and a
    // lane converts 16 consecutive channels into ONE 16-byte store: 2*WN lanes cover a row
    // segment (WN = 4: a full 128-byte line) instead of 4-byte stores in 32-byte pieces.
    constexpr int RW = WN * 32;                                  // floats per scratch row
    constexpr int GPR = WN * 2, RPI = 64 / GPR;                  // 16-channel groups per row, rows per store instruction
    float* scratch = (float*)(lds0 + wave * (WN * 4096));
    // stores through a buffer resource over the tile's real rows (no per-store predicates, as in the float path)
    const int tile_rows8 = m_end - m0 < BM ? m_end - m0 : BM;
    const rsrc_t ro8_tile = make_rsrc((int8_t*)out + (size_t)m0 * (size_t)A.N, (uint32_t)tile_rows8 * (uint32_t)A.N);
    const uint32_t lane_off8 = n0 + wn * RW + (lane % GPR) * 16 < A.N
                                   ? (uint32_t)(lane / GPR) * (uint32_t)A.N + (uint32_t)(n0 + wn * RW + (lane % GPR) * 16)
                                   : kOobOffset;                  // N % 16 == 0: whole group or nothing
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const rsrc_t ro8 = TILE2D ? make_rsrc((int8_t*)out + (size_t)blk_m(i) * (size_t)A.N, (uint32_t)blk_valid(i) * (uint32_t)A.N) : ro8_tile;
      const uint32_t blk_off8 = TILE2D ? 0u : (uint32_t)((wm * WM + i) * 32) * (uint32_t)A.N;
      if (sign_words != nullptr) {
        // with a second output the transform happens in place first (the accumulators become the values the
        // rounding will see), then their "below the zero point" bits, then the transpose
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = mul_then_add(med3(acc[i][j][r], cminf, cmaxf), mj[j], bj[j]);
        bit_rows(i, StepSteady{}, sign_words);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * half) * RW + j * 32 + l31] = acc[i][j][r];
      } else {
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float x = med3(acc[i][j][r], cminf, cmaxf);
            scratch[row * RW + j * 32 + l31] = mul_then_add(x, mj[j], bj[j]);
          }
      }
      wave_lds_fence();
      const int g = lane % GPR;
      // all LDS reads of a batch first (one LDS latency per batch, not one per store), then convert + store
      constexpr int NK = 32 / RPI, KB = NK < 4 ? NK : 4;
#pragma unroll
      for (int k0 = 0; k0 < NK; k0 += KB) {
        f32x4 y[KB][4];
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          const f32x4* src = (const f32x4*)(scratch + (lane / GPR + (k0 + k) * RPI) * RW + g * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) y[k][q] = src[q];
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          u32x4 pk;
#pragma unroll
          for (int q = 0; q < 4; q += 2) {
            f32x4 a = y[k][q], b = y[k][q + 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = med3(a[i], -128.0f, 127.0f); b[i] = med3(b[i], -128.0f, 127.0f); }
            uint32_t lo, hi;
            round_pack8_i8_clamped(a, b, lo, hi);
            pk[q] = lo;
            pk[q + 1] = hi;
          }
          buf_store(ro8, lane_off8 + blk_off8 + (uint32_t)((k0 + k) * RPI) * (uint32_t)A.N, pk);
        }
      }
      wave_lds_fence();
    }
  } else {
    float* scratch = (float*)(lds0 + wave * 4096);              // [32 rows][32 channels]
    const int trow = lane >> 3, tcol = (lane & 7) * 4;          // after the transpose
    const bool vec_ok = (A.N & 3) == 0;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int nbase = n0 + (wn * WN + j) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          const float x = med3(acc[i][j][r], cminf, cmaxf);  // = float(clamp(accum << 1))
          scratch[row * 32 + l31] = mul_then_add(x, mj[j], bj[j]);
        }
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int row = trow + 8 * k;
          const int m = blk_m(i) + row;
          const int n = nbase + tcol;
          f32x4 y = *(const f32x4*)(scratch + row * 32 + tcol);
          if ((TILE2D ? row < blk_valid(i) : m < m_end) && n < A.N) {
            if constexpr (DST == kDstFloat) {
              if constexpr (CORR) {                             // optimized_bgemm.h:153-177
                const uint32_t rw_ = fastdiv((uint32_t)m, A.div_ow);
                const int ox = m - (int)rw_ * A.OW;
                const int oy = (int)(rw_ - fastdiv(rw_, A.div_oh) * (uint32_t)A.OH);
                const int zrow = zero_pad_cache_row(A, oy, ox);
                if (zrow >= 0) {
#pragma unroll
                  for (int c = 0; c < 4; ++c)
                    if (n + c < A.N) y[c] = __fadd_rn(y[c], zpc[zrow + n + c]);
                }
              }
              float* o = (float*)out + (size_t)m * (size_t)A.N + (size_t)n;
              if (vec_ok && n + 4 <= A.N) {
                store_streaming((f32x4*)o, y);
              } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                  if (n + c < A.N) o[c] = y[c];
              }
            } else {
              const uint32_t pk = pack4_u8(round_sat_i8(y[0]), round_sat_i8(y[1]), round_sat_i8(y[2]), round_sat_i8(y[3]));
              int8_t* o = (int8_t*)out + (size_t)m * (size_t)A.N + (size_t)n;
              if (vec_ok && n + 4 <= A.N) {
                *(uint32_t*)o = pk;
              } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                  if (n + c < A.N) o[c] = (int8_t)(pk >> (8 * c));
              }
            }
          }
        }
        wave_lds_fence();
      }
    }
  }
  LCE_PH(4);
#ifdef LCE_PHASES
  __builtin_amdgcn_s_waitcnt(0);   // the block's stores have been acknowledged
  LCE_PH(5);
#endif
}

}  // namespace lce
