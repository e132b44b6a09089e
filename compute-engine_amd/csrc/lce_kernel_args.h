// Plain-old-data launch descriptors shared by the host planner (lce_plan.cpp) and the
// device kernels (lce_kernels.h).  No HIP types here: this header is also compiled by
// plain g++ for the host-side planner tests.
#pragma once
#include <stdint.h>
#include "lce_experiments.h"

namespace lce {

// Destination types of LceBconv2d (tflite/kernels/bconv2d.cc:550-564).
enum : int { kDstFloat = 0, kDstInt8 = 1, kDstBitpacked = 2 };

// How SAME padding with pad_values == 0 is realised.
enum : int {
  kZeroPadNone = 0,        // VALID, or SAME with pad_values == 1 (out-of-image word = 0 = +1)
  kZeroPadExact = 1,       // reference kernel: accum += (Cin/G)/2 per outside tap (reference.h:76-103)
  kZeroPadCorrection = 2,  // optimized kernels: one-padded conv, float correction after the
                           // output transform (zero_padding_correction.h:178-297)
};

// Unsigned division of n < 2^31 by a launch-constant d via one v_mul_hi + shift
// (q = mulhi(n, magic) >> shift; d == 1 is flagged by magic == 0).
struct FastDiv {
  uint32_t magic;
  uint32_t shift;
};

inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f{0u, 0u};
  if (d <= 1) return f;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;  // l = ceil(log2 d) >= 1
  const uint64_t two = 1ull << (31 + l);
  f.magic = (uint32_t)((two + d - 1) / d);  // ceil(2^(31+l) / d) < 2^32
  f.shift = l - 1;
  return f;
}
// The streaming kernel's form: branch-free, (mulhi(n, magic) >> shift) + (n & pass), pass = all ones for a divisor <= 1 -- one scalar
// register per divisor where `magic == 0 ? n : ...` keeps a 64-bit lane mask of the comparison live through the K loop.
struct FastDivNB {
  uint32_t magic;
  uint32_t shift;
  uint32_t pass;
};
inline FastDivNB make_fastdiv_nb(uint32_t d) {
  const FastDiv f = make_fastdiv(d);
  return FastDivNB{f.magic, f.shift, f.magic == 0u ? 0xffffffffu : 0u};
}

struct ConvArgs {
  // geometry of this launch
  int32_t H, W;        // input height / width
  int32_t Cw, Cwg;     // bitpacked words per input pixel: all groups / one group
  int32_t OH, OW;      // output height / width
  int32_t N, Npg;      // output channels: total / per group
  int32_t KH, KW, SH, SW, DH, DW, PH, PW;
  int32_t M;           // output pixels of this launch = batch * OH * OW
  int32_t PT, NT;      // fast kernel: pixel tiles (64*TM pixels each), channel tiles (TN each)
  int32_t Wout;        // bitpacked output words per pixel = ceil(N / 32)
  uint32_t in_bytes;   // bytes bound to the input buffer resource (< 2^31)
  FastDiv div_ow, div_oh;
  // output transform (core/bconv2d/output_transform.h)
  int32_t clamp_min, clamp_max;
  // SAME-zero padding
  int32_t zero_pad_mode;   // kZeroPad*
  int32_t bzp;             // (Cin/G)/2, reference.h:76-77
  int32_t eKH, eKW;        // effective (dilated) filter extent
  int32_t left_off, top_off;  // zero_padding_correction.h:189-194
};


// Extra launch constants of the matrix-core engine (lce_kernels_mfma.h).
struct MfmaArgs {
  int32_t H, W, Cw, Cin;   // source (bitpacked) tensor
  int32_t Hp, Wp, PH, PW;  // spatially padded FP4 workspace; input (0,0) sits at (PH,PW)
  uint32_t NPIX;           // workspace pixels of this launch = batch * Hp * Wp
  int32_t CPW;             // 16-byte word planes of the workspace = Cpad / 32
  int32_t KCH;             // K-steps (of 64 channels) per filter tap = Cpad / 64
  int32_t Npad;            // output channels padded to the block tile
  int32_t zero_border;     // 1: border pixels are 0 (exact SAME-zero), 0: +1 (one-padding)
  uint32_t x_bytes, w_bytes;
  FastDiv div_npix, div_wp, div_hp;
  float a_bt;              // KH*KW*Cin as float (back-transform constant)
  float cmin, cmax;        // output-transform clamps as floats (exact integers)
  // direct variant only (bconv2d_mfma<..., DIRECT>): a block's input halo lives in LDS
  int32_t TPI;             // pixel tiles per image = ceil(OH*OW / BM); tiles never cross an image
  int32_t OHOW;            // output pixels per image
  int32_t halo_rows;       // input rows staged per tile (enough for any tile of the image)
  int32_t PS;              // LDS bytes per halo pixel = CPW*16 + 16 (the +16 staggers banks)
  int32_t halo_bytes;      // halo_rows * Wp * PS, rounded up to 1 KiB; the weight ring follows
  int32_t QG;              // 16-byte groups of input words per pixel = ceil(CPW / 4)
  int32_t IPT;             // whole images per tile (> 1 only when OH*OW <= BM / 2; then TPI == 1)
  int32_t B;               // images of this launch
  int32_t HPIX;            // halo pixels per image = halo_rows * (strip tiles: Wp; 2-D tiles: Wh)
  int32_t f32_wide;        // float epilogue may transpose the WN tiles of a row block together
  int32_t i8_wide;         // int8 epilogue may use WN*4 KiB of LDS scratch per wave (16-byte row stores)
  int32_t noclamp;         // the output transform's clamp is the identity on [0, 2*K_bt] (activation NONE)
  float bit_thr;           // second output (LceQuantize of the value just produced): bit = value < bit_thr
                           // (float output: 0; int8 output: the planner's threshold for "rounds below the zero point")
  FastDiv div_tpi, div_qg, div_ohow, div_hpix;
  // grouped convolutions: a block's channels lie in ONE group g = n0 / Npg, whose input channels are the
  // slice [g*Cin_g, (g+1)*Cin_g) of the same pixels; its K loop covers the KCH 64-channel chunks that
  // contain the slice, starting at chunk (g*Cwg)/2 (weights outside the slice are FP4 zeros)
  FastDiv div_npg;
  // 2-D tiles of the direct variant (bconv2d_mfma<..., TILE2D>; wide images, where a strip of BM consecutive pixels
  // would stage whole image rows): TH = BM/32 output rows x 32 output columns, TX of them across the image,
  // TPI = TX * ceil(OH / TH); the halo is the tile's own neighbourhood, Wh = 31 * SW + effective filter width wide.
  // (Kept at the end: the strip variant's kernel-argument loads stay as they were.)
  int32_t Wh, TX;
  FastDiv div_wh, div_tx;
};


// Launch constants of the pointwise kernel.
struct PwArgs {
  int32_t M;            // pixels (rows) of this launch
  int32_t N;            // output channels
  int32_t Npad;         // padded channel count of the FP4 weight image and the parameter tables
  int32_t Cw;           // input words per pixel
  int32_t Cin;          // input channels
  int32_t Wout;         // bitpacked output words per pixel
  int32_t tiles;        // ceil(M / 32)
  int32_t noclamp;      // the clamp is the identity on [0, 2*K_bt]
  uint32_t in_bytes;    // M * Cw * 4
  uint32_t out_bytes;   // bytes of the output rows of this launch
  float a_bt;           // K_bt = Cin as float
  float cmin, cmax;     // clamps as floats (exact integers)
  float bit_thr;        // second output: bit = value < bit_thr (as MfmaArgs::bit_thr)
  // strided 1x1 only: output pixel (b, oy, ox) reads input pixel (b, oy * SH, ox * SW); M counts OUTPUT pixels
  int32_t OW, OHW;      // output width, output pixels per image
  int32_t IW, IHW;      // input width, input pixels per image
  int32_t SH, SW;
  FastDiv div_ow, div_ohw;
};



// Launch constants of the weight-stationary streaming kernel (lce_kernels_stream.h).  A launch is cut into
// SEGMENTS = (image, run of RS output rows); a persistent block owns SPB consecutive segments and walks them as ONE
// stream of 32-pixel blocks, fed by ONE stream of input rows (a segment contributes SRS = (RS-1)*SH + KH of them,
// its halo included) that are expanded to FP4 into a ring of R row slots in LDS.  An ITEM = 16 bytes (4 words) of
// one input pixel.  The planner simulates the stream and uploads tables: sched[T] = items that must be resident
// before tile step T (4 block steps of 2^pph_log pixel blocks each) begins -- quotas of 256 items (one per lane)
// wherever that keeps up, 512 where it does not --, and per (pixel block, lane) the LDS addresses and output offset.
// RS divides OH, so every segment has RS * OW output pixels and output offsets are linear in the segment index.
constexpr int kStreamItemsPerLane = 2;   // items a lane can produce per tile step (the schedule is smoothed to fit)
struct StreamArgs {
  int32_t H, W, Cw, Cin;       // source (bitpacked) tensor
  int32_t OH, OW, N, Npad, Wout;
  int32_t SH, SW, PH, PW;
  int32_t B;                   // images of this launch
  int32_t Wp;                  // pixels per ring row slot (padded width)
  int32_t pitch;               // bytes from one ring row slot to the next: Wp pixels + the bank skew (lce_plan.cpp, plan_stream)
  int32_t R;                   // ring row slots
  int32_t ring_bytes;          // R * Wp * PS rounded up to 1 KiB; 4 x 8 KiB epilogue scratch (none: bitpacked output) + 4 KiB dump follow
  int32_t zero_border;         // 1: padding is 0 (exact SAME-zero), 0: +1 (one-padding)
  int32_t QG;                  // items per pixel = ceil(padded words / 4)
  int32_t IPR;                 // items per stream row = W * QG
  int32_t RS;                  // output rows per segment
  int32_t SPI;                 // segments per image = ceil(OH / RS)
  int32_t SRS;                 // stream rows per segment
  int32_t PBS;                 // pixel blocks per segment = ceil(RS * OW / 32)
  int32_t S;                   // segments of this launch = B * SPI
  int32_t SPB;                 // segments per block
  // Which segments a block owns (round 5).  GSTR == 1: block b the SPB CONSECUTIVE segments from b * SPB (G0M = SPB) -- with whole
  // images as segments, block b writes image b row by row, 256 write streams megabytes apart.  GSTR == GX > 1 ("interleaved"): block b
  // the segments b, b + GX, b + 2 GX, ... (G0M = 1): at any moment the launch's blocks write GX consecutive segments, one compact
  // window that moves through the output (profiles/r05/store_window.txt).  The context table's output offsets carry gl * GSTR.
  int32_t GSTR, G0M, GX;       // segment stride inside a block's run, block index -> first segment, blocks of this launch (grid.x)
  int32_t pph_log;             // log2 of the pixel blocks per block step: 4 waves = (4 >> pph_log) channel slices x that
  // STRIPS (images too wide for a whole padded row per ring slot): a segment is RS output rows x WSo output COLUMNS; segments are
  // numbered (image, strip, row segment), a block's run stays inside one strip.  A ring row then holds the strip's Wp = (WSo-1)*SW+KW
  // input columns, halo included; columns outside the image read as padding like rows do.  NSTRIP = 1: off (WSo = OW, RSEG = SPI).
  int32_t NSTRIP, RSEG, WSo, XS0;   // strips per image, row segments per image (SPI = NSTRIP * RSEG), output columns per strip, left padding of the image
  int32_t flat;                // 1: pixel blocks are cut from the block's segments laid end to end (NPX pixels each)
  int32_t NPX;                 // output pixels per segment = RS * OW
  int32_t NQ;                  // pixel blocks of a full block's stream = rows of the context table
  uint32_t need0;              // = sched[0]: items resident before tile step 0 (here, so that the first rows' loads wait for the kernel arguments only)
  uint32_t in_bytes, w_bytes, out_bytes;   // bytes bound to the input / weight / output buffer resources (< 2^31)
  // the planner's tables (one buffer): sched at byte 0, one dword per tile step; lim: one dword per pixel block (its
  // last real row); ctx: 16 bytes per (pixel block, lane) = {tap-row LDS addresses 0..2, output byte offset}
  // sgn (float / int8 plans): one dword per (pixel block, lane): the pixel row's byte offset in the sign-word tensor
  uint32_t tab_bytes, tab_lim, tab_ctx, tab_sgn;
  uint32_t tab_seg;                        // STRIPS: one dword per pixel block: the local segment it lies in
  uint32_t sign_bytes;                     // bytes of the second output of this launch (B * OH * OW * Wout * 4)
  float a_bt, cmin, cmax, bit_thr;
  FastDivNB div_ipr, div_qg, div_srs, div_spi, div_r, div_rseg, div_gstr;
};


// Launch constants of the weight-STREAMING kernel (lce_kernels_wstream.h, round 5): the mirror image of the kernel above for
// launches that are over after a handful of block steps (14x14 / 7x7 images at one image per CU), where loading a resident
// filter bank (295 KB per CU, ~5 k cycles of the CU's vector memory path) in FRONT of the matrix work costs a third of the
// block's life.  Here the ACTIVATIONS are stationary -- a block expands the whole images of its GROUP (IPB consecutive images)
// into LDS once -- and the weights stream from L2 straight into registers, two fragments per K-step and wave, a few K-steps
// ahead of their use, while the matrix cores run.  The group's output pixels (NHWC: contiguous) are cut into NQ 32-pixel blocks
// laid end to end; a block owns up to NB of them (a PART of the group; the parts of a group differ by at most one pixel block),
// wave w the 64-channel slice w of the block's 256 channels.  Blocks are numbered part-major (block = part * GROUPS + group), so
// that the two blocks a CU holds at a time are a long and a short part.
struct WsArgs {
  int32_t H, W, Cw, Cin;       // source (bitpacked) tensor
  int32_t OH, OW, N, Npad, Wout;
  int32_t SH, SW, PH, PW;
  int32_t B;                   // images of this launch
  int32_t IPB;                 // images per group
  int32_t GROUPS;              // ceil(B / IPB)
  int32_t PARTS;               // blocks per group; grid.x = PARTS * GROUPS
  int32_t NPXG;                // output pixels per group = IPB * OH * OW
  int32_t NQ;                  // pixel blocks per group = ceil(NPXG / 32)
  int32_t Hp, Wp;              // rows / pixels per row of one image in LDS (padding included)
  int32_t pitch, img_pitch;    // LDS bytes per row (Wp pixels + the bank skew) / per image
  int32_t QG;                  // 16-byte items per input pixel = ceil(padded words / 4)
  int32_t items;               // IPB * Hp * Wp * QG: the expansion's work list
  int32_t zero_border;         // 1: padding is 0 (exact SAME-zero), 0: +1 (one-padding)
  int32_t noclamp;             // float output: the clamp is the identity on [0, 2 * K_bt]
  uint32_t lds_images;         // bytes of the group's images in LDS (1 KiB multiple); 4 x 8 KiB of epilogue scratch follow
  uint32_t in_bytes, w_bytes, out_bytes, sign_bytes, tab_bytes;
  // the planner's tables (one buffer): part: two dwords per part {first pixel block, pixel blocks}; ctx: 16 bytes per (pixel block
  // of the group, lane) = {LDS byte address of the lane's pixel in tap rows 0..2 (K-half included), 0}
  uint32_t tab_part, tab_ctx;
  float a_bt, cmin, cmax, bit_thr;
  FastDivNB div_qg, div_wp, div_hp, div_groups;
};

}  // namespace lce
