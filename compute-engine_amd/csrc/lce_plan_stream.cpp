// Launch planners of the two streaming kernels (see lce_plan.h): the weight-stationary kernel (lce_kernels_stream.h) -- segments, blocks,
// the ring and its production schedule from a simulation of the stream, the context tables -- and the weight-streaming kernel
// (lce_kernels_wstream.h) -- image groups, parts, the LDS images.  Host-only C++.
#include "lce_plan.h"
#include "lce_plan_internal.h"

#include <limits.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace lce {

// ------------------------------------------------------------------------------------
// weight-stationary streaming kernel (lce_kernels_stream.h)
// ------------------------------------------------------------------------------------
bool stream_supported(const HostPlan& p) {
  const lce_hip_bconv2d_desc& d = p.d;
  if (!mfma_supported(p) || d.groups != 1) return false;
  if (d.filter_height != 3 || d.filter_width != 3) return false;          // the instantiated filter extents
  if (d.dilation_height != 1 || d.dilation_width != 1) return false;       // tap offsets are instruction immediates
  if (p.zero_pad_mode == kZeroPadCorrection) return false;                 // that epilogue lives in the block GEMM
  // a lane stores 16 bytes of one pixel's channels: whole groups of 4 floats / 16 int8 only
  if (d.dst_type == LCE_HIP_F32 && d.channels_out % 4) return false;
  if (d.dst_type == LCE_HIP_I8 && d.channels_out % 16) return false;
  const int kch = stream_chunks(d);
  // the filter bank must fit the register file: up to 4 chunks of 64 input channels per wave, or 8 split over a pair of
  // waves (KSPLIT, lce_kernels_stream.h)
  return kch == 1 || kch == 2 || kch == 4 || kch == 8;
}

// Simulates a block's stream for segments of `rs` output rows, `spb` segments per block: what each tile step needs
// resident, a production schedule in quotas of 256 items (one per lane; 512 where 256 would fall behind) that meets
// it, and the ring rows that keep every row a tile step reads apart from every row it writes.
// flat: pixel blocks are cut from the CONCATENATED pixels of the block's segments (whole images whose pixel count is not a
// multiple of 32 -- 7x7: 49 pixels would fill 77 % of two blocks): a block may then read rows of two segments.
// ow / in_w: output columns and input columns (halo included) of a segment -- the whole row, or one column strip of it.
static bool simulate_stream(const HostPlan& p, int rs, int spb, int pph_log, bool flat, int ow, int in_w, int* ring_rows,
                            std::vector<uint32_t>* sched) {
  const lce_hip_bconv2d_desc& d = p.d;
  const int kh = d.filter_height, sh = d.stride_height;
  const int srs = (rs - 1) * sh + kh, pbs = ceil_div(rs * ow, 32);
  const int cpw = stream_chunks(d) * 2, qg = ceil_div(cpw, 4);
  const int64_t ipr = (int64_t)in_w * qg;
  const int pph = 1 << pph_log;
  const int64_t npx = (int64_t)rs * ow;
  const int64_t nblk = flat ? (spb * npx + 31) / 32 : (int64_t)spb * pbs;
  const int64_t usteps = (nblk + pph - 1) / pph, ntile = (usteps + 3) / 4;
  const int64_t total = (int64_t)spb * srs * ipr;
  if (ntile < 1 || total >= (1ll << 31) || ntile > (1 << 20)) return false;
  std::vector<int64_t> need(ntile + 2), first(ntile);
  for (int64_t t = 0; t < ntile; ++t) {
    int64_t hi = 0, lo = INT64_MAX;
    for (int64_t q = 4 * t * pph; q < std::min<int64_t>(nblk, 4 * (t + 1) * pph); ++q) {
      if (flat) {
        const int64_t p0 = q * 32, p1 = std::min<int64_t>(q * 32 + 31, spb * npx - 1);   // first / last pixel of the block
        lo = std::min(lo, (p0 / npx) * srs + (p0 % npx) / ow * sh);
        hi = std::max(hi, (p1 / npx) * srs + (p1 % npx) / ow * sh + kh - 1);
        continue;
      }
      const int64_t gl = q / pbs, pb = q % pbs;
      const int64_t r_first = std::min<int64_t>(pb * 32 / ow, rs - 1), r_last = std::min<int64_t>((pb * 32 + 31) / ow, rs - 1);
      lo = std::min(lo, gl * srs + r_first * sh);
      hi = std::max(hi, gl * srs + r_last * sh + kh - 1);
    }
    need[t] = std::min(total, (hi + 1) * ipr);
    first[t] = lo;
  }
  need[ntile] = need[ntile + 1] = need[ntile - 1];
  for (int64_t t = 1; t < ntile; ++t) need[t] = std::max(need[t], need[t - 1]);
  // latest production that still works with at most 512 items per tile step (768 on the K-split kernel, whose pixels are
  // four items wide: it weaves a second item between the MFMAs of block steps 2 and 3) ...
  const int64_t cap = stream_ksplit(p) ? 768 : 512;
  std::vector<int64_t> m(ntile + 2), s(ntile + 2);
  m[ntile + 1] = m[ntile] = need[ntile];
  for (int64_t t = ntile - 1; t >= 0; --t) m[t] = std::max(need[t], m[t + 1] - cap);
  // ... and going forward, the smallest quota (nothing, one item per lane, two) that keeps up with it
  s[0] = m[0];
  for (int64_t t = 0; t <= ntile; ++t) {
    int64_t inc = 0;
    while (s[t] + inc < m[t + 1]) inc += 256;
    s[t + 1] = std::min(total, s[t] + inc);
    if (s[t + 1] < m[t + 1] || inc > cap) return false;   // cannot happen (m is feasible by construction)
  }
  int64_t rows = kh;
  for (int64_t t = 0; t < ntile; ++t)
    if (s[t + 1] > 0) rows = std::max(rows, (s[t + 1] - 1) / ipr - first[t] + 1);
  *ring_rows = (int)rows;
  sched->resize(ntile + 2);
  for (int64_t t = 0; t < ntile + 2; ++t) (*sched)[t] = (uint32_t)s[t];
  return true;
}

static bool plan_stream_geometry(HostPlan& p, int batch_chunk, int wso, std::string* why);

std::string plan_stream(HostPlan& p, int batch_chunk) {
  const lce_hip_bconv2d_desc& d = p.d;
  if (!stream_supported(p))
    return "bconv2d: the streaming kernel runs ungrouped 3x3 convolutions without dilation, with at most 512 input "
           "channels (on its 64-, 128-, 256- or 512-channel instance) and whole 16-byte groups of output channels (float: a multiple of 4, "
           "int8: of 16), and not the SAME-zero correction semantics";
  if (p.stream_occ_pref == 2 && stream_blocks_per_cu_max(p) < 2)
    return "bconv2d: stream_blocks_per_cu=2: only the bitpacked-output instance of the 64-input-channel bank is compiled for two blocks per CU";
  const uint32_t row_bytes = stream_row_bytes(p);
  if ((int64_t)batch_chunk * p.out_h * p.out_w * row_bytes >= (1ll << 31))
    return "bconv2d: the streaming kernel binds the whole output of a launch to one buffer resource (< 2 GiB)";
  // Whole rows first; an image too wide for that (the ring holds 9-12 padded rows: 224 x 144 B x 12 does not fit) is cut into
  // column strips of 64 or 32 output columns (round 4; instances exist for the 256-channel bank).
  std::vector<int> widths;
  if (p.stream_strip_pref <= 0) widths.push_back(0);
  // (a strip width forced on a layer whose bank is not the 256-channel one: say so, instead of "the ring does not fit")
  if (p.stream_strip_pref > 0 && stream_chunks(d) != 4)
    return "bconv2d: stream_strip: column strips exist for the 256-channel filter bank only (193..256 input channels)";
  if (stream_chunks(d) == 4 && p.stream_strip_pref != 0) {
    if (p.stream_strip_pref > 0) {
      if (p.stream_strip_pref % 32 != 0 || p.out_w % p.stream_strip_pref != 0)
        return "bconv2d: stream_strip must be a multiple of 32 that divides the output width";
      widths.push_back(p.stream_strip_pref);
    } else if (p.out_w % 32 == 0 && p.out_w > 64) {
      if (p.out_w % 64 == 0) widths.push_back(64);
      widths.push_back(32);
    }
  }
  std::string why = p.stream_occ_pref == 2 ? "bconv2d: stream_blocks_per_cu=2: two blocks' row rings do not fit a CU's LDS for this layer"
                                           : "bconv2d: the streaming kernel's row ring does not fit LDS for this layer";
  for (int wso : widths)
    if (plan_stream_geometry(p, batch_chunk, wso, &why)) return "";
  return why;
}

// One attempt: segments of whole rows (wso == 0) or of column strips `wso` output columns wide.
static bool plan_stream_geometry(HostPlan& p, int batch_chunk, int wso, std::string* why) {
  const lce_hip_bconv2d_desc& d = p.d;
  const uint32_t row_bytes = stream_row_bytes(p);
  const bool strips = wso > 0;
  const int nstrip = strips ? p.out_w / wso : 1;
  const int ow_seg = strips ? wso : p.out_w;                                                  // output columns of a segment
  const int in_w_seg = strips ? (wso - 1) * d.stride_width + d.filter_width : d.in_width;     // input columns a ring row holds
  const int nsl = ceil_div(d.channels_out, 64);
  // waves of a block = (64-channel slices) x (pixel phases).  By default all four waves take slices when there are >= 3 of
  // them; `stream_pixel_phases` forces the split (256 channels as 2 slices x 2 phases puts half the filter bank on a CU and
  // two images' rows through it: the bank's arrival -- 295 KB per CU otherwise -- is what a single-round launch waits for)
  int pph_log = nsl >= 3 ? 0 : nsl == 2 ? 1 : 2;
  if (p.stream_phases_pref > 0) pph_log = std::max(pph_log, p.stream_phases_pref == 4 ? 2 : p.stream_phases_pref == 2 ? 1 : 0);
  const bool ksplit = stream_ksplit(p);     // 512 input channels: waves = 2 slices x 2 K-halves, one pixel block per step
  if (ksplit) pph_log = 0;
  const int nslb = ksplit ? 2 : 4 >> pph_log, ny = ceil_div(nsl, nslb), pph = 1 << pph_log;
  const int wp = strips ? in_w_seg
                        : (int)std::max<int64_t>(p.pad_w + d.in_width, (int64_t)(p.out_w - 1) * d.stride_width + d.filter_width);
  const int kch = stream_chunks(d), ps = kch * 32 + 16;
  // Ring row pitch.  An A-fragment read is one 16-byte piece per lane, lane = pixel; the LDS serves 16 lanes per pass without
  // conflicts when their 16-byte units differ mod 16.  Along a row consecutive pixels are ps / 16 (odd) units apart: fine.
  // Where a 32-pixel block wraps to the next output row the unit jumps by pitch / 16 - (OW - 1) * SW * ps / 16 instead, and
  // with pitch = wp * ps (two padding columns) two lanes of the pass collide: every wrap costs a second pass (profiles/r03
  // PMC: 21 % of the LDS pipe's cycles on L0, 39 % on 14x14x256).  A skew of < 256 bytes per row makes the sequence continue
  // across the wrap: SH * pitch / 16 = OW * SW * ps / 16 (mod 16).  (Solvable when SH is odd; otherwise no skew.)
  int skew16 = 0;
  if (d.stride_height % 2 == 1 && !strips) {     // (a strip is a multiple of 32 columns: a pixel block never wraps)
    const int want = (int)(((int64_t)p.out_w * d.stride_width * (ps / 16)) % 16);
    for (int k = 0; k < 16; ++k)
      if (((int64_t)d.stride_height * ((int64_t)wp * (ps / 16) + k)) % 16 == want) { skew16 = k; break; }
  }
#ifdef LCE_STREAM_NO_SKEW   // (A/B aid)
  skew16 = 0;
#endif
  const int pitch = wp * ps + skew16 * 16;
  // blocks per CU: two where the instance is compiled for it, the plan asks for it and (below) both blocks' LDS fit
  const int occ = !strips && p.stream_occ_pref == 2 && stream_blocks_per_cu_max(p) >= 2 ? 2 : 1;
  const int cus = std::max(1, p.num_cus * occ / ny);
  // segment size (a divisor of the output height: every segment is whole): the fewest block steps on the busiest
  // block (ties: the longer segment, whose halo is re-expanded less)
  // segments per block: as many as spread the launch over the CUs (a strip run may pass into the next strip or image: the
  // kernel works out every segment's place in the output)
  auto run_length = [&](int rseg) -> int64_t {
    const int64_t s = (int64_t)batch_chunk * nstrip * rseg, gx = std::min<int64_t>(s, cus);
    const int64_t spb = (s + gx - 1) / gx;
    return strips ? std::min<int64_t>(spb, 128) : spb;      // (the kernel's per-run segment table holds 128 entries)
  };
  struct Cand { int rs; int64_t cost; };
  std::vector<Cand> cands;
  for (int rs = p.out_h; rs >= 1; --rs) {
    if (p.out_h % rs) continue;
    if (p.stream_rows_pref > 0 && rs != p.stream_rows_pref) continue;
    const int64_t spb = run_length(p.out_h / rs), s = (int64_t)batch_chunk * nstrip * (p.out_h / rs);
    // (the K-split instances are the ones built for it)
    const bool flat_c = !strips && rs == p.out_h && spb > 1 && (rs * p.out_w) % 32 != 0 && !p.stream_noflat && ksplit;
    const int64_t blocks = flat_c ? (spb * rs * p.out_w + 31) / 32 : spb * ceil_div(rs * ow_seg, 32);
    const int64_t rounds = (ceil_div((int)s, (int)spb) + cus - 1) / cus;      // (strips: more blocks than CUs run in rounds)
    const int64_t steps = (blocks + pph - 1) / pph;
    // (ties, strips: segments of about 14 rows -- 224x224x256 measured 0.194 ms with 14-row segments, 0.201 with 28 / 56 / 112,
    //  profiles/r04/strips_224.txt; whole rows: the longer one, whose halo rows are re-expanded less)
    cands.push_back(Cand{rs, (steps + 4) * rounds * 4096 + (strips ? std::abs(rs - 14) : 0)});
  }
  if (cands.empty()) { *why = "bconv2d: stream_rows must divide the output height"; return false; }
  std::stable_sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.cost < b.cost; });
  for (const Cand& c : cands) {
    const int rs = c.rs, rseg = p.out_h / rs, spi = nstrip * rseg;
    const int64_t s = (int64_t)batch_chunk * spi, spb = run_length(rseg);
    int rows = 0;
    std::vector<uint32_t> sched;
    // whole small images whose pixels do not fill 32-pixel blocks: cut the blocks from the block's images laid end to end
    // (the output tensor is laid out that way: NHWC with nothing between images)
    // (the K-split instances are the ones built for it)
    const bool flat = !strips && rs == p.out_h && spb > 1 && (rs * p.out_w) % 32 != 0 && !p.stream_noflat && ksplit;
    if (!simulate_stream(p, rs, (int)spb, pph_log, flat, ow_seg, in_w_seg, &rows, &sched)) continue;
    const int64_t ring = ((int64_t)rows * pitch + 1023) / 1024 * 1024;
    if ((ring + stream_lds_extra(p)) * occ > 160 * 1024) continue;
    p.st_occ = occ;
    const int pbs = ceil_div(rs * ow_seg, 32);
    // (the strips epilogue's out-of-line path does not add the segment's place: a strip is a multiple of 32 columns, so no block is
    // partial)
    if (strips && (rs * ow_seg) % 32 != 0) continue;
    const int64_t nq = flat ? (spb * (int64_t)rs * p.out_w + 31) / 32 : spb * pbs;
    p.st_flat = flat ? 1 : 0;
    p.st_nq = (int)nq;
    // Interleaved runs (round 5): block b owns segments b, b + gx, b + 2 gx, ... instead of spb consecutive ones, so that at any
    // moment the launch writes gx CONSECUTIVE segments -- one window of gx * rs * OW pixels moving through the output -- instead
    // of gx streams a whole run apart.  Same segments, same ring schedule, same tables but for the output offsets.  (Flat runs
    // cut pixel blocks across consecutive images: they stay consecutive.)
    const int64_t gx_plan = ceil_div((int)s, (int)spb);
    const int64_t gstr = (p.stream_interleave_pref > 0 && !flat && spb > 1) ? gx_plan : 1;
    p.st_gstr = (int)gstr;
    if (nq * 1024 > (64ll << 20)) continue;               // the context table: 1 KiB per pixel block
    p.st_rs = rs; p.st_spi = spi; p.st_srs = (rs - 1) * d.stride_height + d.filter_height;
    p.st_pbs = pbs; p.st_pph_log = pph_log; p.st_ny = ny;
    p.st_qg = ceil_div(kch * 2, 4); p.st_ipr = in_w_seg * p.st_qg;
    p.st_nstrip = nstrip; p.st_rseg = rseg; p.st_wso = ow_seg;
    p.st_spb = (int)spb; p.st_gx = (int)ceil_div((int)s, (int)spb); p.st_rows = rows; p.st_ring_bytes = (int)ring;
    p.st_batch = batch_chunk;
    p.wp = wp;
    if (p.dbg_level >= 2)
      fprintf(stderr,
              "[lce plan] stream geometry: rows/segment %d, ring %d rows x %d B = %lld B (+%d), blocks %d x %d, segments/block %lld\n",
              rs, rows, pitch, (long long)ring, stream_lds_extra(p), p.st_gx, ny, (long long)spb);
    p.st_pitch = pitch;
    // ---- the tables: [sched | lim | ctx] ----
    const size_t n_sched = (sched.size() + 3) / 4 * 4, n_lim = ((size_t)nq + 3) / 4 * 4;
    const size_t n_sgn = d.dst_type == LCE_HIP_BITPACKED ? 0 : (size_t)nq * 64;
    const size_t n_seg = strips ? ((size_t)nq + 3) / 4 * 4 : 0;
    p.st_tab_lim = (uint32_t)(n_sched * 4);
    p.st_tab_ctx = (uint32_t)((n_sched + n_lim) * 4);
    p.st_tab_sgn = (uint32_t)((n_sched + n_lim + (size_t)nq * 256) * 4);
    p.st_tab_seg = (uint32_t)((n_sched + n_lim + (size_t)nq * 256 + n_sgn) * 4);
    p.st_tabs.assign(n_sched + n_lim + (size_t)nq * 256 + n_sgn + n_seg, 0u);
    std::copy(sched.begin(), sched.end(), p.st_tabs.begin());
    for (size_t i = sched.size(); i < n_sched; ++i) p.st_tabs[i] = sched.back();
    const int npx = rs * ow_seg, sh = d.stride_height, sw = d.stride_width;
    const int64_t total_px = spb * (int64_t)npx;                             // flat: pixels of a full block's stream
    const bool ragged = flat ? total_px % 32 != 0 : npx % 32 != 0;
    // lanes that share a stored pixel row (lce_kernels_stream.h, LPR): the K-split kernel stores 32 channels per wave
    const int lpr = d.dst_type == LCE_HIP_F32 ? (ksplit ? 8 : 16) : d.dst_type == LCE_HIP_I8 ? (ksplit ? 2 : 4) : 0;
    for (int64_t q = 0; q < nq; ++q) {
      // pixel block q = pixels [first, first + 32) of segment gl (flat: of the block's segments laid end to end)
      const int64_t gl = flat ? 0 : q / pbs, pb = flat ? q : q % pbs;
      const int64_t seg_px = flat ? total_px : npx;
      const bool partial = ragged && (flat ? q == nq - 1 : pb == pbs - 1);
      p.st_tabs[n_sched + q] = (uint32_t)std::min<int64_t>(31, seg_px - pb * 32 - 1);
      for (int lane = 0; lane < 64; ++lane) {
        const int l31 = lane & 31, half = lane >> 5;
        int64_t pix = std::min<int64_t>(pb * 32 + l31, seg_px - 1);          // rows past the segment re-read its last pixel
        const int64_t sg_ = flat ? pix / npx : gl;                           // the segment the pixel lies in
        if (flat) pix %= npx;
        const int64_t r = pix / ow_seg, ox = pix % ow_seg;
        const int64_t s0 = sg_ * p.st_srs + r * sh;
        uint32_t* e = &p.st_tabs[n_sched + n_lim + ((size_t)q * 64 + lane) * 4];
        for (int fy = 0; fy < 3; ++fy)
          e[fy] = (uint32_t)(((s0 + fy) % rows) * pitch + ox * sw * ps + half * 16);
        const int rowl = lpr ? lane / lpr : l31;
        // output pixel of the lane's first stored row, relative to the run's first pixel: segments and their pixels follow each
        // other in memory -- or (strips) a segment's rows are OW pixels apart and a pixel block lies inside one of them
        // (strips: relative to the SEGMENT's first pixel; the kernel adds the segment's place)
        // (interleaved runs: the block's local segment gl is segment g0 + gl * gstr of the launch)
        const int64_t blk_px = strips ? ((pb * 32) / ow_seg) * (int64_t)p.out_w + (pb * 32) % ow_seg : gl * gstr * npx + pb * 32;
        e[3] = (uint32_t)((blk_px + rowl) * (int64_t)row_bytes);
        if (partial) e[3] |= 0x80000000u;   // a partial pixel block: its stores go out of line
        if (strips) p.st_tabs[n_sched + n_lim + (size_t)nq * 256 + n_sgn + q] = (uint32_t)gl;
        if (n_sgn) {
          uint32_t& sg = p.st_tabs[n_sched + n_lim + (size_t)nq * 256 + (size_t)q * 64 + lane];
          sg = (uint32_t)((blk_px + l31) * (int64_t)p.wout * 4);
          if (partial) sg |= 0x80000000u;
        }
      }
    }
    return true;
  }
  return false;
}

StreamArgs make_stream_args(const HostPlan& p, int batch_chunk) {
  const lce_hip_bconv2d_desc& d = p.d;
  StreamArgs G{};
  G.H = d.in_height; G.W = d.in_width; G.Cw = p.cw; G.Cin = d.channels_in;
  G.OH = p.out_h; G.OW = p.out_w; G.N = d.channels_out; G.Npad = p.npad; G.Wout = p.wout;
  G.SH = d.stride_height; G.SW = d.stride_width; G.PH = p.pad_h;
  G.PW = p.st_nstrip > 1 ? 0 : p.pad_w;      // (a strip's ring row starts at its first input column, halo or padding)
  G.NSTRIP = p.st_nstrip; G.RSEG = p.st_rseg; G.WSo = p.st_wso; G.XS0 = p.pad_w;
  G.B = batch_chunk;
  G.Wp = p.wp; G.pitch = p.st_pitch; G.R = p.st_rows; G.ring_bytes = p.st_ring_bytes;
  G.zero_border = p.zero_pad_mode == kZeroPadExact ? 1 : 0;
  G.QG = p.st_qg; G.IPR = p.st_ipr; G.RS = p.st_rs; G.SPI = p.st_spi; G.SRS = p.st_srs; G.PBS = p.st_pbs;
  G.S = batch_chunk * p.st_spi;
  // a smaller launch than the one planned for (the last chunk of a batch): the same segments and tables, fewer per block
  const int gx = std::min(G.S, std::max(1, p.num_cus * p.st_occ / p.st_ny));
  G.SPB = std::min(p.st_spb, ceil_div(G.S, std::max(1, gx)));
  // (flat pixel blocks are cut for runs of exactly st_spb segments: a shorter run's last block would spill into the next
  //  block's pixels, so a smaller launch keeps the planned run length and uses fewer blocks)
  if (p.st_flat) G.SPB = p.st_spb;
  G.GSTR = 1; G.G0M = G.SPB; G.GX = ceil_div(G.S, std::max(1, G.SPB));
  if (p.st_gstr > 1) {
    // interleaved runs: the tables' output offsets carry the PLANNED stride, so a smaller launch keeps it and its blocks' runs
    // end earlier (block b: segments b, b + gstr, ... below S)
    G.GSTR = p.st_gstr; G.G0M = 1; G.SPB = p.st_spb; G.GX = std::min(G.S, p.st_gstr);
  }
  G.pph_log = p.st_pph_log;
  G.flat = p.st_flat;
  G.NPX = p.st_rs * p.out_w;
  G.NQ = p.st_nq;
  G.need0 = p.st_tabs.empty() ? 0u : p.st_tabs[0];
  G.in_bytes = (uint32_t)((int64_t)batch_chunk * d.in_height * d.in_width * p.cw * 4);
  G.w_bytes = (uint32_t)p.wq.size();
  G.out_bytes = (uint32_t)((int64_t)batch_chunk * p.out_h * p.out_w * stream_row_bytes(p));
  G.tab_bytes = (uint32_t)(p.st_tabs.size() * 4);
  G.tab_lim = p.st_tab_lim;
  G.tab_ctx = p.st_tab_ctx;
  G.tab_sgn = p.st_tab_sgn;
  G.tab_seg = p.st_tab_seg;
  G.sign_bytes = (uint32_t)((int64_t)batch_chunk * p.out_h * p.out_w * p.wout * 4);
  G.bit_thr = p.bit_thr;
  G.a_bt = (float)p.backtransform_add;
  G.cmin = (float)p.clamp_min;
  G.cmax = (float)p.clamp_max;
  G.div_ipr = make_fastdiv_nb((uint32_t)G.IPR);
  G.div_qg = make_fastdiv_nb((uint32_t)G.QG);
  G.div_srs = make_fastdiv_nb((uint32_t)G.SRS);
  G.div_spi = make_fastdiv_nb((uint32_t)G.SPI);
  G.div_r = make_fastdiv_nb((uint32_t)G.R);
  G.div_rseg = make_fastdiv_nb((uint32_t)std::max(1, G.RSEG));
  G.div_gstr = make_fastdiv_nb((uint32_t)G.GSTR);
  return G;
}

// ------------------------------------------------------------------------------------
// weight-streaming kernel (lce_kernels_wstream.h)

// ------------------------------------------------------------------------------------
bool wstream_supported(const HostPlan& p) {
  // 3x3, no dilation, no groups, whole 16-byte channel groups, not the correction semantics
  if (!stream_supported(p)) return false;
  const int kch = stream_chunks(p.d);
  return kch == 2 || kch == 4 || kch == 8;                    // the instantiated K depths (128 / 256 / 512 input channels)
}

// Cycle model of one launch, used to pick the group size and (select_kernel) to rank this kernel against the weight-stationary
// one.  Calibrated on profiles/r05/wstream_phases.txt: an MFMA of a K loop costs ~34 cycles of its SIMD whichever of the two
// resident blocks issues it; a block's prologue (expansion of its group's images: a global round trip + ~70 VALU per item) and
// epilogue (~450 cycles per pixel block) are hidden by the co-resident block except for the first prologue and the last epilogue.
static int64_t wstream_cost(int64_t blocks, int cus, int occupancy, const std::vector<int>& nb_of_block, int ks, int items_per_lane) {
  // blocks are dispatched in index order, round-robin over the CUs
  std::vector<int64_t> load(cus, 0);
  int64_t worst = 0, last_nb = 0;
  for (int64_t b = 0; b < blocks; ++b) {
    load[b % cus] += nb_of_block[b];
    worst = std::max(worst, load[b % cus]);
  }
  for (int64_t b = std::max<int64_t>(0, blocks - cus); b < blocks; ++b) last_nb = std::max<int64_t>(last_nb, nb_of_block[b]);
  const int64_t rounds = (blocks + (int64_t)cus * occupancy - 1) / ((int64_t)cus * occupancy);
  const int64_t prologue = 2200 + 300 * items_per_lane, epilogue = 450 * last_nb;
  return worst * ks * 2 * 34 + rounds * prologue + epilogue + (occupancy < 2 ? (blocks + cus - 1) / cus * (prologue + epilogue) : 0);
}

std::string plan_wstream(HostPlan& p, int batch_chunk) {
  const lce_hip_bconv2d_desc& d = p.d;
  if (!wstream_supported(p))
    return "bconv2d: the weight-streaming kernel runs ungrouped 3x3 convolutions without dilation over 65 .. 512 input channels "
           "(on its 128-, 256- or 512-channel instance) and whole 16-byte groups of output channels (float: a multiple of 4, int8: of 16), "
           "and not the SAME-zero correction semantics";
  if ((int64_t)batch_chunk * p.out_h * p.out_w * stream_row_bytes(p) >= (1ll << 31))
    return "bconv2d: the weight-streaming kernel binds the whole output of a launch to one buffer resource (< 2 GiB)";
  const int kch = stream_chunks(d), ps = kch * 32 + 16, ks = 9 * kch;
  const int hp = (p.out_h - 1) * d.stride_height + d.filter_height, wp = (p.out_w - 1) * d.stride_width + d.filter_width;
  // row pitch: a skew of < 256 bytes so that a 32-pixel block that wraps to the next output row keeps hitting distinct LDS banks
  // (the streaming kernel's rule, plan_stream_geometry)
  int skew16 = 0;
  if (d.stride_height % 2 == 1) {
    const int want = (int)(((int64_t)p.out_w * d.stride_width * (ps / 16)) % 16);
    for (int k = 0; k < 16; ++k)
      if (((int64_t)d.stride_height * ((int64_t)wp * (ps / 16) + k)) % 16 == want) { skew16 = k; break; }
  }
  const int pitch = wp * ps + skew16 * 16, img_pitch = hp * pitch;
  const int qg = ceil_div(kch * 2, 4), ohw = p.out_h * p.out_w;
  const int ny = ceil_div(ceil_div(d.channels_out, 64), 4), cus = std::max(1, p.num_cus);
  int best_ipb = 0;
  int64_t best_cost = 0;
  const int nbmax = p.ws_blocks_pref > 0 ? p.ws_blocks_pref : 4;       // pixel blocks per block (tuning aid: wstream_blocks)
  for (int ipb = 1; ipb <= std::min(batch_chunk, 64); ++ipb) {
    if (p.ws_images_pref > 0 && ipb != p.ws_images_pref) continue;
    const int64_t lds_images = ((int64_t)ipb * img_pitch + 1023) / 1024 * 1024;
    if (lds_images + kWsLdsExtra > 160 * 1024) break;
    const int occupancy = (int)std::min<int64_t>(2, (160 * 1024) / (lds_images + kWsLdsExtra));
    const int nq = ceil_div(ipb * ohw, 32), parts = ceil_div(nq, nbmax), groups = ceil_div(batch_chunk, ipb);
    std::vector<int> nb_of_block;
    for (int y = 0; y < ny; ++y)
      for (int part = 0; part < parts; ++part)
        for (int g = 0; g < groups; ++g) nb_of_block.push_back(nq / parts + (part < nq % parts ? 1 : 0));
    const int items_per_lane = ceil_div(ipb * hp * wp * qg, 256);
    const int64_t cost = wstream_cost((int64_t)nb_of_block.size(), cus, occupancy, nb_of_block, ks, items_per_lane);
    if (best_ipb == 0 || cost < best_cost) { best_ipb = ipb; best_cost = cost; }
  }
  if (best_ipb == 0) return "bconv2d: one image of this layer does not fit the weight-streaming kernel's LDS (whole images are resident)";
  const int ipb = best_ipb;
  const int nq = ceil_div(ipb * ohw, 32), parts = ceil_div(nq, nbmax);
  p.ws_ipb = ipb; p.ws_parts = parts; p.ws_nq = nq; p.ws_npxg = ipb * ohw; p.ws_nb = ceil_div(nq, parts); p.ws_ny = ny;
  p.ws_hp = hp; p.ws_wp = wp; p.ws_pitch = pitch; p.ws_img_pitch = img_pitch; p.ws_qg = qg;
  p.ws_lds_images = (int)(((int64_t)ipb * img_pitch + 1023) / 1024 * 1024);
  p.ws_occupancy = (int)std::min<int64_t>(2, (160 * 1024) / (p.ws_lds_images + kWsLdsExtra));
  p.ws_cost = best_cost;
  p.st_batch = batch_chunk;
  // ---- the tables: [part | ctx] ----
  const size_t n_part = ((size_t)parts * 2 + 3) / 4 * 4;
  p.ws_tab_part = 0;
  p.ws_tab_ctx = (uint32_t)(n_part * 4);
  p.st_tabs.assign(n_part + (size_t)nq * 256, 0u);
  int q0 = 0;
  for (int part = 0; part < parts; ++part) {
    const int nb = nq / parts + (part < nq % parts ? 1 : 0);
    p.st_tabs[2 * part] = (uint32_t)q0;
    p.st_tabs[2 * part + 1] = (uint32_t)nb;
    q0 += nb;
  }
  for (int q = 0; q < nq; ++q)
    for (int lane = 0; lane < 64; ++lane) {
      const int l31 = lane & 31, half = lane >> 5;
      const int pix = std::min(q * 32 + l31, ipb * ohw - 1);      // rows past the group re-read its last pixel (never stored)
      const int img = pix / ohw, oy = (pix % ohw) / p.out_w, ox = pix % p.out_w;
      uint32_t* e = &p.st_tabs[n_part + ((size_t)q * 64 + lane) * 4];
      for (int fy = 0; fy < 3; ++fy)
        e[fy] = (uint32_t)((int64_t)img * img_pitch + (int64_t)(oy * d.stride_height + fy) * pitch + (int64_t)ox * d.stride_width * ps +
                           half * 16);
    }
  return "";
}

WsArgs make_ws_args(const HostPlan& p, int batch_chunk) {
  const lce_hip_bconv2d_desc& d = p.d;
  WsArgs G{};
  G.H = d.in_height; G.W = d.in_width; G.Cw = p.cw; G.Cin = d.channels_in;
  G.OH = p.out_h; G.OW = p.out_w; G.N = d.channels_out; G.Npad = p.npad; G.Wout = p.wout;
  G.SH = d.stride_height; G.SW = d.stride_width; G.PH = p.pad_h; G.PW = p.pad_w;
  G.B = batch_chunk;
  G.IPB = p.ws_ipb; G.GROUPS = ceil_div(batch_chunk, p.ws_ipb); G.PARTS = p.ws_parts;
  G.NPXG = p.ws_npxg; G.NQ = p.ws_nq;
  G.Hp = p.ws_hp; G.Wp = p.ws_wp; G.pitch = p.ws_pitch; G.img_pitch = p.ws_img_pitch;
  G.QG = p.ws_qg; G.items = p.ws_ipb * p.ws_hp * p.ws_wp * p.ws_qg;
  G.zero_border = p.zero_pad_mode == kZeroPadExact ? 1 : 0;
  G.noclamp = (p.clamp_min <= 0 && p.clamp_max >= 2 * p.backtransform_add) ? 1 : 0;
  G.lds_images = (uint32_t)p.ws_lds_images;
  G.in_bytes = (uint32_t)((int64_t)batch_chunk * d.in_height * d.in_width * p.cw * 4);
  G.w_bytes = (uint32_t)p.wq.size();
  G.out_bytes = (uint32_t)((int64_t)batch_chunk * p.out_h * p.out_w * stream_row_bytes(p));
  G.sign_bytes = (uint32_t)((int64_t)batch_chunk * p.out_h * p.out_w * p.wout * 4);
  G.tab_bytes = (uint32_t)(p.st_tabs.size() * 4);
  G.tab_part = p.ws_tab_part;
  G.tab_ctx = p.ws_tab_ctx;
  G.a_bt = (float)p.backtransform_add;
  G.cmin = (float)p.clamp_min;
  G.cmax = (float)p.clamp_max;
  G.bit_thr = p.bit_thr;
  G.div_qg = make_fastdiv_nb((uint32_t)G.QG);
  G.div_wp = make_fastdiv_nb((uint32_t)G.Wp);
  G.div_hp = make_fastdiv_nb((uint32_t)G.Hp);
  G.div_groups = make_fastdiv_nb((uint32_t)G.GROUPS);
  return G;
}

}  // namespace lce
