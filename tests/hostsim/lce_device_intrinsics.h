// HOST replacement of compute-engine_amd/csrc/lce_device_intrinsics.h -- TEST ONLY.
//
// The CPU-only test suite compiles the real kernel bodies (lce_kernels.h) against this
// header and executes them thread by thread (lanes that need collectives: as fibers in lock step), so index arithmetic, padding, grouping and
// the fused output transforms are exercised without a GPU.  Nothing in the product
// includes this file; it is not a fallback path (the shipped library is built only from
// the gfx950 header and fails loudly without a device).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <sys/mman.h>

#include <functional>
#include <deque>
#include <vector>

#define LCE_DEVICE inline
#define LCE_LAMBDA_INLINE
#define LCE_KERNEL inline
#define __restrict__
#define __launch_bounds__(...)

inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }

namespace lce_dev {

struct u32x2 { uint32_t v[2]; uint32_t& operator[](int i) { return v[i]; } uint32_t operator[](int i) const { return v[i]; } };
struct u32x4 { uint32_t v[4]; uint32_t& operator[](int i) { return v[i]; } uint32_t operator[](int i) const { return v[i]; } };
struct f32x2 { float v[2]; float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };
struct f32x4 { float v[4]; float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };

constexpr int kWave = 64;

struct f32x16 { float v[16]; float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };

// ---- cooperative lanes ------------------------------------------------------------------------------------------------
// The simulated threads of a wave / block that need collectives (ballots, shuffles, the MFMA emulation, LDS fences, block barriers)
// run as FIBERS of one OS thread: a barrier is a counter, waiting is a switch to the next fiber (a dozen instructions: callee-saved
// registers and the stack pointer).  (Until round 5 every lane was an OS thread on a std::barrier: 256 threads on 8 cores spent the
// suite's time in futex calls -- 14 of its 18 minutes.)
#if !defined(__x86_64__)
#error "the host simulation's fiber switch is written for x86-64 (the container's and the driver's CPU)"
#endif
extern "C" void lce_fiber_switch(void** save_sp, void* load_sp);
// (.weak: every translation unit of the simulation carries this definition -- the build is cut into parts -- and the linker keeps one)
asm(R"(
.text
.weak lce_fiber_switch
.type lce_fiber_switch,@function
lce_fiber_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size lce_fiber_switch,.-lce_fiber_switch
)");

struct FiberBarrier {
  int n, arrived = 0;
  unsigned gen = 0;
  explicit FiberBarrier(int n_) : n(n_) {}
  inline void arrive_and_wait();
};

struct ThreadCtx {
  int tid_x = 0, bid_x = 0, bid_y = 0, bdim_x = 1, gdim_x = 1;
  // block-level state (lock-step block mode): LDS, block barrier, per-wave MFMA exchange
  uint8_t* lds = nullptr;
  FiberBarrier* block_bar = nullptr;
  uint32_t* mfma_xchg = nullptr;  // per wave: 64 lanes x 8 dwords (a[4], b[4])
  // wave collectives (only valid when the 64 lanes of a wave run in lock step)
  FiberBarrier* bar = nullptr;
  uint32_t* xchg = nullptr;  // 64 slots shared by the wave
  // LDS-DMA in flight (this lane's 16 bytes of every piece its wave issued), oldest first
  struct PendingDma { uint8_t* dst; uint8_t data[16]; };
  std::deque<PendingDma> dma;
};
inline thread_local ThreadCtx g_main_ctx;             // the OS thread's own context (sequential launches)
inline thread_local ThreadCtx* g_ctxp = &g_main_ctx;  // the running fiber's
#define g_ctx (*::lce_dev::g_ctxp)

struct FiberSet {
  static constexpr size_t kStack = 1u << 20;          // per fiber, reserved lazily; its lowest page is a guard (an overflow faults instead of running into the neighbour)
  struct Fiber { void* sp = nullptr; ThreadCtx ctx; bool done = false; };
  std::vector<Fiber> f;
  std::function<void(int)> body;
  int cur = 0, alive = 0;
  void* main_sp = nullptr;
  uint8_t* stacks = nullptr;
};
inline thread_local FiberSet* g_fibers = nullptr;

inline void fiber_switch_to(int to) {
  FiberSet& s = *g_fibers;
  const int from = s.cur;
  s.cur = to;
  g_ctxp = &s.f[to].ctx;
  lce_fiber_switch(&s.f[from].sp, s.f[to].sp);
}
inline void fiber_yield() {
  FiberSet& s = *g_fibers;
  const int n = (int)s.f.size();
  int to = s.cur;
  do { to = to + 1 == n ? 0 : to + 1; } while (s.f[to].done && to != s.cur);
  if (to != s.cur) fiber_switch_to(to);
}
inline void FiberBarrier::arrive_and_wait() {
  const unsigned g = gen;
  if (++arrived == n) { arrived = 0; ++gen; return; }
  while (gen == g) fiber_yield();
}
extern "C" inline void lce_fiber_entry() {
  FiberSet& s = *g_fibers;
  const int me = s.cur;
  s.body(me);
  s.f[me].done = true;
  if (--s.alive == 0) {
    void* dummy;
    g_ctxp = &g_main_ctx;
    lce_fiber_switch(&dummy, s.main_sp);
  }
  fiber_yield();      // to a fiber that is not done; never resumed
  __builtin_trap();
}
// Runs body(t) for t in [0, n) as n fibers of the calling thread, round-robin at every wait, until all have returned.
// `setup(t, ctx)` fills each fiber's context first.
template <typename Setup>
inline void run_fibers(int n, Setup&& setup, std::function<void(int)> body) {
  FiberSet s;
  s.f.resize(n);
  s.body = std::move(body);
  s.alive = n;
  const size_t bytes = (size_t)n * FiberSet::kStack;
  s.stacks = (uint8_t*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (s.stacks == (uint8_t*)MAP_FAILED) __builtin_trap();
  for (int t = 0; t < n; ++t) {
    setup(t, s.f[t].ctx);
    mprotect(s.stacks + (size_t)t * FiberSet::kStack, 4096, PROT_NONE);
    // the frame lce_fiber_switch pops: six callee-saved registers, then the entry point as its return address; above it one slot
    // so that the entry point starts with the stack alignment of a called function
    uint64_t* top = (uint64_t*)(s.stacks + (size_t)(t + 1) * FiberSet::kStack);
    top[-1] = 0;
    top[-2] = (uint64_t)(void*)&lce_fiber_entry;
    for (int r = 3; r <= 8; ++r) top[-r] = 0;
    s.f[t].sp = (void*)(top - 8);
  }
  FiberSet* outer = g_fibers;
  ThreadCtx* outer_ctx = g_ctxp;
  g_fibers = &s;
  s.cur = 0;
  g_ctxp = &s.f[0].ctx;
  lce_fiber_switch(&s.main_sp, s.f[0].sp);
  g_fibers = outer;
  g_ctxp = outer_ctx;
  munmap(s.stacks, bytes);
}


inline int thread_idx_x() { return g_ctx.tid_x; }
inline int block_idx_x() { return g_ctx.bid_x; }
inline int block_idx_y() { return g_ctx.bid_y; }
inline int block_dim_x() { return g_ctx.bdim_x; }
inline int grid_dim_x() { return g_ctx.gdim_x; }

inline uint32_t uniform(uint32_t x) { return x; }
inline int uniform(int x) { return x; }

struct rsrc_t { const uint8_t* base; uint32_t bytes; };
inline rsrc_t make_rsrc(const void* base, uint32_t bytes) { return rsrc_t{(const uint8_t*)base, bytes}; }
constexpr uint32_t kOobOffset = 0x80000000u;

template <typename V>
inline V buf_load_impl(rsrc_t r, uint32_t off) {
  V v;
  memset(&v, 0, sizeof v);
  // raw buffer, stride 0: the access is out of range iff off + size > num_records -> 0
  if ((uint64_t)off + sizeof(V) <= (uint64_t)r.bytes) memcpy(&v, r.base + off, sizeof v);
  return v;
}
inline uint32_t buf_load(rsrc_t r, uint32_t off, uint32_t*) { return buf_load_impl<uint32_t>(r, off); }
inline u32x2 buf_load(rsrc_t r, uint32_t off, u32x2*) { return buf_load_impl<u32x2>(r, off); }
inline u32x4 buf_load(rsrc_t r, uint32_t off, u32x4*) { return buf_load_impl<u32x4>(r, off); }
// the scalar offset moves the address but is outside the range check (as on the hardware)
inline u32x4 buf_load_so(rsrc_t r, uint32_t lane_off, uint32_t uniform_off, u32x4*) {
  u32x4 v;
  memset(&v, 0, sizeof v);
  if ((uint64_t)lane_off + 16 <= (uint64_t)r.bytes) memcpy(&v, r.base + lane_off + uniform_off, 16);
  return v;
}

inline uint32_t mulhi_u32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
inline int popc(uint32_t x) { return __builtin_popcount(x); }
inline float mul_then_add(float a, float b, float c) { volatile float p = a * b; return p + c; }

inline f32x2 mul_then_add2(f32x2 a, float b, float c) { f32x2 r; r[0] = mul_then_add(a[0], b, c); r[1] = mul_then_add(a[1], b, c); return r; }
inline f32x2 add2(f32x2 a, f32x2 b) { f32x2 r; { volatile float s = a[0] + b[0]; r[0] = s; } { volatile float s = a[1] + b[1]; r[1] = s; } return r; }

inline float fma1(float a, float b, float c) { return fmaf(a, b, c); }
inline f32x2 fma2(f32x2 a, float b, float c) { f32x2 r; r[0] = fmaf(a[0], b, c); r[1] = fmaf(a[1], b, c); return r; }
inline f32x2 mul_then_add_pk(f32x2 a, f32x2 b, f32x2 c) { f32x2 r; r[0] = mul_then_add(a[0], b[0], c[0]); r[1] = mul_then_add(a[1], b[1], c[1]); return r; }
inline void keep_alive(const u32x4&) {}

inline unsigned long long wave_ballot(bool p) {
  if (!g_ctx.bar) return p ? ~0ull : 0ull;  // sequential mode: only used via wave_any
  const int lane = g_ctx.tid_x & 63;
  g_ctx.xchg[lane] = p ? 1u : 0u;
  g_ctx.bar->arrive_and_wait();
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) m |= (unsigned long long)g_ctx.xchg[l] << l;
  g_ctx.bar->arrive_and_wait();
  return m;
}
// sequential mode: answering "true" is always safe -- the guarded code masks per lane
inline bool wave_any(bool p) { return g_ctx.bar ? wave_ballot(p) != 0ull : true; }
template <int LANE>
inline uint32_t write_lane(uint32_t value, uint32_t old) {
  return (g_ctx.tid_x & 63) == LANE ? value : old;
}
template <int LANE> inline uint32_t write_lane_settled(uint32_t value, uint32_t old) { return write_lane<LANE>(value, old); }
template <int LANE> inline uint32_t write_lane_settled_pad1(uint32_t value, uint32_t old) { return write_lane<LANE>(value, old); }
template <int N> inline void settle_ballots(unsigned long long (&)[N]) {}
inline void hold_until(unsigned long long&, unsigned long long, unsigned long long) {}
inline void hold_until(unsigned long long&, float, float) {}
inline uint32_t shfl_xor(uint32_t v, int mask) {
  const int lane = g_ctx.tid_x & 63;
  g_ctx.xchg[lane] = v;
  g_ctx.bar->arrive_and_wait();
  const uint32_t r = g_ctx.xchg[lane ^ mask];
  g_ctx.bar->arrive_and_wait();
  return r;
}

inline void xor_popc_acc(int& c0, uint32_t w, uint32_t a0) { c0 += popc(a0 ^ w); }
inline void xor_popc_acc(int& c0, int& c1, uint32_t w, uint32_t a0, uint32_t a1) {
  c0 += popc(a0 ^ w); c1 += popc(a1 ^ w);
}
inline void xor_popc_acc(int& c0, int& c1, int& c2, int& c3, uint32_t w, uint32_t a0, uint32_t a1,
                         uint32_t a2, uint32_t a3) {
  c0 += popc(a0 ^ w); c1 += popc(a1 ^ w); c2 += popc(a2 ^ w); c3 += popc(a3 ^ w);
}


// ---- matrix-core path (lock-step block mode only) ----------------------------------
// FP4 E2M1 code -> value for the codes the kernels use (0 -> 0, 0x2 -> +1, 0xA -> -1).
inline int fp4_pm1(uint32_t nib) { return nib == 0x2 ? 1 : nib == 0xA ? -1 : 0; }
inline f32x16 mfma_fp4_32x32x64(u32x4 a, u32x4 b, f32x16 c);
inline f32x16 mfma_fp4_32x32x64_unscaled(u32x4 a, u32x4 b, f32x16 c) { return mfma_fp4_32x32x64(a, b, c); }
// Every lane first reduces its 32 codes of A and of B to two 32-bit masks each (non-zero, negative), so that a dot product over the
// 64 K positions is three popcounts: sum = #(both non-zero) - 2 * #(both non-zero, signs differ).
inline void fp4_masks(const u32x4& v, uint32_t& nz, uint32_t& neg) {
  nz = 0; neg = 0;
  for (int j = 0; j < 32; ++j) {
    const int val = fp4_pm1((v[j / 8] >> (4 * (j % 8))) & 0xF);
    if (val != 0) nz |= 1u << j;
    if (val < 0) neg |= 1u << j;
  }
}
inline f32x16 mfma_fp4_32x32x64(u32x4 a, u32x4 b, f32x16 c) {
  const int lane = g_ctx.tid_x & 63;
  uint32_t* x = g_ctx.mfma_xchg;      // per lane: [a.nz, a.neg, b.nz, b.neg, ...]
  fp4_masks(a, x[lane * 8 + 0], x[lane * 8 + 1]);
  fp4_masks(b, x[lane * 8 + 2], x[lane * 8 + 3]);
  g_ctx.bar->arrive_and_wait();
  const int col = lane & 31;
  const uint64_t bnz = (uint64_t)x[col * 8 + 2] | (uint64_t)x[(col + 32) * 8 + 2] << 32;
  const uint64_t bneg = (uint64_t)x[col * 8 + 3] | (uint64_t)x[(col + 32) * 8 + 3] << 32;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const uint64_t anz = (uint64_t)x[row * 8 + 0] | (uint64_t)x[(row + 32) * 8 + 0] << 32;
    const uint64_t aneg = (uint64_t)x[row * 8 + 1] | (uint64_t)x[(row + 32) * 8 + 1] << 32;
    const uint64_t both = anz & bnz;
    const int sum = __builtin_popcountll(both) - 2 * __builtin_popcountll(both & (aneg ^ bneg));
    c[r] += (float)sum;
  }
  g_ctx.bar->arrive_and_wait();
  return c;
}
inline f32x16 f32x16_fill(float v) { f32x16 z; for (int i = 0; i < 16; ++i) z[i] = v; return z; }
inline f32x16 f32x16_zero() { f32x16 z; for (int i = 0; i < 16; ++i) z[i] = 0.0f; return z; }
inline uint8_t* lds_base() { return g_ctx.lds; }
// The simulated LDS-DMA is as asynchronous as the hardware allows, in the most hostile legal
// way: the destination is POISONED the moment the piece is issued (a reader that has not
// finished with that stage, or that reads before the data is due, sees garbage) and the
// real bytes land only when the issuing wave executes the s_waitcnt vmcnt(N) that retires
// the piece.  A missing or miscounted wait, or a refill of a stage still being read, turns
// into a wrong result instead of a lucky pass.
template <int IMM>
inline void buf_load_to_lds16(rsrc_t r, uint8_t* lds_dst, uint32_t lane_off, uint32_t uniform_off) {
  // the range check covers lane_off + IMM (the scalar offset is outside it on the hardware; the kernels never
  // rely on the check for these copies)
  const u32x4 v = buf_load_impl<u32x4>(r, lane_off + uniform_off + IMM);
  ThreadCtx::PendingDma p;
  p.dst = lds_dst + IMM + 16 * (g_ctx.tid_x & 63);
  memcpy(p.data, &v, 16);
  memset(p.dst, 0xEE, 16);
  g_ctx.dma.push_back(p);
}
inline void block_barrier_keep_vm() { g_ctx.block_bar->arrive_and_wait(); }
template <int N> inline void wait_vmcnt() {     // retire the oldest pieces until at most N are in flight
  while ((int)g_ctx.dma.size() > N) {
    memcpy(g_ctx.dma.front().dst, g_ctx.dma.front().data, 16);
    g_ctx.dma.pop_front();
  }
}
inline void wave_lds_fence() { if (g_ctx.bar) { g_ctx.bar->arrive_and_wait(); } }
inline void wave_lds_order() { if (g_ctx.bar) { g_ctx.bar->arrive_and_wait(); } }
inline void sched_fence() {}
inline void pin(f32x16&) {}
inline void keep_in_agpr(u32x4&) {}
inline void keep_in_vgpr(u32x4&) {}
inline void keep_in_vgpr(float&) {}
inline void keep_in_vgpr(int&) {}
inline uint32_t sat_add_u32(uint32_t a, uint32_t b) { const uint64_t r = (uint64_t)a + b; return r > 0xffffffffull ? 0xffffffffu : (uint32_t)r; }
inline void wave_lds_scratch_fence() { if (g_ctx.bar) { g_ctx.bar->arrive_and_wait(); } }
template <int N> inline void yield_issue_slots() {}
template <int N> inline void interleave_mfma_ldsread() {}
template <int NMFMA, int NDS, int NVMEM> inline void interleave_step() {}
inline void store_streaming(f32x4* p, f32x4 v) { *p = v; }
inline void store_streaming(u32x4* p, u32x4 v) { *p = v; }
inline void buf_store_streaming(rsrc_t r, uint32_t lane_off, f32x4 v) {
  if ((uint64_t)lane_off + 16 <= (uint64_t)r.bytes) memcpy(const_cast<uint8_t*>(r.base) + lane_off, &v, 16);
}
inline void buf_store_through(rsrc_t r, uint32_t lane_off, f32x4 v) { buf_store_streaming(r, lane_off, v); }
inline void store_words(uint32_t* p, uint32_t v) { *p = v; }
inline void store_words(uint32_t* p, u32x2 v) { memcpy(p, &v, 8); }
inline void store_words(uint32_t* p, u32x4 v) { memcpy(p, &v, 16); }
inline void buf_store(rsrc_t r, uint32_t lane_off, u32x4 v) {
  if ((uint64_t)lane_off + 16 <= (uint64_t)r.bytes) memcpy(const_cast<uint8_t*>(r.base) + lane_off, &v, 16);
}
// the scalar offset moves the address but is outside the range check (as on the hardware)
inline void buf_store_streaming_so(rsrc_t r, uint32_t lane_off, uint32_t uniform_off, f32x4 v) {
  if ((uint64_t)lane_off + 16 <= (uint64_t)r.bytes) memcpy(const_cast<uint8_t*>(r.base) + lane_off + uniform_off, &v, 16);
}
inline void buf_store_so(rsrc_t r, uint32_t lane_off, uint32_t uniform_off, u32x4 v) {
  if ((uint64_t)lane_off + 16 <= (uint64_t)r.bytes) memcpy(const_cast<uint8_t*>(r.base) + lane_off + uniform_off, &v, 16);
}
inline void buf_store2(rsrc_t r, uint32_t lane_off, u32x2 v) {
  if ((uint64_t)lane_off + 8 <= (uint64_t)r.bytes) memcpy(const_cast<uint8_t*>(r.base) + lane_off, &v, 8);
}
inline void buf_store1(rsrc_t r, uint32_t lane_off, uint32_t v) {
  if ((uint64_t)lane_off + 4 <= (uint64_t)r.bytes) memcpy(const_cast<uint8_t*>(r.base) + lane_off, &v, 4);
}
inline f32x4 load_streaming(const f32x4* p) { return *p; }
inline u32x4 load_streaming(const u32x4* p) { return *p; }
inline float med3(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) {
  const uint64_t pool = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) {
    const uint32_t s = (sel >> (8 * i)) & 0xff;
    const uint32_t b = s < 8 ? (uint32_t)(pool >> (8 * s)) & 0xff : 0u;   // only 0-7 and 0x0c (zero) are used
    r |= b << (8 * i);
  }
  return r;
}
template <int SLO, int SHI>
inline uint32_t pk_lshr_b16(uint32_t v) { return ((v & 0xffffu) >> SLO) | (((v >> 16) >> SHI) << 16); }
inline uint32_t pack4_u8(int q0, int q1, int q2, int q3) {
  return (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
}

inline void cvt_pack8_i8(const f32x4& a, const f32x4& b, uint32_t& lo, uint32_t& hi) {
  lo = pack4_u8((int)a[0], (int)a[1], (int)a[2], (int)a[3]);
  hi = pack4_u8((int)b[0], (int)b[1], (int)b[2], (int)b[3]);
}

inline void cvt_rpi_pack8_i8(const f32x4& a, const f32x4& b, uint32_t& lo, uint32_t& hi) {      // floor(x + 0.5), exactly
  auto r = [](float x) { return (int)floor((double)x + 0.5); };
  lo = pack4_u8(r(a[0]), r(a[1]), r(a[2]), r(a[3]));
  hi = pack4_u8(r(b[0]), r(b[1]), r(b[2]), r(b[3]));
}

}  // namespace lce_dev
