// gfx950 (CDNA4) device intrinsics used by the LCE kernels.
//
// Everything ISA-specific that the kernels need is funnelled through this header so
// that the kernel bodies in lce_kernels.h read as plain index arithmetic.  (The test
// tree has a lock-step-free host replacement of this one header, tests/hostsim/, which
// lets the CPU-only test suite execute the very same kernel bodies thread by thread to
// check their index/padding/epilogue logic; the product is only ever built with this
// file.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lce_experiments.h"

#define LCE_DEVICE __device__ __forceinline__
#define LCE_LAMBDA_INLINE __attribute__((always_inline))
#define LCE_KERNEL __global__

namespace lce_dev {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;  // CDNA wavefront width

LCE_DEVICE int thread_idx_x() { return (int)threadIdx.x; }
LCE_DEVICE int block_idx_x() { return (int)blockIdx.x; }
LCE_DEVICE int block_idx_y() { return (int)blockIdx.y; }
LCE_DEVICE int block_dim_x() { return (int)blockDim.x; }
LCE_DEVICE int grid_dim_x() { return (int)gridDim.x; }

// Promote a value the programmer knows to be wave-uniform into an SGPR so that the
// loads indexed by it become scalar (s_load) and the VALU ops take it as an SGPR operand.
LCE_DEVICE uint32_t uniform(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
LCE_DEVICE int uniform(int x) { return (int)__builtin_amdgcn_readfirstlane((uint32_t)x); }

// Raw buffer resource over [base, base+bytes): loads whose byte offset falls outside
// return 0 -- which is exactly the "+1" padding word of the reference
// (core/bconv2d/reference.h:105-106), so out-of-image taps need no select.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
LCE_DEVICE rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride=*/0, (int)bytes,
                                           /*flags: DATA_FORMAT=32*/ 0x00020000);
}
constexpr uint32_t kOobOffset = 0x80000000u;  // > any num_records we ever bind (< 2 GiB)

LCE_DEVICE uint32_t buf_load(rsrc_t r, uint32_t byte_off, uint32_t*) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0);
}
LCE_DEVICE u32x2 buf_load(rsrc_t r, uint32_t byte_off, u32x2*) {
  return __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0);
}
LCE_DEVICE u32x4 buf_load(rsrc_t r, uint32_t byte_off, u32x4*) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}
// ... with a wave-uniform byte offset on top of the per-lane one (an SGPR operand of the instruction: no vector add).  Only the
// per-lane offset is range-checked, as for the stores below.
LCE_DEVICE u32x4 buf_load_so(rsrc_t r, uint32_t lane_off, uint32_t uniform_off, u32x4*) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uniform_off, 0);
}

LCE_DEVICE uint32_t mulhi_u32(uint32_t a, uint32_t b) { return __umulhi(a, b); }
LCE_DEVICE int popc(uint32_t x) { return __popc(x); }
LCE_DEVICE bool wave_any(bool p) { return __ballot(p) != 0ull; }
LCE_DEVICE unsigned long long wave_ballot(bool p) { return __ballot(p); }
// v_writelane_b32: returns `old` with lane LANE (a compile-time constant) replaced by the
// wave-uniform `value`.  HAZARD: an SGPR written by a VALU compare must not be read by
// v_writelane for 4 wait states, and hipcc pads nothing inside or in front of inline asm (without
// the padding lanes came back stale on gfx950).  write_lane() pads itself; write_lane_settled()
// does not and may only take values that went through settle_ballots() after the compares.
template <int LANE>
LCE_DEVICE uint32_t write_lane(uint32_t value, uint32_t old) {
  asm("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(value), "n"(LANE));
  return old;
}
template <int LANE>
LCE_DEVICE uint32_t write_lane_settled(uint32_t value, uint32_t old) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(value), "n"(LANE));
  return old;
}
// ... preceded by ONE wait state: for pipelined gathers whose worst-case distance to the compare is exactly the four instructions of
// the rule (gather_tile_bits, lce_kernels_mfma.h) -- the margin of the old `s_nop 4` (five wait states) for one cycle.
template <int LANE>
LCE_DEVICE uint32_t write_lane_settled_pad1(uint32_t value, uint32_t old) {
  asm("s_nop 0\n\tv_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(value), "n"(LANE));
  return old;
}
// One padded point for a group of ballot results: the "+s" operands order every compare in front
// of it and every write_lane_settled() of these values behind it.
template <int N>
LCE_DEVICE void settle_ballots(unsigned long long (&b)[N]) {
  static_assert(N >= 1 && N <= 4, "one operand list per N below");
  if constexpr (N == 1) asm("s_nop 4" : "+s"(b[0]));
  else if constexpr (N == 2) asm("s_nop 4" : "+s"(b[0]), "+s"(b[1]));
  else if constexpr (N == 3) asm("s_nop 4" : "+s"(b[0]), "+s"(b[1]), "+s"(b[2]));
  else asm("s_nop 4" : "+s"(b[0]), "+s"(b[1]), "+s"(b[2]), "+s"(b[3]));
}
// Ordering aids for ballots that are written to their lanes one K-step AFTER the compares (lce_kernels_stream.h: the MFMAs and the
// next unit's work in between are the wait states): `held` cannot be consumed before the values on the right exist.
LCE_DEVICE void hold_until(unsigned long long& held, unsigned long long a, unsigned long long b) { asm("" : "+s"(held) : "s"(a), "s"(b)); }
LCE_DEVICE void hold_until(unsigned long long& held, float a, float b) { asm("" : "+s"(held) : "v"(a), "v"(b)); }
LCE_DEVICE uint32_t shfl_xor(uint32_t v, int mask) { return (uint32_t)__shfl_xor((int)v, mask, 64); }

// a * b + c with TWO roundings, as the reference's portable C++ computes it
// (core/bconv2d/output_transform.h:105).  hipcc's default -ffp-contract=fast fuses `a * b + c`
// -- and even __fadd_rn(__fmul_rn(a, b), c) -- into v_fma_f32 / v_pk_fma_f32; the pragma
// switches contraction off for this function whatever the command line says (the library is
// built with -ffp-contract=off as well).  An empty-asm fence on the product did the same job
// but pinned the order of 128 multiply-adds per lane and kept them out of v_pk_mul_f32 /
// v_pk_add_f32: without it the float epilogue is 3 % faster.  The GPU parity tests are
// bit-exact, so a build that fused would fail them.
LCE_DEVICE float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
  const float p = a * b;
  return p + c;
}

// Two of them at once: v_pk_mul_f32 + v_pk_add_f32 (each element rounded twice, exactly as above).
#ifdef LCE_NO_PK_F32   // experiment: the same pairs with scalar instructions
LCE_DEVICE f32x2 mul_then_add2(f32x2 a, float b, float c) {
  f32x2 r = {mul_then_add(a[0], b, c), mul_then_add(a[1], b, c)};
  asm volatile("" : "+v"(r));
  return r;
}
LCE_DEVICE f32x2 add2(f32x2 a, f32x2 b) {
  float x = a[0] + b[0], y = a[1] + b[1];
  asm volatile("" : "+v"(x), "+v"(y));
  return f32x2{x, y};
}
#else
LCE_DEVICE f32x2 mul_then_add2(f32x2 a, float b, float c) {
#pragma clang fp contract(off)
  const f32x2 bb = {b, b}, cc = {c, c};
  const f32x2 p = a * bb;
  return p + cc;
}
LCE_DEVICE f32x2 add2(f32x2 a, f32x2 b) { return a + b; }   // v_pk_add_f32
#endif
// ... with the per-channel constants already duplicated into register pairs
LCE_DEVICE f32x2 mul_then_add_pk(f32x2 a, f32x2 b, f32x2 c) {
#pragma clang fp contract(off)
  const f32x2 p = a * b;
  return p + c;
}
// ONE rounding: v_fma_f32 / v_pk_fma_f32.  Not the reference's transform (two roundings, mul_then_add): only the int8 instances whose plan
// the planner has PROVEN byte-identical with it use these (I8F: lce_plan.cpp, prepare_int8_epilogue).
LCE_DEVICE float fma1(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
LCE_DEVICE f32x2 fma2(f32x2 a, float b, float c) {
  const f32x2 bb = {b, b}, cc = {c, c};
  return __builtin_elementwise_fma(a, bb, cc);
}
// A use of `v` that generates nothing: keeps its registers allocated (and unmodified) up to this point.
LCE_DEVICE void keep_alive(const u32x4& v) { asm volatile("" ::"v"(v)); }

// acc_i += popcount(a_i ^ w) for TM independent activations against ONE weight word.
// Hand-written so that (1) the weight word stays in an SGPR (VOP2 src0), (2) the
// accumulate is the free addend of v_bcnt_u32_b32 instead of a separate v_add (hipcc
// re-associates `acc += __popc(..)` chains into bcnt(x,0) + v_add3), and (3) the TM
// xors are issued ahead of the TM dependent bcnts.
LCE_DEVICE void xor_popc_acc(int& c0, uint32_t w, uint32_t a0) {
  uint32_t t0;
  asm("v_xor_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0"
      : "+v"(c0), "=&v"(t0)
      : "s"(w), "v"(a0));
}
LCE_DEVICE void xor_popc_acc(int& c0, int& c1, uint32_t w, uint32_t a0, uint32_t a1) {
  uint32_t t0, t1;
  asm("v_xor_b32 %2, %4, %5\n\tv_xor_b32 %3, %4, %6\n\t"
      "v_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %1, %3, %1"
      : "+v"(c0), "+v"(c1), "=&v"(t0), "=&v"(t1)
      : "s"(w), "v"(a0), "v"(a1));
}
LCE_DEVICE void xor_popc_acc(int& c0, int& c1, int& c2, int& c3, uint32_t w, uint32_t a0,
                             uint32_t a1, uint32_t a2, uint32_t a3) {
  uint32_t t0, t1, t2, t3;
  asm("v_xor_b32 %4, %8, %9\n\tv_xor_b32 %5, %8, %10\n\t"
      "v_xor_b32 %6, %8, %11\n\tv_xor_b32 %7, %8, %12\n\t"
      "v_bcnt_u32_b32 %0, %4, %0\n\tv_bcnt_u32_b32 %1, %5, %1\n\t"
      "v_bcnt_u32_b32 %2, %6, %2\n\tv_bcnt_u32_b32 %3, %7, %3"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "s"(w), "v"(a0), "v"(a1), "v"(a2), "v"(a3));
}


// ---- matrix-core path -------------------------------------------------------------
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// D(32x32) += A(32x64) * B(64x32) with FP4 (E2M1) operands and unit block scales
// (v_mfma_scale_f32_32x32x64_f8f6f4, cbsz = blgp = 4, E8M0 scale 0x7F = 2^0).
// Operand layout (verified on gfx950 by tools/probes/mfma_fp4_probe.hip): lane l supplies
// row/column (l & 31) and the 32 K-values 32*(l >> 5) + j in nibble j (LSB first) of its
// 16 bytes; D: column = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) for register r.
// Products of +-1 / 0 accumulate exactly in fp32 (|sum| < 2^24).
LCE_DEVICE f32x16 mfma_fp4_32x32x64(u32x4 a, u32x4 b, f32x16 c) {
  i32x8 va = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0};
  i32x8 vb = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);   // E8M0 0x7F = 1.0
}
// The same product from the UNSCALED instruction: scale operands 0 make the compiler select v_mfma_f32_32x32x64_f8f6f4
// (cbsz:4 blgp:4, one 8-byte instruction, inputs taken at scale 1) instead of the scaled form above (16 bytes:
// v_mfma_ld_scale_b32 + v_mfma).  For a wave that is ALONE on its SIMD and fills the gaps behind its MFMAs with other work
// the scaled form costs ~2 more issue cycles per MFMA and one of the filler slots (tools/probes/mfma_gap.hip: 39.2 vs 37.2
// cycles per MFMA with four v_mul behind it): the streaming kernel runs L0 1.7 % (float) / 4.7 % (int8) / 3 % (bitpacked)
// faster with it, the pointwise kernel 2 %.  The block GEMM -- two waves per SIMD, bound by the matrix pipe -- is 4 %
// SLOWER with it on 7x7x512 and keeps the scaled form (profiles/r03/mfma_scaled_vs_unscaled.txt).  Bit-identical results.
LCE_DEVICE f32x16 mfma_fp4_32x32x64_unscaled(u32x4 a, u32x4 b, f32x16 c) {
  i32x8 va = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0};
  i32x8 vb = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], 0, 0, 0, 0};
#ifdef LCE_MFMA_SCALED   // (A/B aid)
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
#else
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 0, 0, 0);
#endif
}
LCE_DEVICE f32x16 f32x16_fill(float v) {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = v;
  return z;
}
LCE_DEVICE f32x16 f32x16_zero() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}

// All LDS of a kernel lives in ONE dynamic array (16-byte aligned carve base).
LCE_DEVICE uint8_t* lds_base() {
  extern __shared__ __attribute__((aligned(16))) uint8_t lce_lds[];
  return lce_lds;
}
// Asynchronous global -> LDS copy (buffer_load_dwordx4 ... lds): every lane fetches 16 bytes
// at its own buffer offset; the wave's 64 x 16 bytes land CONTIGUOUSLY at lds_dst + 16*lane
// (lds_dst is wave-uniform).  No VGPR round trip, no ds_write.  Completion is tracked by
// vmcnt: wait_vmcnt<N>() below, then a barrier, before another wave reads the bytes.
// lane_off: per-lane byte offset (VGPR); uniform_off: wave-uniform byte offset (SGPR, no VALU add); IMM: an
// instruction-immediate byte offset (< 4096) that moves BOTH the source and the LDS destination.
template <int IMM>
LCE_DEVICE void buf_load_to_lds16(rsrc_t r, uint8_t* lds_dst, uint32_t lane_off, uint32_t uniform_off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, lane_off, uniform_off, IMM, 0);
}
// Barrier that does NOT drain the VM counter: LDS-DMA copies issued for later pipeline
// stages stay in flight across it.  Pair with wait_vmcnt<N>() for the stage being consumed.
// (The waits use the s_waitcnt BUILTIN, not inline asm, so that hipcc's own wait-count
//  bookkeeping sees them; with asm it re-waits conservatively in front of the MFMAs.)
// gfx9 s_waitcnt immediate: vmcnt = [15:14]:[3:0], expcnt = [6:4], lgkmcnt = [11:8].
LCE_DEVICE void block_barrier_keep_vm() {
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
}
template <int N>
LCE_DEVICE void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt((((N >> 4) & 3) << 14) | (0xF << 8) | (0x7 << 4) | (N & 0xF));
}
// Orders a wave's own LDS writes before its own later LDS reads (and vice versa) when the
// accesses go through differently-typed pointers: wait for the LDS queue, and stop the
// compiler from moving LDS accesses across.
LCE_DEVICE void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Orders a wave's own LDS accesses in PROGRAM order without waiting: the LDS executes one wave's DS operations in
// issue order, so a later read sees an earlier write of another lane of the same wave; only the compiler must be
// stopped from moving accesses across.  (The host simulation, whose lanes are threads, synchronises here.)
LCE_DEVICE void wave_lds_order() { asm volatile("" ::: "memory"); }
// Nothing is scheduled across this point (the stream kernel places its fillers K-step by K-step).
LCE_DEVICE void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// s_sleep N: the wave gives up its issue slots for ~64*N cycles
template <int N>
LCE_DEVICE void yield_issue_slots() { __builtin_amdgcn_s_sleep(N); }
// Pins an accumulator tile at this point of the program: the MFMAs that produce it cannot
// be sunk below (hipcc otherwise moves the register-only MFMAs of a K-step past the NEXT
// step's barrier, which serialises LDS latency and matrix work).
LCE_DEVICE void pin(f32x16& c) { asm volatile("" : "+v"(c)); }
// Register-class hints for operands that stay resident for the life of a wave (the stream kernel's filter bank):
// the MFMA A/B operands may be AGPRs on gfx950, and a value that passes through an "a"-constrained asm lives in the
// accumulator half of the unified register file from then on -- without it the allocator keeps such values in VGPRs,
// runs out, and copies them through v_accvgpr_read in front of every use.
LCE_DEVICE void keep_in_agpr(u32x4& v) { asm volatile("" : "+a"(v)); }
LCE_DEVICE void keep_in_vgpr(u32x4& v) { asm volatile("" : "+v"(v)); }
LCE_DEVICE void keep_in_vgpr(float& v) { asm volatile("" : "+v"(v)); }
LCE_DEVICE void keep_in_vgpr(int& v) { asm volatile("" : "+v"(v)); }   // a uniform value kept per lane (spares a scalar register)
// a + b, saturating at 2^32 - 1 (v_add_u32 ... clamp): two "out of range" markers must not add up to an in-range offset
LCE_DEVICE uint32_t sat_add_u32(uint32_t a, uint32_t b) { return __builtin_elementwise_add_sat(a, b); }
// Between a wave's scratch writes and its reads of other lanes' values.  The LDS runs one wave's DS operations in
// issue order, so ordering the instructions is enough (LCE_STREAM_LDS_WAIT: also wait for the LDS queue).
#ifdef LCE_STREAM_LDS_WAIT
LCE_DEVICE void wave_lds_scratch_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
LCE_DEVICE void wave_lds_scratch_fence() { asm volatile("" ::: "memory"); }
#endif
// Ask the scheduler for the issue pattern {1 MFMA, 1 LDS read} x n inside the current region.
template <int N>
LCE_DEVICE void interleave_mfma_ldsread() {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
  }
}
// Issue pattern of a steady K-step: MFMA, LDS-DMA, MFMA, LDS-DMA, MFMA, 2 LDS reads, MFMA, 2 LDS reads, ...
// (masks: 0x8 MFMA, 0x10 vector memory, 0x100 LDS read).  Scalar / vector ALU instructions are
// left to the scheduler; the ones an address depends on end up in front of that access.
template <int M, int NMFMA, int NDS, int NVMEM>
LCE_DEVICE void interleave_step_from() {
  if constexpr (M < NMFMA) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (M < NVMEM) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    // fragment reads as early as the DMAs allow, two per MFMA (measured against an even spread: -1 %;
    // DMAs later in the step: +2.5...5 %)
    constexpr int ds = (M >= NVMEM && 2 * (M - NVMEM) < NDS) ? ((NDS - 2 * (M - NVMEM)) >= 2 ? 2 : 1) : 0;
    if constexpr (ds > 0) __builtin_amdgcn_sched_group_barrier(0x100, ds, 0);
    interleave_step_from<M + 1, NMFMA, NDS, NVMEM>();
  }
}
template <int NMFMA, int NDS, int NVMEM>
LCE_DEVICE void interleave_step() { interleave_step_from<0, NMFMA, NDS, NVMEM>(); }
// Streaming (non-temporal) 16-byte store for outputs that are written once and not re-read
// by this kernel: keeps the L2 for the operands that ARE re-read.
LCE_DEVICE void store_streaming(f32x4* p, f32x4 v) { __builtin_nontemporal_store(v, p); }
LCE_DEVICE void store_streaming(u32x4* p, u32x4 v) { __builtin_nontemporal_store(v, p); }
// 16-byte stores through a buffer resource (the per-lane offset is range-checked by the hardware):
// non-temporal for float rows that are written once, plain for the rest
#ifndef LCE_STORE_AUX
#define LCE_STORE_AUX 2   /* nt */
#endif
LCE_DEVICE void buf_store_streaming(rsrc_t r, uint32_t lane_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, lane_off, 0, LCE_STORE_AUX);
}
// ... write-through (agent scope, sc1: the line goes on to memory and is allocated in the Infinity Cache instead of waiting
// dirty in this XCD's L2).  An A/B aid for float rows, NOT what is built: timed as repeated launches of one layer it looks 4-12 %
// faster than nt (the launches rewrite ONE buffer, which the 256 MB Infinity Cache then absorbs), timed on QuickNet's chain of
// 16 layers with their own buffers it is 16 % SLOWER (profiles/r03/store_cache_policy{,_chains}.txt)
#ifndef LCE_STORE_THROUGH_AUX
#define LCE_STORE_THROUGH_AUX 2    /* nt; 16 = sc1 */
#endif
LCE_DEVICE void buf_store_through(rsrc_t r, uint32_t lane_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, lane_off, 0, LCE_STORE_THROUGH_AUX);
}
// int8 rows: write-through.  On the chain of config 5's twelve int8 layers -4 % against plain write-back stores (isolated
// layers: -4...-5 % on the block GEMM, -13...-15 % on the pointwise kernel, neutral on the streaming kernel)
// Bitpacked words (the bitpacked output type, the second output's sign words): a relaxed agent-scope atomic store IS a
// global_store with sc1 -- write-through; measured no better than ordinary write-back stores (profiles/r03/store_cache_policy.txt),
// so that is an A/B aid (-DLCE_WORDS_THROUGH) and the default is a plain store
LCE_DEVICE void store_words(uint32_t* p, uint32_t v) {
#ifndef LCE_WORDS_THROUGH
  *p = v;
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
LCE_DEVICE void store_words(uint32_t* p, u32x2 v) {      // p is 8-byte aligned
#ifndef LCE_WORDS_THROUGH
  *(u32x2*)p = v;
#else
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)v[0] | ((unsigned long long)v[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
LCE_DEVICE void store_words(uint32_t* p, u32x4 v) {      // p is 16-byte aligned
#ifndef LCE_WORDS_THROUGH
  *(u32x4*)p = v;
#else
  store_words(p, u32x2{v[0], v[1]});
  store_words(p + 2, u32x2{v[2], v[3]});
#endif
}
#ifndef LCE_STORE8_AUX
#define LCE_STORE8_AUX 16  /* sc1 */
#endif
LCE_DEVICE void buf_store(rsrc_t r, uint32_t lane_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, lane_off, 0, LCE_STORE8_AUX);
}
// ... with a scalar byte offset on top of the per-lane one (a K-step's row offset costs no vector instruction)
LCE_DEVICE void buf_store_streaming_so(rsrc_t r, uint32_t lane_off, uint32_t uniform_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, lane_off, uniform_off, LCE_STORE_AUX);
}
LCE_DEVICE void buf_store_so(rsrc_t r, uint32_t lane_off, uint32_t uniform_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, lane_off, uniform_off, LCE_STORE8_AUX);
}
#ifndef LCE_STOREW_AUX
#define LCE_STOREW_AUX 0
#endif
LCE_DEVICE void buf_store2(rsrc_t r, uint32_t lane_off, u32x2 v) { __builtin_amdgcn_raw_buffer_store_b64(v, r, lane_off, 0, LCE_STOREW_AUX); }
LCE_DEVICE void buf_store1(rsrc_t r, uint32_t lane_off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, lane_off, 0, LCE_STOREW_AUX); }
// ... and the matching load for inputs that are read exactly once
LCE_DEVICE f32x4 load_streaming(const f32x4* p) { return __builtin_nontemporal_load(p); }
LCE_DEVICE u32x4 load_streaming(const u32x4* p) { return __builtin_nontemporal_load(p); }
LCE_DEVICE float med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
// v_perm_b32: result byte i = byte sel[i] of the 8-byte pool {hi (4-7), lo (0-3)}; 0x0c = 0x00
LCE_DEVICE uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// v_pk_lshrrev_b16: the two 16-bit halves shifted right by their own amounts
template <int SLO, int SHI>
LCE_DEVICE uint32_t pk_lshr_b16(uint32_t v) {
  typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
  u16x2_t x = __builtin_bit_cast(u16x2_t, v);
  const u16x2_t sh = {(unsigned short)SLO, (unsigned short)SHI};
  x = x >> sh;
  return __builtin_bit_cast(uint32_t, x);
}
// low bytes of four ints -> one dword (v_perm_b32: selector byte 0-3 = src1 bytes, 4-7 = src0, 0x0c = 0)
LCE_DEVICE uint32_t pack4_u8(int q0, int q1, int q2, int q3) {
  const uint32_t lo = __builtin_amdgcn_perm((uint32_t)q1, (uint32_t)q0, 0x0c0c0400u);
  const uint32_t hi = __builtin_amdgcn_perm((uint32_t)q3, (uint32_t)q2, 0x0c0c0400u);
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// Eight floats -> the low bytes of their truncated integers, in two dwords.  v_cvt_i32_f32 with an SDWA destination byte converts AND
// packs (four v_cvt + three v_perm per dword otherwise).  gfx940+: a VALU read of a register needs one wait state behind a dst_sel
// write of it (LLVM's DstSelForwardingHazard -- its recognizer cannot see into inline asm; UNUSED_PRESERVE reads the destination): the
// two dwords' conversions alternate and an s_nop ends the block.
LCE_DEVICE void cvt_pack8_i8(const f32x4& a, const f32x4& b, uint32_t& lo, uint32_t& hi) {
  asm("v_cvt_i32_f32_sdwa %0, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
      "v_cvt_i32_f32_sdwa %1, %6 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
      "v_cvt_i32_f32_sdwa %0, %3 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_cvt_i32_f32_sdwa %1, %7 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_cvt_i32_f32_sdwa %0, %4 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_cvt_i32_f32_sdwa %1, %8 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_cvt_i32_f32_sdwa %0, %5 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_cvt_i32_f32_sdwa %1, %9 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "s_nop 0"
      : "=&v"(lo), "=&v"(hi)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}

// The same eight conversions with v_cvt_rpi_i32_f32 = floor(x + 0.5) -- EXACTLY, for every float of magnitude <= 129, plain and in
// this SDWA byte form (tools/probes/cvt_rpi.hip runs all of them) -- instead of truncation: round-half-up in one instruction per value,
// no copysign / add in front of it.  Differs from the reference's round-half-AWAY only at exact negative ties, which the planner rules
// out per plan before it selects a kernel built with it (lce_plan.cpp, prepare_int8_epilogue).
LCE_DEVICE void cvt_rpi_pack8_i8(const f32x4& a, const f32x4& b, uint32_t& lo, uint32_t& hi) {
  asm("v_cvt_rpi_i32_f32_sdwa %0, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
      "v_cvt_rpi_i32_f32_sdwa %1, %6 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
      "v_cvt_rpi_i32_f32_sdwa %0, %3 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_cvt_rpi_i32_f32_sdwa %1, %7 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_cvt_rpi_i32_f32_sdwa %0, %4 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_cvt_rpi_i32_f32_sdwa %1, %8 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_cvt_rpi_i32_f32_sdwa %0, %5 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "v_cvt_rpi_i32_f32_sdwa %1, %9 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
      "s_nop 0"
      : "=&v"(lo), "=&v"(hi)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}

}  // namespace lce_dev
