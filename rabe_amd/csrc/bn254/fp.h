// BN254 prime-field arithmetic for gfx950: 8 x 32-bit limbs, Montgomery form (R = 2^256).
//
// This is the bottom of the engine that replaces the `rabe_bn` Fr / Fq arithmetic the
// reference reaches through `use rabe_bn::{Fr, G1, G2, Gt, pairing}` (src/schemes/ac17/mod.rs:42).
// The limb width is 32 bits because CDNA4 has no 64-bit integer multiplier: a limb product is one
// `v_mad_u64_u32` (32x32+64 -> 64).  All loops are fully unrolled with compile-time indices so the
// limbs live in VGPRs (cdna_hip_programming.md rule 20: runtime-indexed arrays go to scratch).
//
// The same template serves Fp (base field) and Fr (scalar field); only the constants differ.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "constants.h"

// Inlining policy (code size / compile time vs call overhead on gfx950):
//   RB_HD     small helpers (add/sub/neg/select, constants): always inlined.
//   RB_FN     real functions.  Field multiplications are functions at Fp and Fp2 granularity whose
//             operands are single-member structs of an 8-lane vector type, which the AMDGPU calling
//             convention passes in VGPRs (no scratch traffic); everything from Fp6 upwards (tower,
//             curve, pairing steps) is a function taking references -- its operands are 48..96 dwords
//             and the memory traffic is <5% of the multiplications inside.
#define RB_HD __host__ __device__ __forceinline__
#define RB_FN __host__ __device__ inline __attribute__((noinline))
#define RB_HD_NOINLINE RB_FN
#if defined(__HIP_DEVICE_COMPILE__)
#define RB_MID __host__ __device__ __forceinline__
#else
#define RB_MID RB_FN
#endif

#if defined(RB_COUNT_MULS) && !defined(__HIP_DEVICE_COMPILE__)
extern "C" unsigned long long rb_mul_counter;
#endif

namespace rabe { namespace bn254 {

struct FpParams {
  static RB_HD constexpr uint32_t mod(int i) { constexpr uint32_t m[8] = RB_FP_MOD; return m[i]; }
  static RB_HD constexpr uint32_t one(int i) { constexpr uint32_t m[8] = RB_FP_ONE; return m[i]; }
  static RB_HD constexpr uint32_t r2(int i) { constexpr uint32_t m[8] = RB_FP_R2; return m[i]; }
  static RB_HD constexpr uint32_t exp_inv(int i) { constexpr uint32_t m[8] = RB_FP_PM2; return m[i]; }
  static constexpr uint32_t INV = RB_FP_INV32;
};
struct FrParams {
  static RB_HD constexpr uint32_t mod(int i) { constexpr uint32_t m[8] = RB_FR_MOD; return m[i]; }
  static RB_HD constexpr uint32_t one(int i) { constexpr uint32_t m[8] = RB_FR_ONE; return m[i]; }
  static RB_HD constexpr uint32_t r2(int i) { constexpr uint32_t m[8] = RB_FR_R2; return m[i]; }
  static RB_HD constexpr uint32_t exp_inv(int i) { constexpr uint32_t m[8] = RB_FR_RM2; return m[i]; }
  static constexpr uint32_t INV = RB_FR_INV32;
};

// A field element in Montgomery form, fully reduced (< modulus).
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
template <class M>
struct Mont {
  u32x8 v;   // single-member struct of a vector: passed and returned in 8 VGPRs
};
typedef Mont<FpParams> Fp;
typedef Mont<FrParams> Fr;

// ---------------------------------------------------------------------------------------------
// raw 256-bit helpers: explicit carry chains (clang lowers __builtin_addc/__builtin_subc to
// v_add_co_u32 / v_addc_co_u32 / v_subb_co_u32 on gfx950 -- one instruction per limb).
RB_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t& carry) { return __builtin_addc(a, b, carry, &carry); }
RB_HD uint32_t subb32(uint32_t a, uint32_t b, uint32_t& borrow) { return __builtin_subc(a, b, borrow, &borrow); }

template <class M>
RB_HD bool is_zero(const Mont<M>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i];
  return o == 0;
}
template <class M>
RB_HD bool eq(const Mont<M>& a, const Mont<M>& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}
template <class M>
RB_HD Mont<M> zero() {
  Mont<M> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = 0;
  return r;
}
template <class M>
RB_HD Mont<M> one() {
  Mont<M> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = M::one(i);
  return r;
}

// RB_SAFE_CARRY (a build switch, `python -m rabe_amd.build --safe`): every carry dependency gets the wait states LLVM's gfx940+
// hazard table asks for -- the additive chains fall back to compiler-scheduled code (which hipcc pads itself), the multiply-
// accumulate statements put `s_nop 1` between a carry-writing instruction and its first reader (RB_CPAD), fp_lin9 falls back to
// the doubling form.  The fast build relies on the hardware interlocking these dependencies (tools/ubench_addc.hip);
// tests/test_gpu_carry_interlock.py runs the same vectors through both builds and requires identical bytes.
#if defined(RB_SAFE_CARRY)
#define RB_CPAD "s_nop 1\n\t"
#ifndef RB_NO_LIN9
#define RB_NO_LIN9
#endif
#else
#define RB_CPAD ""
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RB_SAFE_CARRY)
// ---- gfx950 carry chains.  hipcc pads every carry-dependent v_addc / v_subb with two wait states (LLVM's gfx940+ rule "VALU
// writes SGPR/VCC -> VALU reads it"); with one wave per SIMD each pad costs ~6 cycles, which doubles the time of an add / sub
// chain (tools/ubench_addc.hip: 113 vs 55 cycles per 9-instruction chain) -- ~15 % of a pairing kernel.  The hardware interlocks
// the dependency itself: a chain without the pads runs at 6.1 cycles per instruction against ~4 for independent ones, and its
// results are identical on 7.5e11 dependent instructions (same tool; the multiply-accumulate chains above have relied on the
// same interlock since round 1).  So the 8-limb chains are written here as single asm statements.
#define RB_CH8(first_, next_, lit_) \
  first_ " %0, vcc, " lit_ "%8, %0\n\t" next_ " %1, vcc, " lit_ "%9, %1, vcc\n\t" next_ " %2, vcc, " lit_ "%10, %2, vcc\n\t" \
  next_ " %3, vcc, " lit_ "%11, %3, vcc\n\t" next_ " %4, vcc, " lit_ "%12, %4, vcc\n\t" next_ " %5, vcc, " lit_ "%13, %5, vcc\n\t" \
  next_ " %6, vcc, " lit_ "%14, %6, vcc\n\t" next_ " %7, vcc, " lit_ "%15, %7, vcc"
// t += b  (8 limbs, carry out dropped: the callers' sums stay below 2^256)
RB_HD void limbs_add8(uint32_t* t, const uint32_t* b) {
  asm(RB_CH8("v_add_co_u32", "v_addc_co_u32", "")
      : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7])
      : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
}
// t <- t - mod if t >= mod, for t < 2*mod < 2^255 (so t itself never carries out of 256 bits)
template <class M>
RB_HD void cond_sub_mod(uint32_t* t, uint32_t /*hi: always 0*/) {
  uint32_t d0, d1, d2, d3, d4, d5, d6, d7;
  asm("v_subrev_co_u32 %8, vcc, %16, %0\n\tv_subbrev_co_u32 %9, vcc, %17, %1, vcc\n\tv_subbrev_co_u32 %10, vcc, %18, %2, vcc\n\t"
      "v_subbrev_co_u32 %11, vcc, %19, %3, vcc\n\tv_subbrev_co_u32 %12, vcc, %20, %4, vcc\n\tv_subbrev_co_u32 %13, vcc, %21, %5, vcc\n\t"
      "v_subbrev_co_u32 %14, vcc, %22, %6, vcc\n\tv_subbrev_co_u32 %15, vcc, %23, %7, vcc\n\t"
      // borrow: t < mod, keep t
      "v_cndmask_b32 %0, %8, %0, vcc\n\tv_cndmask_b32 %1, %9, %1, vcc\n\tv_cndmask_b32 %2, %10, %2, vcc\n\tv_cndmask_b32 %3, %11, %3, vcc\n\t"
      "v_cndmask_b32 %4, %12, %4, vcc\n\tv_cndmask_b32 %5, %13, %5, vcc\n\tv_cndmask_b32 %6, %14, %6, vcc\n\tv_cndmask_b32 %7, %15, %7, vcc"
      : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "=&v"(d0), "=&v"(d1), "=&v"(d2),
        "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
      // a carry-consuming VOP2 cannot also read a literal (one constant-bus operand on gfx9): limbs 1..7 sit in VGPRs
      : "i"(M::mod(0)), "v"(M::mod(1)), "v"(M::mod(2)), "v"(M::mod(3)), "v"(M::mod(4)), "v"(M::mod(5)), "v"(M::mod(6)), "v"(M::mod(7))
      : "vcc");
}
// a + b mod m: both < m < 2^254, so the sum needs no ninth limb
template <class M>
RB_HD Mont<M> add(const Mont<M>& a, const Mont<M>& b) {
  uint32_t t[8], u[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { t[i] = a.v[i]; u[i] = b.v[i]; }
  limbs_add8(t, u);
  cond_sub_mod<M>(t, 0);
  Mont<M> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return r;
}
template <class M>
RB_HD Mont<M> sub(const Mont<M>& a, const Mont<M>& b) {
  uint32_t t0 = a.v[0], t1 = a.v[1], t2 = a.v[2], t3 = a.v[3], t4 = a.v[4], t5 = a.v[5], t6 = a.v[6], t7 = a.v[7], m;
  // t = a - b; m = -(borrow)
  asm("v_sub_co_u32 %0, vcc, %0, %9\n\tv_subb_co_u32 %1, vcc, %1, %10, vcc\n\tv_subb_co_u32 %2, vcc, %2, %11, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %3, %12, vcc\n\tv_subb_co_u32 %4, vcc, %4, %13, vcc\n\tv_subb_co_u32 %5, vcc, %5, %14, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %6, %15, vcc\n\tv_subb_co_u32 %7, vcc, %7, %16, vcc\n\tv_subb_co_u32_e64 %8, vcc, 0, 0, vcc"
      : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7), "=&v"(m)
      : "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]) : "vcc");
  // add the modulus back when the subtraction wrapped: r = t + (mod & m)
  uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
  asm("v_and_b32 %0, %17, %16\n\tv_and_b32 %1, %18, %16\n\tv_and_b32 %2, %19, %16\n\tv_and_b32 %3, %20, %16\n\t"
      "v_and_b32 %4, %21, %16\n\tv_and_b32 %5, %22, %16\n\tv_and_b32 %6, %23, %16\n\tv_and_b32 %7, %24, %16\n\t"
      "v_add_co_u32 %0, vcc, %0, %8\n\tv_addc_co_u32 %1, vcc, %1, %9, vcc\n\tv_addc_co_u32 %2, vcc, %2, %10, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %11, vcc\n\tv_addc_co_u32 %4, vcc, %4, %12, vcc\n\tv_addc_co_u32 %5, vcc, %5, %13, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %6, %14, vcc\n\tv_addc_co_u32 %7, vcc, %7, %15, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
      : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7), "v"(m), "i"(M::mod(0)), "i"(M::mod(1)), "i"(M::mod(2)),
        "i"(M::mod(3)), "i"(M::mod(4)), "i"(M::mod(5)), "i"(M::mod(6)), "i"(M::mod(7))
      : "vcc");
  Mont<M> r;
  r.v[0] = r0; r.v[1] = r1; r.v[2] = r2; r.v[3] = r3; r.v[4] = r4; r.v[5] = r5; r.v[6] = r6; r.v[7] = r7;
  return r;
}
template <class M>
RB_HD Mont<M> neg(const Mont<M>& a) {
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) nz |= a.v[i];
  const uint32_t m = nz ? 0xFFFFFFFFu : 0u;
  // (mod & m) - a
  uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
  asm("v_and_b32 %0, %17, %16\n\tv_and_b32 %1, %18, %16\n\tv_and_b32 %2, %19, %16\n\tv_and_b32 %3, %20, %16\n\t"
      "v_and_b32 %4, %21, %16\n\tv_and_b32 %5, %22, %16\n\tv_and_b32 %6, %23, %16\n\tv_and_b32 %7, %24, %16\n\t"
      "v_sub_co_u32 %0, vcc, %0, %8\n\tv_subb_co_u32 %1, vcc, %1, %9, vcc\n\tv_subb_co_u32 %2, vcc, %2, %10, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %3, %11, vcc\n\tv_subb_co_u32 %4, vcc, %4, %12, vcc\n\tv_subb_co_u32 %5, vcc, %5, %13, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %6, %14, vcc\n\tv_subb_co_u32 %7, vcc, %7, %15, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]), "v"(m), "i"(M::mod(0)),
        "i"(M::mod(1)), "i"(M::mod(2)), "i"(M::mod(3)), "i"(M::mod(4)), "i"(M::mod(5)), "i"(M::mod(6)), "i"(M::mod(7))
      : "vcc");
  Mont<M> r;
  r.v[0] = r0; r.v[1] = r1; r.v[2] = r2; r.v[3] = r3; r.v[4] = r4; r.v[5] = r5; r.v[6] = r6; r.v[7] = r7;
  return r;
}
#undef RB_CH8
#else
// t <- t - mod if t >= mod, for t < 2*mod < 2^255 (so t itself never carries out of 256 bits)
template <class M>
RB_HD void cond_sub_mod(uint32_t* t, uint32_t hi) {
  uint32_t d[8];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d[i] = subb32(t[i], M::mod(i), borrow);
  const bool keep = (borrow != 0) & (hi == 0);   // borrow: t < mod
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = keep ? t[i] : d[i];
}

// a + b mod m: both < m < 2^254, so the sum needs no ninth limb
template <class M>
RB_HD Mont<M> add(const Mont<M>& a, const Mont<M>& b) {
  uint32_t t[8];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = addc32(a.v[i], b.v[i], c);
  cond_sub_mod<M>(t, 0);
  Mont<M> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return r;
}
template <class M>
RB_HD Mont<M> sub(const Mont<M>& a, const Mont<M>& b) {
  uint32_t t[8];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = subb32(a.v[i], b.v[i], borrow);
  // add the modulus back when the subtraction wrapped
  const uint32_t mask = 0u - borrow;
  uint32_t c = 0;
  Mont<M> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = addc32(t[i], M::mod(i) & mask, c);
  return r;
}
template <class M>
RB_HD Mont<M> neg(const Mont<M>& a) {
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) nz |= a.v[i];
  const uint32_t mask = nz ? 0xFFFFFFFFu : 0u;
  uint32_t borrow = 0;
  Mont<M> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = subb32(M::mod(i) & mask, a.v[i], borrow);
  return r;
}
#endif
// a / 2 mod m: (a + (a odd ? m : 0)) >> 1 -- a + m < 2^255, so nothing leaves the 8 limbs.  Division by two is linear, so it is the same
// in the Montgomery domain; it stands where a multiplication by the constant 1/2 stood (the G2 doubling step: two per Fq2 halving).
template <class M>
RB_HD Mont<M> half(const Mont<M>& a) {
  const uint32_t mask = 0u - (a.v[0] & 1u);
  uint32_t t[8], u[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { t[i] = a.v[i]; u[i] = M::mod(i) & mask; }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RB_SAFE_CARRY)
  limbs_add8(t, u);
#else
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = addc32(t[i], u[i], c);
#endif
  Mont<M> r;
#pragma unroll
  for (int i = 0; i < 7; i++) r.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
  r.v[7] = t[7] >> 1;
  return r;
}
// 2a mod m: shift left by one, then one conditional subtraction
template <class M>
RB_HD Mont<M> dbl(const Mont<M>& a) {
  uint32_t t[8];
#pragma unroll
  for (int i = 7; i > 0; i--) t[i] = (a.v[i] << 1) | (a.v[i - 1] >> 31);
  t[0] = a.v[0] << 1;
  cond_sub_mod<M>(t, 0);
  Mont<M> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return r;
}

// ---------------------------------------------------------------------------------------------
// Montgomery multiplication, CIOS over 32-bit limbs.  136 limb products (64 a*b, 64 m*p, 8 m).
// Inputs < mod, output < mod.  Because mod < 2^254 the running value never exceeds 2*mod, so the
// ninth accumulator word stays zero and only one conditional subtraction is needed at the end.
// Which limb products of the device form below can issue WITHOUT a carry capture (see the comment there): compile-time data
// of the modulus, kept outside the device-only block so that tests/hostsim can hand the very table the kernels use to the
// exact-integer bound check in tests/test_mac_plan.py.
template <class M>
struct ColumnPlan {
  // safe[k] bit i: the reduction product m_i * mod(k - i) of column k issues without a carry capture.
  // last_safe bit k (k < 8): so does m_k * mod(0), the product that closes column k (it needs every other product of the
  // column to be in the safe set).  `top2` tells how much of the budget the top-limb products of the a*b part take.
  uint16_t safe[16];
  uint16_t last_safe;
  constexpr ColumnPlan(bool with_ab, uint32_t top_a, uint32_t top_b) : safe{}, last_safe(0) {
    constexpr uint64_t LIMIT = 0xFFFFFF00ull;     // sum of the bounding limbs, in units of 2^32; 2^8 units of slack cover the
                                                  // incoming < 2^37 and the one plain 32-bit addend of redc2
    for (int k = 0; k < 16; k++) {
      uint64_t used = 0;
      if (with_ab && k >= 7 && k <= 14) used = (k == 14) ? (uint64_t)top_a : (uint64_t)top_a + top_b;   // a7 b(k-7) [+ a(k-7) b7]
      const int lo = k < 8 ? 0 : k - 7, hi = k < 8 ? k - 1 : 7;        // i range of the m_i * mod(k-i) products, m_k * mod(0) excluded
      bool taken[8] = {false, false, false, false, false, false, false, false};
      int n_taken = 0;
      for (;;) {
        int best = -1;
        for (int i = lo; i <= hi; i++)
          if (!taken[i] && (best < 0 || M::mod(k - i) < M::mod(k - best))) best = i;
        if (best < 0 || used + M::mod(k - best) > LIMIT) break;
        used += M::mod(k - best);
        taken[best] = true;
        n_taken++;
        safe[k] = (uint16_t)(safe[k] | (1u << best));
      }
      // the closing product is safe only when nothing in the column banks a carry before it
      const int n_mp = hi - lo + 1;
      const int n_ab_unsafe = !with_ab ? 0 : (k < 7 ? k + 1 : (k == 14 ? 0 : (15 - k) - 2));
      if (k < 8 && n_taken == n_mp && n_ab_unsafe == 0 && used + M::mod(0) <= LIMIT) last_safe = (uint16_t)(last_safe | (1u << k));
    }
  }
};
// plan of a bare reduction (redc2) / of a full product of two operands < mod (top limb <= mod(7)) / of one < mod with an
// arbitrary 256-bit second operand (to_mont_reduce256: only a's top limb is bounded, and only a7 * b(k-7) is taken as safe)
template <class M> RB_HD constexpr bool plan_redc_safe(int k, int i) { constexpr ColumnPlan<M> t(false, 0, 0); return (t.safe[k] >> i) & 1; }
template <class M> RB_HD constexpr bool plan_redc_last_safe(int k) { constexpr ColumnPlan<M> t(false, 0, 0); return (t.last_safe >> k) & 1; }
template <class M> RB_HD constexpr bool plan_mul_safe(int k, int i) { constexpr ColumnPlan<M> t(true, M::mod(7) + 1, M::mod(7) + 1); return (t.safe[k] >> i) & 1; }
template <class M> RB_HD constexpr bool plan_mul_last_safe(int k) { constexpr ColumnPlan<M> t(true, M::mod(7) + 1, M::mod(7) + 1); return (t.last_safe >> k) & 1; }
// a_i * b_(k-i) is a top-limb product of column k
RB_HD constexpr bool ab_top(int k, int i) { return k >= 7 && (i == 7 || k - i == 7); }

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 form: product scanning (FIPS).  A limb product is ONE `v_mad_u64_u32` into a 64-bit column accumulator; a
// product that can carry out of the 64 bits is followed by ONE `v_addc_co_u32` that banks the carry in a third word.
// Which products can NOT carry is known when the code is written, and those issue the mad alone:
//   * a column starts from the previous column's high word + its banked carries: < 2^37;
//   * reduction products m_i * p_j are bounded by p_j * 2^32, so a set of them whose p_j sum to < 2^32 - 2^8 cannot carry
//     (ColumnPlan picks, per column, the largest such set in ascending p_j -- pure compile-time data of the modulus);
//   * a product with a TOP limb of an operand < 2p is < 2^31 * 2^32, so the (at most two) top-limb products of a column
//     cannot carry either (the callers' operands are reduced field elements, or sums of two of them in fp2_mul_lazy_raw).
// Every other product is carry-banked; the first banking addc of a column also DEFINES the carry word (0 + 0 + carry), so no
// register is zeroed per column.  ~20 % fewer instructions per multiplication than mad + addc for every product.
// one chain: mad alone / mad + addc into an existing carry word / mad + addc defining the carry word
#define RB_MAD(acc_, x_, y_) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc_) : "v"(x_), "v"(y_) : "vcc")
#define RB_MAD_S(acc_, x_, y_) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc_) : "v"(x_), "s"(y_) : "vcc")
#define RB_MAC(acc_, ovf_, x_, y_) \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc_), "+v"(ovf_) : "v"(x_), "v"(y_) : "vcc")
#define RB_MAC_S(acc_, ovf_, x_, y_) \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc_), "+v"(ovf_) : "v"(x_), "s"(y_) : "vcc")
#define RB_MACF(acc_, ovf_, x_, y_) \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc" : "+v"(acc_), "=&v"(ovf_) : "v"(x_), "v"(y_) : "vcc")
#define RB_MACF_S(acc_, ovf_, x_, y_) \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc" : "+v"(acc_), "=&v"(ovf_) : "v"(x_), "s"(y_) : "vcc")
// carry-banked product, defining the carry word when it is the column's first
#define RB_BANK(have_, acc_, ovf_, x_, y_) do { if (have_) { RB_MAC(acc_, ovf_, x_, y_); } else { RB_MACF(acc_, ovf_, x_, y_); have_ = true; } } while (0)
#define RB_BANK_S(have_, acc_, ovf_, x_, y_) do { if (have_) { RB_MAC_S(acc_, ovf_, x_, y_); } else { RB_MACF_S(acc_, ovf_, x_, y_); have_ = true; } } while (0)
// next column: drop the finished word, the banked carries become the high word
#define RB_SHIFT(have_, acc_, ovf_) do { acc_ = (acc_ >> 32) | ((have_) ? ((uint64_t)(ovf_) << 32) : 0ull); } while (0)

// B_REDUCED: b < mod as well (false: b is any 256-bit integer, to_mont_reduce256)
template <class M, bool B_REDUCED = true, class A, class B>
RB_HD void mont_mul_raw(uint32_t* r, const A& a, const B& b) {
  uint32_t m[8];
  uint32_t t[8];
  uint64_t acc = 0;
  uint32_t ovf;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    bool have = false;
    const int lo = k < 8 ? 0 : k - 7, hi = k < 8 ? k : 7;
    // products that cannot carry first
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      const bool top = B_REDUCED ? ab_top(k, i) : (i == 7);
      if (top) { const uint32_t x = a[i], y = b[k - i]; RB_MAD(acc, x, y); }
    }
#pragma unroll
    for (int i = lo; i <= hi; i++)
      if (!(k < 8 && i == k) && plan_mul_safe<M>(k, i)) { const uint32_t x = m[i], y = M::mod(k - i); RB_MAD_S(acc, x, y); }
    // the carry-banked rest
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      const bool top = B_REDUCED ? ab_top(k, i) : (i == 7);
      if (!top) { const uint32_t x = a[i], y = b[k - i]; RB_BANK(have, acc, ovf, x, y); }
    }
#pragma unroll
    for (int i = lo; i <= hi; i++)
      if (!(k < 8 && i == k) && !plan_mul_safe<M>(k, i)) { const uint32_t x = m[i], y = M::mod(k - i); RB_BANK_S(have, acc, ovf, x, y); }
    if (k < 8) {
      m[k] = (uint32_t)acc * M::INV;
      const uint32_t x = m[k], y = M::mod(0);
      if (!have && plan_mul_last_safe<M>(k)) RB_MAD_S(acc, x, y);
      else RB_BANK_S(have, acc, ovf, x, y);
    } else {
      t[k - 8] = (uint32_t)acc;
    }
    RB_SHIFT(have, acc, ovf);
  }
  // value < 2*mod < 2^255: nothing left in acc
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = t[i];
  cond_sub_mod<M>(r, 0);
}
// Two / three INDEPENDENT Montgomery multiplications in lockstep.  A single multiplication is one serial
// dependency chain (mad -> addc -> mad ...); a kernel that runs one wave per SIMD (the Miller loop and the
// final exponentiation keep ~500 registers of state) cannot hide that latency with other waves, so the Fp2
// routines interleave their independent Fp products instead: each step issues N mads then N addcs on N
// accumulators, with the carries in N different SGPR pairs (tools/ubench_mac.hip: 11.4 cycles per product at N = 3
// against 23 at N = 1, one wave per SIMD).
#define RB_MAD2(A0, X0, Y0, A1, X1, Y1)                                                               \
  asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1"                          \
      : "+v"(A0), "+v"(A1), "=&s"(c0_) : "v"(X0), "v"(Y0), "v"(X1), "v"(Y1))
#define RB_MAD2_S(A0, X0, A1, X1, Y)                                                                  \
  asm("v_mad_u64_u32 %0, %2, %3, %5, %0\n\tv_mad_u64_u32 %1, %2, %4, %5, %1"                          \
      : "+v"(A0), "+v"(A1), "=&s"(c0_) : "v"(X0), "v"(X1), "s"(Y))
#define RB_MAC2(A0, O0, X0, Y0, A1, O1, X1, Y1)                                                       \
  asm("v_mad_u64_u32 %0, %4, %6, %7, %0\n\tv_mad_u64_u32 %2, %5, %8, %9, %2\n\t"                      \
      "" RB_CPAD "v_addc_co_u32_e64 %1, %4, 0, %1, %4\n\tv_addc_co_u32_e64 %3, %5, 0, %3, %5"                    \
      : "+v"(A0), "+v"(O0), "+v"(A1), "+v"(O1), "=&s"(c0_), "=&s"(c1_)                                \
      : "v"(X0), "v"(Y0), "v"(X1), "v"(Y1))
#define RB_MAC2_S(A0, O0, X0, A1, O1, X1, Y)                                                          \
  asm("v_mad_u64_u32 %0, %4, %6, %8, %0\n\tv_mad_u64_u32 %2, %5, %7, %8, %2\n\t"                      \
      "" RB_CPAD "v_addc_co_u32_e64 %1, %4, 0, %1, %4\n\tv_addc_co_u32_e64 %3, %5, 0, %3, %5"                    \
      : "+v"(A0), "+v"(O0), "+v"(A1), "+v"(O1), "=&s"(c0_), "=&s"(c1_)                                \
      : "v"(X0), "v"(X1), "s"(Y))
#define RB_MAC2F(A0, O0, X0, Y0, A1, O1, X1, Y1)                                                      \
  asm("v_mad_u64_u32 %0, %4, %6, %7, %0\n\tv_mad_u64_u32 %2, %5, %8, %9, %2\n\t"                      \
      "" RB_CPAD "v_addc_co_u32_e64 %1, %4, 0, 0, %4\n\tv_addc_co_u32_e64 %3, %5, 0, 0, %5"                      \
      : "+v"(A0), "=&v"(O0), "+v"(A1), "=&v"(O1), "=&s"(c0_), "=&s"(c1_)                              \
      : "v"(X0), "v"(Y0), "v"(X1), "v"(Y1))
#define RB_MAC2F_S(A0, O0, X0, A1, O1, X1, Y)                                                         \
  asm("v_mad_u64_u32 %0, %4, %6, %8, %0\n\tv_mad_u64_u32 %2, %5, %7, %8, %2\n\t"                      \
      "" RB_CPAD "v_addc_co_u32_e64 %1, %4, 0, 0, %4\n\tv_addc_co_u32_e64 %3, %5, 0, 0, %5"                      \
      : "+v"(A0), "=&v"(O0), "+v"(A1), "=&v"(O1), "=&s"(c0_), "=&s"(c1_)                              \
      : "v"(X0), "v"(X1), "s"(Y))
#define RB_MAD3(A0, X0, Y0, A1, X1, Y1, A2, X2, Y2)                                                   \
  asm("v_mad_u64_u32 %0, %3, %4, %5, %0\n\tv_mad_u64_u32 %1, %3, %6, %7, %1\n\tv_mad_u64_u32 %2, %3, %8, %9, %2" \
      : "+v"(A0), "+v"(A1), "+v"(A2), "=&s"(c0_) : "v"(X0), "v"(Y0), "v"(X1), "v"(Y1), "v"(X2), "v"(Y2))
#define RB_MAD3_S(A0, X0, A1, X1, A2, X2, Y)                                                          \
  asm("v_mad_u64_u32 %0, %3, %4, %7, %0\n\tv_mad_u64_u32 %1, %3, %5, %7, %1\n\tv_mad_u64_u32 %2, %3, %6, %7, %2" \
      : "+v"(A0), "+v"(A1), "+v"(A2), "=&s"(c0_) : "v"(X0), "v"(X1), "v"(X2), "s"(Y))
#define RB_MAC3(A0, O0, X0, Y0, A1, O1, X1, Y1, A2, O2, X2, Y2)                                       \
  asm("v_mad_u64_u32 %0, %6, %9, %10, %0\n\tv_mad_u64_u32 %2, %7, %11, %12, %2\n\t"                   \
      "v_mad_u64_u32 %4, %8, %13, %14, %4\n\t"                                                        \
      "" RB_CPAD "v_addc_co_u32_e64 %1, %6, 0, %1, %6\n\tv_addc_co_u32_e64 %3, %7, 0, %3, %7\n\t"                \
      "v_addc_co_u32_e64 %5, %8, 0, %5, %8"                                                           \
      : "+v"(A0), "+v"(O0), "+v"(A1), "+v"(O1), "+v"(A2), "+v"(O2), "=&s"(c0_), "=&s"(c1_), "=&s"(c2_) \
      : "v"(X0), "v"(Y0), "v"(X1), "v"(Y1), "v"(X2), "v"(Y2))
#define RB_MAC3_S(A0, O0, X0, A1, O1, X1, A2, O2, X2, Y)                                              \
  asm("v_mad_u64_u32 %0, %6, %9, %12, %0\n\tv_mad_u64_u32 %2, %7, %10, %12, %2\n\t"                   \
      "v_mad_u64_u32 %4, %8, %11, %12, %4\n\t"                                                        \
      "" RB_CPAD "v_addc_co_u32_e64 %1, %6, 0, %1, %6\n\tv_addc_co_u32_e64 %3, %7, 0, %3, %7\n\t"                \
      "v_addc_co_u32_e64 %5, %8, 0, %5, %8"                                                           \
      : "+v"(A0), "+v"(O0), "+v"(A1), "+v"(O1), "+v"(A2), "+v"(O2), "=&s"(c0_), "=&s"(c1_), "=&s"(c2_) \
      : "v"(X0), "v"(X1), "v"(X2), "s"(Y))
#define RB_MAC3F(A0, O0, X0, Y0, A1, O1, X1, Y1, A2, O2, X2, Y2)                                      \
  asm("v_mad_u64_u32 %0, %6, %9, %10, %0\n\tv_mad_u64_u32 %2, %7, %11, %12, %2\n\t"                   \
      "v_mad_u64_u32 %4, %8, %13, %14, %4\n\t"                                                        \
      "" RB_CPAD "v_addc_co_u32_e64 %1, %6, 0, 0, %6\n\tv_addc_co_u32_e64 %3, %7, 0, 0, %7\n\t"                  \
      "v_addc_co_u32_e64 %5, %8, 0, 0, %8"                                                            \
      : "+v"(A0), "=&v"(O0), "+v"(A1), "=&v"(O1), "+v"(A2), "=&v"(O2), "=&s"(c0_), "=&s"(c1_), "=&s"(c2_) \
      : "v"(X0), "v"(Y0), "v"(X1), "v"(Y1), "v"(X2), "v"(Y2))
#define RB_MAC3F_S(A0, O0, X0, A1, O1, X1, A2, O2, X2, Y)                                             \
  asm("v_mad_u64_u32 %0, %6, %9, %12, %0\n\tv_mad_u64_u32 %2, %7, %10, %12, %2\n\t"                   \
      "v_mad_u64_u32 %4, %8, %11, %12, %4\n\t"                                                        \
      "" RB_CPAD "v_addc_co_u32_e64 %1, %6, 0, 0, %6\n\tv_addc_co_u32_e64 %3, %7, 0, 0, %7\n\t"                  \
      "v_addc_co_u32_e64 %5, %8, 0, 0, %8"                                                            \
      : "+v"(A0), "=&v"(O0), "+v"(A1), "=&v"(O1), "+v"(A2), "=&v"(O2), "=&s"(c0_), "=&s"(c1_), "=&s"(c2_) \
      : "v"(X0), "v"(X1), "v"(X2), "s"(Y))
#define RB_BANK2(have_, A0, O0, X0, Y0, A1, O1, X1, Y1) \
  do { if (have_) { RB_MAC2(A0, O0, X0, Y0, A1, O1, X1, Y1); } else { RB_MAC2F(A0, O0, X0, Y0, A1, O1, X1, Y1); have_ = true; } } while (0)
#define RB_BANK2_S(have_, A0, O0, X0, A1, O1, X1, Y) \
  do { if (have_) { RB_MAC2_S(A0, O0, X0, A1, O1, X1, Y); } else { RB_MAC2F_S(A0, O0, X0, A1, O1, X1, Y); have_ = true; } } while (0)
#define RB_BANK3(have_, A0, O0, X0, Y0, A1, O1, X1, Y1, A2, O2, X2, Y2) \
  do { if (have_) { RB_MAC3(A0, O0, X0, Y0, A1, O1, X1, Y1, A2, O2, X2, Y2); } else { RB_MAC3F(A0, O0, X0, Y0, A1, O1, X1, Y1, A2, O2, X2, Y2); have_ = true; } } while (0)
#define RB_BANK3_S(have_, A0, O0, X0, A1, O1, X1, A2, O2, X2, Y) \
  do { if (have_) { RB_MAC3_S(A0, O0, X0, A1, O1, X1, A2, O2, X2, Y); } else { RB_MAC3F_S(A0, O0, X0, A1, O1, X1, A2, O2, X2, Y); have_ = true; } } while (0)

// operands < mod (every caller passes reduced field elements); the Fp instance is the generated mont_mul2_fp below
template <class M, class A>
RB_HD void mont_mul2_generic(uint32_t* r0, uint32_t* r1, const A& a0, const A& b0, const A& a1, const A& b1) {
  uint32_t m0[8], m1[8];
  uint64_t acc0 = 0, acc1 = 0, c0_, c1_;
  uint32_t ovf0, ovf1;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    bool have = false;
    const int lo = k < 8 ? 0 : k - 7, hi = k < 8 ? k : 7;
#pragma unroll
    for (int i = lo; i <= hi; i++)
      if (ab_top(k, i)) { const uint32_t x0 = a0[i], y0 = b0[k - i], x1 = a1[i], y1 = b1[k - i]; RB_MAD2(acc0, x0, y0, acc1, x1, y1); }
#pragma unroll
    for (int i = lo; i <= hi; i++)
      if (!(k < 8 && i == k) && plan_mul_safe<M>(k, i)) { const uint32_t x0 = m0[i], x1 = m1[i], y = M::mod(k - i); RB_MAD2_S(acc0, x0, acc1, x1, y); }
#pragma unroll
    for (int i = lo; i <= hi; i++)
      if (!ab_top(k, i)) { const uint32_t x0 = a0[i], y0 = b0[k - i], x1 = a1[i], y1 = b1[k - i]; RB_BANK2(have, acc0, ovf0, x0, y0, acc1, ovf1, x1, y1); }
#pragma unroll
    for (int i = lo; i <= hi; i++)
      if (!(k < 8 && i == k) && !plan_mul_safe<M>(k, i)) { const uint32_t x0 = m0[i], x1 = m1[i], y = M::mod(k - i); RB_BANK2_S(have, acc0, ovf0, x0, acc1, ovf1, x1, y); }
    if (k < 8) {
      m0[k] = (uint32_t)acc0 * M::INV;
      m1[k] = (uint32_t)acc1 * M::INV;
      const uint32_t x0 = m0[k], x1 = m1[k], y = M::mod(0);
      if (!have && plan_mul_last_safe<M>(k)) RB_MAD2_S(acc0, x0, acc1, x1, y);
      else RB_BANK2_S(have, acc0, ovf0, x0, acc1, ovf1, x1, y);
    } else {
      r0[k - 8] = (uint32_t)acc0;
      r1[k - 8] = (uint32_t)acc1;
    }
    RB_SHIFT(have, acc0, ovf0);
    RB_SHIFT(have, acc1, ovf1);
  }
  cond_sub_mod<M>(r0, 0);
  cond_sub_mod<M>(r1, 0);
}
template <class M, class A>
RB_HD void mont_mul3_raw(uint32_t* r0, uint32_t* r1, uint32_t* r2, const A& a0, const A& b0, const A& a1, const A& b1, const A& a2,
                         const A& b2) {
  uint32_t m0[8], m1[8], m2[8];
  uint64_t acc0 = 0, acc1 = 0, acc2 = 0, c0_, c1_, c2_;
  uint32_t ovf0, ovf1, ovf2;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    bool have = false;
    const int lo = k < 8 ? 0 : k - 7, hi = k < 8 ? k : 7;
#pragma unroll
    for (int i = lo; i <= hi; i++)
      if (ab_top(k, i)) {
        const uint32_t x0 = a0[i], y0 = b0[k - i], x1 = a1[i], y1 = b1[k - i], x2 = a2[i], y2 = b2[k - i];
        RB_MAD3(acc0, x0, y0, acc1, x1, y1, acc2, x2, y2);
      }
#pragma unroll
    for (int i = lo; i <= hi; i++)
      if (!(k < 8 && i == k) && plan_mul_safe<M>(k, i)) {
        const uint32_t x0 = m0[i], x1 = m1[i], x2 = m2[i], y = M::mod(k - i);
        RB_MAD3_S(acc0, x0, acc1, x1, acc2, x2, y);
      }
#pragma unroll
    for (int i = lo; i <= hi; i++)
      if (!ab_top(k, i)) {
        const uint32_t x0 = a0[i], y0 = b0[k - i], x1 = a1[i], y1 = b1[k - i], x2 = a2[i], y2 = b2[k - i];
        RB_BANK3(have, acc0, ovf0, x0, y0, acc1, ovf1, x1, y1, acc2, ovf2, x2, y2);
      }
#pragma unroll
    for (int i = lo; i <= hi; i++)
      if (!(k < 8 && i == k) && !plan_mul_safe<M>(k, i)) {
        const uint32_t x0 = m0[i], x1 = m1[i], x2 = m2[i], y = M::mod(k - i);
        RB_BANK3_S(have, acc0, ovf0, x0, acc1, ovf1, x1, acc2, ovf2, x2, y);
      }
    if (k < 8) {
      m0[k] = (uint32_t)acc0 * M::INV;
      m1[k] = (uint32_t)acc1 * M::INV;
      m2[k] = (uint32_t)acc2 * M::INV;
      const uint32_t x0 = m0[k], x1 = m1[k], x2 = m2[k], y = M::mod(0);
      if (!have && plan_mul_last_safe<M>(k)) RB_MAD3_S(acc0, x0, acc1, x1, acc2, x2, y);
      else RB_BANK3_S(have, acc0, ovf0, x0, acc1, ovf1, x1, acc2, ovf2, x2, y);
    } else {
      r0[k - 8] = (uint32_t)acc0;
      r1[k - 8] = (uint32_t)acc1;
      r2[k - 8] = (uint32_t)acc2;
    }
    RB_SHIFT(have, acc0, ovf0);
    RB_SHIFT(have, acc1, ovf1);
    RB_SHIFT(have, acc2, ovf2);
  }
  cond_sub_mod<M>(r0, 0);
  cond_sub_mod<M>(r1, 0);
  cond_sub_mod<M>(r2, 0);
}
// ---- Fp2 multiplication with lazy reduction (Fp only): three plain 256x256 -> 512-bit products in lockstep,
// the Karatsuba combination on the 512-bit values, then TWO Montgomery reductions in lockstep:
//   W0 = a0 b0 - a1 b1 + p^2   in (0, 2p^2)        W1 = (a0+a1)(b0+b1) - a0 b0 - a1 b1 = a0 b1 + a1 b0  in [0, 2p^2)
//   c0 = W0 / 2^256 mod p,  c1 = W1 / 2^256 mod p   (REDC output < 1.38 p: one conditional subtraction each)
// 192 + 128 MACs instead of 3 x 128 + ... = 384 for three full Montgomery products.
// wide_mul3 (operands a0, b0, a1, b1 < p and a2, b2 < 2p: top limbs < 2^31, so the two top-limb products of a column sum to
// < 2^63.6), redc2_fp and mont_mul2_fp come from tools/gen_fp_asm.py: the same plan as the loops above, laid out as straight-line
// code with as many products per asm statement as its 30 operands allow (hipcc pads every statement boundary with a wait state).
#include "fp_gfx950_gen.h"      // inside namespace rabe::bn254, device-only
template <class M> struct IsFp { static constexpr bool value = false; };
template <> struct IsFp<FpParams> { static constexpr bool value = true; };
template <class M, class A>
RB_HD void mont_mul2_raw(uint32_t* r0, uint32_t* r1, const A& a0, const A& b0, const A& a1, const A& b1) {
#ifndef RB_NO_GEN_MUL2
  if constexpr (IsFp<M>::value) mont_mul2_fp(r0, r1, a0, b0, a1, b1);
  else
#endif
    mont_mul2_generic<M>(r0, r1, a0, b0, a1, b1);
}
// (a0 + a1 u)(b0 + b1 u) over Fp with lazy reduction; all operands / results fully reduced Montgomery values
template <class A>
RB_HD void fp2_mul_lazy_raw(uint32_t* c0, uint32_t* c1, const A& a0, const A& a1, const A& b0, const A& b1) {
  constexpr uint32_t p2[16] = RB_FP_P2;
  uint32_t sa[8], sb[8];
  { uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sa[i] = addc32(a0[i], a1[i], c); }      // < 2p < 2^255: no carry out
  { uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sb[i] = addc32(b0[i], b1[i], c); }
  uint32_t T0[16], T1[16], T2[16];
  wide_mul3(T0, T1, T2, a0, b0, a1, b1, sa, sb);
  uint32_t W0[16], W1[16];
  { uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) W0[i] = subb32(T0[i], T1[i], br); }
  { uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) W0[i] = addc32(W0[i], p2[i], c); }
  { uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) W1[i] = subb32(T2[i], T0[i], br); }
  { uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) W1[i] = subb32(W1[i], T1[i], br); }
  redc2_fp(c0, c1, W0, W1);
}
#undef RB_MAD
#undef RB_MAD_S
#undef RB_MACF
#undef RB_MACF_S
#undef RB_BANK
#undef RB_BANK_S
#undef RB_SHIFT
#undef RB_MAD2
#undef RB_MAD2_S
#undef RB_MAC2F
#undef RB_MAC2F_S
#undef RB_MAD3
#undef RB_MAD3_S
#undef RB_MAC3F
#undef RB_MAC3F_S
#undef RB_BANK2
#undef RB_BANK2_S
#undef RB_BANK3
#undef RB_BANK3_S
#undef RB_MAC2
#undef RB_MAC2_S
#undef RB_MAC3
#undef RB_MAC3_S
#undef RB_MAC
#undef RB_MAC_S
#else
// host: the interleaved forms are just independent multiplications
template <class M, bool B_REDUCED = true, class A, class B> RB_HD void mont_mul_raw(uint32_t* r, const A& a, const B& b);
template <class M, class A>
RB_HD void mont_mul2_raw(uint32_t* r0, uint32_t* r1, const A& a0, const A& b0, const A& a1, const A& b1) {
  mont_mul_raw<M>(r0, a0, b0);
  mont_mul_raw<M>(r1, a1, b1);
}
template <class M, class A>
RB_HD void mont_mul3_raw(uint32_t* r0, uint32_t* r1, uint32_t* r2, const A& a0, const A& b0, const A& a1, const A& b1, const A& a2,
                         const A& b2) {
  mont_mul_raw<M>(r0, a0, b0);
  mont_mul_raw<M>(r1, a1, b1);
  mont_mul_raw<M>(r2, a2, b2);
}
// portable form (host build of the same headers: tests/hostsim)
template <class M, bool B_REDUCED, class A, class B>
RB_HD void mont_mul_raw(uint32_t* r, const A& a, const B& b) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    const uint32_t bi = b[i];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint64_t x = (uint64_t)a[j] * bi + t[j] + c;
      t[j] = (uint32_t)x;
      c = x >> 32;
    }
    t[8] = (uint32_t)c;  // previous t[8] is always 0 here (value < 2*mod < 2^255 after each reduction step)
    const uint32_t m = t[0] * M::INV;
    uint64_t x = (uint64_t)m * M::mod(0) + t[0];
    c = x >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      x = (uint64_t)m * M::mod(j) + t[j] + c;
      t[j - 1] = (uint32_t)x;
      c = x >> 32;
    }
    x = (uint64_t)t[8] + c;
    t[7] = (uint32_t)x;
    // (x >> 32) is 0: see above
  }
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = t[i];
  cond_sub_mod<M>(r, 0);
}
#endif

// Host-only instrumentation for the roofline's algorithmic work count (tests/hostsim builds with
// -DRB_COUNT_MULS; never defined for device code): every Montgomery multiplication bumps a counter.
#if defined(RB_COUNT_MULS) && !defined(__HIP_DEVICE_COMPILE__)
#define RB_COUNT_ONE_MUL() (++::rb_mul_counter)
#else
#define RB_COUNT_ONE_MUL() ((void)0)
#endif

// inlined form (used inside the Fp2-level functions, which are themselves real functions)
template <class M>
RB_HD Mont<M> mul_inl(const Mont<M>& a, const Mont<M>& b) {
  RB_COUNT_ONE_MUL();
  uint32_t t[8];
  mont_mul_raw<M>(t, a.v, b.v);
  Mont<M> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return r;
}
// N independent products at once (see mont_mul2_raw / mont_mul3_raw)
template <class M>
RB_HD void mul2_inl(Mont<M>& r0, Mont<M>& r1, const Mont<M>& a0, const Mont<M>& b0, const Mont<M>& a1, const Mont<M>& b1) {
  RB_COUNT_ONE_MUL(); RB_COUNT_ONE_MUL();
  uint32_t t0[8], t1[8];
  mont_mul2_raw<M>(t0, t1, a0.v, b0.v, a1.v, b1.v);
#pragma unroll
  for (int i = 0; i < 8; i++) { r0.v[i] = t0[i]; r1.v[i] = t1[i]; }
}
template <class M>
RB_HD void mul3_inl(Mont<M>& r0, Mont<M>& r1, Mont<M>& r2, const Mont<M>& a0, const Mont<M>& b0, const Mont<M>& a1, const Mont<M>& b1,
                    const Mont<M>& a2, const Mont<M>& b2) {
  RB_COUNT_ONE_MUL(); RB_COUNT_ONE_MUL(); RB_COUNT_ONE_MUL();
  uint32_t t0[8], t1[8], t2[8];
  mont_mul3_raw<M>(t0, t1, t2, a0.v, b0.v, a1.v, b1.v, a2.v, b2.v);
#pragma unroll
  for (int i = 0; i < 8; i++) { r0.v[i] = t0[i]; r1.v[i] = t1[i]; r2.v[i] = t2[i]; }
}
// out-of-line form: operands and result travel in VGPRs
template <class M>
RB_FN Mont<M> mul(Mont<M> a, Mont<M> b) { return mul_inl(a, b); }
template <class M>
RB_FN Mont<M> sqr(Mont<M> a) { return mul_inl(a, a); }

// canonical integer (little-endian limbs, < mod) <-> Montgomery
template <class M>
RB_HD Mont<M> to_mont(const uint32_t x[8]) {
  uint32_t r2[8];
#pragma unroll
  for (int i = 0; i < 8; i++) r2[i] = M::r2(i);
  Mont<M> xm;
#pragma unroll
  for (int i = 0; i < 8; i++) { xm.v[i] = x[i]; }
  Mont<M> rm;
#pragma unroll
  for (int i = 0; i < 8; i++) { rm.v[i] = r2[i]; }
  // x as the second operand, declared unbounded: exact for ANY 256-bit x (a caller's value is canonical, but a wrong one must
  // not meet an arithmetic precondition)
  RB_COUNT_ONE_MUL();
  uint32_t t[8];
  mont_mul_raw<M, false>(t, rm.v, xm.v);
  Mont<M> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return r;
}
template <class M>
RB_HD void from_mont(uint32_t out[8], const Mont<M>& a) {
  Mont<M> o;
#pragma unroll
  for (int i = 0; i < 8; i++) o.v[i] = (i == 0) ? 1u : 0u;
  Mont<M> r = mul(a, o);
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = r.v[i];
}
// inlined form of from_mont for kernels that must not call out of their hot loop (a call parks every live value in callee-saved
// registers or on the stack: the AC17 row kernels)
template <class M>
RB_HD void from_mont_inl(uint32_t out[8], const Mont<M>& a) {
  Mont<M> o;
#pragma unroll
  for (int i = 0; i < 8; i++) o.v[i] = (i == 0) ? 1u : 0u;
  Mont<M> r = mul_inl(a, o);
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = r.v[i];
}
// Reduce an arbitrary 256-bit integer into Montgomery form: mont_mul(x, R^2) = x*R mod m for any
// x < 2^256 (the CIOS bound only needs one operand < mod).  This is `Fr::from_slice` on a SHA3
// digest (src/utils/hash/mod.rs:16) -- SURVEY.md 8c assumption (i).
template <class M>
RB_HD Mont<M> to_mont_reduce256(const uint32_t x[8]) {
  uint32_t r2[8];
#pragma unroll
  for (int i = 0; i < 8; i++) r2[i] = M::r2(i);
  Mont<M> xm;
#pragma unroll
  for (int i = 0; i < 8; i++) { xm.v[i] = x[i]; }
  Mont<M> rm;
#pragma unroll
  for (int i = 0; i < 8; i++) { rm.v[i] = r2[i]; }
  // x plays `b`: any 256-bit value there keeps the running sum below 2*mod; the a-operand (R2) is < mod
  RB_COUNT_ONE_MUL();
  uint32_t t[8];
  mont_mul_raw<M, false>(t, rm.v, xm.v);
  Mont<M> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return r;
}

// a^(mod-2): Fermat inversion, left-to-right binary over the constant exponent (the instruction
// stream is identical for every lane: no divergence, no table, nothing runtime-indexed).  inv(0) = 0.
template <class M>
RB_HD uint32_t inv_exp_word(int k) {
  switch (k) {
    case 0: return M::exp_inv(0);
    case 1: return M::exp_inv(1);
    case 2: return M::exp_inv(2);
    case 3: return M::exp_inv(3);
    case 4: return M::exp_inv(4);
    case 5: return M::exp_inv(5);
    case 6: return M::exp_inv(6);
    default: return M::exp_inv(7);
  }
}
template <class M>
RB_HD_NOINLINE Mont<M> inv_fermat(Mont<M> a) {
  Mont<M> acc = a;   // top bit (bit 253) of mod-2 is set
  for (int i = 252; i >= 0; i--) {
    acc = sqr(acc);
    const uint32_t bit = (inv_exp_word<M>(i >> 5) >> (i & 31)) & 1u;
    if (bit) acc = mul(acc, a);
  }
  return acc;
}
// The inversion the kernels use: Kaliski's almost-inverse (binary extended Euclid on the 256-bit integers: shifts,
// additions and subtractions only, 254..508 steps of ~35 instructions) followed by the power-of-two correction as two
// Montgomery multiplications -- ~14 k instructions instead of the ~125 k of the exponentiation above.  The inverse of a
// field element is unique, so the result is the same canonical value; inv(0) = 0.  The step sequence depends on the
// data: where a block's ONE inversion is computed by a wave on a wave-uniform value (block_batch_inverse_*) there is no
// divergence; per-lane calls (final exponentiation) diverge over at most four short branches.
RB_HD void inv_shr1(uint32_t x[8]) {
#pragma unroll
  for (int i = 0; i < 7; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
  x[7] >>= 1;
}
RB_HD void inv_shl1(uint32_t x[8]) {
#pragma unroll
  for (int i = 7; i > 0; i--) x[i] = (x[i] << 1) | (x[i - 1] >> 31);
  x[0] <<= 1;
}
template <class M>
RB_HD_NOINLINE Mont<M> inv(Mont<M> a) {
  uint32_t u[8], v[8], r[8], s[8];
  uint32_t any = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { u[i] = M::mod(i); v[i] = a.v[i]; r[i] = 0; s[i] = (i == 0) ? 1u : 0u; any |= a.v[i]; }
  if (!any) return a;
  // u = mod, v = x, r = 0, s = 1: on exit (v = 0, u = 1)  mod - r = x^-1 2^k  with r < 2 mod, s <= 2 mod (255 bits)
  int k = 0;
  for (;;) {
    any = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) any |= v[i];
    if (!any) break;
    k++;
    if (!(u[0] & 1u)) { inv_shr1(u); inv_shl1(s); continue; }
    if (!(v[0] & 1u)) { inv_shr1(v); inv_shl1(r); continue; }
    uint32_t d[8], borrow = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { d[i] = subb32(u[i], v[i], borrow); nz |= d[i]; }
    if (!borrow && nz) {                       // u > v: u = (u - v) / 2, r += s, s *= 2
#pragma unroll
      for (int i = 0; i < 8; i++) u[i] = d[i];
      inv_shr1(u);
      uint32_t c = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) r[i] = addc32(r[i], s[i], c);
      inv_shl1(s);
    } else {                                   // v >= u: v = (v - u) / 2, s += r, r *= 2
      borrow = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = subb32(v[i], u[i], borrow);
      inv_shr1(v);
      uint32_t c = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) s[i] = addc32(s[i], r[i], c);
      inv_shl1(r);
    }
  }
  // y = mod - (r mod mod)
  uint32_t t[8], borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = subb32(r[i], M::mod(i), borrow);
  if (!borrow) {
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = t[i];
  }
  Mont<M> y, r2;
  borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { y.v[i] = subb32(M::mod(i), r[i], borrow); r2.v[i] = M::r2(i); }
  // x = a R, y = x^-1 2^k; wanted a^-1 R = y 2^(512 - k): mul(mul(y, R^2), 2^c) = y 2^c for c <= 253 (2^253 < mod)
  int e = 512 - k;                             // 4 .. 258
  while (e > 0) {
    const int c = e < 253 ? e : 253;
    Mont<M> oh;
#pragma unroll
    for (int i = 0; i < 8; i++) oh.v[i] = ((c >> 5) == i) ? (1u << (c & 31)) : 0u;
    y = mul(mul(y, r2), oh);
    e -= c;
  }
  return y;
}

}}  // namespace rabe::bn254
