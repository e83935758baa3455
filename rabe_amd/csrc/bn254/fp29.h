// BN254 base field in REDUCED RADIX for gfx950: 9 signed limbs of 29 bits, Montgomery form with R = 2^261  (namespace rr).
//
// Why a second representation beside fp.h's 8 x 32-bit limbs: a 32 x 32-bit limb product fills its 64-bit column
// accumulator, so every `v_mad_u64_u32` of fp.h is followed by carry bookkeeping (the lazy Fq2 multiplication there is 352
// mads + 241 carry adds + 99 moves + ~120 instructions of Karatsuba / reduction glue: 883 instructions, 166 registers).
// With 29-bit limbs a product is 58 bits: a signed 64-bit column takes the 18 products of a lazy Fq2 multiplication plus the
// 9 of its reduction WITHOUT any carry instruction, subtraction is limb-wise (signed limbs: no "+ k p" offset), additions
// and subtractions need no reduction, and carries are propagated once per column by a 64-bit shift: 693 instructions (486
// of them multiply-adds), 100 registers, and at one wave per SIMD -- where the Fq12 kernels run -- 3.2 k instead of 4.6 k
// cycles per Fq2 multiplication (tools/ubench_rr29.hip).  Plain C++: no carry flags, no hardware interlock to rely on.
//
// An element is  x = sum l[k] 2^(29 k)  (a signed integer) standing for  x / 2^261 mod p.  Its type carries two bounds,
// checked at compile time wherever values are combined:
//   FB<L, V>:   |l[k]| <= L * U,  U = 2^28 + 2^12        (limbs: what a 64-bit column can take)
//               |x|    <= V * 1.5 p                     (value: what keeps the results of multiplications below 1.5 p)
//   F = FB<1, 1> is what every multiplication returns and what is stored; a sum of two of them is FB<2, 2>, and so on.
// A column of  a b + c d  holds 9 (La Lb + Lc Ld) U^2 + 9 * 2^58 (reduction) + 2^28 + carry < 2^63  for  La Lb + Lc Ld <= 10;
// its result is below  p + (Va Vb + Vc Vd) 2.25 p^2 / 2^261 <= 1.5 p  for  Va Vb + Vc Vd <= 36.
// Deeper sums are brought back by norm(): one parallel carry pass (limbs), plus -- when the value bound needs it -- the
// subtraction of round(x / p) p, estimated from the top limb.
// RB29_CHECK (host builds of the tests) asserts every bound at run time as well, on the values that actually occur
// (tests/test_hostsim_rr.py); tests/test_fp29_bounds.py restates the column, value and quotient-estimate bounds with exact integers.
//
// Replaces, for the kernels that use it, the same `rabe_bn::Fq` arithmetic as fp.h (src/schemes/ac17/mod.rs:42).
#pragma once
#include "tower.h"
#include "constants29.h"
#include <type_traits>
#if defined(RB29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
#include <stdio.h>
#include <stdlib.h>
#endif

#if defined(RB_COUNT_MULS) && !defined(__HIP_DEVICE_COMPILE__)
extern "C" unsigned long long rb_rr_mad_counter;
#endif

namespace rabe { namespace bn254 { namespace rr {

#define RB29_MASK 0x1fffffff
#define RB29_HALF 0x10000000            // 2^28
#define RB29_U (RB29_HALF + 4096)       // unit of the limb bound

// single-member struct of a vector: the AMDGPU calling convention passes and returns it in 9 VGPRs
typedef int32_t i32x9 __attribute__((ext_vector_type(9)));

RB_HD constexpr int32_t rr_p(int i) { constexpr int32_t m[9] = RB29_P; return m[i]; }

#if defined(RB29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
// run-time restatement of the static bounds: limbs, and the value against V * 1.5 p through the top limb (p / 2^232 = 3170894.7)
inline void rr_check(const i32x9& l, int L, int V, const char* what) {
  for (int k = 0; k < 9; k++) {
    const long long a = l[k] < 0 ? -(long long)l[k] : l[k];
    if (a > (long long)L * RB29_U) { fprintf(stderr, "fp29 bound violated (%s): limb %d = %d, L = %d\n", what, k, l[k], L); abort(); }
  }
  double v = 0;
  for (int k = 8; k >= 0; k--) v = v * 536870912.0 + l[k];
  const double p = 21888242871839275222246405745257275088696311157297823662689037894645226208583.0;
  if (v > 1.5 * V * p || v < -1.5 * V * p) { fprintf(stderr, "fp29 value bound violated (%s): x / p = %f, V = %d\n", what, v / p, V); abort(); }
}
#define RR_CHECK(x, what) rr_check((x).l, (x).LB, (x).VB, what)
#else
#define RR_CHECK(x, what) ((void)0)
#endif

template <int L, int V>
struct FB {
  static constexpr int LB = L, VB = V;
  i32x9 l;
  FB() = default;
  template <int L2, int V2, class = typename std::enable_if<(L2 <= L && V2 <= V && (L2 < L || V2 < V))>::type>
  RB_HD FB(const FB<L2, V2>& o) : l(o.l) {}          // widening the bounds is free
};
typedef FB<1, 1> F;

template <int L, int V> RB_HD FB<L, V> mk(const i32x9& l) { FB<L, V> r; r.l = l; return r; }
#define RB29_CONST(name, ...) \
  RB_HD F name() { constexpr int32_t m[9] = __VA_ARGS__; F r; _Pragma("unroll") for (int i = 0; i < 9; i++) r.l[i] = m[i]; return r; }
RB29_CONST(one, RB29_ONE)
RB29_CONST(c266, RB29_C266)
RB29_CONST(c256, RB29_C256)
RB_HD F zero() { F r; r.l = (i32x9)(0); return r; }

template <int L1, int V1, int L2, int V2>
RB_HD FB<L1 + L2, V1 + V2> add(const FB<L1, V1>& a, const FB<L2, V2>& b) {
  static_assert(L1 + L2 <= 7, "fp29: limb bound of a sum exceeds int32 -- normalise an operand first");
  return mk<L1 + L2, V1 + V2>(a.l + b.l);
}
template <int L1, int V1, int L2, int V2>
RB_HD FB<L1 + L2, V1 + V2> sub(const FB<L1, V1>& a, const FB<L2, V2>& b) {
  static_assert(L1 + L2 <= 7, "fp29: limb bound of a difference exceeds int32 -- normalise an operand first");
  return mk<L1 + L2, V1 + V2>(a.l - b.l);
}
template <int L, int V> RB_HD FB<L, V> neg(const FB<L, V>& a) { return mk<L, V>(-a.l); }
template <int L, int V> RB_HD FB<2 * L, 2 * V> dbl(const FB<L, V>& a) { static_assert(2 * L <= 7, "fp29: dbl"); return mk<2 * L, 2 * V>(a.l + a.l); }
template <int L, int V> RB_HD FB<3 * L, 3 * V> tpl(const FB<L, V>& a) { static_assert(3 * L <= 7, "fp29: tpl"); return mk<3 * L, 3 * V>(a.l + a.l + a.l); }

// ---- normalisation.  One parallel pass: limb k keeps its balanced low 29 bits and takes the carry of limb k - 1 (|carry| <= 2^4 for
// int32 inputs, so the result is within U); with REDUCE the value loses q p first, q = round(x / p) from the top limb
// (x / 2^232 = l[8] +- L / 2 against p / 2^232 = 3.17e6: the quotient is off by less than 1e-5 of p), leaving |x| <= 0.51 p.
template <int L, int V>
RB_HD FB<1, (V <= 2 ? V : 1)> norm(const FB<L, V>& a) {
  static_assert(L <= 6, "fp29: norm takes limbs below 6 U (the rounding offset must not overflow)");
  RR_CHECK(a, "norm in");
  i32x9 r;
  if (V <= 2) {
    int32_t c_prev = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int32_t s = a.l[i] + RB29_HALF;
      r[i] = (int32_t)((uint32_t)s & RB29_MASK) - RB29_HALF + c_prev;
      c_prev = s >> 29;
    }
    r[8] = a.l[8] + c_prev;
  } else {
    const int32_t q = (int32_t)(((int64_t)a.l[8] * RB29_QK + ((int64_t)1 << 43)) >> 44);
    int32_t c_prev = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int64_t s = (int64_t)(a.l[i] + RB29_HALF) - (int64_t)q * (int64_t)rr_p(i);
      r[i] = (int32_t)((uint32_t)s & RB29_MASK) - RB29_HALF + c_prev;
      c_prev = (int32_t)(s >> 29);
    }
    r[8] = a.l[8] - q * rr_p(8) + c_prev;
  }
  const FB<1, (V <= 2 ? V : 1)> out = mk<1, (V <= 2 ? V : 1)>(r);
  RR_CHECK(out, "norm out");
  return out;
}
// base + 9 y, normalised and reduced in one pass (the multiplication by xi = 9 + u of the tower); base may be any sum within int32
template <int L1, int V1, int L2, int V2>
RB_HD F norm_lin9(const FB<L1, V1>& base, const FB<L2, V2>& y) {
  static_assert(L1 <= 6 && L2 <= 7, "fp29: norm_lin9");
  RR_CHECK(base, "lin9 base"); RR_CHECK(y, "lin9 y");
  const int32_t top = base.l[8] + 9 * y.l[8];          // top limbs are tiny (value / 2^232)
  const int32_t q = (int32_t)(((int64_t)top * RB29_QK + ((int64_t)1 << 43)) >> 44);
  i32x9 r;
  int32_t c_prev = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int64_t s = (int64_t)(base.l[i] + RB29_HALF) + (int64_t)y.l[i] * 9 - (int64_t)q * (int64_t)rr_p(i);
    r[i] = (int32_t)((uint32_t)s & RB29_MASK) - RB29_HALF + c_prev;
    c_prev = (int32_t)(s >> 29);
  }
  r[8] = top - q * rr_p(8) + c_prev;
  const F out = mk<1, 1>(r);
  RR_CHECK(out, "lin9 out");
  return out;
}
// x / 2 mod p: add p (balanced limbs) to an odd value, then shift; a limb's low bit moves down as 2^28
template <int L, int V>
RB_HD FB<(L + 2) / 2 + 1, (V + 2) / 2> half(const FB<L, V>& a) {
  static_assert(L <= 5, "fp29: half");
  constexpr int32_t pb[9] = RB29_PBAL;
  const int32_t odd = -(a.l[0] & 1);
  i32x9 s, r;
#pragma unroll
  for (int i = 0; i < 9; i++) s[i] = a.l[i] + (pb[i] & odd);
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = (s[i] >> 1) + ((s[i + 1] & 1) << 28);
  r[8] = s[8] >> 1;
  return mk<(L + 2) / 2 + 1, (V + 2) / 2>(r);
}

// ---- multiplication.  Column sums t[0..16] (+ the carry column t[17]), schoolbook, no carries.
RB_HD void cols_init(int64_t* t) {
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
  for (int i = 9; i < 17; i++) t[i] = RB29_HALF;          // the output limbs are balanced: redc's rounding offset, folded in here
  t[17] = 0;
}
RB_HD void cols_mac(int64_t* t, const i32x9& a, const i32x9& b) {
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) t[i + j] += (int64_t)a[i] * (int64_t)b[j];
}
// Montgomery reduction: r = T / 2^261 mod p with balanced limbs, |r| <= |T| / 2^261 + p
RB_HD i32x9 redc(int64_t* t) {
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int32_t m = (int32_t)(((uint32_t)t[i] * RB29_PINV) & RB29_MASK);
#pragma unroll
    for (int j = 0; j < 9; j++) t[i + j] += (int64_t)m * (int64_t)rr_p(j);
    t[i + 1] += t[i] >> 29;                                // exact: the low 29 bits are zero now
  }
  i32x9 r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    r[k] = (int32_t)((uint32_t)t[9 + k] & RB29_MASK) - RB29_HALF;
    t[10 + k] += t[9 + k] >> 29;
  }
  r[8] = (int32_t)t[17];
  return r;
}
// host-only instrumentation (tests/hostsim builds with -DRB_COUNT_MULS): multiply-add instructions this core issues -- 81 per schoolbook
// product and 81 per reduction -- for the roofline's work count of the reduced-radix kernels (tests/count_muls.py)
#if defined(RB_COUNT_MULS) && !defined(__HIP_DEVICE_COMPILE__)
#define RR_COUNT(n) (::rb_rr_mad_counter += 81ull * (n))
#else
#define RR_COUNT(n) ((void)0)
#endif
RB_HD i32x9 mul_raw(const i32x9& a, const i32x9& b) {
  RR_COUNT(2);
  int64_t t[18];
  cols_init(t);
  cols_mac(t, a, b);
  return redc(t);
}
RB_HD i32x9 mac2_raw(const i32x9& a, const i32x9& b, const i32x9& c, const i32x9& d) {          // (a b + c d) / R
  RR_COUNT(3);
  int64_t t[18];
  cols_init(t);
  cols_mac(t, a, b);
  cols_mac(t, c, d);
  return redc(t);
}
// ---- out-of-line forms for the device.  The Fq2-level routines are the call granularity of the pairing kernels, as in tower.h
// (a kernel body has ~130 Fq2 multiplication sites; inlined they would be ~90 k instructions streaming through a 64 KB instruction
// cache).  The calling convention passes 31 argument dwords and returns 16 in VGPRs; an Fq2 product has 36 in and 18 out.  The
// remainder travels through a per-lane LDS slot (written before the call, read after it: the LDS serves a wave's accesses in
// order) instead of the stack, which is HBM-backed scratch: measured 3.39 k cycles per multiplication against 3.17 k fully
// inlined and 4.4 k with stack arguments (tools/ubench_rr29.hip).  Blocks of at most RB29_SIDE_LANES threads.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RB29_INLINE_ALL)
#define RB29_CALLS 1
#ifndef RB29_SIDE_LANES
#define RB29_SIDE_LANES 256
#endif
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x8 __attribute__((ext_vector_type(8)));
static __shared__ uint32_t rr_side[5 * RB29_SIDE_LANES];
struct Out16 { i32x8 lo0, lo1; };
__device__ __forceinline__ i32x9 side_join(const i32x4& lo, const uint32_t* side) {
  i32x9 b;
  b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3];
  b[4] = (int32_t)side[0]; b[5] = (int32_t)side[RB29_SIDE_LANES]; b[6] = (int32_t)side[2 * RB29_SIDE_LANES]; b[7] = (int32_t)side[3 * RB29_SIDE_LANES];
  b[8] = (int32_t)side[4 * RB29_SIDE_LANES];
  return b;
}
__device__ __forceinline__ i32x4 side_split(const i32x9& b, uint32_t* side) {
  side[0] = (uint32_t)b[4]; side[RB29_SIDE_LANES] = (uint32_t)b[5]; side[2 * RB29_SIDE_LANES] = (uint32_t)b[6]; side[3 * RB29_SIDE_LANES] = (uint32_t)b[7];
  side[4 * RB29_SIDE_LANES] = (uint32_t)b[8];
  i32x4 lo;
  lo[0] = b[0]; lo[1] = b[1]; lo[2] = b[2]; lo[3] = b[3];
  return lo;
}
__device__ __forceinline__ Out16 side_ret(const i32x9& c0, const i32x9& c1, uint32_t* side) {
  side[0] = (uint32_t)c0[8]; side[RB29_SIDE_LANES] = (uint32_t)c1[8];
  Out16 o;
#pragma unroll
  for (int i = 0; i < 8; i++) { o.lo0[i] = c0[i]; o.lo1[i] = c1[i]; }
  return o;
}
#define RB29_TAKE(o, c0, c1, side)                                          \
  ::rabe::bn254::rr::i32x9 c0, c1;                                                           \
  _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) { c0[i_] = o.lo0[i_]; c1[i_] = o.lo1[i_]; } \
  c0[8] = (int32_t)side[0]; c1[8] = (int32_t)side[RB29_SIDE_LANES];
// (a0 + a1 u)(b0 + b1 u)
__device__ __attribute__((noinline)) Out16 mul2_core(i32x9 a0, i32x9 a1, i32x9 b0, i32x4 b1lo) {
  uint32_t* side = rr_side + threadIdx.x;
  const i32x9 b1 = side_join(b1lo, side);
  return side_ret(mac2_raw(a0, b0, -a1, b1), mac2_raw(a0, b1, a1, b0), side);
}
// (s d) + (2 a0 a1) u with s = a0 + a1, d = a0 - a1 formed by the caller, or a0 a0 - a1 a1 when the bounds ask for it
__device__ __attribute__((noinline)) Out16 sqr2_sd_core(i32x9 s, i32x9 d, i32x9 a0x2, i32x4 a1lo) {
  uint32_t* side = rr_side + threadIdx.x;
  const i32x9 a1 = side_join(a1lo, side);
  return side_ret(mul_raw(s, d), mul_raw(a0x2, a1), side);
}
__device__ __attribute__((noinline)) Out16 sqr2_mac_core(i32x9 a0, i32x9 a1) {
  uint32_t* side = rr_side + threadIdx.x;
  return side_ret(mac2_raw(a0, a0, -a1, a1), mul_raw(a0 + a0, a1), side);
}
__device__ __attribute__((noinline)) Out16 mul2_fp_core(i32x9 a0, i32x9 a1, i32x9 k) {
  uint32_t* side = rr_side + threadIdx.x;
  return side_ret(mul_raw(a0, k), mul_raw(a1, k), side);
}
struct Out9 { i32x9 v; };
__device__ __attribute__((noinline)) Out9 mul_core(i32x9 a, i32x9 b) { return Out9{mul_raw(a, b)}; }
__device__ __attribute__((noinline)) Out9 mac2_core(i32x9 a, i32x9 b, i32x9 c, i32x4 dlo) {
  uint32_t* side = rr_side + threadIdx.x;
  return Out9{mac2_raw(a, b, c, side_join(dlo, side))};
}
#endif

template <int L1, int V1, int L2, int V2>
RB_HD F mul(const FB<L1, V1>& a, const FB<L2, V2>& b) {
  static_assert(L1 * L2 <= 10, "fp29: limb bounds of a product overflow the 64-bit column");
  static_assert(V1 * V2 <= 36, "fp29: value bounds of a product");
  RR_CHECK(a, "mul a"); RR_CHECK(b, "mul b");
#ifdef RB29_CALLS
  const F r = mk<1, 1>(mul_core(a.l, b.l).v);
#else
  const F r = mk<1, 1>(mul_raw(a.l, b.l));
#endif
  RR_CHECK(r, "mul out");
  return r;
}
template <int L1, int V1, int L2, int V2, int L3, int V3, int L4, int V4>
RB_HD F mac2(const FB<L1, V1>& a, const FB<L2, V2>& b, const FB<L3, V3>& c, const FB<L4, V4>& d) {
  static_assert(L1 * L2 + L3 * L4 <= 10, "fp29: limb bounds of a two-product sum overflow the 64-bit column");
  static_assert(V1 * V2 + V3 * V4 <= 36, "fp29: value bounds of a two-product sum");
  RR_CHECK(a, "mac2 a"); RR_CHECK(b, "mac2 b"); RR_CHECK(c, "mac2 c"); RR_CHECK(d, "mac2 d");
#ifdef RB29_CALLS
  const F r = mk<1, 1>(mac2_core(a.l, b.l, c.l, side_split(d.l, rr_side + threadIdx.x)).v);
#else
  const F r = mk<1, 1>(mac2_raw(a.l, b.l, c.l, d.l));
#endif
  RR_CHECK(r, "mac2 out");
  return r;
}

// (x0 y0 + x1 y1 + x2 y2) over Fq2 on ONE pair of column sets: twelve schoolbook products, two reductions -- the form in which the
// sparse line products of the Miller loop are taken (pairing29.h): no Karatsuba sums, nothing to normalise afterwards
RB_HD void dot3_raw(i32x9& c0, i32x9& c1, const i32x9& x0a, const i32x9& x0b, const i32x9& y0a, const i32x9& y0b, const i32x9& x1a, const i32x9& x1b,
                    const i32x9& y1a, const i32x9& y1b, const i32x9& x2a, const i32x9& x2b, const i32x9& y2a, const i32x9& y2b) {
  RR_COUNT(14);
  {
    int64_t t[18];
    cols_init(t);
    cols_mac(t, x0a, y0a); cols_mac(t, -x0b, y0b);
    cols_mac(t, x1a, y1a); cols_mac(t, -x1b, y1b);
    cols_mac(t, x2a, y2a); cols_mac(t, -x2b, y2b);
    c0 = redc(t);
  }
  {
    int64_t t[18];
    cols_init(t);
    cols_mac(t, x0a, y0b); cols_mac(t, x0b, y0a);
    cols_mac(t, x1a, y1b); cols_mac(t, x1b, y1a);
    cols_mac(t, x2a, y2b); cols_mac(t, x2b, y2a);
    c1 = redc(t);
  }
}

// the same with an Fp factor in the first term: (x0 s + x1 y1 + x2 y2), s in Fq -- ten products, two reductions.  The line of a PREPARED
// pair is stored with a unit y-coefficient (engine_rr.hip: k_lines_to_rr), so that what meets the accumulator's k-th coefficient is y_P itself
RB_HD void dot3s_raw(i32x9& c0, i32x9& c1, const i32x9& x0a, const i32x9& x0b, const i32x9& s, const i32x9& x1a, const i32x9& x1b,
                     const i32x9& y1a, const i32x9& y1b, const i32x9& x2a, const i32x9& x2b, const i32x9& y2a, const i32x9& y2b) {
  RR_COUNT(12);
  {
    int64_t t[18];
    cols_init(t);
    cols_mac(t, x0a, s);
    cols_mac(t, x1a, y1a); cols_mac(t, -x1b, y1b);
    cols_mac(t, x2a, y2a); cols_mac(t, -x2b, y2b);
    c0 = redc(t);
  }
  {
    int64_t t[18];
    cols_init(t);
    cols_mac(t, x0b, s);
    cols_mac(t, x1a, y1b); cols_mac(t, x1b, y1a);
    cols_mac(t, x2a, y2b); cols_mac(t, x2b, y2a);
    c1 = redc(t);
  }
}

// ---- conversions from / to fp.h's canonical Montgomery form (x 2^256 mod p in 8 x 32-bit limbs)
RB_HD FB<2, 1> unpack(const Fp& x) {          // the 256-bit integer cut into 29-bit pieces (non-negative, < 2^29)
  i32x9 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int bit = 29 * k, w = bit >> 5, s = bit & 31;
    uint32_t v = x.v[w] >> s;
    if (s > 3 && w + 1 < 8) v |= x.v[w + 1] << (32 - s);
    r[k] = (int32_t)(v & RB29_MASK);
  }
  return mk<2, 1>(r);
}
RB_HD F from_fp(const Fp& x) { return mul(unpack(x), c266()); }
RB_HD Fp to_fp(const F& a) {
  const F v = mul(a, c256());                 // |v| <= 1.01 p
  // v + 2 p in (0.99 p, 3.01 p), carried into non-negative 29-bit limbs, repacked, then at most three subtractions of p
  uint32_t u[9];
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    acc += (int64_t)v.l[k] + 2 * (int64_t)rr_p(k);
    u[k] = (k < 8) ? (uint32_t)acc & RB29_MASK : (uint32_t)acc;
    acc >>= 29;
  }
  uint32_t t[8];
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const int bit = 32 * w, k = bit / 29, s = bit - 29 * k;
    uint32_t x = u[k] >> s;
    x |= u[k + 1] << (29 - s);
    if (58 - s < 32 && k + 2 < 9) x |= u[k + 2] << (58 - s);
    t[w] = x;
  }
  cond_sub_mod<FpParams>(t, 0);
  cond_sub_mod<FpParams>(t, 0);
  cond_sub_mod<FpParams>(t, 0);
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return r;
}

// ============================================================================ Fq2 = Fp[u]/(u^2 + 1)
template <int L, int V>
struct F2B {
  FB<L, V> c0, c1;
  F2B() = default;
  RB_HD F2B(const FB<L, V>& a, const FB<L, V>& b) : c0(a), c1(b) {}
  template <int L2, int V2, class = typename std::enable_if<(L2 <= L && V2 <= V && (L2 < L || V2 < V))>::type>
  RB_HD F2B(const F2B<L2, V2>& o) : c0(o.c0), c1(o.c1) {}
};
typedef F2B<1, 1> F2;
template <int L, int V> RB_HD F2B<L, V> mk2(const FB<L, V>& a, const FB<L, V>& b) { return F2B<L, V>(a, b); }
RB_HD F2 zero2() { return mk2(zero(), zero()); }
RB_HD F2 one2() { return mk2(one(), zero()); }
template <int L1, int V1, int L2, int V2> RB_HD F2B<L1 + L2, V1 + V2> add2(const F2B<L1, V1>& a, const F2B<L2, V2>& b) { return mk2(add(a.c0, b.c0), add(a.c1, b.c1)); }
template <int L1, int V1, int L2, int V2> RB_HD F2B<L1 + L2, V1 + V2> sub2(const F2B<L1, V1>& a, const F2B<L2, V2>& b) { return mk2(sub(a.c0, b.c0), sub(a.c1, b.c1)); }
template <int L, int V> RB_HD F2B<L, V> neg2(const F2B<L, V>& a) { return mk2(neg(a.c0), neg(a.c1)); }
template <int L, int V> RB_HD F2B<L, V> conj2(const F2B<L, V>& a) { return mk2(a.c0, neg(a.c1)); }
template <int L, int V> RB_HD F2B<2 * L, 2 * V> dbl2(const F2B<L, V>& a) { return mk2(dbl(a.c0), dbl(a.c1)); }
template <int L, int V> RB_HD F2B<3 * L, 3 * V> tpl2(const F2B<L, V>& a) { return mk2(tpl(a.c0), tpl(a.c1)); }
template <int L, int V> RB_HD auto norm2(const F2B<L, V>& a) { return mk2(norm(a.c0), norm(a.c1)); }
template <int L, int V> RB_HD auto half2(const F2B<L, V>& a) { return mk2(half(a.c0), half(a.c1)); }
// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u: four schoolbook products on two column sets, two reductions
template <int L1, int V1, int L2, int V2>
RB_HD F2 mul2(const F2B<L1, V1>& a, const F2B<L2, V2>& b) {
#ifdef RB29_CALLS
  static_assert(2 * L1 * L2 <= 10 && 2 * V1 * V2 <= 36, "fp29: bounds of an Fq2 product");
  uint32_t* side = rr_side + threadIdx.x;
  const Out16 o = mul2_core(a.c0.l, a.c1.l, b.c0.l, side_split(b.c1.l, side));
  RB29_TAKE(o, c0, c1, side)
  return mk2(mk<1, 1>(c0), mk<1, 1>(c1));
#else
  return mk2(mac2(a.c0, b.c0, neg(a.c1), b.c1), mac2(a.c0, b.c1, a.c1, b.c0));
#endif
}
// (a0 + a1)(a0 - a1) + 2 a0 a1 u where the bounds allow the sum and the difference as operands, a0 a0 - a1 a1 otherwise
template <int L, int V>
RB_HD F2 sqr2(const F2B<L, V>& a) {
#ifdef RB29_CALLS
  static_assert(2 * L * L <= 10 && 2 * V * V <= 36, "fp29: bounds of an Fq2 square");
  uint32_t* side = rr_side + threadIdx.x;
  if constexpr (4 * L * L <= 10 && 4 * V * V <= 36) {
    const Out16 o = sqr2_sd_core(a.c0.l + a.c1.l, a.c0.l - a.c1.l, a.c0.l + a.c0.l, side_split(a.c1.l, side));
    RB29_TAKE(o, c0, c1, side)
    return mk2(mk<1, 1>(c0), mk<1, 1>(c1));
  } else {
    const Out16 o = sqr2_mac_core(a.c0.l, a.c1.l);
    RB29_TAKE(o, c0, c1, side)
    return mk2(mk<1, 1>(c0), mk<1, 1>(c1));
  }
#else
  if constexpr (4 * L * L <= 10 && 4 * V * V <= 36) return mk2(mul(add(a.c0, a.c1), sub(a.c0, a.c1)), mul(dbl(a.c0), a.c1));
  else return mk2(mac2(a.c0, a.c0, neg(a.c1), a.c1), mul(dbl(a.c0), a.c1));
#endif
}
template <int L1, int V1, int L2, int V2>
RB_HD F2 mul2_fp(const F2B<L1, V1>& a, const FB<L2, V2>& k) {
#ifdef RB29_CALLS
  static_assert(L1 * L2 <= 10 && V1 * V2 <= 36, "fp29: bounds of an Fq2 x Fp product");
  uint32_t* side = rr_side + threadIdx.x;
  const Out16 o = mul2_fp_core(a.c0.l, a.c1.l, k.l);
  RB29_TAKE(o, c0, c1, side)
  return mk2(mk<1, 1>(c0), mk<1, 1>(c1));
#else
  return mk2(mul(a.c0, k), mul(a.c1, k));
#endif
}
template <int LA, int VA, int LB, int VB, int LC, int VC, int LD, int VD, int LE, int VE, int LF, int VF>
RB_HD F2 dot3(const F2B<LA, VA>& x0, const F2B<LB, VB>& y0, const F2B<LC, VC>& x1, const F2B<LD, VD>& y1, const F2B<LE, VE>& x2, const F2B<LF, VF>& y2) {
  static_assert(2 * (LA * LB + LC * LD + LE * LF) <= 10, "fp29: limb bounds of an Fq2 dot product overflow the 64-bit column");
  static_assert(2 * (VA * VB + VC * VD + VE * VF) <= 36, "fp29: value bounds of an Fq2 dot product");
  RR_CHECK(x0.c0, "dot3"); RR_CHECK(x0.c1, "dot3"); RR_CHECK(y0.c0, "dot3"); RR_CHECK(y0.c1, "dot3"); RR_CHECK(x1.c0, "dot3"); RR_CHECK(x1.c1, "dot3");
  RR_CHECK(y1.c0, "dot3"); RR_CHECK(y1.c1, "dot3"); RR_CHECK(x2.c0, "dot3"); RR_CHECK(x2.c1, "dot3"); RR_CHECK(y2.c0, "dot3"); RR_CHECK(y2.c1, "dot3");
  i32x9 c0, c1;
  dot3_raw(c0, c1, x0.c0.l, x0.c1.l, y0.c0.l, y0.c1.l, x1.c0.l, x1.c1.l, y1.c0.l, y1.c1.l, x2.c0.l, x2.c1.l, y2.c0.l, y2.c1.l);
  const F2 r = mk2(mk<1, 1>(c0), mk<1, 1>(c1));
  RR_CHECK(r.c0, "dot3 out"); RR_CHECK(r.c1, "dot3 out");
  return r;
}
template <int LA, int VA, int LB, int VB, int LC, int VC, int LD, int VD, int LE, int VE, int LF, int VF>
RB_HD F2 dot3s(const F2B<LA, VA>& x0, const FB<LB, VB>& s, const F2B<LC, VC>& x1, const F2B<LD, VD>& y1, const F2B<LE, VE>& x2, const F2B<LF, VF>& y2) {
  static_assert(LA * LB + 2 * (LC * LD + LE * LF) <= 10, "fp29: limb bounds of an Fq2 dot product overflow the 64-bit column");
  static_assert(VA * VB + 2 * (VC * VD + VE * VF) <= 36, "fp29: value bounds of an Fq2 dot product");
  RR_CHECK(x0.c0, "dot3s"); RR_CHECK(x0.c1, "dot3s"); RR_CHECK(s, "dot3s"); RR_CHECK(x1.c0, "dot3s"); RR_CHECK(x1.c1, "dot3s");
  RR_CHECK(y1.c0, "dot3s"); RR_CHECK(y1.c1, "dot3s"); RR_CHECK(x2.c0, "dot3s"); RR_CHECK(x2.c1, "dot3s"); RR_CHECK(y2.c0, "dot3s"); RR_CHECK(y2.c1, "dot3s");
  i32x9 c0, c1;
  dot3s_raw(c0, c1, x0.c0.l, x0.c1.l, s.l, x1.c0.l, x1.c1.l, y1.c0.l, y1.c1.l, x2.c0.l, x2.c1.l, y2.c0.l, y2.c1.l);
  const F2 r = mk2(mk<1, 1>(c0), mk<1, 1>(c1));
  RR_CHECK(r.c0, "dot3s out"); RR_CHECK(r.c1, "dot3s out");
  return r;
}
// x + xi y,  xi = 9 + u:  (x0 + 9 y0 - y1) + (x1 + 9 y1 + y0) u, normalised
template <int L1, int V1, int L2, int V2>
RB_HD F2 add_mul_xi2(const F2B<L1, V1>& x, const F2B<L2, V2>& y) {
  return mk2(norm_lin9(sub(x.c0, y.c1), y.c0), norm_lin9(add(x.c1, y.c0), y.c1));
}
template <int L, int V>
RB_HD F2 mul_xi2(const F2B<L, V>& y) { return mk2(norm_lin9(neg(y.c1), y.c0), norm_lin9(y.c0, y.c1)); }

RB_HD F2 from_fp2(const Fp2& x) { return mk2(from_fp(x.c0), from_fp(x.c1)); }
RB_HD Fp2 to_fp2(const F2& a) { return Fp2{to_fp(a.c0), to_fp(a.c1)}; }
#define RB29_CONST2(name, C0, C1) \
  RB29_CONST(name##_c0, C0) RB29_CONST(name##_c1, C1) RB_HD F2 name() { return mk2(name##_c0(), name##_c1()); }
RB29_CONST2(twist_b, RB29_TWIST_B_C0, RB29_TWIST_B_C1)
RB29_CONST2(gamma1_1, RB29_GAMMA1_1_C0, RB29_GAMMA1_1_C1)
RB29_CONST2(gamma1_2, RB29_GAMMA1_2_C0, RB29_GAMMA1_2_C1)
RB29_CONST2(gamma1_3, RB29_GAMMA1_3_C0, RB29_GAMMA1_3_C1)
RB29_CONST2(gamma1_4, RB29_GAMMA1_4_C0, RB29_GAMMA1_4_C1)
RB29_CONST2(gamma1_5, RB29_GAMMA1_5_C0, RB29_GAMMA1_5_C1)
RB29_CONST2(gamma3_1, RB29_GAMMA3_1_C0, RB29_GAMMA3_1_C1)
RB29_CONST2(gamma3_2, RB29_GAMMA3_2_C0, RB29_GAMMA3_2_C1)
RB29_CONST2(gamma3_3, RB29_GAMMA3_3_C0, RB29_GAMMA3_3_C1)
RB29_CONST2(gamma3_4, RB29_GAMMA3_4_C0, RB29_GAMMA3_4_C1)
RB29_CONST2(gamma3_5, RB29_GAMMA3_5_C0, RB29_GAMMA3_5_C1)
RB29_CONST(gamma2_1, RB29_GAMMA2_1_C0)
RB29_CONST(gamma2_2, RB29_GAMMA2_2_C0)
RB29_CONST(gamma2_3, RB29_GAMMA2_3_C0)
RB29_CONST(gamma2_4, RB29_GAMMA2_4_C0)
RB29_CONST(gamma2_5, RB29_GAMMA2_5_C0)

} } }   // namespace rabe::bn254::rr
