// Six-lane cooperative Fq12 arithmetic: ONE Fq12 value spread over six adjacent lanes of a wave, one Fq2 coefficient per lane.
//
// Why (SURVEY.md section 7 "Alternative to evaluate"): the one-lane-per-item pairing kernels keep a whole
// Fq12 accumulator (96 registers) plus the Fq6 temporaries of its Karatsuba steps in one lane -- one wave per SIMD, ~490
// registers, and a ~15 k-multiplication dependency chain per item (the latency of `ac17::cp_decrypt`, src/schemes/ac17/mod.rs:385-430,
// when it is called for ONE ciphertext).  Here the value is seen as  f = sum_k a_k w^k  over Fq2[w]/(w^6 - xi)  (the same six
// Fq2 coefficients as the tower Fq2 -> Fq6 -> Fq12 of tower.h, only indexed by the power of w: c0 = (a0, a2, a4), c1 = (a1, a3, a5)),
// lane k of a group owns a_k, and every Fq12 operation is a DOT PRODUCT per lane
//       r_k = sum_s x_s * y_s          (2 .. 6 Fq2 products, operands fetched from the group's slots in LDS)
// accumulated as three unreduced 512-bit sums (Karatsuba over Fq2 on the wide values) and reduced ONCE:
//       MUL    r_k = sum_i b_i f'_(k-i)          6 products       (f' = xi f when the index wraps: rows F and FX hold both)
//       LINE   r_k = l0 f_k + l1 f'_(k-1) + l3 f'_(k-3)            the sparse line of the D-type twist, 3 products
//       SQR    r_k = 2 sum_(i<j, i+j=k mod 6) a_i a'_j + a_i a'_i   4 products (halved squares, doubled after the reduction)
//       CSQR   Granger-Scott: the Fq4 squarings of the pairs (a_k, a_(k+3)), 2 products
// 36 / 18 / 24 / 12 Fq2 products per operation against 18 / 13 / 12 / 9 in one lane -- but 12 instead of 36 Montgomery reductions
// for a multiplication, no Fq6 / Fq12 Karatsuba glue, ~130 live registers per lane (two to four waves per SIMD) and a dependency
// chain six times shorter.  Values are exactly those of tower.h / pairing.h (field elements are unique): the kernels built on
// this header are byte-identical to the one-lane kernels (tests/test_hostsim_coop6.py, tests/test_gpu_coop6.py).
//
// The code is written against a lane context CX so that the host build (tests/hostsim: six threads and a barrier per group) runs
// the very same functions:
//   int  role() const                         k: the power of w this lane owns
//   Fp2  ld(int row, int lane) const          slot of lane `lane` of my group in row `row`
//   void st(int row, const Fp2&) const        my slot of row `row`
//   bool all(bool) const                      true when every lane of the group passes true (device: a ballot; host: slots + barriers)
//   void sync() const                         orders the group's slot traffic (device: the wave runs in lockstep and the LDS keeps a
//                                             wave's accesses in order, so this is a compiler barrier only; host: a thread barrier)
// Discipline: a slot written between two sync()s is not read by another lane between the same two.
#pragma once
#include "pairing.h"

namespace rabe { namespace bn254 {

enum { C6_F = 0, C6_FX = 1, C6_L0 = 2, C6_L1 = 3, C6_L3 = 4, C6_ROWS = 5, C6_B = C6_L0 };
enum { C6_OP_MUL = 0, C6_OP_LINE, C6_OP_LINE024, C6_OP_SQR, C6_OP_CSQR };

// ---- three unreduced sums of 256 x 256-bit products: t0 += x0 y0, t1 += x1 y1, t2 += (x0 + x1)(y0 + y1)
struct Wide3 {
  uint32_t t0[16], t1[16], t2[16];
};
#if defined(__HIP_DEVICE_COMPILE__)
// wide_mac3 comes from tools/gen_fp_asm.py (fp_gfx950_gen.h): wide_mul3 with the running sums added in column by column
template <bool FIRST>
RB_HD void wide3_mac_t(Wide3& T, const Fp2& x, const Fp2& y) {
  uint32_t sa[8], sb[8];
  { uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sa[i] = addc32(x.c0.v[i], x.c1.v[i], c); }      // < 2p < 2^255: no carry out
  { uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sb[i] = addc32(y.c0.v[i], y.c1.v[i], c); }
  if (FIRST) wide_mul3(T.t0, T.t1, T.t2, x.c0.v, y.c0.v, x.c1.v, y.c1.v, sa, sb);        // the first product DEFINES the sums
  else wide_mac3(T.t0, T.t1, T.t2, x.c0.v, y.c0.v, x.c1.v, y.c1.v, sa, sb);
}
RB_HD void wide3_mac(Wide3& T, const Fp2& x, const Fp2& y) { wide3_mac_t<false>(T, x, y); }
RB_HD void wide3_first(Wide3& T, const Fp2& x, const Fp2& y) { wide3_mac_t<true>(T, x, y); }
#else
RB_HD void wide_mac1_portable(uint32_t* T, const uint32_t* a, const uint32_t* b) {     // T += a b mod 2^512
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    for (int j = 0; j < 8; j++) {
      const uint64_t x = (uint64_t)a[i] * b[j] + T[i + j] + c;
      T[i + j] = (uint32_t)x;
      c = x >> 32;
    }
    for (int k = i + 8; k < 16 && c; k++) {
      const uint64_t x = (uint64_t)T[k] + c;
      T[k] = (uint32_t)x;
      c = x >> 32;
    }
  }
}
RB_HD void wide3_mac(Wide3& T, const Fp2& x, const Fp2& y) {
  RB_COUNT_ONE_MUL(); RB_COUNT_ONE_MUL(); RB_COUNT_ONE_MUL();
  uint32_t a0[8], a1[8], b0[8], b1[8], sa[8], sb[8];
  uint32_t ca = 0, cb = 0;
  for (int i = 0; i < 8; i++) {
    a0[i] = x.c0.v[i]; a1[i] = x.c1.v[i]; b0[i] = y.c0.v[i]; b1[i] = y.c1.v[i];
    sa[i] = addc32(a0[i], a1[i], ca);
    sb[i] = addc32(b0[i], b1[i], cb);
  }
  wide_mac1_portable(T.t0, a0, b0);
  wide_mac1_portable(T.t1, a1, b1);
  wide_mac1_portable(T.t2, sa, sb);
}
RB_HD void wide3_first(Wide3& T, const Fp2& x, const Fp2& y) {
  for (int i = 0; i < 16; i++) { T.t0[i] = 0; T.t1[i] = 0; T.t2[i] = 0; }
  wide3_mac(T, x, y);
}
RB_HD void redc_portable(uint32_t* r, const uint32_t* W) {       // W < 2^512, (W + m p) / 2^256 < 2^256 for every caller here
  uint32_t t[17];
  for (int i = 0; i < 16; i++) t[i] = W[i];
  t[16] = 0;
  for (int i = 0; i < 8; i++) {
    const uint32_t m = t[i] * FpParams::INV;
    uint64_t c = 0;
    for (int j = 0; j < 8; j++) {
      const uint64_t x = (uint64_t)m * FpParams::mod(j) + t[i + j] + c;
      t[i + j] = (uint32_t)x;
      c = x >> 32;
    }
    for (int k = i + 8; k < 17 && c; k++) {
      const uint64_t x = (uint64_t)t[k] + c;
      t[k] = (uint32_t)x;
      c = x >> 32;
    }
  }
  for (int i = 0; i < 8; i++) r[i] = t[8 + i];
  cond_sub_mod<FpParams>(r, 0);
}
#endif
RB_HD void wide3_zero(Wide3& T) {
#pragma unroll
  for (int i = 0; i < 16; i++) { T.t0[i] = 0; T.t1[i] = 0; T.t2[i] = 0; }
}
// sum of n <= 6 products (every operand < p): c0 = (T0 - T1 + 6 p^2) / R, c1 = (T2 - T0 - T1) / R mod p.
// T2 < 24 p^2 < 2^512 (tools/gen_constants.py asserts it); both numerators are in [0, 12 p^2), so a reduction leaves < 3.27 p:
// three conditional subtractions (one inside the reduction routine).
// n: the number of products in the sums -- it bounds the value a reduction leaves, i.e. how many conditional subtractions follow
// (with the fixed offset 6 p^2: c0 < ((n + 6) 0.1892 + 1) p, c1 < (2 n 0.1892 + 1) p)
RB_HD Fp2 wide3_finish(const Wide3& T, int n = 6) {
  constexpr uint32_t off[16] = RB_FP_6P2;
  uint32_t W0[16], W1[16];
  { uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) W0[i] = subb32(T.t0[i], T.t1[i], br); }
  { uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) W0[i] = addc32(W0[i], off[i], c); }
  { uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) W1[i] = subb32(T.t2[i], T.t0[i], br); }
  { uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) W1[i] = subb32(W1[i], T.t1[i], br); }
  uint32_t c0[8], c1[8];
#if defined(__HIP_DEVICE_COMPILE__)
  redc2_fp(c0, c1, W0, W1);
#else
  redc_portable(c0, W0);
  redc_portable(c1, W1);
#endif
  cond_sub_mod<FpParams>(c0, 0);                 // n <= 4: c0 < 2.9 p, one more than the reduction's own
  if (n > 4) cond_sub_mod<FpParams>(c0, 0);      // n <= 6: c0 < 3.27 p
  if (n > 2) cond_sub_mod<FpParams>(c1, 0);      // n <= 2: c1 < 1.76 p (none); n <= 5: c1 < 2.9 p
  if (n > 5) cond_sub_mod<FpParams>(c1, 0);      // n = 6: c1 < 3.27 p
  Fp2 r;
#pragma unroll
  for (int i = 0; i < 8; i++) { r.c0.v[i] = c0[i]; r.c1.v[i] = c1[i]; }
  return r;
}

// ---- which two slots lane k multiplies in step s of an operation
struct C6Slot {
  int xrow, xlane, yrow, ylane, flag;      // flag 1: y is halved (a square inside a doubled sum); 2: the step is empty (y = 0)
};
RB_HD int c6_op_slots(int op) { return op == C6_OP_MUL ? 6 : (op == C6_OP_SQR ? 4 : (op == C6_OP_CSQR ? 2 : 3)); }
// SQR, lane k, step s: xlane | ylane << 3 | (y from FX) << 6 | flag << 7
//   r_k = sum_(i+j=k) a_i a_j + xi sum_(i+j=k+6) a_i a_j = 2 * ( cross terms + halved squares )
RB_HD int c6_sqr_entry(int k, int s) {
  constexpr unsigned short t[6][4] = {
      {1 | 5 << 3 | 1 << 6, 2 | 4 << 3 | 1 << 6, 0 | 0 << 3 | 1 << 7, 3 | 3 << 3 | 1 << 6 | 1 << 7},
      {0 | 1 << 3, 2 | 5 << 3 | 1 << 6, 3 | 4 << 3 | 1 << 6, 2 << 7},
      {0 | 2 << 3, 3 | 5 << 3 | 1 << 6, 1 | 1 << 3 | 1 << 7, 4 | 4 << 3 | 1 << 6 | 1 << 7},
      {0 | 3 << 3, 1 | 2 << 3, 4 | 5 << 3 | 1 << 6, 2 << 7},
      {0 | 4 << 3, 1 | 3 << 3, 2 | 2 << 3 | 1 << 7, 5 | 5 << 3 | 1 << 6 | 1 << 7},
      {0 | 5 << 3, 1 | 4 << 3, 2 | 3 << 3, 2 << 7}};
  return t[k][s];
}
// CSQR (tower.h: fp12_cyclotomic_sqr; the Fq4 pairs are (a_k, a_(k+3)), s^2 = xi):
//   k = 0, 2, 4 (with the pair (a, b) = (a_0,a_3), (a_1,a_4), (a_2,a_5) feeding lanes 0 / 2 / 4):  t = a^2 + xi b^2,  r_k = 3 t - 2 a_k
//   k = 3, 5:  t = 2 a b of the pairs (a_0,a_3), (a_1,a_4);   k = 1:  t = xi 2 a_2 a_5;          r_k = 3 t + 2 a_k
RB_HD int c6_csqr_entry(int k, int s) {
  constexpr unsigned short t[6][2] = {{0 | 0 << 3, 3 | 3 << 3 | 1 << 6}, {2 | 5 << 3 | 1 << 6, 2 | 5 << 3 | 1 << 6},
                                      {1 | 1 << 3, 4 | 4 << 3 | 1 << 6}, {0 | 3 << 3, 0 | 3 << 3},
                                      {2 | 2 << 3, 5 | 5 << 3 | 1 << 6}, {1 | 4 << 3, 1 | 4 << 3}};
  return t[k][s];
}
RB_HD C6Slot c6_slot(int op, int s, int k, int j) {
  if (op == C6_OP_SQR || op == C6_OP_CSQR) {
    const int e = op == C6_OP_SQR ? c6_sqr_entry(k, s) : c6_csqr_entry(k, s);
    return C6Slot{C6_F, e & 7, ((e >> 6) & 1) ? C6_FX : C6_F, (e >> 3) & 7, e >> 7};
  }
  // x = coefficient of w^d of the multiplier: broadcast; y = f_(k-d), or xi f_(k-d+6) when the power wraps
  int d, xrow, xlane;
  if (op == C6_OP_MUL) { d = s; xrow = C6_B; xlane = s; }
  else if (op == C6_OP_LINE) { d = s == 2 ? 3 : s; xrow = C6_L0 + s; xlane = j; }
  else { d = 2 * s; xrow = C6_L0 + s; xlane = j; }
  const int t = k - d;
  return C6Slot{xrow, xlane, t < 0 ? C6_FX : C6_F, t < 0 ? t + 6 : t, 0};
}
RB_HD Fp2 fp2_select(bool c, const Fp2& a, const Fp2& b) {
  Fp2 r;
#pragma unroll
  for (int i = 0; i < 8; i++) { r.c0.v[i] = c ? a.c0.v[i] : b.c0.v[i]; r.c1.v[i] = c ? a.c1.v[i] : b.c1.v[i]; }
  return r;
}
// the lane's dot product of operation `op` (j: the lane whose line / sparse multiplier is used)
template <class CX>
RB_FN Fp2 c6_dot(CX cx, int op, int j) {
  Wide3 T;
  const int n = c6_op_slots(op), k = cx.role();
  {          // the first product defines the sums (no flags: SQR's halved / empty steps are its last two)
    const C6Slot e = c6_slot(op, 0, k, j);
    wide3_first(T, cx.ld(e.xrow, e.xlane), cx.ld(e.yrow, e.ylane));
  }
#pragma unroll 1
  for (int s = 1; s < n; s++) {
    const C6Slot e = c6_slot(op, s, k, j);
    const Fp2 x = cx.ld(e.xrow, e.xlane);
    Fp2 y = cx.ld(e.yrow, e.ylane);
    if (op == C6_OP_SQR && s >= 2) {
      y = fp2_select(e.flag == 1, fp2_half(y), y);
      y = fp2_select(e.flag == 2, fp2_zero(), y);
    }
    wide3_mac(T, x, y);
  }
  return wide3_finish(T, n);
}
// publish the group's new value: row F gets r, row FX gets xi r
template <class CX>
RB_HD void c6_put_f(CX cx, const Fp2& r) {
  cx.sync();
  cx.st(C6_F, r);
  cx.st(C6_FX, fp2_mul_xi(r));
  cx.sync();
}
template <class CX>
RB_HD void c6_put(CX cx, int row, const Fp2& v) {
  cx.sync();
  cx.st(row, v);
  cx.sync();
}
template <class CX> RB_HD Fp2 c6_mine(CX cx) { return cx.ld(C6_F, cx.role()); }
// f <- f^2 (general), f <- f^2 (cyclotomic subgroup), f <- f * b (b in row B), f <- f * line of lane j
template <class CX> RB_HD void c6_sqr(CX cx) { c6_put_f(cx, fp2_dbl(c6_dot(cx, C6_OP_SQR, 0))); }
template <class CX> RB_HD void c6_csqr(CX cx) {
  const Fp2 t = c6_dot(cx, C6_OP_CSQR, 0), z = c6_mine(cx);
  const bool plus = cx.role() & 1;
  const Fp2 u = fp2_select(plus, fp2_add(t, z), fp2_sub(t, z));       // 3 t -+ 2 z = t + 2 (t -+ z)
  c6_put_f(cx, fp2_add(fp2_dbl(u), t));
}
template <class CX> RB_HD void c6_mul_b(CX cx) { c6_put_f(cx, c6_dot(cx, C6_OP_MUL, 0)); }
// conjugation over Fq6 (w -> -w): the odd coefficients change sign
template <class CX> RB_HD Fp2 c6_conj(CX cx, const Fp2& a) { return (cx.role() & 1) ? fp2_neg(a) : a; }
// a^(p^e) on the lane's coefficient: conjugate (e odd) and multiply by gamma_(e,k) (tower.h: fp12_frob1/2/3)
RB_HD Fp2 c6_gamma(int e, int k) {
  Fp2 g = fp2_one();
  if (e == 1) g = k == 1 ? gamma1_1() : k == 2 ? gamma1_2() : k == 3 ? gamma1_3() : k == 4 ? gamma1_4() : k == 5 ? gamma1_5() : g;
  else if (e == 3) g = k == 1 ? gamma3_1() : k == 2 ? gamma3_2() : k == 3 ? gamma3_3() : k == 4 ? gamma3_4() : k == 5 ? gamma3_5() : g;
  else g.c0 = k == 1 ? gamma2_1() : k == 2 ? gamma2_2() : k == 3 ? gamma2_3() : k == 4 ? gamma2_4() : k == 5 ? gamma2_5() : g.c0;
  return g;
}
template <class CX> RB_FN Fp2 c6_frob(CX cx, Fp2 a, int e) {
  const int k = cx.role();
  if (e != 2) a = fp2_conj(a);
  const Fp2 r = fp2_mul(a, c6_gamma(e, k));
  return k == 0 ? a : r;
}
// value helpers: set the group's accumulator / multiplier from the lanes' coefficients
template <class CX> RB_HD Fp2 c6_mul(CX cx, const Fp2& a, const Fp2& b) {      // a * b; leaves the product in the accumulator rows
  c6_put_f(cx, a);
  c6_put(cx, C6_B, b);
  const Fp2 r = c6_dot(cx, C6_OP_MUL, 0);
  c6_put_f(cx, r);
  return r;
}

// ---- one line event of ONE pair on TWO lanes (A = even lane, B = its odd neighbour): the G2 step of pairing.h (g2hom_double /
// g2hom_add: the same field elements, so the same line and the same running point) with its ten to fifteen Fq2 products split over the
// two lanes.  Lanes of a wave share one instruction stream, so the split is written as ROUNDS: in every round both lanes execute the
// same Fq2 multiplication (or squaring) on operands chosen by their role -- two different code paths would simply run one after the
// other.  Intermediate values are handed over through the lanes' slots of rows L0 / L1 / L3; lane A ends with the scaled line
// (l0, l1, l3) in ITS slots of those rows.  The running point lives where acc keeps it, coordinate by coordinate.
//   doubling (2 squarings + 4 products deep, against 6 + 4 + 2 scalings in one lane):
//     A:  C = Z^2      E = 3 b' C    E^2      | G^2          (-H) yP      (3 J) xP        -> Y3 = G^2 - 3 E^2, l3 = E - B
//     B:  J = X^2      Y Z           B = Y^2  | X Y          A (B - F)    B H             -> X3, Z3            (A = X Y / 2, H = 2 Y Z)
//   addition (1 squaring + 7 products deep, against 2 + 11 + 2 scalings):
//     A:  qy Z | theta^2    theta qx    Z c    lambda yP | theta (g - h)    e Y    (-theta) xP    -> Y3, l3 = theta qx - lambda qy
//     B:  qx Z | lambda^2   lambda qy   e      g = X d   | lambda h         Z e    --             -> X3, Z3
// A pair that replays prepared lines joins the two scaling rounds; a pair without a step runs along on zeros.
// ACC additionally provides  Fp2 ld_tc(int j, int c) / void st_tc(int j, int c, const Fp2&)  (c = 0, 1, 2: X, Y, Z of pair j's point).
RB_HD Fp2 fp2_from_fp(const Fp& a) { return Fp2{a, zero<FpParams>()}; }
template <class CX, class ACC>
RB_HD void c6_pair_step(CX cx, ACC acc, int j, bool have, int kind, int mode, int ln) {
  const int k = cx.role();
  const bool isB = k & 1, walk = have && kind == MP_WALK, lines = have && kind == MP_LINES && !isB;
  const int other = isB ? k - 1 : k + 1;
  const Fp2 Z0 = fp2_zero();
  MillerP p{zero<FpParams>(), zero<FpParams>(), zero<FpParams>(), false};
  LineCoeffs pl{Z0, Z0, Z0};
  if (have && !isB && kind != MP_SKIP) p = acc.p(j);
  if (lines) pl = acc.line(j, ln);
  Fp2 l0 = Z0, l1 = Z0, l3 = pl.c0;
  if (mode == MS_DBL) {
    Fp2 x = Z0, y = Z0, z = Z0;
    if (walk) { z = acc.ld_tc(j, 2); if (isB) { x = acc.ld_tc(j, 0); y = acc.ld_tc(j, 1); } }
    const Fp2 r1 = fp2_sqr(isB ? x : z);                                   // A: C      B: J
    const Fp2 r2 = fp2_mul(isB ? y : twist_b(), isB ? z : fp2_add(fp2_dbl(r1), r1));      // A: E      B: Y Z
    const Fp2 r3 = fp2_sqr(isB ? y : r2);                                  // A: E^2    B: B
    const Fp2 f3 = fp2_add(fp2_dbl(r2), r2), h2 = fp2_dbl(r2);             // A: F = 3 E            B: H = 2 Y Z
    if (walk) { cx.st(C6_L0, isB ? r3 : f3); if (isB) { cx.st(C6_L1, h2); cx.st(C6_L3, r1); } }
    cx.sync();
    Fp2 o0 = Z0, o1 = Z0, o2 = Z0;
    if (walk) { o0 = cx.ld(C6_L0, other); if (!isB) { o1 = cx.ld(C6_L1, other); o2 = cx.ld(C6_L3, other); } }       // A: B, H, J    B: F
    cx.sync();
    const Fp2 g = fp2_half(fp2_add(o0, f3));
    const Fp2 r4 = fp2_mul(isB ? x : g, isB ? y : g);                      // A: G^2    B: X Y
    const Fp2 cy = lines ? pl.cy : fp2_neg(o1), cxl = lines ? pl.cx : fp2_add(fp2_dbl(o2), o2);
    const Fp2 r5 = fp2_mul(isB ? fp2_half(r4) : cy, isB ? fp2_sub(r3, o0) : fp2_from_fp(p.py));      // A: l0     B: X3 = A (B - F)
    const Fp2 r6 = fp2_mul(isB ? r3 : cxl, isB ? h2 : fp2_from_fp(p.px));                            // A: l1     B: Z3 = B H
    if (walk && isB) { acc.st_tc(j, 0, r5); acc.st_tc(j, 2, r6); }
    if (walk && !isB) { acc.st_tc(j, 1, fp2_sub(r4, fp2_add(fp2_dbl(r3), r3))); l3 = fp2_sub(r2, o0); }
    l0 = r5; l1 = r6;
  } else {
    G2Aff q{Z0, Z0};
    if (walk) {
      q = acc.q(j);
      if (mode == MS_ADD_NEG) q.y = fp2_neg(q.y);
      else if (mode == MS_FROB1) q = g2_frob1(q);
      else if (mode == MS_FROB2) q = aff_neg(g2_frob2(q));
    }
    Fp2 z = Z0, xy = Z0;
    if (walk) { z = acc.ld_tc(j, 2); xy = acc.ld_tc(j, isB ? 0 : 1); }    // B: X, A: Y
    const Fp2 t0 = fp2_sub(xy, fp2_mul(isB ? q.x : q.y, z));               // A: theta  B: lambda
    if (walk) cx.st(C6_L0, t0);
    cx.sync();
    Fp2 t1 = Z0;
    if (walk) t1 = cx.ld(C6_L0, other);                                    // A: lambda B: theta
    cx.sync();
    const Fp2 r2 = fp2_sqr(t0);                                            // A: c      B: d
    const Fp2 r3 = fp2_mul(t0, isB ? q.y : q.x);                           // A: theta qx           B: lambda qy
    const Fp2 r4 = fp2_mul(isB ? t0 : z, r2);                              // A: f = Z c            B: e = lambda d
    const Fp2 r5 = fp2_mul(isB ? xy : (lines ? pl.cy : t1), isB ? r2 : fp2_from_fp(p.py));           // A: l0 (cy = lambda)   B: g = X d
    if (walk) { cx.st(C6_L0, r4); if (isB) { cx.st(C6_L1, r5); cx.st(C6_L3, r3); } }
    cx.sync();
    Fp2 o0 = Z0, o1 = Z0, o2 = Z0;
    if (walk) { o0 = cx.ld(C6_L0, other); if (!isB) { o1 = cx.ld(C6_L1, other); o2 = cx.ld(C6_L3, other); } }       // A: e, g, lambda qy    B: f
    cx.sync();
    const Fp2 e = isB ? r4 : o0, f = isB ? o0 : r4, g = isB ? r5 : o1;
    const Fp2 h = fp2_sub(fp2_add(e, f), fp2_dbl(g));
    const Fp2 r6 = fp2_mul(t0, isB ? h : fp2_sub(g, h));                   // A: theta (g - h)      B: X3 = lambda h
    const Fp2 r7 = fp2_mul(isB ? z : e, isB ? e : xy);                     // A: e Y                B: Z3 = Z e
    const Fp2 r8 = fp2_mul(lines ? pl.cx : fp2_neg(t0), fp2_from_fp(p.px));                          // A: l1 (cx = -theta)
    if (walk && isB) { acc.st_tc(j, 0, r6); acc.st_tc(j, 2, r7); }
    if (walk && !isB) { acc.st_tc(j, 1, fp2_sub(r6, r7)); l3 = fp2_sub(r3, o2); }
    l0 = r5; l1 = r8;
  }
  if ((walk || lines) && !isB) { cx.st(C6_L0, l0); cx.st(C6_L1, l1); cx.st(C6_L3, l3); }
  cx.sync();
}

// ---- Miller loop, any number of pairs of ONE group on one accumulator: the same value as miller_loop_multi (pairing.h), the same
// ACC interface.  The group's pairs are taken three at a time: pair base + p belongs to lanes 2p and 2p + 1, which walk its G2 point
// together (c6_pair_step) or fetch its prepared line and scale it by P -- then the group multiplies the accumulator by the three lines
// one after the other.  cmax: the largest pair count among the groups that run in lockstep with this one (device: of the wave; host: count()).
template <class CX, class ACC>
RB_FN Fp2 c6_miller_loop_multi(CX cx, ACC acc, int cmax) {
  const int n = acc.count(), k = cx.role();
  c6_put_f(cx, k == 0 ? fp2_one() : fp2_zero());
  for (int j = k; j < n; j += 6) {
    if (acc.kind(j) == MP_WALK) {
      const G2Aff q = acc.q(j);
      acc.st_tc(j, 0, q.x); acc.st_tc(j, 1, q.y); acc.st_tc(j, 2, fp2_one());
    }
  }
  cx.sync();
  int i = RB_ATE_NAF_LEN - 2;
  bool add_pending = false;
#pragma unroll 1
  for (int ln = 0; ln < RB_MILLER_LINES; ln++) {
    int mode;
    if (i >= 0) {
      const bool pos = (i < 64) && ((RB_ATE_NAF_POS >> i) & 1ull);
      const bool ngt = (i < 64) && ((RB_ATE_NAF_NEG >> i) & 1ull);
      if (!add_pending) {
        c6_sqr(cx);
        mode = MS_DBL;
        if (pos | ngt) add_pending = true; else i--;
      } else {
        mode = pos ? MS_ADD_POS : MS_ADD_NEG;
        add_pending = false;
        i--;
      }
    } else {
      mode = (i == -1) ? MS_FROB1 : MS_FROB2;
      i--;
    }
#pragma unroll 1
    for (int base = 0; base < cmax; base += 3) {
      const int j = base + (k >> 1);
      const bool have = j < n;
      const int kind = have ? acc.kind(j) : MP_SKIP;
      c6_pair_step(cx, acc, j, have, kind, mode, ln);
      const int m = cmax - base < 3 ? cmax - base : 3;
#pragma unroll 1
      for (int jj = 0; jj < m; jj++) {
        const bool active = base + jj < n && acc.kind(base + jj) != MP_SKIP;       // the same for the six lanes of a group
        const Fp2 r = c6_dot(cx, C6_OP_LINE, 2 * jj);
        if (active) c6_put_f(cx, r);
      }
      cx.sync();
    }
  }
  return c6_mine(cx);
}

// ---- final exponentiation of the value in the accumulator rows: the chain of final_exponentiation (pairing.h), value by value
// f^u over the width-3 NAF of u (pairing.h: wsx_exp_u); src: the lanes' coefficients of f.  Leaves f^u in the accumulator rows.
template <class CX> RB_FN Fp2 c6_exp_u(CX cx, const Fp2& src) {
  constexpr signed char SQ[RB_U_WNAF_STEPS] = RB_U_WNAF_SQ;
  constexpr signed char DG[RB_U_WNAF_STEPS] = RB_U_WNAF_DG;
  c6_put_f(cx, src);
  c6_csqr(cx);
  c6_put(cx, C6_B, src);
  c6_mul_b(cx);
  const Fp2 cube = c6_mine(cx);                                        // f^3 = f^2 * f
  if (RB_U_WNAF_TOP != 3) c6_put_f(cx, src);
#pragma unroll 1
  for (int i = 0; i < RB_U_WNAF_STEPS; i++) {
    const int d = DG[i];
#pragma unroll 1
    for (int q = 0; q < SQ[i]; q++) c6_csqr(cx);
    const Fp2 m = (d == 1 || d == -1) ? src : cube;
    c6_put(cx, C6_B, d < 0 ? c6_conj(cx, m) : m);
    c6_mul_b(cx);
  }
#pragma unroll 1
  for (int q = 0; q < RB_U_WNAF_TAIL; q++) c6_csqr(cx);
  return c6_mine(cx);
}
// 1 / N for N = a_0 + a_2 w^2 + a_4 w^4 in Fq6 (the norm of an Fq12 value): lane 0 inverts (fp6_inv, tower.h) and leaves the three
// coefficients in its slots of rows L0 / L1 / L3 -- where LINE024 reads a sparse multiplier.  n_k: the lanes' coefficients of N.
template <class CX> RB_FN void c6_inv6_to_line0(CX cx, const Fp2& n_k) {
  c6_put(cx, C6_L3, n_k);
  Fp6 ni = fp6_zero();
  if (cx.role() == 0) ni = fp6_inv(Fp6{cx.ld(C6_L3, 0), cx.ld(C6_L3, 2), cx.ld(C6_L3, 4)});
  cx.sync();
  if (cx.role() == 0) { cx.st(C6_L0, ni.a0); cx.st(C6_L1, ni.a1); cx.st(C6_L3, ni.a2); }
  cx.sync();
}
template <class CX> RB_FN Fp2 c6_final_exponentiation(CX cx, const Fp2& f_in) {
  // easy part: f^(p^6-1) = conj(f)^2 / N, N = f conj(f) in Fq6
  const Fp2 fc = c6_conj(cx, f_in);
  const Fp2 nrm = c6_mul(cx, f_in, fc);
  c6_inv6_to_line0(cx, nrm);
  c6_put_f(cx, fc);
  c6_put_f(cx, c6_dot(cx, C6_OP_LINE024, 0));                           // conj(f) / N = 1 / f
  c6_put(cx, C6_B, fc);
  c6_mul_b(cx);
  const Fp2 f1 = c6_mine(cx);
  const Fp2 f = c6_mul(cx, c6_frob(cx, f1, 2), f1);                     // ^(p^2+1)
  // hard part (libff / zcash-bn last chunk; names as in final_exponentiation)
  const Fp2 a = c6_conj(cx, c6_exp_u(cx, f));
  c6_put_f(cx, a);
  c6_csqr(cx);
  const Fp2 b = c6_mine(cx);
  c6_csqr(cx);
  c6_put(cx, C6_B, b);
  c6_mul_b(cx);
  const Fp2 d = c6_mine(cx);                                           // d = b^2 * b
  const Fp2 e = c6_conj(cx, c6_exp_u(cx, d));
  c6_put_f(cx, e);
  c6_csqr(cx);
  const Fp2 g = c6_conj(cx, c6_exp_u(cx, c6_mine(cx)));
  const Fp2 kk = c6_mul(cx, c6_mul(cx, c6_conj(cx, g), e), c6_conj(cx, d));
  const Fp2 l = c6_mul(cx, kk, b);
  const Fp2 nn = c6_mul(cx, c6_mul(cx, kk, e), f);
  const Fp2 r = c6_mul(cx, c6_frob(cx, kk, 2), c6_mul(cx, c6_frob(cx, l, 1), nn));
  const Fp2 t = c6_mul(cx, c6_conj(cx, f), l);
  return c6_mul(cx, c6_frob(cx, t, 3), r);
}

// ---- membership in Gt, the order-r subgroup of Fq12* (engine_jobs.hip: k_gt_is_member, mode 0, value by value): the cyclotomic test
// f^(p^4) f = f^(p^2), then the BN order test f^p = f^(6 u^2).  ok_k: what the lane already knows about its own coefficient (canonical
// words).  CX additionally provides  bool all(bool) const  -- true when every lane of the group passes true.
template <class CX> RB_FN bool c6_gt_is_member(CX cx, const Fp2& f, bool ok_k) {
  const bool good = cx.all(ok_k) && !cx.all(fp2_is_zero(f));
  const Fp2 f2 = c6_frob(cx, f, 2), f4 = c6_frob(cx, f2, 2);
  const Fp2 t = c6_mul(cx, f4, f);
  const bool cyc = cx.all(fp2_eq(t, f2));
  const Fp2 h = c6_exp_u(cx, c6_exp_u(cx, f));                       // f^(u^2)
  c6_put_f(cx, h);
  c6_csqr(cx);
  const Fp2 h2 = c6_mine(cx);
  c6_csqr(cx);
  const Fp2 r = c6_mul(cx, h2, c6_mine(cx));                          // h^2 * h^4 = f^(6 u^2)
  const bool ord = cx.all(fp2_eq(r, c6_frob(cx, f, 1)));
  return good && cyc && ord;
}

// tower position (Fq2 index in the 6-coefficient order c0.a0, c0.a1, c0.a2, c1.a0, c1.a1, c1.a2) of the coefficient of w^k
RB_HD int c6_tower_index(int k) { return (k & 1) ? 3 + (k >> 1) : (k >> 1); }
RB_HD Fp2 c6_coeff(const Fp12& f, int k) {
  return k == 0 ? f.c0.a0 : k == 1 ? f.c1.a0 : k == 2 ? f.c0.a1 : k == 3 ? f.c1.a1 : k == 4 ? f.c0.a2 : f.c1.a2;
}

}}  // namespace rabe::bn254
