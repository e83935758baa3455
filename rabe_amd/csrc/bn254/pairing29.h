// The Miller loop of pairing.h (miller_loop_multi: any number of pairings on one Fq12 accumulator) on the reduced-radix
// field core of fp29.h -- the same tower (Fq2 = Fp[u]/(u^2+1), Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v)), the same
// Costello-Lange-Naehrig steps, the same sparse line products, formula by formula, so that every value is the same field
// element as in pairing.h and the canonical bytes that leave the kernel are identical (tests/test_hostsim_rr.py,
// tests/test_gpu_rr.py).  What differs is the bookkeeping a redundant representation needs: additions and subtractions are
// limb-wise and unreduced, the bounds travel in the types (fp29.h: FB<L, V>), and a value is normalised where a bound
// would otherwise be exceeded -- the static_asserts of fp29.h decide where, not the author.
//
// `rabe_bn::pairing` call sites of the reference: src/schemes/ac17/mod.rs:415-418, bsw/mod.rs:291-294,308,
// lsw/mod.rs:275-280, aw11/mod.rs:340-350.
#pragma once
#include "fp29.h"
#include "pairing.h"

namespace rabe { namespace bn254 { namespace rr {

// always to F = FB<1, 1> (what is stored): norm() keeps a value bound of 2 where it is given one
template <int L, int V> RB_HD F normf(const FB<L, V>& a) { return norm(FB<(L < 6 ? L : 6), (V < 3 ? 3 : V)>(a)); }
RB_HD F normf(const F& a) { return a; }
template <int L, int V> RB_HD F2 normf2(const F2B<L, V>& a) { return mk2(normf(a.c0), normf(a.c1)); }

// ============================================================================ Fq6: three Fq2 coefficients, each with its own bounds
template <class A0, class A1, class A2>
struct F6T {
  A0 a0; A1 a1; A2 a2;
};
typedef F6T<F2, F2, F2> F6;
template <class A0, class A1, class A2> RB_HD F6T<A0, A1, A2> mk6(const A0& a0, const A1& a1, const A2& a2) { return F6T<A0, A1, A2>{a0, a1, a2}; }
RB_HD F6 zero6() { return mk6(zero2(), zero2(), zero2()); }
RB_HD F6 one6() { return mk6(one2(), zero2(), zero2()); }
template <class A, class B> RB_HD auto add6(const A& a, const B& b) { return mk6(add2(a.a0, b.a0), add2(a.a1, b.a1), add2(a.a2, b.a2)); }
template <class A> RB_HD auto norm6(const A& a) { return mk6(norm2(a.a0), norm2(a.a1), norm2(a.a2)); }

// Karatsuba, 6 Fq2 multiplications; the coefficients of the operands may be weakly normalised sums (FB<1, 2>)
template <class A, class B>
RB_HD F6 mul6(const A& a, const B& b) {
  const F2 v0 = mul2(a.a0, b.a0);
  const F2 v1 = mul2(a.a1, b.a1);
  const F2 v2 = mul2(a.a2, b.a2);
  const F2 t0 = mul2(add2(a.a1, a.a2), add2(b.a1, b.a2));
  const F2 t1 = mul2(add2(a.a0, a.a1), add2(b.a0, b.a1));
  const F2 t2 = mul2(add2(a.a0, a.a2), add2(b.a0, b.a2));
  return mk6(add_mul_xi2(v0, sub2(sub2(t0, v1), v2)), add_mul_xi2(sub2(sub2(t1, v0), v1), v2), normf2(add2(sub2(sub2(t2, v0), v2), v1)));
}
// a (b0 + b1 v): 5 Fq2 multiplications; the last coefficient comes back as the plain sum of two products (FB<2, 2>)
template <class A, class B0, class B1>
RB_HD F6T<F2, F2, F2B<2, 2>> mul6_by_01(const A& a, const B0& b0, const B1& b1) {
  const F2 v0 = mul2(a.a0, b0);
  const F2 v1 = mul2(a.a1, b1);
  const F2 t0 = mul2(add2(a.a1, a.a2), b1);
  const F2 t1 = mul2(add2(a.a0, a.a1), add2(b0, b1));
  const F2 t2 = mul2(a.a2, b0);
  return mk6(add_mul_xi2(v0, sub2(t0, v1)), normf2(sub2(sub2(t1, v0), v1)), add2(t2, v1));
}
template <class A, class B> RB_HD F6 mul6_fp2(const A& a, const B& b) { return mk6(mul2(a.a0, b), mul2(a.a1, b), mul2(a.a2, b)); }

// ============================================================================ the accumulator f = c0 + c1 w in its home
// FA provides  F6 ld_f6(int half) const, void st_f6(int half, const F6&) const, F6 ld_x() const, void st_x(const F6&) const, void fence() const
// (pairing.h: the same interface on the 8 x 32-bit types).  Stored values are F = FB<1, 1>.
template <class FA> RB_HD void facc_set_one(FA a) { a.st_f6(0, one6()); a.st_f6(1, zero6()); }
// r.c0 = t0 + v t1 ; r.c1 = t2 - t0 - t1, t0 parked
template <class FA, class T1, class T2>
RB_HD void facc_finish(FA a, const T1& t1, const T2& t2) {
  const F6 t0 = a.ld_x();
  a.st_f6(0, mk6(add_mul_xi2(t0.a0, t1.a2), normf2(add2(t0.a1, t1.a0)), normf2(add2(t0.a2, t1.a1))));
  a.st_f6(1, mk6(normf2(sub2(sub2(t2.a0, t0.a0), t1.a0)), normf2(sub2(sub2(t2.a1, t0.a1), t1.a1)), normf2(sub2(sub2(t2.a2, t0.a2), t1.a2))));
}
// complex squaring: ab = c0 c1, t = (c0 + c1)(c0 + v c1);  c0' = t - ab - v ab, c1' = 2 ab
template <class FA> RB_HD void facc_sqr(FA a) {
  { const F6 ab = mul6(a.ld_f6(0), a.ld_f6(1)); a.st_x(ab); }
  a.fence();
  F6 t;
  {
    const F6 c0 = a.ld_f6(0), c1 = a.ld_f6(1);
    t = mul6(norm6(add6(c0, c1)), mk6(add_mul_xi2(c0.a0, c1.a2), norm2(add2(c0.a1, c1.a0)), norm2(add2(c0.a2, c1.a1))));
  }
  a.fence();
  const F6 ab = a.ld_x();
  a.st_f6(0, mk6(add_mul_xi2(sub2(t.a0, ab.a0), neg2(ab.a2)), normf2(sub2(sub2(t.a1, ab.a1), ab.a0)), normf2(sub2(sub2(t.a2, ab.a2), ab.a1))));
  a.st_f6(1, mk6(normf2(dbl2(ab.a0)), normf2(dbl2(ab.a1)), normf2(dbl2(ab.a2))));
}
// f (l0 + l1 w + l3 w^3) coefficient by coefficient over the basis 1, w, ..., w^5 (w^6 = xi):
//   out_k = f_k l0 + f_(k-1) l1 + f_(k-3) l3,  an index below zero wraps to +6 with a factor xi
// -- six dot products of three Fq2 products each, ONE pair of reductions per coefficient and no Karatsuba bookkeeping: 18 products + 6
// reduction pairs instead of 13 products + 13 reduction pairs + ~2.5 k instructions of sums and normalisations (11.5 + 11.5 when two
// lines are merged first), and a third of the calls.  The same field elements as fp12_mul_by_line (tower.h).
// FA provides, beside the interface above:  void set_y(int slot, const F2&)  (slot 1, 2: the second and third right-hand operand of the dot
// products that follow) and  F2 dot3(const F2& y0, int ia, int ib, int ic)  =  f[ia] y0 + f[ib] y[1] + f[ic] y[2]  with f[i] = coefficient i
// of the accumulator in the order c0.a0, c0.a1, c0.a2, c1.a0, c1.a1, c1.a2, and  void st_f2(int i, const F2&).
enum { W0 = 0, W1 = 3, W2 = 1, W3 = 4, W4 = 2, W5 = 5 };          // home index of the coefficient of w^k
template <class FA> RB_HD void facc_mul_by_line(FA a, const F2& l0, const F2& l1, const F2& l3) {
  a.set_y(1, mul_xi2(l1));
  a.set_y(2, mul_xi2(l3));
  const F2 o0 = a.dot3(l0, W0, W5, W3);
  a.set_y(1, l1);
  const F2 o1 = a.dot3(l0, W1, W0, W4);
  const F2 o2 = a.dot3(l0, W2, W1, W5);
  a.set_y(2, l3);
  const F2 o3 = a.dot3(l0, W3, W2, W0);
  const F2 o4 = a.dot3(l0, W4, W3, W1);
  const F2 o5 = a.dot3(l0, W5, W4, W2);
  a.fence();
  a.st_f2(W0, o0); a.st_f2(W1, o1); a.st_f2(W2, o2); a.st_f2(W3, o3); a.st_f2(W4, o4); a.st_f2(W5, o5);
  a.fence();
}

// The same product for a line with a UNIT y-coefficient, l = s + l1 w + l3 w^3 with s = y_P in Fq: the first term of every dot product is
// an Fq2 x Fq product (ten schoolbook products per coefficient instead of twelve).  FA provides  F2 dot3s(const F& s, int ia, int ib, int ic).
template <class FA> RB_HD void facc_mul_by_line_s(FA a, const F& s, const F2& l1, const F2& l3) {
  a.set_y(1, mul_xi2(l1));
  a.set_y(2, mul_xi2(l3));
  const F2 o0 = a.dot3s(s, W0, W5, W3);
  a.set_y(1, l1);
  const F2 o1 = a.dot3s(s, W1, W0, W4);
  const F2 o2 = a.dot3s(s, W2, W1, W5);
  a.set_y(2, l3);
  const F2 o3 = a.dot3s(s, W3, W2, W0);
  const F2 o4 = a.dot3s(s, W4, W3, W1);
  const F2 o5 = a.dot3s(s, W5, W4, W2);
  a.fence();
  a.st_f2(W0, o0); a.st_f2(W1, o1); a.st_f2(W2, o2); a.st_f2(W3, o3); a.st_f2(W4, o4); a.st_f2(W5, o5);
  a.fence();
}

// ============================================================================ G2 steps (pairing.h: g2hom_double / g2hom_add)
struct G2Hom29 { F2 x, y, z; };
struct G2Aff29 { F2 x, y; };
struct Line29 { F2B<3, 3> cy, cx; F2 c0; };          // cy, cx are scaled by y_P, x_P before they meet the accumulator
// A PREPARED line, divided by its y-coefficient when the key's lines were converted (cx / cy, c0 / cy): the factor 1 / cy lies in Fq2, a
// proper subfield of Fq12, so the final exponentiation removes it -- the pairing is the same element, the Miller value is not.
struct LineU29 { F2 cx, c0; };
struct MillerP29 { F px, py; };

RB_MID Line29 g2hom_double(G2Hom29& r) {
  const auto a = half2(mul2(r.x, r.y));                        // X Y / 2                     FB<2, 1>
  const F2 b = sqr2(r.y);
  const F2 c = sqr2(r.z);
  const F2 e = mul2(twist_b(), tpl2(c));                       // 3 b' Z^2
  const auto f = tpl2(e);                                      // 9 b' Z^2                    FB<3, 3>
  const F2 g = normf2(half2(add2(b, f)));
  const auto h = sub2(sqr2(add2(r.y, r.z)), add2(b, c));       // 2 Y Z                       FB<3, 3>
  const F2 i = normf2(sub2(e, b));
  const F2 j = sqr2(r.x);
  const F2 e2 = sqr2(e);
  r.x = mul2(a, normf2(sub2(b, f)));
  r.y = normf2(sub2(sqr2(g), tpl2(e2)));
  r.z = mul2(b, h);
  Line29 l;
  l.cy = neg2(h);
  l.cx = tpl2(j);
  l.c0 = i;
  return l;
}
RB_MID Line29 g2hom_add(G2Hom29& r, const G2Aff29& q) {
  const auto theta = sub2(r.y, mul2(q.y, r.z));                // FB<2, 2>
  const auto lambda = sub2(r.x, mul2(q.x, r.z));
  const F2 c = sqr2(theta);
  const F2 d = sqr2(lambda);
  const F2 e = mul2(lambda, d);
  const F2 f = mul2(r.z, c);
  const F2 g = mul2(r.x, d);
  const F2 h = normf2(sub2(add2(e, f), dbl2(g)));
  const F2 ry = r.y;
  r.x = mul2(lambda, h);
  r.y = normf2(sub2(mul2(theta, sub2(g, h)), mul2(e, ry)));
  r.z = mul2(r.z, e);
  Line29 l;
  l.cy = lambda;
  l.cx = neg2(theta);
  l.c0 = normf2(sub2(mul2(theta, q.x), mul2(lambda, q.y)));
  return l;
}
RB_HD G2Aff29 g2_frob1(const G2Aff29& q) { return G2Aff29{mul2(conj2(q.x), gamma1_2()), mul2(conj2(q.y), gamma1_3())}; }
RB_HD G2Aff29 g2_frob2_neg(const G2Aff29& q) { return G2Aff29{mul2_fp(q.x, gamma2_2()), mul2_fp(neg2(q.y), gamma2_3())}; }

template <class FA> RB_HD void facc_ell(FA a, const Line29& l, const MillerP29& p) {
  facc_mul_by_line(a, mul2_fp(l.cy, p.py), mul2_fp(l.cx, p.px), l.c0);
}

template <class FA> RB_HD void facc_ell_u(FA a, const LineU29& l, const MillerP29& p) {
  facc_mul_by_line_s(a, p.py, mul2_fp(l.cx, p.px), l.c0);
}

// ============================================================================ the loop (pairing.h: miller_loop_multi, same event order)
// ACC provides, beside the FA interface:  int count(), int kind(int j) (MP_WALK / MP_LINES / MP_SKIP), MillerP29 p(int j),
// G2Aff29 q(int j), LineU29 line_u(int j, int n) (line n of prepared pair j, unit y-coefficient), G2Hom29 ld_t(int j), void st_t(int j, const G2Hom29&), void begin()  (called once before
// the loop: converts the lane's arguments)
// One pair's share of a loop step: its line of the FIRST event (the doubling, or the first Frobenius addition) and, when the step has
// one, of the SECOND (the addition of a non-zero digit, the second Frobenius addition) -- with the running point fetched and stored
// ONCE for both and every argument requested in one batch: the lone wave of a SIMD pays each memory round trip in full (~3 k cycles on
// the device, and a call boundary waits for all of them), so a step makes one instead of two or three per pair.  The lines multiply into
// the accumulator pair by pair instead of event by event -- the same product.  (Written out per case rather than as a loop over the
// step's events: the loop form keeps the running point live across a line product and spills it -- measured 268 against 259 ms on
// config 3's launch set.)
// RB_MILLER_PROF (a diagnostic build, tools/prof_miller.sh): shader cycles per region of the loop, summed per lane in registers and written
// by the kernel -- 0 squaring, 1 prepared pair: loads + scaling, 2 prepared pair: line products, 3 walking pair: loads + G2 step + store,
// 4 walking pair: line products
#if defined(RB_MILLER_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define RB_PROF_DECL unsigned long long rb_prof_t_ = clock64()
#define RB_PROF_MARK(k) do { const unsigned long long n_ = clock64(); acc.prof[k] += n_ - rb_prof_t_; rb_prof_t_ = n_; } while (0)
#else
#define RB_PROF_DECL ((void)0)
#define RB_PROF_MARK(k) ((void)0)
#endif
// (inlined into the loop: out of line -- RB_FN -- each kind of step gets a register allocation of its own, but the calling convention's
// callee-saved registers turn into scratch spills: 177 scratch instructions in the walking step, 42.7 M against 39.5 M cycles per wave)
#ifndef RB_STEP_FN
#define RB_STEP_FN RB_HD
#endif
template <class ACC>
RB_STEP_FN void miller_prepared_step(ACC acc, int j, int second, int ln) {
  const MillerP29 p = acc.p(j);
  RB_PROF_DECL;
  const LineU29 l1 = acc.line_u(j, ln);
  if (second >= 0) {
    const LineU29 l2 = acc.line_u(j, ln + 1);
#if defined(RB_MILLER_PROF) && defined(__HIP_DEVICE_COMPILE__)
    const F2 s1 = mul2_fp(l1.cx, p.px), s2 = mul2_fp(l2.cx, p.px);
    RB_PROF_MARK(1);
    facc_mul_by_line_s(acc, p.py, s1, l1.c0);
    facc_mul_by_line_s(acc, p.py, s2, l2.c0);
    RB_PROF_MARK(2);
#else
    facc_ell_u(acc, l1, p);
    facc_ell_u(acc, l2, p);
#endif
  } else {
#if defined(RB_MILLER_PROF) && defined(__HIP_DEVICE_COMPILE__)
    const F2 s1 = mul2_fp(l1.cx, p.px);
    RB_PROF_MARK(1);
    facc_mul_by_line_s(acc, p.py, s1, l1.c0);
    RB_PROF_MARK(2);
#else
    facc_ell_u(acc, l1, p);
#endif
  }
}
template <class ACC>
RB_STEP_FN void miller_walking_step(ACC acc, int j, int first, int second) {
  const MillerP29 p = acc.p(j);
  RB_PROF_DECL;
  G2Hom29 t = acc.ld_t(j);
  G2Aff29 q;
  if (second >= 0 || first != MS_DBL) q = acc.q(j);
  Line29 l1, l2;
  if (first == MS_DBL) l1 = g2hom_double(t);
  else l1 = g2hom_add(t, g2_frob1(q));
  if (second >= 0) {
    G2Aff29 q2 = q;
    if (second == MS_ADD_NEG) q2.y = neg2(q.y);
    else if (second == MS_FROB2) q2 = g2_frob2_neg(q);
    l2 = g2hom_add(t, q2);
  }
  acc.st_t(j, t);
  RB_PROF_MARK(3);
  facc_ell(acc, l1, p);
  if (second >= 0) facc_ell(acc, l2, p);
  RB_PROF_MARK(4);
}
template <class ACC>
RB_HD void miller_pair_step(ACC acc, int j, int first, int second, int ln) {
  const int kind = acc.kind(j);          // (the device accessor keeps the kinds in registers: nothing to wait for)
  if (kind == MP_SKIP) return;
  if (kind == MP_LINES) miller_prepared_step(acc, j, second, ln);
  else miller_walking_step(acc, j, first, second);
}
// The loop: 65 doubling steps (21 of them with an addition), then the two Frobenius additions -- the line events of pairing.h's
// miller_loop_multi in the same order per pair, prepared lines numbered the same way.
template <class ACC>
RB_MID void miller_loop_multi(ACC acc) {
  const int n = acc.count();
  facc_set_one(acc);
  acc.begin();
  int ln = 0;
  for (int i = RB_ATE_NAF_LEN - 2; i >= 0; i--) {
    const bool pos = (i < 64) && ((RB_ATE_NAF_POS >> i) & 1ull);
    const bool ngt = (i < 64) && ((RB_ATE_NAF_NEG >> i) & 1ull);
    const int second = pos ? MS_ADD_POS : ngt ? MS_ADD_NEG : -1;
    { RB_PROF_DECL; facc_sqr(acc); RB_PROF_MARK(0); }
    for (int j = 0; j < n; j++) miller_pair_step(acc, j, MS_DBL, second, ln);
    ln += (second >= 0) ? 2 : 1;
  }
  for (int j = 0; j < n; j++) miller_pair_step(acc, j, MS_FROB1, MS_FROB2, ln);
}

// ============================================================================ final exponentiation (pairing.h: final_exponentiation_ws)
// The same chain value by value -- easy part, then the hard part's three exponentiations by u over the width-3 NAF with Granger-Scott
// squarings -- on values that live in numbered workspace slots; only the representation of the field elements differs.
struct F12 { F6 c0, c1; };
RB_HD F6 neg6(const F6& a) { return mk6(neg2(a.a0), neg2(a.a1), neg2(a.a2)); }
RB_HD F12 from_fp12(const Fp12& x) {
  return F12{mk6(from_fp2(x.c0.a0), from_fp2(x.c0.a1), from_fp2(x.c0.a2)), mk6(from_fp2(x.c1.a0), from_fp2(x.c1.a1), from_fp2(x.c1.a2))};
}
RB_HD Fp12 to_fp12(const F12& x) {
  return Fp12{Fp6{to_fp2(x.c0.a0), to_fp2(x.c0.a1), to_fp2(x.c0.a2)}, Fp6{to_fp2(x.c1.a0), to_fp2(x.c1.a1), to_fp2(x.c1.a2)}};
}
// f *= y for a general y; Y provides  F6 half(int h) const
template <class FA, class Y> RB_HD void facc_mul(FA a, Y y) {
  { const F6 t0 = mul6(a.ld_f6(0), y.half(0)); a.st_x(t0); }
  a.fence();
  const F6 t1 = mul6(a.ld_f6(1), y.half(1));
  a.fence();
  const F6 t2 = mul6(norm6(add6(a.ld_f6(0), a.ld_f6(1))), norm6(add6(y.half(0), y.half(1))));
  a.fence();
  rr::facc_finish(a, t1, t2);
  a.fence();
}
// Granger-Scott squaring (tower.h: fp12_cyclotomic_sqr)
RB_HD void fp4_sqr(F2& r0, F2& r1, const F2& a, const F2& b) {
  const F2 t0 = sqr2(a);
  const F2 t1 = sqr2(b);
  r0 = add_mul_xi2(t0, t1);
  r1 = normf2(sub2(sub2(sqr2(add2(a, b)), t0), t1));
}
RB_MID F12 cyclotomic_sqr(const F12& f) {
  const F2 z0 = f.c0.a0, z4 = f.c0.a1, z3 = f.c0.a2, z2 = f.c1.a0, z1 = f.c1.a1, z5 = f.c1.a2;
  F2 t0, t1, t2, t3, t4, t5;
  fp4_sqr(t0, t1, z0, z1);
  fp4_sqr(t2, t3, z2, z3);
  fp4_sqr(t4, t5, z4, z5);
  F12 r;
  r.c0.a0 = normf2(add2(dbl2(sub2(t0, z0)), t0));
  r.c1.a1 = normf2(add2(dbl2(add2(t1, z1)), t1));
  const F2 x5 = mul_xi2(t5);
  r.c1.a0 = normf2(add2(dbl2(add2(x5, z2)), x5));
  r.c0.a2 = normf2(add2(dbl2(sub2(t4, z3)), t4));
  r.c0.a1 = normf2(add2(dbl2(sub2(t2, z4)), t2));
  r.c1.a2 = normf2(add2(dbl2(add2(t3, z5)), t3));
  return r;
}
RB_MID F12 frob(const F12& a, int k) {
  F12 r;
  if (k == 2) {
    r.c0 = mk6(a.c0.a0, mul2_fp(a.c0.a1, gamma2_2()), mul2_fp(a.c0.a2, gamma2_4()));
    r.c1 = mk6(mul2_fp(a.c1.a0, gamma2_1()), mul2_fp(a.c1.a1, gamma2_3()), mul2_fp(a.c1.a2, gamma2_5()));
  } else if (k == 1) {
    r.c0 = mk6(conj2(a.c0.a0), mul2(conj2(a.c0.a1), gamma1_2()), mul2(conj2(a.c0.a2), gamma1_4()));
    r.c1 = mk6(mul2(conj2(a.c1.a0), gamma1_1()), mul2(conj2(a.c1.a1), gamma1_3()), mul2(conj2(a.c1.a2), gamma1_5()));
  } else {
    r.c0 = mk6(conj2(a.c0.a0), mul2(conj2(a.c0.a1), gamma3_2()), mul2(conj2(a.c0.a2), gamma3_4()));
    r.c1 = mk6(mul2(conj2(a.c1.a0), gamma3_1()), mul2(conj2(a.c1.a1), gamma3_3()), mul2(conj2(a.c1.a2), gamma3_5()));
  }
  return r;
}
// WS provides  F12 ld(int slot) const, void st(int slot, const F12&) const, F6 ld6(int slot, int half) const, void st6(int slot, int half, const F6&) const
// and home() -- an FA for the value being worked on
template <class WS> struct WsOperand29 {
  WS ws; int slot; bool conj;
  RB_HD F6 half(int h) const { const F6 v = ws.ld6(slot, h); return (conj && h == 1) ? neg6(v) : v; }
};
template <class WS> RB_HD void wsx_to_home(WS ws, int a, bool conj) {
  auto h = ws.home();
  const WsOperand29<WS> y{ws, a, conj};
  h.st_f6(0, y.half(0)); h.st_f6(1, y.half(1)); h.fence();
}
template <class WS> RB_HD void wsx_from_home(WS ws, int dst) {
  auto h = ws.home();
  ws.st6(dst, 0, h.ld_f6(0));
  ws.st6(dst, 1, h.ld_f6(1));
  h.fence();
}
template <class WS> RB_FN void wsx_mul(WS ws, int dst, int a, bool conj_a, int b, bool conj_b) {
  rr::wsx_to_home(ws, a, conj_a);
  rr::facc_mul(ws.home(), WsOperand29<WS>{ws, b, conj_b});
  rr::wsx_from_home(ws, dst);
}
template <class WS> RB_FN void wsx_csqr(WS ws, int dst, int a, bool conj_a) {
  F12 x = ws.ld(a);
  if (conj_a) x.c1 = neg6(x.c1);
  ws.st(dst, cyclotomic_sqr(x));
}
template <class WS> RB_FN void wsx_frob_mul(WS ws, int dst, int a, int k, int b) {          // dst = a^(p^k) * b
  {
    const F12 x = frob(ws.ld(a), k);
    auto h = ws.home();
    h.st_f6(0, x.c0);
    h.st_f6(1, x.c1);
    h.fence();
  }
  rr::facc_mul(ws.home(), WsOperand29<WS>{ws, b, false});
  rr::wsx_from_home(ws, dst);
}
// the one inversion of the chain (a twelfth of a percent of its multiplications are inside the Fp inversion's addition chain) goes through
// the 8 x 32-bit core's fp12_inv: converted in, inverted, converted back
template <class WS> RB_FN void wsx_inv(WS ws, int dst, int a) { ws.st(dst, from_fp12(fp12_inv(to_fp12(ws.ld(a))))); }
template <class WS> RB_MID void wsx_sqrn_mul(WS ws, int dst, int a, int n, int b, bool conj_b) {
  F12 x = ws.ld(a);
#pragma unroll 1
  for (int i = 0; i < n; i++) x = cyclotomic_sqr(x);
  if (b < 0) { ws.st(dst, x); return; }
  auto h = ws.home();
  h.st_f6(0, x.c0);
  h.st_f6(1, x.c1);
  h.fence();
  rr::facc_mul(h, WsOperand29<WS>{ws, b, conj_b});
  rr::wsx_from_home(ws, dst);
}
template <class WS> RB_FN void wsx_exp_u(WS ws, int dst, int src, int cube) {
  constexpr signed char SQ[RB_U_WNAF_STEPS] = RB_U_WNAF_SQ;
  constexpr signed char DG[RB_U_WNAF_STEPS] = RB_U_WNAF_DG;
  rr::wsx_sqrn_mul(ws, cube, src, 1, src, false);               // f^3 = f^2 * f
  int cur = (RB_U_WNAF_TOP == 3) ? cube : src;
  for (int i = 0; i < RB_U_WNAF_STEPS; i++) {
    const int d = DG[i];
    rr::wsx_sqrn_mul(ws, dst, cur, SQ[i], (d == 1 || d == -1) ? src : cube, d < 0);
    cur = dst;
  }
  if (RB_U_WNAF_TAIL) rr::wsx_sqrn_mul(ws, dst, cur, RB_U_WNAF_TAIL, -1, false);
}
// in: slot FE_T0 = the Miller value; out: slot FE_T1 (pairing.h: final_exponentiation_ws, step by step)
template <class WS> RB_FN void final_exponentiation_ws(WS ws) {
  rr::wsx_inv(ws, FE_T1, FE_T0);
  rr::wsx_mul(ws, FE_T1, FE_T0, true, FE_T1, false);       // f^(p^6-1) = conj(f) * f^-1
  rr::wsx_frob_mul(ws, FE_F, FE_T1, 2, FE_T1);             // ^(p^2+1)                                   F
  rr::wsx_exp_u(ws, FE_T0, FE_F, FE_T1);                   // f^u        (a = conj of it)
  rr::wsx_csqr(ws, FE_B, FE_T0, true);                     // b = a^2                                    B
  rr::wsx_csqr(ws, FE_T0, FE_B, false);                    // c = b^2
  rr::wsx_mul(ws, FE_D, FE_T0, false, FE_B, false);        // d = c*b                                    D
  rr::wsx_exp_u(ws, FE_E, FE_D, FE_T1);                    // d^u        (e = conj of it)
  rr::wsx_csqr(ws, FE_T0, FE_E, true);                     // f' = e^2
  rr::wsx_exp_u(ws, FE_T1, FE_T0, FE_K);                   // f'^u = conj(g) = i
  rr::wsx_mul(ws, FE_T0, FE_T1, false, FE_E, true);        // j = i*e
  rr::wsx_mul(ws, FE_K, FE_T0, false, FE_D, true);         // k = j*h, h = d^-1                          K
  rr::wsx_mul(ws, FE_L, FE_K, false, FE_B, false);         // l = k*b                                    L
  rr::wsx_mul(ws, FE_T0, FE_K, false, FE_E, true);         // m = k*e
  rr::wsx_mul(ws, FE_T0, FE_T0, false, FE_F, false);       // n = m*f
  rr::wsx_frob_mul(ws, FE_T1, FE_L, 1, FE_T0);             // p = l^p * n
  rr::wsx_frob_mul(ws, FE_T0, FE_K, 2, FE_T1);             // r = k^(p^2) * p
  rr::wsx_mul(ws, FE_T1, FE_F, true, FE_L, false);         // t = f^-1 * l
  rr::wsx_frob_mul(ws, FE_T1, FE_T1, 3, FE_T0);            // v = t^(p^3) * r
}

} } }   // namespace rabe::bn254::rr
