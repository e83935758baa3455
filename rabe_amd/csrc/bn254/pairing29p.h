// The Miller loop of pairing29.h with ONE (item, chunk) on TWO adjacent lanes of a wave -- f = c0 + c1 w, lane 0 of the pair owns c0, lane 1
// owns c1 -- so that the accumulator's home is 216 B of LDS per lane instead of 432, eight waves fit a CU and every SIMD runs TWO waves
// (round 6).  Same tower, same Costello-Lange-Naehrig steps, same line values and the same event order per pair
// as miller_loop_multi: the Miller value is the same field element and the canonical bytes that leave the kernel are identical
// (tests/test_gpu_rr2.py; the whole GPU suite cross-checks this family on every pairing launch).
//
// What the two lanes share and how (a wave's LDS traffic is served in order, so no barrier is involved anywhere):
//  * squaring (complex): lane 0 computes t = (c0 + c1)(c0 + v c1), lane 1 computes ab = c0 c1 -- ONE Karatsuba Fq6 product per lane, both
//    read both halves from the LDS home -- then the products cross over with one DPP move per dword and lane 0 forms t - ab - v ab,
//    lane 1 forms 2 ab;
//  * a line product f (l0 + l1 w + l3 w^3) is six three-term dot products (pairing29.h: facc_mul_by_line), three per lane: lane 0 the
//    even powers of w (its own coefficients), lane 1 the odd ones, in three lock-step rounds.  The pair has TWO operand slots in LDS and
//    each dot product one operand in registers; the terms are arranged so that two slots suffice:
//        round 0   slots (l0, xi l3)   registers: lane 0  xi l1,   lane 1  l1
//        round 1   slots (l0, l1)      registers: lane 0  xi l3,   lane 1  l3
//        round 2   slots (l0, l1)      registers: both    l3
//    (xi l1 is made by lane 0 and xi l3 by lane 1 in one lock-step multiplication by xi; lane 0 gets xi l3 over DPP);
//  * the G2 steps of the WALKING pairs run one pair per lane, two pairs at a time; a pair's line is broadcast to the partner lane
//    (DPP) when its turn to multiply into the accumulator comes.  A PREPARED pair's line is loaded by both lanes.
//
// `rabe_bn::pairing` call sites of the reference: src/schemes/ac17/mod.rs:415-418, bsw/mod.rs:291-294,308, lsw/mod.rs:275-280,
// aw11/mod.rs:340-350.
#pragma once
#include "pairing29.h"

namespace rabe { namespace bn254 { namespace rr {

// ---- lane-by-lane selection between two values of one type
RB_HD i32x9 sel9(bool c, const i32x9& a, const i32x9& b) {
  i32x9 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r[i] = c ? a[i] : b[i];
  return r;
}
template <int L, int V> RB_HD FB<L, V> sel(bool c, const FB<L, V>& a, const FB<L, V>& b) { return mk<L, V>(sel9(c, a.l, b.l)); }
template <int L, int V> RB_HD F2B<L, V> sel2(bool c, const F2B<L, V>& a, const F2B<L, V>& b) { return mk2(sel(c, a.c0, b.c0), sel(c, a.c1, b.c1)); }

// PA (the pair accumulator) provides:
//   bool hi()                                   this lane is lane 1 of its pair (owns c1)
//   F2 ld_co(int i)                             coefficient i (0..5: c0.a0, c0.a1, c0.a2, c1.a0, c1.a1, c1.a2) of the pair's accumulator
//   void st_own(int i, const F2&)               this lane's coefficient i (0..2)
//   void set_slot(int s, const F2&, bool w)     operand slot s (0, 1) of the pair, written by the lanes where w holds
//   void set_slot0_fp(const F&, bool w)         slot 0 holding an Fq element (the unit-y lines of prepared pairs)
//   F2 dotp(const F2& yr, int is0, int ir, int is1)    f[is0] slot0 + f[ir] yr + f[is1] slot1
//   F2 dotps(const F2& yr, int is0, int ir, int is1)   the same with slot 0 in Fq
//   F2 other2(const F2&)                        the partner lane's value
//   template <int O> F2 from2(const F2&)        lane O's value, in both lanes
// and the argument accessors of the one-lane loop: count, kind, p, q, line_u, ld_t, st_t, begin, fence.
template <class PA> RB_HD void pacc_set_one(PA a) {
  const F2 first = sel2(a.hi(), zero2(), one2());
  a.st_own(0, first); a.st_own(1, zero2()); a.st_own(2, zero2());
  a.fence();
}
// f <- f^2:  ab = c0 c1, t = (c0 + c1)(c0 + v c1);  c0' = t - ab - v ab, c1' = 2 ab  (pairing29.h: facc_sqr)
template <class PA> RB_HD void pacc_sqr(PA a) {
  const bool hi = a.hi();
  F6 p;
  {
    const F6 c0 = mk6(a.ld_co(0), a.ld_co(1), a.ld_co(2)), c1 = mk6(a.ld_co(3), a.ld_co(4), a.ld_co(5));
    // lane 0: (c0 + c1)(c0 + v c1);  lane 1: c0 c1 -- the same Karatsuba product on operands picked lane by lane
    const auto s = norm6(add6(c0, c1));
    typedef decltype(s.a0) S2;
    const auto x = mk6(sel2(hi, S2(c0.a0), s.a0), sel2(hi, S2(c0.a1), s.a1), sel2(hi, S2(c0.a2), s.a2));
    const F2 y0 = add_mul_xi2(c0.a0, c1.a2);
    const auto y1 = norm2(add2(c0.a1, c1.a0)), y2 = norm2(add2(c0.a2, c1.a1));
    typedef decltype(y1) Y2;
    const auto y = mk6(sel2(hi, c1.a0, y0), sel2(hi, Y2(c1.a1), y1), sel2(hi, Y2(c1.a2), y2));
    p = mul6(x, y);
  }
  const F6 q = mk6(a.other2(p.a0), a.other2(p.a1), a.other2(p.a2));          // lane 0: ab, lane 1: t (unused)
  // lane 0: t - ab - v ab;  lane 1: 2 ab
  const F2 r0 = sel2(hi, normf2(dbl2(p.a0)), add_mul_xi2(sub2(p.a0, q.a0), neg2(q.a2)));
  const F2 r1 = sel2(hi, normf2(dbl2(p.a1)), normf2(sub2(sub2(p.a1, q.a1), q.a0)));
  const F2 r2 = sel2(hi, normf2(dbl2(p.a2)), normf2(sub2(sub2(p.a2, q.a2), q.a1)));
  a.fence();
  a.st_own(0, r0); a.st_own(1, r1); a.st_own(2, r2);
  a.fence();
}
// f <- f (l0 + l1 w + l3 w^3) with l0 already in slot 0 of the pair (UNIT: an Fq element, the unit-y line of a prepared pair) and l1, l3 in
// BOTH lanes' registers.  out_k = f_k l0 + f_(k-1) l1 + f_(k-3) l3 over the basis 1, w, .., w^5 (an index below zero wraps with a factor xi);
// coefficient of w^k lives at home index (0, 3, 1, 4, 2, 5)[k]: lane 0 computes k = 0, 2, 4 (its own c0.a0, c0.a1, c0.a2), lane 1 k = 1, 3, 5.
template <bool UNIT, class PA> RB_HD void pacc_mul_by_line(PA a, const F2& l1, const F2& l3) {
  const bool hi = a.hi();
  const F2 x = mul_xi2(sel2(hi, l3, l1));          // lane 0: xi l1, lane 1: xi l3
  const F2 xo = a.other2(x);                       // lane 0: xi l3
  a.set_slot(1, x, hi);                            // slot 1 = xi l3
  //            slot 0 term      register term          slot 1 term
  // lane 0, k = 0:  f0 l0   +   f5 (xi l1)        +    f3 (xi l3)          home indices 0, 5, 4
  // lane 1, k = 1:  f1 l0   +   f0 l1             +    f4 (xi l3)                       3, 0, 2
  const F2 yr0 = sel2(hi, l1, x);
  const F2 o0 = UNIT ? a.dotps(yr0, hi ? 3 : 0, hi ? 0 : 5, hi ? 2 : 4) : a.dotp(yr0, hi ? 3 : 0, hi ? 0 : 5, hi ? 2 : 4);
  a.set_slot(1, l1, hi);                           // slot 1 = l1 (the wave's LDS accesses are served in order: round 0 has read xi l3)
  // lane 0, k = 2:  f2 l0   +   f5 (xi l3)        +    f1 l1                             1, 5, 3
  // lane 1, k = 3:  f3 l0   +   f0 l3             +    f2 l1                             4, 0, 1
  const F2 yr1 = sel2(hi, l3, xo);
  const F2 o1 = UNIT ? a.dotps(yr1, hi ? 4 : 1, hi ? 0 : 5, hi ? 1 : 3) : a.dotp(yr1, hi ? 4 : 1, hi ? 0 : 5, hi ? 1 : 3);
  // lane 0, k = 4:  f4 l0   +   f1 l3             +    f3 l1                             2, 3, 4
  // lane 1, k = 5:  f5 l0   +   f2 l3             +    f4 l1                             5, 1, 2
  const F2 o2 = UNIT ? a.dotps(l3, hi ? 5 : 2, hi ? 1 : 3, hi ? 2 : 4) : a.dotp(l3, hi ? 5 : 2, hi ? 1 : 3, hi ? 2 : 4);
  a.fence();
  a.st_own(0, o0); a.st_own(1, o1); a.st_own(2, o2);
  a.fence();
}

// a line of a walking pair as it meets the accumulator: the y- and x-coefficients already scaled by the G1 argument
struct LineS29 { F2 l0, l1, l3; };
RB_HD LineS29 scale_line(const Line29& l, const MillerP29& p) { return LineS29{mul2_fp(l.cy, p.py), mul2_fp(l.cx, p.px), l.c0}; }

// lane O's line multiplies into the pair's accumulator
template <int O, class PA> RB_HD void pacc_ell_from(PA a, const LineS29& mine) {
  a.set_slot(0, mine.l0, a.hi() == (O == 1));
  const F2 l1 = a.template from2<O>(mine.l1), l3 = a.template from2<O>(mine.l3);
  pacc_mul_by_line<false>(a, l1, l3);
}
// a prepared pair's share of a step (both lanes work on the SAME pair): line ln and, when the step has a second event, line ln + 1
template <class PA> RB_HD void pair_prepared_step(PA a, int j, int second, int ln) {
  const MillerP29 p = a.p(j);
  const LineU29 u1 = a.line_u(j, ln);
  if (second >= 0) {
    const LineU29 u2 = a.line_u(j, ln + 1);
    const F2 s1 = mul2_fp(u1.cx, p.px), s2 = mul2_fp(u2.cx, p.px);
    a.set_slot0_fp(p.py, !a.hi());
    pacc_mul_by_line<true>(a, s1, u1.c0);
    pacc_mul_by_line<true>(a, s2, u2.c0);          // slot 0 still holds y_P
  } else {
    const F2 s1 = mul2_fp(u1.cx, p.px);
    a.set_slot0_fp(p.py, !a.hi());
    pacc_mul_by_line<true>(a, s1, u1.c0);
  }
}
// two walking pairs' share of a step: lane 0 walks pair ja, lane 1 walks pair jb (jb < 0: lane 1 has no pair this time).  EVENT BY EVENT: both
// lanes take the step's first G2 operation on their pair, the two lines multiply into the accumulator (lane 0's, then lane 1's), then the
// same for the second operation -- a wave has 256 registers here, and only ONE line per lane waits for its turn (in the one-lane kernel a
// pair's two lines wait together and the running point is fetched once per step; here it makes a second round trip, which the SIMD's other
// wave covers).
template <class PA> RB_HD LineS29 pair_walk_event(PA a, int me, int ev) {
  LineS29 m;
  m.l0 = m.l1 = m.l3 = zero2();
  if (me >= 0) {
    G2Hom29 t = a.ld_t(me);
    Line29 l;
    if (ev == MS_DBL) l = g2hom_double(t);
    else {
      G2Aff29 q = a.q(me);
      if (ev == MS_ADD_NEG) q.y = neg2(q.y);
      else if (ev == MS_FROB1) q = g2_frob1(q);
      else if (ev == MS_FROB2) q = g2_frob2_neg(q);
      l = g2hom_add(t, q);
    }
    a.st_t(me, t);
    m = scale_line(l, a.p(me));
  }
  return m;
}
template <class PA> RB_HD void pair_walking_step(PA a, int ja, int jb, int first, int second) {
  const int me = a.hi() ? jb : ja;
  {
    const LineS29 m = pair_walk_event(a, me, first);
    pacc_ell_from<0>(a, m);
    if (jb >= 0) pacc_ell_from<1>(a, m);
  }
  if (second >= 0) {
    const LineS29 m = pair_walk_event(a, me, second);
    pacc_ell_from<0>(a, m);
    if (jb >= 0) pacc_ell_from<1>(a, m);
  }
}
template <class PA> RB_HD void pair_steps(PA a, int n, int first, int second, int ln) {
  int ja = -1;
  for (int j = 0; j < n; j++) {
    const int kind = a.kind(j);
    if (kind == MP_SKIP) continue;
    if (kind == MP_LINES) { pair_prepared_step(a, j, second, ln); continue; }
    if (ja < 0) { ja = j; continue; }
    pair_walking_step(a, ja, j, first, second);
    ja = -1;
  }
  if (ja >= 0) pair_walking_step(a, ja, -1, first, second);
}
// The loop: 65 doubling steps (21 of them with an addition), then the two Frobenius additions -- miller_loop_multi's events
template <class PA> RB_MID void miller_loop_pair(PA a) {
  const int n = a.count();
  pacc_set_one(a);
  a.begin();
  int ln = 0;
  for (int i = RB_ATE_NAF_LEN - 2; i >= 0; i--) {
    const bool pos = (i < 64) && ((RB_ATE_NAF_POS >> i) & 1ull);
    const bool ngt = (i < 64) && ((RB_ATE_NAF_NEG >> i) & 1ull);
    const int second = pos ? MS_ADD_POS : ngt ? MS_ADD_NEG : -1;
    pacc_sqr(a);
    pair_steps(a, n, MS_DBL, second, ln);
    ln += (second >= 0) ? 2 : 1;
  }
  pair_steps(a, n, MS_FROB1, MS_FROB2, ln);
}

} } }   // namespace rabe::bn254::rr
