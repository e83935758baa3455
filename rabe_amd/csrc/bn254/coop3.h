// Three-lane cooperative Fq12 / G2 arithmetic for the pairing (wavefront-level parallelism inside ONE pairing).
//
// At the batch sizes the reference's callers produce (BASELINE config 2: 4096 items = 24 576 Miller loops = 384
// waves on 1 024 SIMDs) one-lane-per-pairing leaves most of the chip idle and every lane runs a ~10 k Fp-mul
// dependency chain.  Here a TRIPLE of adjacent lanes works on one pairing: the state (f, T, Q, P) is replicated
// in the three lanes, every Karatsuba-shaped step is split into three independent partial products -- lane L of
// the triple computes part L -- and the parts are all-gathered with ds_bpermute.  The split is work-neutral for
// the Fq12 operations (3 Fq6 products per Fq12 product, 3 Fq4 squarings per cyclotomic squaring, 3 sparse
// products per line multiplication) and close to it for the G2 steps.
//
// Every cooperative operation is written as   part_L = <op>_part(L, inputs)   (pure, same instruction stream
// for all lanes, operands chosen with selects)   +   gather   +   <op>_combine(part_0, part_1, part_2)   (pure,
// replicated).  The host build exercises part/combine with L = 0,1,2 (tests/hostsim); only the gather is
// device-specific.
#pragma once
#include "pairing.h"

namespace rabe { namespace bn254 {

// ---- lane-uniform selects (v_cndmask per limb; no divergence)
RB_HD Fp sel3(int L, const Fp& a, const Fp& b, const Fp& c) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = (L == 0) ? a.v[i] : ((L == 1) ? b.v[i] : c.v[i]);
  return r;
}
RB_HD Fp2 sel3(int L, const Fp2& a, const Fp2& b, const Fp2& c) { return Fp2{sel3(L, a.c0, b.c0, c.c0), sel3(L, a.c1, b.c1, c.c1)}; }
RB_HD Fp6 sel3(int L, const Fp6& a, const Fp6& b, const Fp6& c) {
  return Fp6{sel3(L, a.a0, b.a0, c.a0), sel3(L, a.a1, b.a1, c.a1), sel3(L, a.a2, b.a2, c.a2)};
}

// ---- Fq12 = Fq6[w]/(w^2 - v): (a0 + a1 w)(b0 + b1 w) = (a0b0 + v a1b1) + ((a0+a1)(b0+b1) - a0b0 - a1b1) w
RB_MID Fp12 c3_combine_karatsuba(const Fp6& t0, const Fp6& t1, const Fp6& t2) {
  Fp12 r;
  r.c0 = fp6_add(t0, fp6_mul_v(t1));
  r.c1 = fp6_sub(fp6_sub(t2, t0), t1);
  return r;
}
RB_MID Fp6 c3_mul_part(int L, const Fp12& a, const Fp12& b) {
  Fp6 x = sel3(L, a.c0, a.c1, fp6_add(a.c0, a.c1));
  Fp6 y = sel3(L, b.c0, b.c1, fp6_add(b.c0, b.c1));
  return fp6_mul(x, y);
}
// squaring: a0^2, a1^2, (a0+a1)^2
RB_MID Fp6 c3_sqr_part(int L, const Fp12& a) {
  Fp6 x = sel3(L, a.c0, a.c1, fp6_add(a.c0, a.c1));
  return fp6_sqr(x);
}
// f * (l0 + l1 w + l3 v w):  t0 = f0*(l0), t1 = f1*(l1 + l3 v), t2 = (f0+f1)*((l0+l1) + l3 v)
RB_MID Fp6 c3_line_part(int L, const Fp12& f, const Fp2& l0, const Fp2& l1, const Fp2& l3) {
  Fp6 x = sel3(L, f.c0, f.c1, fp6_add(f.c0, f.c1));
  Fp2 b0 = sel3(L, l0, l1, fp2_add(l0, l1));
  Fp2 b1 = sel3(L, fp2_zero(), l3, l3);
  return fp6_mul_by_01(x, b0, b1);
}

// ---- cyclotomic squaring: three Fq4 squarings on the pairs (c0.a0,c1.a1), (c1.a0,c0.a2), (c0.a1,c1.a2)
struct Fp4Pair { Fp2 r0, r1; };
RB_MID Fp4Pair c3_cyc_part(int L, const Fp12& f) {
  Fp2 x = sel3(L, f.c0.a0, f.c1.a0, f.c0.a1);
  Fp2 y = sel3(L, f.c1.a1, f.c0.a2, f.c1.a2);
  Fp4Pair p;
  fp4_sqr(p.r0, p.r1, x, y);
  return p;
}
RB_MID Fp12 c3_cyc_combine(const Fp12& f, const Fp4Pair& p0, const Fp4Pair& p1, const Fp4Pair& p2) {
  const Fp2 &t0 = p0.r0, &t1 = p0.r1, &t2 = p1.r0, &t3 = p1.r1, &t4 = p2.r0, &t5 = p2.r1;
  const Fp2 &z0 = f.c0.a0, &z4 = f.c0.a1, &z3 = f.c0.a2, &z2 = f.c1.a0, &z1 = f.c1.a1, &z5 = f.c1.a2;
  Fp12 r;
  r.c0.a0 = fp2_add(fp2_dbl(fp2_sub(t0, z0)), t0);
  r.c1.a1 = fp2_add(fp2_dbl(fp2_add(t1, z1)), t1);
  Fp2 x5 = fp2_mul_xi(t5);
  r.c1.a0 = fp2_add(fp2_dbl(fp2_add(x5, z2)), x5);
  r.c0.a2 = fp2_add(fp2_dbl(fp2_sub(t4, z3)), t4);
  r.c0.a1 = fp2_add(fp2_dbl(fp2_sub(t2, z4)), t2);
  r.c1.a2 = fp2_add(fp2_dbl(fp2_add(t3, z5)), t3);
  return r;
}

// ---- G2 doubling step, two rounds of two Fq2 products per lane
struct Fp2Pair { Fp2 p, q; };
//   round 1:  lane0: X*Y, (Y+Z)^2   lane1: Y^2, Z^2   lane2: X^2, (spare)
RB_MID Fp2Pair c3_dbl_r1(int L, const G2Hom& t) {
  Fp2 yz = fp2_add(t.y, t.z);
  Fp2 u1 = sel3(L, t.x, t.y, t.x), v1 = sel3(L, t.y, t.y, t.x);
  Fp2 u2 = sel3(L, yz, t.z, t.x), v2 = sel3(L, yz, t.z, t.x);
  return Fp2Pair{fp2_mul(u1, v1), fp2_mul(u2, v2)};
}
struct DblMid { Fp2 a, b, e, f, g, h, i, j; };
RB_MID DblMid c3_dbl_mid(const Fp2Pair& r0, const Fp2Pair& r1, const Fp2Pair& r2) {
  DblMid m;
  m.a = fp2_half(r0.p);                            // XY/2
  m.b = r1.p;                                      // Y^2
  const Fp2& c = r1.q;                             // Z^2
  m.e = fp2_mul(twist_b(), fp2_add(fp2_dbl(c), c));   // 3 b' Z^2   (replicated)
  m.f = fp2_add(fp2_dbl(m.e), m.e);
  m.g = fp2_half(fp2_add(m.b, m.f));
  m.h = fp2_sub(r0.q, fp2_add(m.b, c));            // 2YZ
  m.i = fp2_sub(m.e, m.b);
  m.j = r2.p;                                      // X^2
  return m;
}
//   round 2:  lane0: e^2, g^2   lane1: a*(b-f), (spare)   lane2: b*h, (spare)
RB_MID Fp2Pair c3_dbl_r2(int L, const DblMid& m) {
  Fp2 bf = fp2_sub(m.b, m.f);
  Fp2 u1 = sel3(L, m.e, m.a, m.b), v1 = sel3(L, m.e, bf, m.h);
  Fp2 u2 = sel3(L, m.g, m.a, m.b), v2 = sel3(L, m.g, bf, m.h);
  return Fp2Pair{fp2_mul(u1, v1), fp2_mul(u2, v2)};
}
RB_MID LineCoeffs c3_dbl_finish(G2Hom& t, const DblMid& m, const Fp2Pair& r0, const Fp2Pair& r1, const Fp2Pair& r2) {
  const Fp2& e2 = r0.p;
  t.x = r1.p;
  t.y = fp2_sub(r0.q, fp2_add(fp2_dbl(e2), e2));
  t.z = r2.p;
  LineCoeffs l;
  l.cy = fp2_neg(m.h);
  l.cx = fp2_add(fp2_dbl(m.j), m.j);
  l.c0 = m.i;
  return l;
}

// ---- G2 addition step T <- T + Q: four rounds
//   r1: lane0 qy*Z   lane1 qx*Z          -> theta = Y - qyZ, lambda = X - qxZ
//   r2: lane0 theta^2   lane1 lambda^2   lane2 theta*qx
//   r3: lane0 lambda*d   lane1 Z*c   lane2 X*d
//   r4 (two products): lane0 lambda*h, Z*e   lane1 theta*(g-h), lambda*qy   lane2 e*Y, (spare)
RB_MID Fp2 c3_add_r1(int L, const G2Hom& t, const G2Aff& q) { return fp2_mul(sel3(L, q.y, q.x, q.x), t.z); }
struct AddMid { Fp2 theta, lambda, c, d, j1, e, f, g, h; };
RB_MID Fp2 c3_add_r2(int L, const AddMid& m, const G2Aff& q) {
  return fp2_mul(sel3(L, m.theta, m.lambda, m.theta), sel3(L, m.theta, m.lambda, q.x));
}
RB_MID Fp2 c3_add_r3(int L, const AddMid& m, const G2Hom& t) {
  return fp2_mul(sel3(L, m.lambda, t.z, t.x), sel3(L, m.d, m.c, m.d));
}
RB_MID Fp2Pair c3_add_r4(int L, const AddMid& m, const G2Hom& t, const G2Aff& q) {
  Fp2 gh = fp2_sub(m.g, m.h);
  Fp2 u1 = sel3(L, m.lambda, m.theta, m.e), v1 = sel3(L, m.h, gh, t.y);
  Fp2 u2 = sel3(L, t.z, m.lambda, m.e), v2 = sel3(L, m.e, q.y, t.y);
  return Fp2Pair{fp2_mul(u1, v1), fp2_mul(u2, v2)};
}
RB_MID LineCoeffs c3_add_finish(G2Hom& t, const AddMid& m, const Fp2Pair& r0, const Fp2Pair& r1, const Fp2Pair& r2) {
  t.x = r0.p;
  t.y = fp2_sub(r1.p, r2.p);
  t.z = r0.q;
  LineCoeffs l;
  l.cy = m.lambda;
  l.cx = fp2_neg(m.theta);
  l.c0 = fp2_sub(m.j1, r1.q);
  return l;
}

// ---- line evaluation: lane0 cy*py, lane1 cx*px, lane2 c0*pz3
RB_MID Fp2 c3_ell_part(int L, const LineCoeffs& l, const MillerP& p) {
  return fp2_mul_fp(sel3(L, l.cy, l.cx, l.c0), sel3(L, p.py, p.px, p.pz3));
}

// ---- a communicator abstracts the all-gather: device = ds_bpermute within the triple, host = explicit arrays
template <class COMM>
RB_MID Fp12 c3_fp12_sqr(COMM& cm, const Fp12& a) {
  Fp6 s[3];
  cm.gather(c3_sqr_part(cm.L, a), s);
  return c3_combine_karatsuba(s[0], s[1], s[2]);
}
template <class COMM>
RB_MID Fp12 c3_fp12_mul(COMM& cm, const Fp12& a, const Fp12& b) {
  Fp6 s[3];
  cm.gather(c3_mul_part(cm.L, a, b), s);
  return c3_combine_karatsuba(s[0], s[1], s[2]);
}
template <class COMM>
RB_MID Fp12 c3_cyclotomic_sqr(COMM& cm, const Fp12& a) {
  Fp4Pair s[3];
  cm.gather(c3_cyc_part(cm.L, a), s);
  return c3_cyc_combine(a, s[0], s[1], s[2]);
}
template <class COMM>
RB_MID Fp12 c3_ell(COMM& cm, const Fp12& f, const LineCoeffs& l, const MillerP& p) {
  Fp2 e[3];
  cm.gather(c3_ell_part(cm.L, l, p), e);
  Fp6 s[3];
  cm.gather(c3_line_part(cm.L, f, e[0], e[1], e[2]), s);
  return c3_combine_karatsuba(s[0], s[1], s[2]);
}
template <class COMM>
RB_MID LineCoeffs c3_g2hom_double(COMM& cm, G2Hom& t) {
  Fp2Pair r[3];
  cm.gather(c3_dbl_r1(cm.L, t), r);
  DblMid m = c3_dbl_mid(r[0], r[1], r[2]);
  cm.gather(c3_dbl_r2(cm.L, m), r);
  return c3_dbl_finish(t, m, r[0], r[1], r[2]);
}
template <class COMM>
RB_MID LineCoeffs c3_g2hom_add(COMM& cm, G2Hom& t, const G2Aff& q) {
  Fp2 r[3];
  AddMid m;
  cm.gather(c3_add_r1(cm.L, t, q), r);
  m.theta = fp2_sub(t.y, r[0]);
  m.lambda = fp2_sub(t.x, r[1]);
  cm.gather(c3_add_r2(cm.L, m, q), r);
  m.c = r[0]; m.d = r[1]; m.j1 = r[2];
  cm.gather(c3_add_r3(cm.L, m, t), r);
  m.e = r[0]; m.f = r[1]; m.g = r[2];
  m.h = fp2_sub(fp2_add(m.e, m.f), fp2_dbl(m.g));
  Fp2Pair w[3];
  cm.gather(c3_add_r4(cm.L, m, t, q), w);
  return c3_add_finish(t, m, w[0], w[1], w[2]);
}

// Miller loop, cooperative.  Returns the same value (bit for bit) as miller_loop() in every lane of the triple.
template <class COMM>
RB_FN Fp12 c3_miller_loop(COMM& cm, const MillerP& p, bool p_is_inf, const G2Aff& q) {
  Fp12 f = fp12_one();
  if (p_is_inf || aff_is_inf(q)) return f;      // uniform within the triple (all three lanes hold the same inputs)
  G2Hom t{q.x, q.y, fp2_one()};
  const G2Aff qn = aff_neg(q);
  for (int i = RB_ATE_NAF_LEN - 2; i >= 0; i--) {       // same NAF chain as miller_loop()
    f = c3_fp12_sqr(cm, f);
    LineCoeffs l = c3_g2hom_double(cm, t);
    f = c3_ell(cm, f, l, p);
    const bool pos = (i < 64) && ((RB_ATE_NAF_POS >> i) & 1ull);
    const bool ngt = (i < 64) && ((RB_ATE_NAF_NEG >> i) & 1ull);
    if (pos | ngt) {
      LineCoeffs la = c3_g2hom_add(cm, t, pos ? q : qn);
      f = c3_ell(cm, f, la, p);
    }
  }
  G2Aff q1 = g2_frob1(q);
  G2Aff q2 = aff_neg(g2_frob2(q));
  LineCoeffs l1 = c3_g2hom_add(cm, t, q1);
  f = c3_ell(cm, f, l1, p);
  LineCoeffs l2 = c3_g2hom_add(cm, t, q2);
  f = c3_ell(cm, f, l2, p);
  return f;
}

template <class COMM>
RB_MID Fp12 c3_cyclotomic_exp_u(COMM& cm, const Fp12& f) {
  Fp12 acc = f;
  for (int i = 61; i >= 0; i--) {
    acc = c3_cyclotomic_sqr(cm, acc);
    if ((RB_BN_U >> i) & 1ull) acc = c3_fp12_mul(cm, acc, f);
  }
  return acc;
}
// Final exponentiation, cooperative (same chain as final_exponentiation()).
template <class COMM>
RB_FN Fp12 c3_final_exponentiation(COMM& cm, const Fp12& f_in) {
  Fp12 f = c3_fp12_mul(cm, fp12_conj(f_in), fp12_inv(f_in));
  f = c3_fp12_mul(cm, fp12_frob2(f), f);
  Fp12 a = fp12_conj(c3_cyclotomic_exp_u(cm, f));
  Fp12 b = c3_cyclotomic_sqr(cm, a);
  Fp12 c = c3_cyclotomic_sqr(cm, b);
  Fp12 d = c3_fp12_mul(cm, c, b);
  Fp12 e = fp12_conj(c3_cyclotomic_exp_u(cm, d));
  Fp12 ff = c3_cyclotomic_sqr(cm, e);
  Fp12 g = fp12_conj(c3_cyclotomic_exp_u(cm, ff));
  Fp12 h = fp12_conj(d);
  Fp12 i = fp12_conj(g);
  Fp12 j = c3_fp12_mul(cm, i, e);
  Fp12 k = c3_fp12_mul(cm, j, h);
  Fp12 l = c3_fp12_mul(cm, k, b);
  Fp12 m = c3_fp12_mul(cm, k, e);
  Fp12 n = c3_fp12_mul(cm, m, f);
  Fp12 o = fp12_frob1(l);
  Fp12 pp = c3_fp12_mul(cm, o, n);
  Fp12 q = fp12_frob2(k);
  Fp12 r = c3_fp12_mul(cm, q, pp);
  Fp12 s = fp12_conj(f);
  Fp12 t = c3_fp12_mul(cm, s, l);
  Fp12 uu = fp12_frob3(t);
  return c3_fp12_mul(cm, uu, r);
}

}}  // namespace rabe::bn254
