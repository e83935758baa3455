// Optimal-ate pairing on BN254: `rabe_bn::pairing(G1, G2) -> Gt`
// (call sites: src/schemes/ac17/mod.rs:148,415-416; bsw/mod.rs:108,292-293,308; lsw/mod.rs:106,275-276;
//  aw11/mod.rs:144,263,274,340-341).
//
// Split the way the batched engine needs it:
//   miller_loop(P, Q)          one lane per (P, Q) pair, G2 in homogeneous projective coordinates
//                              (Costello-Lange-Naehrig doubling/addition with line coefficients),
//                              sparse line multiplication; P may be given in Jacobian form (no inversion).
//   final_exponentiation(f)    ONE per product of pairings: easy part, then the libff / zcash-bn hard-part
//                              chain (three exponentiations by u with Granger-Scott squarings).
// A product of pairings  prod e(P_i, Q_i)  is  final_exponentiation(prod miller_loop(P_i, Q_i)).
#pragma once
#include "curve.h"

namespace rabe { namespace bn254 {

// G2 point in homogeneous projective coordinates (x = X/Z, y = Y/Z) for the Miller loop.
struct G2Hom {
  Fp2 x, y, z;
};

// Line through the untwisted points evaluated at P, as (l0, l1, l3) with value l0*yP' + l1*xP' w + l3 w^3.
// The coefficients returned here are (c_y, c_x, c_0): the caller scales c_y by y_P and c_x by x_P.
struct LineCoeffs {
  Fp2 cy, cx, c0;
};

// Doubling step: T <- 2T, returns the tangent line coefficients.  (Costello et al., as used for
// D-type twists: cy = -2YZ, cx = 3X^2, c0 = 3b'Z^2 - Y^2.)
RB_MID LineCoeffs g2hom_double(G2Hom& r) {
  Fp2 a = fp2_half(fp2_mul(r.x, r.y));               // halvings by shift (fp.h: half), not by a multiplication with 1/2
  Fp2 b = fp2_sqr(r.y);
  Fp2 c = fp2_sqr(r.z);
  Fp2 e = fp2_mul(twist_b(), fp2_add(fp2_dbl(c), c));   // 3 b' Z^2
  Fp2 f = fp2_add(fp2_dbl(e), e);                       // 9 b' Z^2
  Fp2 g = fp2_half(fp2_add(b, f));
  Fp2 h = fp2_sub(fp2_sqr(fp2_add(r.y, r.z)), fp2_add(b, c));   // 2YZ
  Fp2 i = fp2_sub(e, b);
  Fp2 j = fp2_sqr(r.x);
  Fp2 e2 = fp2_sqr(e);
  r.x = fp2_mul(a, fp2_sub(b, f));
  r.y = fp2_sub(fp2_sqr(g), fp2_add(fp2_dbl(e2), e2));
  r.z = fp2_mul(b, h);
  LineCoeffs l;
  l.cy = fp2_neg(h);
  l.cx = fp2_add(fp2_dbl(j), j);
  l.c0 = i;
  return l;
}

// Addition step: T <- T + Q (Q affine), returns the chord line coefficients.
RB_MID LineCoeffs g2hom_add(G2Hom& r, const G2Aff& q) {
  Fp2 theta = fp2_sub(r.y, fp2_mul(q.y, r.z));
  Fp2 lambda = fp2_sub(r.x, fp2_mul(q.x, r.z));
  Fp2 c = fp2_sqr(theta);
  Fp2 d = fp2_sqr(lambda);
  Fp2 e = fp2_mul(lambda, d);
  Fp2 f = fp2_mul(r.z, c);
  Fp2 g = fp2_mul(r.x, d);
  Fp2 h = fp2_sub(fp2_add(e, f), fp2_dbl(g));
  Fp2 ry = r.y;
  r.x = fp2_mul(lambda, h);
  r.y = fp2_sub(fp2_mul(theta, fp2_sub(g, h)), fp2_mul(e, ry));
  r.z = fp2_mul(r.z, e);
  LineCoeffs l;
  l.cy = lambda;
  l.cx = fp2_neg(theta);
  l.c0 = fp2_sub(fp2_mul(theta, q.x), fp2_mul(lambda, q.y));
  return l;
}

// pi(Q) and pi^2(Q) in twist coordinates: pi(x', y') = (conj(x') gamma1_2, conj(y') gamma1_3),
// pi^2(x', y') = (x' gamma2_2, y' gamma2_3).
RB_HD G2Aff g2_frob1(const G2Aff& q) {
  return G2Aff{fp2_mul(fp2_conj(q.x), gamma1_2()), fp2_mul(fp2_conj(q.y), gamma1_3())};
}
RB_HD G2Aff g2_frob2(const G2Aff& q) {
  return G2Aff{fp2_mul_fp(q.x, gamma2_2()), fp2_mul_fp(q.y, gamma2_3())};
}

// P for the Miller loop: (px, py, and the scale s applied to the constant coefficient).
// For affine P: px = x, py = y, pz3 = 1.  For Jacobian P = (X, Y, Z): multiply the whole line by Z^3
// (an Fp factor, killed by the final exponentiation): py = Y, px = X*Z, pz3 = Z^3.
struct MillerP {
  Fp px, py, pz3;
  bool scaled;   // false: pz3 == 1 (skip the multiplication)
};
RB_HD MillerP miller_p_from_aff(const G1Aff& p) { return MillerP{p.x, p.y, one<FpParams>(), false}; }
RB_HD MillerP miller_p_from_jac(const G1Jac& p) {
  Fp z2 = sqr(p.z);
  return MillerP{mul(p.x, p.z), p.y, mul(z2, p.z), true};
}

RB_MID Fp12 ell(const Fp12& f, const LineCoeffs& l, const MillerP& p) {
  Fp2 l0 = fp2_mul_fp(l.cy, p.py);
  Fp2 l1 = fp2_mul_fp(l.cx, p.px);
  Fp2 l3 = p.scaled ? fp2_mul_fp(l.c0, p.pz3) : l.c0;
  return fp12_mul_by_line(f, l0, l1, l3);
}

// f * line_A(P_A) * line_B(P_B)
RB_MID Fp12 ell2(const Fp12& f, const LineCoeffs& la, const MillerP& pa, const LineCoeffs& lb, const MillerP& pb) {
  Fp2 a0 = fp2_mul_fp(la.cy, pa.py);
  Fp2 a1 = fp2_mul_fp(la.cx, pa.px);
  Fp2 a3 = pa.scaled ? fp2_mul_fp(la.c0, pa.pz3) : la.c0;
  Fp2 b0 = fp2_mul_fp(lb.cy, pb.py);
  Fp2 b1 = fp2_mul_fp(lb.cx, pb.px);
  Fp2 b3 = pb.scaled ? fp2_mul_fp(lb.c0, pb.pz3) : lb.c0;
  return fp12_mul_by_two_lines(f, a0, a1, a3, b0, b1, b3);
}

// Miller loop.  Either argument at infinity gives 1 (as `pairing` does for zero inputs).
RB_FN Fp12 miller_loop(const MillerP& p, bool p_is_inf, const G2Aff& q) {
  Fp12 f = fp12_one();
  if (p_is_inf || aff_is_inf(q)) return f;
  G2Hom t{q.x, q.y, fp2_one()};
  // 6u+2 in non-adjacent form (66 digits, 22 non-zero): the leading digit is consumed by T = Q, then 65 doubling
  // steps and 21 addition steps with +Q or -Q.  The value differs from the binary chain's only by vertical-line
  // factors in Fq6, which the final exponentiation removes (zcash bn / libff use the same NAF chain).
  const G2Aff qn = aff_neg(q);
  for (int i = RB_ATE_NAF_LEN - 2; i >= 0; i--) {
    f = fp12_sqr(f);
    LineCoeffs l = g2hom_double(t);
    f = ell(f, l, p);
    const bool pos = (i < 64) && ((RB_ATE_NAF_POS >> i) & 1ull);
    const bool ngt = (i < 64) && ((RB_ATE_NAF_NEG >> i) & 1ull);
    if (pos | ngt) {
      LineCoeffs la = g2hom_add(t, pos ? q : qn);
      f = ell(f, la, p);
    }
  }
  G2Aff q1 = g2_frob1(q);
  G2Aff q2 = aff_neg(g2_frob2(q));
  LineCoeffs l1 = g2hom_add(t, q1);
  f = ell(f, l1, p);
  LineCoeffs l2 = g2hom_add(t, q2);
  f = ell(f, l2, p);
  return f;
}

// Out-of-line forms for the final exponentiation: it juggles ~6 live Fq12 values (more than the register file
// holds), so inlining every product only multiplies spill code; as calls, each product's temporaries are private
// to its own frame and the exponentiation keeps a small, explicit set of locals.
RB_FN Fp12 fp12_mul_fn(const Fp12& a, const Fp12& b) { return fp12_mul(a, b); }
RB_FN Fp12 fp12_cyclotomic_sqr_fn(const Fp12& a) { return fp12_cyclotomic_sqr(a); }
RB_FN Fp12 fp12_frob_fn(const Fp12& a, int k) { return k == 1 ? fp12_frob1(a) : (k == 2 ? fp12_frob2(a) : fp12_frob3(a)); }

// ---- prepared G2 argument: when Q is fixed (a secret key's k_0, a public key element) the G2 arithmetic of the
// Miller loop does not depend on P, so its line coefficients are computed once per Q and replayed
// (the role of `G2Precomp` in the zcash-bn lineage).  88 triples (65 doublings + 21 additions + 2 Frobenius steps).
#define RB_MILLER_LINES (RB_ATE_NAF_LEN - 1 + RB_ATE_NAF_ADDS + 2)
RB_FN void g2_prepare_lines(const G2Aff& q, LineCoeffs* out) {
  int n = 0;
  if (aff_is_inf(q)) {
    for (int i = 0; i < RB_MILLER_LINES; i++) out[i] = LineCoeffs{fp2_zero(), fp2_zero(), fp2_zero()};
    return;
  }
  G2Hom t{q.x, q.y, fp2_one()};
  const G2Aff qn = aff_neg(q);
  for (int i = RB_ATE_NAF_LEN - 2; i >= 0; i--) {
    out[n++] = g2hom_double(t);
    const bool pos = (i < 64) && ((RB_ATE_NAF_POS >> i) & 1ull);
    const bool ngt = (i < 64) && ((RB_ATE_NAF_NEG >> i) & 1ull);
    if (pos | ngt) out[n++] = g2hom_add(t, pos ? q : qn);
  }
  G2Aff q1 = g2_frob1(q);
  G2Aff q2 = aff_neg(g2_frob2(q));
  out[n++] = g2hom_add(t, q1);
  out[n++] = g2hom_add(t, q2);
}
// LOAD is a functor  LineCoeffs operator()(int k)  fetching triple k (device: global memory, host: array)
template <class LOAD>
RB_FN Fp12 miller_loop_prepared(const MillerP& p, bool p_is_inf, bool q_is_inf, LOAD load) {
  Fp12 f = fp12_one();
  if (p_is_inf || q_is_inf) return f;
  int n = 0;
  for (int i = RB_ATE_NAF_LEN - 2; i >= 0; i--) {
    f = fp12_sqr(f);
    f = ell(f, load(n++), p);
    const bool nz = (i < 64) && (((RB_ATE_NAF_POS | RB_ATE_NAF_NEG) >> i) & 1ull);
    if (nz) f = ell(f, load(n++), p);
  }
  f = ell(f, load(n++), p);
  f = ell(f, load(n++), p);
  return f;
}

// Two pairings on one accumulator: the Fq12 squaring of every doubling step is paid once for both.  Pairing A
// replays prepared lines (fixed Q), pairing B walks its own G2 point.  A pairing with an argument at infinity
// contributes 1 (its steps are skipped).  Result = miller(pa, A) * miller(pb, qb) up to Fq6 factors.
template <class LOAD>
RB_FN Fp12 miller_loop_pair(const MillerP& pa, bool skip_a, LOAD load, const MillerP& pb, bool pb_is_inf, const G2Aff& qb) {
  const bool skip_b = pb_is_inf || aff_is_inf(qb);
  Fp12 f = fp12_one();
  G2Hom t{qb.x, qb.y, fp2_one()};
  const G2Aff qn = aff_neg(qb);
  int n = 0;
  for (int i = RB_ATE_NAF_LEN - 2; i >= 0; i--) {
    f = fp12_sqr(f);
    if (!skip_a) f = ell(f, load(n), pa);
    n++;
    if (!skip_b) {
      LineCoeffs l = g2hom_double(t);
      f = ell(f, l, pb);
    }
    const bool pos = (i < 64) && ((RB_ATE_NAF_POS >> i) & 1ull);
    const bool ngt = (i < 64) && ((RB_ATE_NAF_NEG >> i) & 1ull);
    if (pos | ngt) {
      if (!skip_a) f = ell(f, load(n), pa);
      n++;
      if (!skip_b) {
        LineCoeffs la = g2hom_add(t, pos ? qb : qn);
        f = ell(f, la, pb);
      }
    }
  }
  if (!skip_a) {
    f = ell(f, load(n), pa);
    f = ell(f, load(n + 1), pa);
  }
  if (!skip_b) {
    G2Aff q1 = g2_frob1(qb);
    G2Aff q2 = aff_neg(g2_frob2(qb));
    LineCoeffs l1 = g2hom_add(t, q1);
    f = ell(f, l1, pb);
    LineCoeffs l2 = g2hom_add(t, q2);
    f = ell(f, l2, pb);
  }
  return f;
}

// The same paired loop with the loop-invariant / once-per-step values parked outside the register file.
// The pairing kernels run one wave per SIMD with the whole 512-entry register file, and still the accumulator, the
// G2 point, both P's, Q and the Fq12 temporaries do not fit; what the compiler spills goes to scratch (HBM-backed,
// ~1 us per reload with nothing else on the SIMD to hide it).  With one wave per SIMD each wave also owns a quarter
// of the CU's 160 KB LDS, so the values that are touched once per step -- P_A, P_B (scaled for Jacobian input),
// Q_B and the running point T -- are parked there explicitly and fetched right where they are used.
// PARK provides  Fp ld(int i) const  /  void st(int i, const Fp&) const  for i < PK_FPS.
enum { PK_PA = 0, PK_PB = 3, PK_QB = 6, PK_T = 10, PK_FPS = 16 };
template <class PARK> RB_HD Fp2 pk_ld2(PARK pk, int i) { return Fp2{pk.ld(i), pk.ld(i + 1)}; }
template <class PARK> RB_HD void pk_st2(PARK pk, int i, const Fp2& a) { pk.st(i, a.c0); pk.st(i + 1, a.c1); }
template <class PARK> RB_HD MillerP pk_ld_p(PARK pk, int i) { return MillerP{pk.ld(i), pk.ld(i + 1), pk.ld(i + 2), true}; }
template <class PARK> RB_HD void pk_st_p(PARK pk, int i, const MillerP& p) { pk.st(i, p.px); pk.st(i + 1, p.py); pk.st(i + 2, p.pz3); }
template <class PARK> RB_HD G2Hom pk_ld_t(PARK pk) { return G2Hom{pk_ld2(pk, PK_T), pk_ld2(pk, PK_T + 2), pk_ld2(pk, PK_T + 4)}; }
template <class PARK> RB_HD void pk_st_t(PARK pk, const G2Hom& t) { pk_st2(pk, PK_T, t.x); pk_st2(pk, PK_T + 2, t.y); pk_st2(pk, PK_T + 4, t.z); }
// caller parks P_A at PK_PA, P_B at PK_PB (pk_st_p) and Q_B at PK_QB (x, y); skip_b must already include Q_B = infinity
template <class LOAD, class PARK>
RB_FN Fp12 miller_loop_pair_parked(PARK pk, bool skip_a, LOAD load, bool skip_b) {
  Fp12 f = fp12_one();
  pk_st_t(pk, G2Hom{pk_ld2(pk, PK_QB), pk_ld2(pk, PK_QB + 2), fp2_one()});
  const bool both = !skip_a && !skip_b;
  int n = 0;
  for (int i = RB_ATE_NAF_LEN - 2; i >= 0; i--) {
    f = fp12_sqr(f);
    if (both) {
      G2Hom t = pk_ld_t(pk);
      LineCoeffs l = g2hom_double(t);
      pk_st_t(pk, t);
      f = ell2(f, load(n), pk_ld_p(pk, PK_PA), l, pk_ld_p(pk, PK_PB));
    } else if (!skip_a) {
      f = ell(f, load(n), pk_ld_p(pk, PK_PA));
    } else if (!skip_b) {
      G2Hom t = pk_ld_t(pk);
      LineCoeffs l = g2hom_double(t);
      pk_st_t(pk, t);
      f = ell(f, l, pk_ld_p(pk, PK_PB));
    }
    n++;
    const bool pos = (i < 64) && ((RB_ATE_NAF_POS >> i) & 1ull);
    const bool ngt = (i < 64) && ((RB_ATE_NAF_NEG >> i) & 1ull);
    if (pos | ngt) {
      LineCoeffs lb;
      if (!skip_b) {
        G2Hom t = pk_ld_t(pk);
        G2Aff q{pk_ld2(pk, PK_QB), pk_ld2(pk, PK_QB + 2)};
        if (ngt) q.y = fp2_neg(q.y);
        lb = g2hom_add(t, q);
        pk_st_t(pk, t);
      }
      if (both) f = ell2(f, load(n), pk_ld_p(pk, PK_PA), lb, pk_ld_p(pk, PK_PB));
      else if (!skip_a) f = ell(f, load(n), pk_ld_p(pk, PK_PA));
      else if (!skip_b) f = ell(f, lb, pk_ld_p(pk, PK_PB));
      n++;
    }
  }
  if (!skip_a) {
    f = ell(f, load(n), pk_ld_p(pk, PK_PA));
    f = ell(f, load(n + 1), pk_ld_p(pk, PK_PA));
  }
  if (!skip_b) {
    G2Hom t = pk_ld_t(pk);
    const G2Aff qb{pk_ld2(pk, PK_QB), pk_ld2(pk, PK_QB + 2)};
    LineCoeffs l1 = g2hom_add(t, g2_frob1(qb));
    f = ell(f, l1, pk_ld_p(pk, PK_PB));
    LineCoeffs l2 = g2hom_add(t, aff_neg(g2_frob2(qb)));
    f = ell(f, l2, pk_ld_p(pk, PK_PB));
  }
  return f;
}

// ---- Fq12 operations on an accumulator that lives outside the register file.
// A lane of the pairing kernels owns the whole 512-entry register file and still an Fq12 operation with its operands,
// three Fq6 temporaries and the Fq2-level callee's 166 registers does not fit; what the register allocator evicts goes to
// scratch, i.e. to HBM, with one wave per SIMD and nothing to hide the reload behind.  These forms keep f = c0 + c1 w in
// a home the caller provides (LDS on the device), fetch each half where it is consumed, and park the one Fq6 value that has
// to survive the other two products in a second slot -- the same formulas as fp12_sqr / fp12_mul_by_line /
// fp12_mul_by_two_lines (tower.h), hence the same canonical values.
// FA provides  Fp6 ld_f6(int half) const, void st_f6(int half, const Fp6&) const, Fp6 ld_x() const, void st_x(const Fp6&) const,
//              Fp2 ld_f2(int i) const, void st_f2(int i, const Fp2&) const   (coefficient i of c0.a0, c0.a1, c0.a2, c1.a0, c1.a1, c1.a2),
//              void fence() const   (keeps the compiler from carrying a fetched half across it instead of fetching again)
template <class FA> RB_HD void facc_set_one(FA a) { a.st_f6(0, fp6_one()); a.st_f6(1, fp6_zero()); }
template <class FA> RB_HD Fp12 facc_get(FA a) { return Fp12{a.ld_f6(0), a.ld_f6(1)}; }
template <class FA> RB_HD void facc_finish(FA a, const Fp6& t1, const Fp6& t2) {     // r.c0 = t0 + v t1 ; r.c1 = t2 - t0 - t1, t0 parked
  const Fp6 t0 = a.ld_x();
  a.st_f6(0, fp6_add(t0, fp6_mul_v(t1)));
  a.st_f6(1, fp6_sub(fp6_sub(t2, t0), t1));
}
template <class FA> RB_HD void facc_sqr(FA a) {
  { const Fp6 ab = fp6_mul(a.ld_f6(0), a.ld_f6(1)); a.st_x(ab); }
  a.fence();
  Fp6 t;
  { const Fp6 c0 = a.ld_f6(0), c1 = a.ld_f6(1); t = fp6_mul(fp6_add(c0, c1), fp6_add(c0, fp6_mul_v(c1))); }
  a.fence();
  const Fp6 ab = a.ld_x();
  a.st_f6(0, fp6_sub(fp6_sub(t, ab), fp6_mul_v(ab)));
  a.st_f6(1, fp6_dbl(ab));
}
template <class FA> RB_HD void facc_mul_by_line(FA a, const Fp2& l0, const Fp2& l1, const Fp2& l3) {
  { const Fp6 t0 = fp6_mul_fp2(a.ld_f6(0), l0); a.st_x(t0); }
  a.fence();
  const Fp6 t1 = fp6_mul_by_01(a.ld_f6(1), l1, l3);
  a.fence();
  const Fp6 t2 = fp6_mul_by_01(fp6_add(a.ld_f6(0), a.ld_f6(1)), fp2_add(l0, l1), l3);
  a.fence();
  facc_finish(a, t1, t2);
}
template <class FA> RB_HD void facc_mul_by_two_lines(FA a, const Fp2& a0, const Fp2& a1, const Fp2& a3, const Fp2& b0, const Fp2& b1, const Fp2& b3) {
  const Fp2 m00 = fp2_mul(a0, b0);
  const Fp2 m11 = fp2_mul(a1, b1);
  const Fp2 m33 = fp2_mul(a3, b3);
  const Fp2 x01 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a0, a1), fp2_add(b0, b1)), m00), m11);
  const Fp2 x03 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a0, a3), fp2_add(b0, b3)), m00), m33);
  const Fp2 x13 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a1, a3), fp2_add(b1, b3)), m11), m33);
  const Fp6 p0{fp2_add_mul_xi(m00, m33), m11, x13};
  { const Fp6 t0 = fp6_mul(a.ld_f6(0), p0); a.st_x(t0); }
  a.fence();
  const Fp6 t1 = fp6_mul_by_01(a.ld_f6(1), x01, x03);
  a.fence();
  const Fp6 t2 = fp6_mul(fp6_add(a.ld_f6(0), a.ld_f6(1)), Fp6{fp2_add(p0.a0, x01), fp2_add(p0.a1, x03), p0.a2});
  a.fence();
  facc_finish(a, t1, t2);
}
template <class FA> RB_HD void facc_ell(FA a, const LineCoeffs& l, const MillerP& p) {
  const Fp2 l0 = fp2_mul_fp(l.cy, p.py);
  const Fp2 l1 = fp2_mul_fp(l.cx, p.px);
  const Fp2 l3 = p.scaled ? fp2_mul_fp(l.c0, p.pz3) : l.c0;
  facc_mul_by_line(a, l0, l1, l3);
}
template <class FA> RB_HD void facc_ell2(FA a, const LineCoeffs& la, const MillerP& pa, const LineCoeffs& lb, const MillerP& pb) {
  const Fp2 a0 = fp2_mul_fp(la.cy, pa.py);
  const Fp2 a1 = fp2_mul_fp(la.cx, pa.px);
  const Fp2 a3 = pa.scaled ? fp2_mul_fp(la.c0, pa.pz3) : la.c0;
  const Fp2 b0 = fp2_mul_fp(lb.cy, pb.py);
  const Fp2 b1 = fp2_mul_fp(lb.cx, pb.px);
  const Fp2 b3 = pb.scaled ? fp2_mul_fp(lb.c0, pb.pz3) : lb.c0;
  facc_mul_by_two_lines(a, a0, a1, a3, b0, b1, b3);
}

// f *= y for a general y; Y provides  Fp6 half(int h) const  (fetched where it is consumed: a table entry or a workspace slot)
template <class FA, class Y> RB_HD void facc_mul(FA a, Y y) {
  { const Fp6 t0 = fp6_mul(a.ld_f6(0), y.half(0)); a.st_x(t0); }
  a.fence();
  const Fp6 t1 = fp6_mul(a.ld_f6(1), y.half(1));
  a.fence();
  const Fp6 t2 = fp6_mul(fp6_add(a.ld_f6(0), a.ld_f6(1)), fp6_add(y.half(0), y.half(1)));
  a.fence();
  facc_finish(a, t1, t2);
  a.fence();
}
template <class FA, class Y> RB_HD void facc_set(FA a, Y y) { a.st_f6(0, y.half(0)); a.st_f6(1, y.half(1)); a.fence(); }
// Granger-Scott squaring in place (fp12_cyclotomic_sqr, tower.h): the pair (c0.a0, c1.a1) is self-contained, the other two
// pairs feed each other's slots and are squared together.
template <class FA> RB_HD void facc_cyclotomic_sqr(FA a) {
  {
    const Fp2 z0 = a.ld_f2(0), z1 = a.ld_f2(4);
    Fp2 t0, t1;
    fp4_sqr(t0, t1, z0, z1);
    a.st_f2(0, fp2_add(fp2_dbl(fp2_sub(t0, z0)), t0));
    a.st_f2(4, fp2_add(fp2_dbl(fp2_add(t1, z1)), t1));
  }
  a.fence();
  {
    const Fp2 z4 = a.ld_f2(1), z3 = a.ld_f2(2), z2 = a.ld_f2(3), z5 = a.ld_f2(5);
    Fp2 t2, t3, t4, t5;
    fp4_sqr(t2, t3, z2, z3);
    fp4_sqr(t4, t5, z4, z5);
    const Fp2 x5 = fp2_mul_xi(t5);
    a.st_f2(3, fp2_add(fp2_dbl(fp2_add(x5, z2)), x5));
    a.st_f2(2, fp2_add(fp2_dbl(fp2_sub(t4, z3)), t4));
    a.st_f2(1, fp2_add(fp2_dbl(fp2_sub(t2, z4)), t2));
    a.st_f2(5, fp2_add(fp2_dbl(fp2_add(t3, z5)), t3));
  }
  a.fence();
}

// ---- any number of pairings on one accumulator: f = prod_j miller(P_j, Q_j) (up to Fq6 factors the final
// exponentiation removes).  The Fq12 squaring of a doubling step is paid once for all of the lane's pairs, and the
// lines of two neighbouring pairs are multiplied together first (ell2).  A pair either walks its own G2 point
// (MP_WALK: the running point T_j lives outside the register file, fetched and written back around its step) or
// replays prepared lines of a fixed Q (MP_LINES: no G2 arithmetic at all); MP_SKIP pairs contribute 1 (an argument
// at infinity).  This is what bsw / lsw / aw11 decrypt need: the 2m+1 pairings of one ciphertext multiply into one
// value (src/schemes/bsw/mod.rs:291-294,308; lsw/mod.rs:275-280; aw11/mod.rs:340-350).
// ACC provides:
//   int count() const                         pairs of this lane
//   int kind(int j) const                     MP_WALK / MP_LINES / MP_SKIP
//   MillerP p(int j) const                    the G1 argument (affine: scaled = false)
//   G2Aff q(int j) const                      MP_WALK: the G2 argument
//   LineCoeffs line(int j, int n) const       MP_LINES: prepared triple n (order of g2_prepare_lines)
//   G2Hom ld_t(int j) const / void st_t(int j, const G2Hom&) const      MP_WALK: the running point
//   the FA interface above (ld_f6 / st_f6 / ld_x / st_x / fence)         the accumulator's home (LDS on the device)
enum { MP_WALK = 0, MP_LINES = 1, MP_SKIP = 2 };
enum { MS_DBL = 0, MS_ADD_POS, MS_ADD_NEG, MS_FROB1, MS_FROB2 };
template <class ACC>
RB_HD bool miller_multi_line(ACC acc, int j, int mode, int ln, LineCoeffs& l) {
  const int kind = acc.kind(j);
  if (kind == MP_SKIP) return false;
  if (kind == MP_LINES) { l = acc.line(j, ln); return true; }
  G2Hom t = acc.ld_t(j);
  if (mode == MS_DBL) {
    l = g2hom_double(t);
  } else {
    G2Aff q = acc.q(j);
    if (mode == MS_ADD_NEG) q.y = fp2_neg(q.y);
    else if (mode == MS_FROB1) q = g2_frob1(q);
    else if (mode == MS_FROB2) q = aff_neg(g2_frob2(q));
    l = g2hom_add(t, q);
  }
  acc.st_t(j, t);
  return true;
}
template <class ACC>
RB_FN Fp12 miller_loop_multi(ACC acc) {
  const int n = acc.count();
  facc_set_one(acc);
  for (int j = 0; j < n; j++) {
    if (acc.kind(j) == MP_WALK) {
      const G2Aff q = acc.q(j);
      acc.st_t(j, G2Hom{q.x, q.y, fp2_one()});
    }
  }
  // one loop over the RB_MILLER_LINES line events (65 doublings, 21 additions, 2 Frobenius additions) so that the step
  // body below exists once in the instruction stream
  int i = RB_ATE_NAF_LEN - 2;
  bool add_pending = false;
  for (int ln = 0; ln < RB_MILLER_LINES; ln++) {
    int mode;
    if (i >= 0) {
      const bool pos = (i < 64) && ((RB_ATE_NAF_POS >> i) & 1ull);
      const bool ngt = (i < 64) && ((RB_ATE_NAF_NEG >> i) & 1ull);
      if (!add_pending) {
        facc_sqr(acc);
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
    for (int j = 0; j < n; j += 2) {
      LineCoeffs la, lb;
      const bool ha = miller_multi_line(acc, j, mode, ln, la);
      const bool hb = (j + 1 < n) && miller_multi_line(acc, j + 1, mode, ln, lb);
      if (ha && hb) facc_ell2(acc, la, acc.p(j), lb, acc.p(j + 1));
      else if (ha) facc_ell(acc, la, acc.p(j));
      else if (hb) facc_ell(acc, lb, acc.p(j + 1));
    }
  }
  return facc_get(acc);
}

// f^u for f in the cyclotomic subgroup (u = 4965661367192848881, 63 bits).
RB_FN Fp12 fp12_cyclotomic_exp_u(const Fp12& f) {
  Fp12 acc = f;   // top bit (bit 62)
  for (int i = 61; i >= 0; i--) {
    acc = fp12_cyclotomic_sqr_fn(acc);
    if ((RB_BN_U >> i) & 1ull) acc = fp12_mul_fn(acc, f);
  }
  return acc;
}

// Final exponentiation: f^((p^6-1)(p^2+1)) then the libff / zcash-bn `final_exponentiation_last_chunk`
// chain, whose exponent is 2u(6u^2+3u+1) * (p^4-p^2+1)/r (oracle/bn254.py FINAL_EXP documents the choice).
RB_FN Fp12 final_exponentiation(const Fp12& f_in) {
  // easy part
  Fp12 f = fp12_mul_fn(fp12_conj(f_in), fp12_inv(f_in));   // f^(p^6-1)
  f = fp12_mul_fn(fp12_frob_fn(f, 2), f);                       // ^(p^2+1)
  // hard part; in the cyclotomic subgroup inversion is conjugation, exp_by_neg_z(x) = conj(x^u).
  // libff names (a..v) in the comments; temporaries are folded so that few Fq12 values are live at once.
  Fp12 b = fp12_cyclotomic_sqr_fn(fp12_conj(fp12_cyclotomic_exp_u(f)));      // a = f^-u ; b = a^2
  Fp12 d = fp12_mul_fn(fp12_cyclotomic_sqr_fn(b), b);                           // c = b^2 ; d = c*b
  Fp12 e = fp12_conj(fp12_cyclotomic_exp_u(d));                           // e = d^-u
  Fp12 g = fp12_conj(fp12_cyclotomic_exp_u(fp12_cyclotomic_sqr_fn(e)));      // f' = e^2 ; g = f'^-u
  Fp12 k = fp12_mul_fn(fp12_mul_fn(fp12_conj(g), e), fp12_conj(d));             // i = g^-1 ; j = i*e ; h = d^-1 ; k = j*h
  Fp12 l = fp12_mul_fn(k, b);                                                // l = k*b
  Fp12 n = fp12_mul_fn(fp12_mul_fn(k, e), f);                                   // m = k*e ; n = m*f
  Fp12 r = fp12_mul_fn(fp12_frob_fn(k, 2), fp12_mul_fn(fp12_frob_fn(l, 1), n));           // o = l^p ; p = o*n ; q = k^(p^2) ; r = q*p
  Fp12 t = fp12_mul_fn(fp12_conj(f), l);                                     // s = f^-1 ; t = s*l
  return fp12_mul_fn(fp12_frob_fn(t, 3), r);                                      // u = t^(p^3) ; v = u*r
}

// ---- the same final exponentiation over an explicit workspace.
// On the device every Fq12 that is live across a call has to sit in memory anyway (96 registers do not travel
// through a call); letting the compiler put them in stack frames made k_final_exp the kernel with the largest
// scratch frame, and the HSA runtime sizes each hardware queue's scratch for the largest frame x every wave slot
// of the chip (32 GB in total over all queues), which capped the number of batches in flight.  Here the values
// live in numbered workspace slots the caller provides (global memory sized by the actual lane count, coalesced
// [slot][word][lane]); each operation loads its operands, works in registers and stores its result.
// WS provides  Fp12 ld(int slot) const  and  void st(int slot, const Fp12&) const.
enum { FE_F = 0, FE_B, FE_D, FE_E, FE_K, FE_L, FE_T0, FE_T1, FE_SLOTS };
// WS provides  Fp12 ld(int slot) const, void st(int slot, const Fp12&) const, Fp6 ld6(int slot, int half) const,
//              void st6(int slot, int half, const Fp6&) const, and  home()  -- an FA (above) for the value being worked on.
template <class WS> struct WsOperand {
  WS ws; int slot; bool conj;
  RB_HD Fp6 half(int h) const { const Fp6 v = ws.ld6(slot, h); return (conj && h == 1) ? fp6_neg(v) : v; }
};
template <class WS> RB_HD void wsx_to_home(WS ws, int a, bool conj) { facc_set(ws.home(), WsOperand<WS>{ws, a, conj}); }
template <class WS> RB_HD void wsx_from_home(WS ws, int dst) {
  auto h = ws.home();
  ws.st6(dst, 0, h.ld_f6(0));
  ws.st6(dst, 1, h.ld_f6(1));
  h.fence();
}
template <class WS> RB_FN void wsx_mul(WS ws, int dst, int a, bool conj_a, int b, bool conj_b) {
  wsx_to_home(ws, a, conj_a);
  facc_mul(ws.home(), WsOperand<WS>{ws, b, conj_b});
  wsx_from_home(ws, dst);
}
template <class WS> RB_FN void wsx_csqr(WS ws, int dst, int a, bool conj_a) {
  Fp12 x = ws.ld(a);
  if (conj_a) x = fp12_conj(x);
  ws.st(dst, fp12_cyclotomic_sqr(x));
}
template <class WS> RB_FN void wsx_frob_mul(WS ws, int dst, int a, int k, int b) {   // dst = a^(p^k) * b
  {
    Fp12 x = ws.ld(a);
    x = (k == 1) ? fp12_frob1(x) : (k == 2) ? fp12_frob2(x) : fp12_frob3(x);
    auto h = ws.home();
    h.st_f6(0, x.c0);
    h.st_f6(1, x.c1);
    h.fence();
  }
  facc_mul(ws.home(), WsOperand<WS>{ws, b, false});
  wsx_from_home(ws, dst);
}
template <class WS> RB_FN void wsx_inv(WS ws, int dst, int a) {      // fp12_inv with the Fq12 operand / result kept out of stack frames
  const Fp12 x = ws.ld(a);
  const Fp6 t = fp6_sub(fp6_sqr(x.c0), fp6_mul_v(fp6_sqr(x.c1)));
  const Fp6 ti = fp6_inv(t);
  ws.st(dst, Fp12{fp6_mul(x.c0, ti), fp6_neg(fp6_mul(x.c1, ti))});
}
// dst = a^(2^n) * (b >= 0 ? b (conjugated if conj_b) : 1): the run of cyclotomic squarings stays in registers (a squaring
// needs few temporaries), the multiplication that ends it works on the home copy
template <class WS> RB_MID void wsx_sqrn_mul(WS ws, int dst, int a, int n, int b, bool conj_b) {
  Fp12 x = ws.ld(a);
#pragma unroll 1
  for (int i = 0; i < n; i++) x = fp12_cyclotomic_sqr(x);
  if (b < 0) { ws.st(dst, x); return; }
  auto h = ws.home();
  h.st_f6(0, x.c0);
  h.st_f6(1, x.c1);
  h.fence();
  facc_mul(h, WsOperand<WS>{ws, b, conj_b});
  wsx_from_home(ws, dst);
}
// dst = src^u over the width-3 NAF of u (digits +-1, +-3; an inverse in the cyclotomic subgroup is a conjugation):
// 62 squarings + 17 multiplications + f^3, instead of 62 + 27 for the binary chain.  dst, src, cube distinct slots.
template <class WS> RB_FN void wsx_exp_u(WS ws, int dst, int src, int cube) {
  constexpr signed char SQ[RB_U_WNAF_STEPS] = RB_U_WNAF_SQ;
  constexpr signed char DG[RB_U_WNAF_STEPS] = RB_U_WNAF_DG;
  wsx_sqrn_mul(ws, cube, src, 1, src, false);               // f^3 = f^2 * f
  int cur = (RB_U_WNAF_TOP == 3) ? cube : src;
  for (int i = 0; i < RB_U_WNAF_STEPS; i++) {
    const int d = DG[i];
    wsx_sqrn_mul(ws, dst, cur, SQ[i], (d == 1 || d == -1) ? src : cube, d < 0);
    cur = dst;
  }
  if (RB_U_WNAF_TAIL) wsx_sqrn_mul(ws, dst, cur, RB_U_WNAF_TAIL, -1, false);
}
// in: slot FE_T0 = the Miller value; out: slot FE_T1.  Same chain as final_exponentiation above.
template <class WS> RB_FN void final_exponentiation_ws(WS ws) {
  wsx_inv(ws, FE_T1, FE_T0);
  wsx_mul(ws, FE_T1, FE_T0, true, FE_T1, false);       // f^(p^6-1) = conj(f) * f^-1
  wsx_frob_mul(ws, FE_F, FE_T1, 2, FE_T1);             // ^(p^2+1)                                   F
  wsx_exp_u(ws, FE_T0, FE_F, FE_T1);                   // f^u        (a = conj of it)
  wsx_csqr(ws, FE_B, FE_T0, true);                     // b = a^2                                    B
  wsx_csqr(ws, FE_T0, FE_B, false);                    // c = b^2
  wsx_mul(ws, FE_D, FE_T0, false, FE_B, false);        // d = c*b                                    D
  wsx_exp_u(ws, FE_E, FE_D, FE_T1);                    // d^u        (e = conj of it)
  wsx_csqr(ws, FE_T0, FE_E, true);                     // f' = e^2
  wsx_exp_u(ws, FE_T1, FE_T0, FE_K);                   // f'^u = conj(g) = i
  wsx_mul(ws, FE_T0, FE_T1, false, FE_E, true);        // j = i*e
  wsx_mul(ws, FE_K, FE_T0, false, FE_D, true);         // k = j*h, h = d^-1                          K
  wsx_mul(ws, FE_L, FE_K, false, FE_B, false);         // l = k*b                                    L
  wsx_mul(ws, FE_T0, FE_K, false, FE_E, true);         // m = k*e
  wsx_mul(ws, FE_T0, FE_T0, false, FE_F, false);       // n = m*f
  wsx_frob_mul(ws, FE_T1, FE_L, 1, FE_T0);             // p = l^p * n
  wsx_frob_mul(ws, FE_T0, FE_K, 2, FE_T1);             // r = k^(p^2) * p
  wsx_mul(ws, FE_T1, FE_F, true, FE_L, false);         // t = f^-1 * l
  wsx_frob_mul(ws, FE_T1, FE_T1, 3, FE_T0);            // v = t^(p^3) * r
}

// Gt exponentiation by a canonical little-endian scalar (`Gt::pow(Fr)`), binary, cyclotomic squarings.
// Valid for elements of Gt (unitary); the reference only ever raises pairing outputs.
RB_FN Fp12 gt_pow_binary(const Fp12& base, const uint32_t k[8]) {
  Fp12 acc = fp12_one();
  for (int w = 7; w >= 0; w--) {
    uint32_t word = 0;
    switch (w) {
      case 0: word = k[0]; break;
      case 1: word = k[1]; break;
      case 2: word = k[2]; break;
      case 3: word = k[3]; break;
      case 4: word = k[4]; break;
      case 5: word = k[5]; break;
      case 6: word = k[6]; break;
      default: word = k[7]; break;
    }
    for (int b = 31; b >= 0; b--) {
      acc = fp12_cyclotomic_sqr(acc);
      if ((word >> b) & 1u) acc = fp12_mul(acc, base);
    }
  }
  return acc;
}

// The same power over signed 4-bit digits of k (an inverse in Gt is a conjugation): 8 table entries base^1..base^8,
// then 4 cyclotomic squarings + at most one multiplication per digit -- 256 squarings + <= 65 + 7 multiplications
// instead of 254 + ~127.  Exact arithmetic: the same field element as gt_pow_binary.
RB_FN Fp12 gt_pow_window(const Fp12& base, const uint32_t k[8]) {
  Fp12 tbl[8];
  tbl[0] = base;
  tbl[1] = fp12_cyclotomic_sqr(base);
  for (int i = 2; i < 8; i++) tbl[i] = fp12_mul_fn(tbl[i - 1], base);
  // digits d_j in [-8, 8), k = sum d_j 16^j, j = 0..64 (the top digit absorbs the last carry)
  signed char dg[65];
  uint32_t carry = 0;
  for (int j = 0; j < 64; j++) {
    uint32_t word = 0;
    switch (j >> 3) {
      case 0: word = k[0]; break;
      case 1: word = k[1]; break;
      case 2: word = k[2]; break;
      case 3: word = k[3]; break;
      case 4: word = k[4]; break;
      case 5: word = k[5]; break;
      case 6: word = k[6]; break;
      default: word = k[7]; break;
    }
    uint32_t v = ((word >> (4 * (j & 7))) & 15u) + carry;
    carry = v >= 8 ? 1u : 0u;
    dg[j] = (signed char)((int)v - (carry ? 16 : 0));
  }
  dg[64] = (signed char)carry;
  Fp12 acc = fp12_one();
  bool started = false;
  for (int j = 64; j >= 0; j--) {
    if (started)
      for (int q = 0; q < 4; q++) acc = fp12_cyclotomic_sqr_fn(acc);
    const int d = dg[j];
    if (d) {
      Fp12 e = tbl[(d > 0 ? d : -d) - 1];
      if (d < 0) e = fp12_conj(e);
      acc = started ? fp12_mul_fn(acc, e) : e;
      started = true;
    }
  }
  return acc;
}

}}  // namespace rabe::bn254
