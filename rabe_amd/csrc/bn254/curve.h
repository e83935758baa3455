// BN254 group arithmetic: G1 over Fp (y^2 = x^3 + 3), G2 over Fp2 (y^2 = x^3 + 3/(9+u)).
// Replaces `rabe_bn::{G1, G2}` `+`, `-`, `* Fr` (call sites: src/schemes/ac17/mod.rs:219-260,300-348,
// 406-415; bsw/mod.rs:138-148,233-242; lsw/mod.rs:141-160,203-206; aw11/mod.rs:146,218,275-276).
// Jacobian coordinates internally (infinity: Z = 0); affine (x, y) at the boundary, infinity = (0, 0).
// One template serves both groups through the f* overload set below.
#pragma once
#include "tower.h"

namespace rabe { namespace bn254 {

// ---- field overload set
RB_HD Fp fadd(const Fp& a, const Fp& b) { return add(a, b); }
RB_HD Fp fsub(const Fp& a, const Fp& b) { return sub(a, b); }
RB_HD Fp fmul(const Fp& a, const Fp& b) { return mul(a, b); }
RB_HD Fp fsqr(const Fp& a) { return sqr(a); }
RB_HD Fp fdbl(const Fp& a) { return dbl(a); }
RB_HD Fp fneg(const Fp& a) { return neg(a); }
RB_HD Fp finv(const Fp& a) { return inv(a); }
RB_HD bool fis_zero(const Fp& a) { return is_zero(a); }
RB_HD bool feq(const Fp& a, const Fp& b) { return eq(a, b); }
RB_HD Fp2 fadd(const Fp2& a, const Fp2& b) { return fp2_add(a, b); }
RB_HD Fp2 fsub(const Fp2& a, const Fp2& b) { return fp2_sub(a, b); }
RB_HD Fp2 fmul(const Fp2& a, const Fp2& b) { return fp2_mul(a, b); }
RB_HD Fp2 fsqr(const Fp2& a) { return fp2_sqr(a); }
RB_HD Fp2 fdbl(const Fp2& a) { return fp2_dbl(a); }
RB_HD Fp2 fneg(const Fp2& a) { return fp2_neg(a); }
RB_HD Fp2 finv(const Fp2& a) { return fp2_inv(a); }
RB_HD bool fis_zero(const Fp2& a) { return fp2_is_zero(a); }
RB_HD bool feq(const Fp2& a, const Fp2& b) { return fp2_eq(a, b); }
template <class F> RB_HD F fzero();
template <class F> RB_HD F fone();
template <> RB_HD Fp fzero<Fp>() { return zero<FpParams>(); }
template <> RB_HD Fp fone<Fp>() { return one<FpParams>(); }
template <> RB_HD Fp2 fzero<Fp2>() { return fp2_zero(); }
template <> RB_HD Fp2 fone<Fp2>() { return fp2_one(); }

template <class F>
struct Aff {
  F x, y;   // infinity: x = y = 0
};
template <class F>
struct Jac {
  F x, y, z;   // infinity: z = 0
};
typedef Aff<Fp> G1Aff;
typedef Jac<Fp> G1Jac;
typedef Aff<Fp2> G2Aff;
typedef Jac<Fp2> G2Jac;

template <class F> RB_HD bool aff_is_inf(const Aff<F>& p) { return fis_zero(p.x) & fis_zero(p.y); }
template <class F> RB_HD bool jac_is_inf(const Jac<F>& p) { return fis_zero(p.z); }
template <class F> RB_HD Jac<F> jac_inf() { return Jac<F>{fone<F>(), fone<F>(), fzero<F>()}; }
template <class F> RB_HD Aff<F> aff_inf() { return Aff<F>{fzero<F>(), fzero<F>()}; }
template <class F> RB_HD Aff<F> aff_neg(const Aff<F>& p) { return Aff<F>{p.x, fneg(p.y)}; }
template <class F> RB_HD Jac<F> jac_neg(const Jac<F>& p) { return Jac<F>{p.x, fneg(p.y), p.z}; }
template <class F>
RB_HD Jac<F> aff_to_jac(const Aff<F>& p) {
  if (aff_is_inf(p)) return jac_inf<F>();
  return Jac<F>{p.x, p.y, fone<F>()};
}

// dbl-2009-l (a = 0): 2M + 5S
template <class F>
RB_MID Jac<F> jac_dbl(const Jac<F>& p) {
  F A = fsqr(p.x);
  F B = fsqr(p.y);
  F C = fsqr(B);
  F D = fsub(fsub(fsqr(fadd(p.x, B)), A), C);
  D = fdbl(D);
  F E = fadd(fdbl(A), A);
  F Fq = fsqr(E);
  Jac<F> r;
  r.x = fsub(Fq, fdbl(D));
  F C8 = fdbl(fdbl(fdbl(C)));
  r.z = fdbl(fmul(p.y, p.z));   // Y = 0 cannot happen on an odd-order group; Z = 0 stays 0
  r.y = fsub(fmul(E, fsub(D, r.x)), C8);
  return r;
}

// madd-2007-bl: Jacobian + affine, 7M + 4S, all special cases handled.
template <class F>
RB_MID Jac<F> jac_add_aff(const Jac<F>& p, const Aff<F>& q) {
  if (aff_is_inf(q)) return p;
  if (jac_is_inf(p)) return Jac<F>{q.x, q.y, fone<F>()};
  F Z1Z1 = fsqr(p.z);
  F U2 = fmul(q.x, Z1Z1);
  F S2 = fmul(fmul(q.y, p.z), Z1Z1);
  F H = fsub(U2, p.x);
  F rr = fsub(S2, p.y);
  if (fis_zero(H)) {
    if (fis_zero(rr)) return jac_dbl(p);
    return jac_inf<F>();
  }
  rr = fdbl(rr);
  F HH = fsqr(H);
  F I = fdbl(fdbl(HH));
  F J = fmul(H, I);
  F V = fmul(p.x, I);
  Jac<F> r;
  r.x = fsub(fsub(fsqr(rr), J), fdbl(V));
  r.y = fsub(fmul(rr, fsub(V, r.x)), fdbl(fmul(p.y, J)));
  r.z = fsub(fsub(fsqr(fadd(p.z, H)), Z1Z1), HH);
  return r;
}

// G1 form of madd-2007-bl for the fixed-base table kernels: the 11 field products are expanded in place (no call,
// so nothing has to sit in callee-saved registers around them) and issued as 1 + 5 interleaved pairs of
// independent products.  Same formulas and special cases as jac_add_aff, hence the same values.
// The doubling case of the mixed addition (never taken with honest inputs) as a function that passes and returns FIELD elements
// in registers, one coordinate per call: a call that takes the accumulator by reference, or returns a point through a memory
// slot, pins the caller's accumulator to the stack -- a store and a reload of the whole point in every loop iteration of the
// fixed-base and NAF kernels (19 GB of scratch write-back per launch of the row kernel).
RB_FN Fp g1_dbl_coord(Fp x, Fp y, Fp z, int which) {
  const Jac<Fp> d = jac_dbl(Jac<Fp>{x, y, z});
  return which == 0 ? d.x : which == 1 ? d.y : d.z;
}
RB_HD Jac<Fp> g1_madd_inl(const Jac<Fp>& p, const Aff<Fp>& q) {
  if (aff_is_inf(q)) return p;
  if (jac_is_inf(p)) return Jac<Fp>{q.x, q.y, fone<Fp>()};
  Fp Z1Z1 = mul_inl(p.z, p.z);
  Fp U2, T;
  mul2_inl(U2, T, q.x, Z1Z1, q.y, p.z);
  Fp H = sub(U2, p.x);
  Fp S2, HH;
  mul2_inl(S2, HH, T, Z1Z1, H, H);
  Fp rr = sub(S2, p.y);
  if (is_zero(H)) {
    if (is_zero(rr)) return Jac<Fp>{g1_dbl_coord(p.x, p.y, p.z, 0), g1_dbl_coord(p.x, p.y, p.z, 1), g1_dbl_coord(p.x, p.y, p.z, 2)};
    return jac_inf<Fp>();
  }
  rr = dbl(rr);
  Fp I = dbl(dbl(HH));
  Fp J, V;
  mul2_inl(J, V, H, I, p.x, I);
  Fp zh = add(p.z, H);
  Fp R2, ZH2;
  mul2_inl(R2, ZH2, rr, rr, zh, zh);
  Jac<Fp> r;
  r.x = sub(sub(R2, J), dbl(V));
  r.z = sub(sub(ZH2, Z1Z1), HH);
  Fp A, B;
  mul2_inl(A, B, rr, sub(V, r.x), p.y, J);
  r.y = sub(A, dbl(B));
  return r;
}

// add-2007-bl: Jacobian + Jacobian, 11M + 5S.
template <class F>
RB_MID Jac<F> jac_add(const Jac<F>& p, const Jac<F>& q) {
  if (jac_is_inf(q)) return p;
  if (jac_is_inf(p)) return q;
  F Z1Z1 = fsqr(p.z);
  F Z2Z2 = fsqr(q.z);
  F U1 = fmul(p.x, Z2Z2);
  F U2 = fmul(q.x, Z1Z1);
  F S1 = fmul(fmul(p.y, q.z), Z2Z2);
  F S2 = fmul(fmul(q.y, p.z), Z1Z1);
  F H = fsub(U2, U1);
  F rr = fsub(S2, S1);
  if (fis_zero(H)) {
    if (fis_zero(rr)) return jac_dbl(p);
    return jac_inf<F>();
  }
  rr = fdbl(rr);
  F I = fsqr(fdbl(H));
  F J = fmul(H, I);
  F V = fmul(U1, I);
  Jac<F> r;
  r.x = fsub(fsub(fsqr(rr), J), fdbl(V));
  r.y = fsub(fmul(rr, fsub(V, r.x)), fdbl(fmul(S1, J)));
  r.z = fmul(fsub(fsub(fsqr(fadd(p.z, q.z)), Z1Z1), Z2Z2), H);
  return r;
}

// to affine with a caller-supplied inverse of z (batch inversion happens outside)
template <class F>
RB_HD Aff<F> jac_to_aff_with_zinv(const Jac<F>& p, const F& zinv) {
  if (jac_is_inf(p)) return aff_inf<F>();
  F zi2 = fsqr(zinv);
  return Aff<F>{fmul(p.x, zi2), fmul(p.y, fmul(zi2, zinv))};
}
template <class F>
RB_FN Aff<F> jac_to_aff(const Jac<F>& p) {
  if (jac_is_inf(p)) return aff_inf<F>();
  return jac_to_aff_with_zinv(p, finv(p.z));
}

// on-curve test for affine inputs
template <class F> RB_HD F curve_b();
template <> RB_HD Fp curve_b<Fp>() { return fp_three(); }
template <> RB_HD Fp2 curve_b<Fp2>() { return twist_b(); }
template <class F>
RB_HD bool aff_on_curve(const Aff<F>& p) {
  if (aff_is_inf(p)) return true;
  return feq(fsqr(p.y), fadd(fmul(fsqr(p.x), p.x), curve_b<F>()));
}

// ---------------------------------------------------------------------------------------------
// Variable-base scalar multiplication, left-to-right binary double-and-add over a canonical
// (non-Montgomery) little-endian scalar k < 2^256.  No table, nothing runtime-indexed; lanes with
// different scalars diverge only on the (masked) add.  The value equals the reference's `G * Fr`.
template <class F>
RB_FN Jac<F> jac_mul_binary(const Aff<F>& base, const uint32_t k[8]) {
  Jac<F> acc = jac_inf<F>();
  for (int w = 7; w >= 0; w--) {
    uint32_t word = 0;
    switch (w) {   // keep k[] indices compile-time constant
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
      acc = jac_dbl(acc);   // doubling infinity (Z = 0) stays infinity
      if ((word >> b) & 1u) acc = jac_add_aff(acc, base);
    }
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// Variable-base scalar multiplication over the non-adjacent form of k: digit_i = h_(i+1) - k_(i+1) with h = 3k
// (two bit masks, no table, nothing runtime-indexed): 254 doublings + ~85 mixed additions with +-base instead of
// 254 + ~127.  Leading zero digits cost nothing, so small coefficients (the Lagrange coefficients of binary gates
// are 2 and -1 -- callers hand -1 over as "negate the base, k = 1") are cheap.  Same group element as `G * Fr`.
RB_HD void naf_masks(const uint32_t k[8], uint32_t pos[8], uint32_t neg[8]) {
  uint32_t h[8];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t two = (k[i] << 1) | (i ? (k[i - 1] >> 31) : 0u);
    h[i] = addc32(k[i], two, c);          // k < 2^254: 3k < 2^256, no carry out
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t hp = (h[i] >> 1) | (i < 7 ? (h[i + 1] << 31) : 0u);
    const uint32_t kp = (k[i] >> 1) | (i < 7 ? (k[i + 1] << 31) : 0u);
    pos[i] = hp & ~kp;
    neg[i] = ~hp & kp;
  }
}
RB_HD uint32_t word_sel8(const uint32_t a[8], int w) {
  switch (w) {   // keeps the indices compile-time constant (registers, not scratch)
    case 0: return a[0];
    case 1: return a[1];
    case 2: return a[2];
    case 3: return a[3];
    case 4: return a[4];
    case 5: return a[5];
    case 6: return a[6];
    default: return a[7];
  }
}
template <class F> RB_HD Jac<F> jac_madd_fast(const Jac<F>& p, const Aff<F>& q) { return jac_add_aff(p, q); }
// G1 form of dbl-2009-l: the seven products as three interleaved pairs + one (same formulas as jac_dbl, hence the same values);
// the variable-base kernels run two waves per SIMD, where a lone product chain leaves issue slots empty (tools/ubench_mac.hip)
RB_HD Jac<Fp> g1_dbl_inl(const Jac<Fp>& p) {
  Fp A, B;
  mul2_inl(A, B, p.x, p.x, p.y, p.y);
  Fp C, YZ;
  mul2_inl(C, YZ, B, B, p.y, p.z);
  const Fp xb = add(p.x, B);
  const Fp E = add(dbl(A), A);
  Fp T, Fq;
  mul2_inl(T, Fq, xb, xb, E, E);
  const Fp D = dbl(sub(sub(T, A), C));
  Jac<Fp> r;
  r.x = sub(Fq, dbl(D));
  r.z = dbl(YZ);                                  // Y = 0 cannot happen on an odd-order group; Z = 0 stays 0
  const Fp C8 = dbl(dbl(dbl(C)));
  r.y = sub(mul_inl(E, sub(D, r.x)), C8);
  return r;
}
template <class F> RB_HD Jac<F> jac_dbl_fast(const Jac<F>& p) { return jac_dbl(p); }
template <> RB_HD Jac<Fp> jac_dbl_fast<Fp>(const Jac<Fp>& p) { return g1_dbl_inl(p); }
template <> RB_HD Jac<Fp> jac_madd_fast<Fp>(const Jac<Fp>& p, const Aff<Fp>& q) { return g1_madd_inl(p, q); }
template <class F>
RB_FN Jac<F> jac_mul_naf_plain(const Aff<F>& base, const uint32_t k[8]) {
  uint32_t pos[8], neg[8];
  naf_masks(k, pos, neg);
  Jac<F> acc = jac_inf<F>();
  if (aff_is_inf(base)) return acc;
  bool started = false;
  for (int w = 7; w >= 0; w--) {
    const uint32_t pw = word_sel8(pos, w), nw = word_sel8(neg, w);
    if (!started && !(pw | nw)) continue;
    for (int b = 31; b >= 0; b--) {
      if (started) acc = jac_dbl_fast(acc);
      const uint32_t pb = (pw >> b) & 1u, nb = (nw >> b) & 1u;
      if (pb | nb) {
        Aff<F> q = base;
        if (nb) q.y = fneg(q.y);
        acc = started ? jac_madd_fast(acc, q) : Jac<F>{q.x, q.y, fone<F>()};
        started = true;
      }
    }
  }
  return acc;
}


// ---------------------------------------------------------------------------------------------
// GLV on G1: the curve y^2 = x^3 + 3 has the endomorphism phi(x, y) = (beta x, y) = lambda (x, y) (beta, lambda: cube roots of
// unity mod p, mod r).  k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^130, so k P = k1 P + k2 phi(P) runs as a two-term sum with
// shared doublings: ~128 doublings + ~85 mixed additions instead of 254 + ~85.  The decomposition rounds with the precomputed
// g_i = floor(2^256 |b_i| / r) (constants.h); whatever the rounding, k1 + k2 lambda = k (mod r) holds exactly, so the result is the
// same group element as `G * Fr` -- only the size of k1, k2 depends on it.  Small scalars come out as k1 = k, k2 = 0.
template <int NA, int NB>
RB_HD void limbs_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {      // out[NA + NB] = a[NA] * b[NB]
#pragma unroll
  for (int i = 0; i < NA + NB; i++) out[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
      out[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    out[i + NB] = (uint32_t)carry;
  }
}
// |x| and the sign of a 256-bit two's-complement value
RB_HD bool limbs_abs8(uint32_t x[8]) {
  const bool neg_ = (x[7] >> 31) != 0;
  if (neg_) {
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = subb32(0u, x[i], borrow);
  }
  return neg_;
}
RB_HD void glv_decompose(const uint32_t k[8], uint32_t k1[8], bool& neg1, uint32_t k2[8], bool& neg2) {
  constexpr uint32_t A1[2] = RB_GLV_A1, NB1[4] = RB_GLV_NEG_B1, A2[4] = RB_GLV_A2, B2[2] = RB_GLV_B2, G1C[3] = RB_GLV_G1, G2C[5] = RB_GLV_G2;
  uint32_t a1[2], nb1[4], a2[4], b2[2], g1[3], g2[5];
#pragma unroll
  for (int i = 0; i < 2; i++) { a1[i] = A1[i]; b2[i] = B2[i]; }
#pragma unroll
  for (int i = 0; i < 4; i++) { nb1[i] = NB1[i]; a2[i] = A2[i]; }
#pragma unroll
  for (int i = 0; i < 3; i++) g1[i] = G1C[i];
#pragma unroll
  for (int i = 0; i < 5; i++) g2[i] = G2C[i];
  uint32_t kk[8];
#pragma unroll
  for (int i = 0; i < 8; i++) kk[i] = k[i];
  uint32_t p1[11], p2[13];
  limbs_mul<8, 3>(kk, g1, p1);
  limbs_mul<8, 5>(kk, g2, p2);
  uint32_t c1[3], c2[5];                     // c_i = floor(k g_i / 2^256)
#pragma unroll
  for (int i = 0; i < 3; i++) c1[i] = p1[8 + i];
#pragma unroll
  for (int i = 0; i < 5; i++) c2[i] = p2[8 + i];
  uint32_t t1[5], t2[9], u1[7], u2[7];
  limbs_mul<3, 2>(c1, a1, t1);               // c1 a1
  limbs_mul<5, 4>(c2, a2, t2);               // c2 a2   (< 2^256)
  limbs_mul<3, 4>(c1, nb1, u1);              // c1 |b1|
  limbs_mul<5, 2>(c2, b2, u2);               // c2 b2
  // k1 = k - c1 a1 - c2 a2, k2 = c1 |b1| - c2 b2: small signed values, exact in 256-bit two's complement
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) k1[i] = subb32(kk[i], i < 5 ? t1[i] : 0u, borrow);
  borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) k1[i] = subb32(k1[i], t2[i], borrow);
  borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) k2[i] = subb32(i < 7 ? u1[i] : 0u, i < 7 ? u2[i] : 0u, borrow);
  neg1 = limbs_abs8(k1);
  neg2 = limbs_abs8(k2);
}
RB_FN Jac<Fp> jac_mul_glv_g1(const Aff<Fp>& base, const uint32_t k[8]) {
  Jac<Fp> acc = jac_inf<Fp>();
  if (aff_is_inf(base)) return acc;
  uint32_t k1[8], k2[8];
  bool neg1, neg2;
  glv_decompose(k, k1, neg1, k2, neg2);
  uint32_t p1[8], n1[8], p2[8], n2[8];
  naf_masks(k1, p1, n1);
  naf_masks(k2, p2, n2);
  constexpr uint32_t BETA[8] = RB_GLV_BETA;
  Fp beta;
#pragma unroll
  for (int i = 0; i < 8; i++) beta.v[i] = BETA[i];
  Aff<Fp> b1 = base, b2{mul(base.x, beta), base.y};
  if (neg1) b1.y = neg(b1.y);
  if (neg2) b2.y = neg(b2.y);
  bool started = false;
  for (int w = 4; w >= 0; w--) {             // |k1|, |k2| < 2^130: NAF digits up to bit 130
    const uint32_t pw1 = word_sel8(p1, w), nw1 = word_sel8(n1, w), pw2 = word_sel8(p2, w), nw2 = word_sel8(n2, w);
    if (!started && !(pw1 | nw1 | pw2 | nw2)) continue;
    for (int b = 31; b >= 0; b--) {
      if (started) acc = g1_dbl_inl(acc);
      const uint32_t d1p = (pw1 >> b) & 1u, d1n = (nw1 >> b) & 1u, d2p = (pw2 >> b) & 1u, d2n = (nw2 >> b) & 1u;
      if (d1p | d1n) {
        Aff<Fp> q = b1;
        if (d1n) q.y = neg(q.y);
        acc = g1_madd_inl(acc, q);           // handles acc = infinity and the doubling / cancelling cases
        started = true;
      }
      if (d2p | d2n) {
        Aff<Fp> q = b2;
        if (d2n) q.y = neg(q.y);
        acc = g1_madd_inl(acc, q);
        started = true;
      }
    }
  }
  return acc;
}
// Which one where (measured): the decrypt kernels scale by Lagrange coefficients, which for the gates the benchmarks use are
// signed binomials (<= 97 bits at 100 leaves, <= 196 at 200) -- the plain chain skips their leading zeros and GLV, whose halves are
// ~127 bits whatever the size of k, gains nothing there (k_bsw_dec_pairs 27.7 ms either way, k_lsw_dec_pairs 27.5 vs 26.9 ms).
// GLV serves `rhip_g1_mul` (uniform 254-bit scalars: 1.8 k instead of 3.2 k Fp multiplications for the binary chain).
template <class F> RB_HD Jac<F> jac_mul_naf(const Aff<F>& base, const uint32_t k[8]) { return jac_mul_naf_plain(base, k); }

// Multi-scalar multiplication  sum_j k_j * P_j  with the doublings shared (Straus over the NAFs of the k_j):
// 254 doublings for the whole sum + ~85 mixed additions per term, instead of a full multiplication per term.
// TERMS provides  int count() const;  Aff<F> base(int j) const;  uint32_t pos_word(int j, int w) / neg_word(int j, int w) const
// (word w of term j's NAF masks: computed once per term by the caller -- naf_masks -- and kept outside the register file).
template <class F, class TERMS>
RB_FN Jac<F> jac_msm_naf(TERMS terms) {
  const int n = terms.count();
  Jac<F> acc = jac_inf<F>();
  bool started = false;
  for (int w = 7; w >= 0; w--) {
    for (int b = 31; b >= 0; b--) {
      if (started) acc = jac_dbl_fast(acc);
      for (int j = 0; j < n; j++) {
        const uint32_t pw = terms.pos_word(j, w), nw = terms.neg_word(j, w);
        const uint32_t pb = (pw >> b) & 1u, nb = (nw >> b) & 1u;
        if (pb | nb) {
          Aff<F> q = terms.base(j);
          if (aff_is_inf(q)) continue;
          if (nb) q.y = fneg(q.y);
          acc = jac_madd_fast(acc, q);        // handles acc = infinity and the doubling / cancelling cases
          started = true;
        }
      }
    }
  }
  return acc;
}

}}  // namespace rabe::bn254
