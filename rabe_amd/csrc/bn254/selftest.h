// Known-answer self-test of the field / tower / curve arithmetic, run once per device when the first context is created
// (engine.hip: rhip_ctx_create -> k_selftest).
//
// Why: the fast build issues carry-dependent v_addc / v_subb / v_mad_u64_u32 back to back inside single asm statements (fp.h:
// RB_CH8, the generated multiply-accumulate routines), i.e. without the wait states LLVM's gfx940+ hazard table asks for -- measured
// safe (tools/ubench_addc.hip, tests/test_gpu_carry_interlock.py), but measured on the boxes the tests ran on.  This check runs the
// same instruction forms on adversarial limb patterns on the device the product actually opened, on every SIMD, and compares with
// constants computed with exact integers by tools/gen_selftest.py (selftest_gen.h).  A mismatch refuses the context.
//
// Lane l of a wave works on a = V[l & 7], b = V[(l >> 3) & 7] (V: Montgomery residues with saturated limbs, p - 1, p - 2, ...) and
// folds every result limb into a 32-bit digest; the expected digests of the 64 lanes are compiled in.
#pragma once
#include "coop6.h"
#include "selftest_gen.h"

namespace rabe { namespace bn254 {

RB_HD Fp st_vec(int i) {
  constexpr uint32_t v[8][8] = RB_SELFTEST_VECS;
  Fp r;
#pragma unroll
  for (int k = 0; k < 8; k++)
    r.v[k] = i == 0 ? v[0][k] : i == 1 ? v[1][k] : i == 2 ? v[2][k] : i == 3 ? v[3][k] : i == 4 ? v[4][k] : i == 5 ? v[5][k] : i == 6 ? v[6][k] : v[7][k];
  return r;
}
RB_HD uint32_t st_fold(uint32_t h, const Fp& x) {
#pragma unroll
  for (int i = 0; i < 8; i++) h = (((h << 5) | (h >> 27)) ^ x.v[i]) + 0x9e3779b9u;
  return h;
}
RB_HD uint32_t st_fold(uint32_t h, const Fp2& x) { return st_fold(st_fold(h, x.c0), x.c1); }
RB_HD uint32_t st_fold(uint32_t h, const Fp6& x) { return st_fold(st_fold(st_fold(h, x.a0), x.a1), x.a2); }
RB_HD uint32_t st_fold(uint32_t h, const Fp12& x) { return st_fold(st_fold(h, x.c0), x.c1); }
RB_FN uint32_t selftest_digest(int lane) {
  const Fp a = st_vec(lane & 7), b = st_vec((lane >> 3) & 7);
  uint32_t h = 0x6a09e667u ^ (uint32_t)lane;
  // Fp: one-chain products, the additive chains
  const Fp m = mul(a, b), s = sqr(a);
  h = st_fold(h, m); h = st_fold(h, s);
  h = st_fold(h, add(a, b)); h = st_fold(h, sub(a, b)); h = st_fold(h, neg(a)); h = st_fold(h, dbl(b)); h = st_fold(h, half(a));
  // Fq2: lazy three-chain product + two-chain reduction, two-chain squaring, the one-pass xi reductions
  const Fp2 X{a, b}, Y{b, m};
  const Fp2 P = fp2_mul(X, Y), Q = fp2_sqr(X), Xi = fp2_mul_xi(X), Ax = fp2_add_mul_xi(Y, X), Kf = fp2_mul_fp(X, s);
  h = st_fold(h, P); h = st_fold(h, Q); h = st_fold(h, Xi); h = st_fold(h, Ax); h = st_fold(h, Kf);
  // a dependent chain through the Montgomery product
  Fp x = a;
#pragma unroll 1
  for (int i = 0; i < 12; i++) { x = add(mul(x, x), b); h = st_fold(h, x); }
  h = st_fold(h, inv(add(a, b)));
  // Fq12 and the G2 doubling step (the formulas are field operations: the inputs need not be points or cyclotomic)
  const Fp12 f{Fp6{X, Y, P}, Fp6{Q, Xi, Ax}}, g{Fp6{P, Q, X}, Fp6{Y, Ax, Xi}};
  h = st_fold(h, fp12_mul(f, g));
  h = st_fold(h, fp12_sqr(f));
  h = st_fold(h, fp12_mul_by_line(f, X, Y, P));
  h = st_fold(h, fp12_cyclotomic_sqr(g));
  G2Hom T{X, Y, Fp2{b, a}};
  const LineCoeffs l = g2hom_double(T);
  h = st_fold(h, T.x); h = st_fold(h, T.y); h = st_fold(h, T.z);
  h = st_fold(h, l.cy); h = st_fold(h, l.cx); h = st_fold(h, l.c0);
  // the multi-product lazy sums of the six-lane kernels (coop6.h)
  Wide3 W;
  wide3_zero(W);
  wide3_mac(W, X, Y); wide3_mac(W, P, Q); wide3_mac(W, Xi, Ax); wide3_mac(W, Kf, X); wide3_mac(W, Y, Y); wide3_mac(W, Q, P);
  h = st_fold(h, wide3_finish(W));
  return h;
}

}}  // namespace rabe::bn254
