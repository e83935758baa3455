// BN254 extension tower on top of fp.h:
//   Fp2 = Fp[u]/(u^2+1),  Fp6 = Fp2[v]/(v^3 - xi),  Fp12 = Fp6[w]/(w^2 - v),  xi = 9 + u.
// This is the `Gt` arithmetic of the reference's math backend (`rabe_bn::Gt`: `*`, `pow`, `inverse`,
// used at src/schemes/ac17/mod.rs:357-360,415-418; bsw/mod.rs:234,291-294,308) plus what the
// Miller loop and the final exponentiation need.  Basis of Fp12 over Fp2: 1, v, v^2, w, vw, v^2 w.
#pragma once
#include "fp.h"

namespace rabe { namespace bn254 {

// ============================================================================ Fp2
struct Fp2 {
  Fp c0, c1;
};

RB_HD Fp2 fp2_zero() { return Fp2{zero<FpParams>(), zero<FpParams>()}; }
RB_HD Fp2 fp2_one() { return Fp2{one<FpParams>(), zero<FpParams>()}; }
RB_HD bool fp2_is_zero(const Fp2& a) { return is_zero(a.c0) & is_zero(a.c1); }
RB_HD bool fp2_eq(const Fp2& a, const Fp2& b) { return eq(a.c0, b.c0) & eq(a.c1, b.c1); }
RB_HD Fp2 fp2_add(const Fp2& a, const Fp2& b) { return Fp2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
RB_HD Fp2 fp2_sub(const Fp2& a, const Fp2& b) { return Fp2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
RB_HD Fp2 fp2_neg(const Fp2& a) { return Fp2{neg(a.c0), neg(a.c1)}; }
RB_HD Fp2 fp2_dbl(const Fp2& a) { return Fp2{dbl(a.c0), dbl(a.c1)}; }
RB_HD Fp2 fp2_half(const Fp2& a) { return Fp2{half(a.c0), half(a.c1)}; }
RB_HD Fp2 fp2_conj(const Fp2& a) { return Fp2{a.c0, neg(a.c1)}; }

// Karatsuba: 3 Fp multiplications.  Out of line; the four Fp operands travel in 32 VGPRs.
RB_FN Fp2 fp2_mul_regs(Fp a0, Fp a1, Fp b0, Fp b1) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RB_NO_LAZY_FP2)
  // lazy reduction: 3 plain products + 2 Montgomery reductions (fp.h: fp2_mul_lazy_raw); same canonical result
  RB_COUNT_ONE_MUL(); RB_COUNT_ONE_MUL(); RB_COUNT_ONE_MUL();
  uint32_t c0[8], c1[8];
  fp2_mul_lazy_raw(c0, c1, a0.v, a1.v, b0.v, b1.v);
  Fp2 rl;
#pragma unroll
  for (int i = 0; i < 8; i++) { rl.c0.v[i] = c0[i]; rl.c1.v[i] = c1[i]; }
  return rl;
#else
  Fp t0, t1, t2;
#ifdef RB_FP2_MUL_2WAY
  mul2_inl(t0, t1, a0, b0, a1, b1);
  t2 = mul_inl(add(a0, a1), add(b0, b1));
#else
  mul3_inl(t0, t1, t2, a0, b0, a1, b1, add(a0, a1), add(b0, b1));     // three independent products, interleaved
#endif
  Fp2 r;
  r.c0 = sub(t0, t1);
  r.c1 = sub(sub(t2, t0), t1);
  return r;
#endif
}
RB_HD Fp2 fp2_mul(const Fp2& a, const Fp2& b) { return fp2_mul_regs(a.c0, a.c1, b.c0, b.c1); }
// (a0+a1)(a0-a1) + 2 a0 a1 u: 2 Fp multiplications.
RB_FN Fp2 fp2_sqr_regs(Fp a0, Fp a1) {
  Fp t0, t1;
  mul2_inl(t0, t1, add(a0, a1), sub(a0, a1), a0, a1);
  return Fp2{t0, dbl(t1)};
}
RB_HD Fp2 fp2_sqr(const Fp2& a) { return fp2_sqr_regs(a.c0, a.c1); }
RB_FN Fp2 fp2_mul_fp_regs(Fp a0, Fp a1, Fp k) {
  Fp2 r;
  mul2_inl(r.c0, r.c1, a0, k, a1, k);
  return r;
}
RB_HD Fp2 fp2_mul_fp(const Fp2& a, const Fp& k) { return fp2_mul_fp_regs(a.c0, a.c1, k); }
// (c0 + c1 u)(9 + u) = (9 c0 - c1) + (c0 + 9 c1) u
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RB_NO_LIN9)
// x + 9 y -+ z mod p in ONE reduction (x optional): the value is built as a 9-limb integer below 12 p, its quotient by p is
// estimated from the top 26 bits (constants.h: RB_FP_LIN_RECIP -- floor(v / p) or one less, checked exhaustively by
// tools/gen_constants.py), q p is subtracted and one conditional subtraction finishes.  ~95 instructions instead of the ~190 of
// three doublings, two additions / subtractions and their five reductions.  Carry chains as in fp.h (no compiler wait states).
template <bool MINUS, bool HAS_X>
RB_HD Fp fp_lin9(const Fp& x, const Fp& y, const Fp& z) {
  uint32_t s0, s1, s2, s3, s4, s5, s6, s7, s8;
  // s = y << 3
  s0 = y.v[0] << 3;
  s1 = (y.v[1] << 3) | (y.v[0] >> 29); s2 = (y.v[2] << 3) | (y.v[1] >> 29); s3 = (y.v[3] << 3) | (y.v[2] >> 29);
  s4 = (y.v[4] << 3) | (y.v[3] >> 29); s5 = (y.v[5] << 3) | (y.v[4] >> 29); s6 = (y.v[6] << 3) | (y.v[5] >> 29);
  s7 = (y.v[7] << 3) | (y.v[6] >> 29); s8 = y.v[7] >> 29;
#define RB_ADD9(B)                                                                                                        \
  asm("v_add_co_u32 %0, vcc, %0, %9\n\tv_addc_co_u32 %1, vcc, %1, %10, vcc\n\tv_addc_co_u32 %2, vcc, %2, %11, vcc\n\t"       \
      "v_addc_co_u32 %3, vcc, %3, %12, vcc\n\tv_addc_co_u32 %4, vcc, %4, %13, vcc\n\tv_addc_co_u32 %5, vcc, %5, %14, vcc\n\t" \
      "v_addc_co_u32 %6, vcc, %6, %15, vcc\n\tv_addc_co_u32 %7, vcc, %7, %16, vcc\n\tv_addc_co_u32 %8, vcc, 0, %8, vcc"        \
      : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7), "+v"(s8)                        \
      : "v"(B.v[0]), "v"(B.v[1]), "v"(B.v[2]), "v"(B.v[3]), "v"(B.v[4]), "v"(B.v[5]), "v"(B.v[6]), "v"(B.v[7]) : "vcc")
  RB_ADD9(y);                       // 9 y
  if (HAS_X) RB_ADD9(x);
  if (!MINUS) {
    RB_ADD9(z);                     // < 11 p
  } else {
    // s -= z; negative (> -p): add p back
    uint32_t m;
    asm("v_sub_co_u32 %0, vcc, %0, %10\n\tv_subb_co_u32 %1, vcc, %1, %11, vcc\n\tv_subb_co_u32 %2, vcc, %2, %12, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %3, %13, vcc\n\tv_subb_co_u32 %4, vcc, %4, %14, vcc\n\tv_subb_co_u32 %5, vcc, %5, %15, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %6, %16, vcc\n\tv_subb_co_u32 %7, vcc, %7, %17, vcc\n\tv_subbrev_co_u32 %8, vcc, 0, %8, vcc\n\t"
        "v_subb_co_u32_e64 %9, vcc, 0, 0, vcc"
        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7), "+v"(s8), "=&v"(m)
        : "v"(z.v[0]), "v"(z.v[1]), "v"(z.v[2]), "v"(z.v[3]), "v"(z.v[4]), "v"(z.v[5]), "v"(z.v[6]), "v"(z.v[7]) : "vcc");
    uint32_t p0, p1, p2, p3, p4, p5, p6, p7;
    asm("v_and_b32 %9, %18, %17\n\tv_and_b32 %10, %19, %17\n\tv_and_b32 %11, %20, %17\n\tv_and_b32 %12, %21, %17\n\t"
        "v_and_b32 %13, %22, %17\n\tv_and_b32 %14, %23, %17\n\tv_and_b32 %15, %24, %17\n\tv_and_b32 %16, %25, %17\n\t"
        "v_add_co_u32 %0, vcc, %0, %9\n\tv_addc_co_u32 %1, vcc, %1, %10, vcc\n\tv_addc_co_u32 %2, vcc, %2, %11, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %12, vcc\n\tv_addc_co_u32 %4, vcc, %4, %13, vcc\n\tv_addc_co_u32 %5, vcc, %5, %14, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %15, vcc\n\tv_addc_co_u32 %7, vcc, %7, %16, vcc\n\tv_addc_co_u32 %8, vcc, 0, %8, vcc"
        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7), "+v"(s8), "=&v"(p0), "=&v"(p1), "=&v"(p2),
          "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7)
        : "v"(m), "i"(FpParams::mod(0)), "i"(FpParams::mod(1)), "i"(FpParams::mod(2)), "i"(FpParams::mod(3)), "i"(FpParams::mod(4)),
          "i"(FpParams::mod(5)), "i"(FpParams::mod(6)), "i"(FpParams::mod(7))
        : "vcc");
  }
#undef RB_ADD9
  // q = floor(s / p) or one less, from the top 26 bits (s < 12 p < 2^258)
  const uint32_t top = (s8 << 24) | (s7 >> 8);
  const uint32_t q = __umulhi(top << 6, (uint32_t)RB_FP_LIN_RECIP << 6) >> 24;
  // s -= q p   (q <= 11: the product's limbs come out of one 64-bit multiply-add chain)
  uint32_t w[9];
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    acc = (uint64_t)q * FpParams::mod(i) + (acc >> 32);
    w[i] = (uint32_t)acc;
  }
  w[8] = (uint32_t)(acc >> 32);
  asm("v_sub_co_u32 %0, vcc, %0, %9\n\tv_subb_co_u32 %1, vcc, %1, %10, vcc\n\tv_subb_co_u32 %2, vcc, %2, %11, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %3, %12, vcc\n\tv_subb_co_u32 %4, vcc, %4, %13, vcc\n\tv_subb_co_u32 %5, vcc, %5, %14, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %6, %15, vcc\n\tv_subb_co_u32 %7, vcc, %7, %16, vcc\n\tv_subb_co_u32 %8, vcc, %8, %17, vcc"
      : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7), "+v"(s8)
      : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]) : "vcc");
  // now s < 2 p (s8 == 0): one conditional subtraction
  uint32_t t[8] = {s0, s1, s2, s3, s4, s5, s6, s7};
  cond_sub_mod<FpParams>(t, 0);
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return r;
}
RB_HD Fp2 fp2_mul_xi(const Fp2& a) { return Fp2{fp_lin9<true, false>(a.c0, a.c0, a.c1), fp_lin9<false, false>(a.c1, a.c1, a.c0)}; }
// x + xi a
RB_HD Fp2 fp2_add_mul_xi(const Fp2& x, const Fp2& a) { return Fp2{fp_lin9<true, true>(x.c0, a.c0, a.c1), fp_lin9<false, true>(x.c1, a.c1, a.c0)}; }
#else
RB_HD Fp2 fp2_mul_xi(const Fp2& a) {
  Fp t0 = dbl(dbl(dbl(a.c0)));   // 8 c0
  Fp t1 = dbl(dbl(dbl(a.c1)));   // 8 c1
  Fp2 r;
  r.c0 = sub(add(t0, a.c0), a.c1);
  r.c1 = add(add(t1, a.c1), a.c0);
  return r;
}
RB_HD Fp2 fp2_add_mul_xi(const Fp2& x, const Fp2& a) { return fp2_add(x, fp2_mul_xi(a)); }
#endif
RB_FN Fp2 fp2_inv(const Fp2& a) {
  Fp n = add(sqr(a.c0), sqr(a.c1));
  Fp ni = inv(n);
  return Fp2{mul(a.c0, ni), neg(mul(a.c1, ni))};
}

// ============================================================================ Fp6
struct Fp6 {
  Fp2 a0, a1, a2;
};
RB_HD Fp6 fp6_zero() { return Fp6{fp2_zero(), fp2_zero(), fp2_zero()}; }
RB_HD Fp6 fp6_one() { return Fp6{fp2_one(), fp2_zero(), fp2_zero()}; }
RB_HD Fp6 fp6_add(const Fp6& a, const Fp6& b) { return Fp6{fp2_add(a.a0, b.a0), fp2_add(a.a1, b.a1), fp2_add(a.a2, b.a2)}; }
RB_HD Fp6 fp6_sub(const Fp6& a, const Fp6& b) { return Fp6{fp2_sub(a.a0, b.a0), fp2_sub(a.a1, b.a1), fp2_sub(a.a2, b.a2)}; }
RB_HD Fp6 fp6_neg(const Fp6& a) { return Fp6{fp2_neg(a.a0), fp2_neg(a.a1), fp2_neg(a.a2)}; }
RB_HD Fp6 fp6_dbl(const Fp6& a) { return Fp6{fp2_dbl(a.a0), fp2_dbl(a.a1), fp2_dbl(a.a2)}; }
// multiply by v: (a0 + a1 v + a2 v^2) v = xi a2 + a0 v + a1 v^2
RB_HD Fp6 fp6_mul_v(const Fp6& a) { return Fp6{fp2_mul_xi(a.a2), a.a0, a.a1}; }
RB_HD bool fp6_eq(const Fp6& a, const Fp6& b) { return fp2_eq(a.a0, b.a0) & fp2_eq(a.a1, b.a1) & fp2_eq(a.a2, b.a2); }

// Karatsuba, 6 Fp2 multiplications.
RB_MID Fp6 fp6_mul(const Fp6& a, const Fp6& b) {
  Fp2 v0 = fp2_mul(a.a0, b.a0);
  Fp2 v1 = fp2_mul(a.a1, b.a1);
  Fp2 v2 = fp2_mul(a.a2, b.a2);
  Fp2 t0 = fp2_mul(fp2_add(a.a1, a.a2), fp2_add(b.a1, b.a2));
  Fp2 t1 = fp2_mul(fp2_add(a.a0, a.a1), fp2_add(b.a0, b.a1));
  Fp2 t2 = fp2_mul(fp2_add(a.a0, a.a2), fp2_add(b.a0, b.a2));
  Fp6 r;
  r.a0 = fp2_add_mul_xi(v0, fp2_sub(fp2_sub(t0, v1), v2));
  r.a1 = fp2_add_mul_xi(fp2_sub(fp2_sub(t1, v0), v1), v2);
  r.a2 = fp2_add(fp2_sub(fp2_sub(t2, v0), v2), v1);
  return r;
}
// a * (b0 + b1 v): 5 Fp2 multiplications
RB_MID Fp6 fp6_mul_by_01(const Fp6& a, const Fp2& b0, const Fp2& b1) {
  Fp2 v0 = fp2_mul(a.a0, b0);
  Fp2 v1 = fp2_mul(a.a1, b1);
  Fp2 t0 = fp2_mul(fp2_add(a.a1, a.a2), b1);                    // a1 b1 + a2 b1
  Fp2 t1 = fp2_mul(fp2_add(a.a0, a.a1), fp2_add(b0, b1));       // a0b0 + a0b1 + a1b0 + a1b1
  Fp2 t2 = fp2_mul(a.a2, b0);
  Fp6 r;
  r.a0 = fp2_add_mul_xi(v0, fp2_sub(t0, v1));                   // a0b0 + xi a2b1
  r.a1 = fp2_sub(fp2_sub(t1, v0), v1);                          // a0b1 + a1b0
  r.a2 = fp2_add(t2, v1);                                       // a2b0 + a1b1
  return r;
}
RB_MID Fp6 fp6_mul_fp2(const Fp6& a, const Fp2& b) { return Fp6{fp2_mul(a.a0, b), fp2_mul(a.a1, b), fp2_mul(a.a2, b)}; }
// CH-SQR2
RB_MID Fp6 fp6_sqr(const Fp6& a) {
  Fp2 s0 = fp2_sqr(a.a0);
  Fp2 ab = fp2_mul(a.a0, a.a1);
  Fp2 s1 = fp2_dbl(ab);
  Fp2 s2 = fp2_sqr(fp2_add(fp2_sub(a.a0, a.a1), a.a2));
  Fp2 bc = fp2_mul(a.a1, a.a2);
  Fp2 s3 = fp2_dbl(bc);
  Fp2 s4 = fp2_sqr(a.a2);
  Fp6 r;
  r.a0 = fp2_add_mul_xi(s0, s3);
  r.a1 = fp2_add_mul_xi(s1, s4);
  r.a2 = fp2_sub(fp2_sub(fp2_add(fp2_add(s1, s2), s3), s0), s4);
  return r;
}
RB_FN Fp6 fp6_inv(const Fp6& a) {
  Fp2 c0 = fp2_sub(fp2_sqr(a.a0), fp2_mul_xi(fp2_mul(a.a1, a.a2)));
  Fp2 c1 = fp2_sub(fp2_mul_xi(fp2_sqr(a.a2)), fp2_mul(a.a0, a.a1));
  Fp2 c2 = fp2_sub(fp2_sqr(a.a1), fp2_mul(a.a0, a.a2));
  Fp2 t = fp2_add(fp2_mul(a.a0, c0), fp2_mul_xi(fp2_add(fp2_mul(a.a2, c1), fp2_mul(a.a1, c2))));
  Fp2 ti = fp2_inv(t);
  return Fp6{fp2_mul(c0, ti), fp2_mul(c1, ti), fp2_mul(c2, ti)};
}

// ============================================================================ Fp12
struct Fp12 {
  Fp6 c0, c1;
};
RB_HD Fp12 fp12_one() { return Fp12{fp6_one(), fp6_zero()}; }
RB_HD bool fp12_eq(const Fp12& a, const Fp12& b) { return fp6_eq(a.c0, b.c0) & fp6_eq(a.c1, b.c1); }
RB_HD bool fp12_is_zero(const Fp12& a) {
  return fp2_is_zero(a.c0.a0) & fp2_is_zero(a.c0.a1) & fp2_is_zero(a.c0.a2) & fp2_is_zero(a.c1.a0) & fp2_is_zero(a.c1.a1) & fp2_is_zero(a.c1.a2);
}
RB_HD Fp12 fp12_conj(const Fp12& a) { return Fp12{a.c0, fp6_neg(a.c1)}; }

RB_MID Fp12 fp12_mul(const Fp12& a, const Fp12& b) {
  Fp6 t0 = fp6_mul(a.c0, b.c0);
  Fp6 t1 = fp6_mul(a.c1, b.c1);
  Fp6 t2 = fp6_mul(fp6_add(a.c0, a.c1), fp6_add(b.c0, b.c1));
  Fp12 r;
  r.c0 = fp6_add(t0, fp6_mul_v(t1));
  r.c1 = fp6_sub(fp6_sub(t2, t0), t1);
  return r;
}
// complex squaring: 2 Fp6 multiplications
RB_MID Fp12 fp12_sqr(const Fp12& a) {
  Fp6 ab = fp6_mul(a.c0, a.c1);
  Fp6 t = fp6_mul(fp6_add(a.c0, a.c1), fp6_add(a.c0, fp6_mul_v(a.c1)));
  Fp12 r;
  r.c0 = fp6_sub(fp6_sub(t, ab), fp6_mul_v(ab));
  r.c1 = fp6_dbl(ab);
  return r;
}
RB_FN Fp12 fp12_inv(const Fp12& a) {
  Fp6 t = fp6_sub(fp6_sqr(a.c0), fp6_mul_v(fp6_sqr(a.c1)));
  Fp6 ti = fp6_inv(t);
  return Fp12{fp6_mul(a.c0, ti), fp6_neg(fp6_mul(a.c1, ti))};
}

// f * (l0 + l1 w + l3 w^3)  [w^3 = v w]: the sparse line value of the D-type twist.
// In Fp6[w] the line is c0 = (l0,0,0), c1 = (l1,l3,0).  13 Fp2 multiplications.
RB_MID Fp12 fp12_mul_by_line(const Fp12& f, const Fp2& l0, const Fp2& l1, const Fp2& l3) {
  Fp6 t0 = fp6_mul_fp2(f.c0, l0);                                 // f0 * c0
  Fp6 t1 = fp6_mul_by_01(f.c1, l1, l3);                           // f1 * c1
  Fp6 t2 = fp6_mul_by_01(fp6_add(f.c0, f.c1), fp2_add(l0, l1), l3);   // (f0+f1)(c0+c1)
  Fp12 r;
  r.c0 = fp6_add(t0, fp6_mul_v(t1));
  r.c1 = fp6_sub(fp6_sub(t2, t0), t1);
  return r;
}

// f * (a0 + a1 w + a3 w^3) * (b0 + b1 w + b3 w^3): the two sparse lines are multiplied first (6 Fp2 products, the
// result has c1.a2 = 0), then one 17-product multiplication -- 23 instead of 2 x 13.
//   lineA * lineB = (a0b0 + xi a3b3, a1b1, a1b3 + a3b1) + (a0b1 + a1b0, a0b3 + a3b0, 0) w
RB_MID Fp12 fp12_mul_by_two_lines(const Fp12& f, const Fp2& a0, const Fp2& a1, const Fp2& a3, const Fp2& b0, const Fp2& b1, const Fp2& b3) {
  Fp2 m00 = fp2_mul(a0, b0);
  Fp2 m11 = fp2_mul(a1, b1);
  Fp2 m33 = fp2_mul(a3, b3);
  Fp2 x01 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a0, a1), fp2_add(b0, b1)), m00), m11);
  Fp2 x03 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a0, a3), fp2_add(b0, b3)), m00), m33);
  Fp2 x13 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a1, a3), fp2_add(b1, b3)), m11), m33);
  Fp6 p0{fp2_add_mul_xi(m00, m33), m11, x13};
  Fp6 t0 = fp6_mul(f.c0, p0);
  Fp6 t1 = fp6_mul_by_01(f.c1, x01, x03);
  Fp6 t2 = fp6_mul(fp6_add(f.c0, f.c1), Fp6{fp2_add(p0.a0, x01), fp2_add(p0.a1, x03), p0.a2});
  Fp12 r;
  r.c0 = fp6_add(t0, fp6_mul_v(t1));
  r.c1 = fp6_sub(fp6_sub(t2, t0), t1);
  return r;
}

// ---------------------------------------------------------------------------- Frobenius
#define RB_FP_CONST(name, ...)                          \
  RB_HD Fp name() {                                    \
    constexpr uint32_t m[8] = __VA_ARGS__;             \
    Fp r;                                              \
    _Pragma("unroll") for (int i = 0; i < 8; i++) r.v[i] = m[i]; \
    return r;                                          \
  }
#define RB_FP2_CONST(name) \
  RB_HD Fp2 name() { return Fp2{name##_c0(), name##_c1()}; }
RB_FP_CONST(gamma1_1_c0, RB_GAMMA1_1_C0)
RB_FP_CONST(gamma1_1_c1, RB_GAMMA1_1_C1)
RB_FP2_CONST(gamma1_1)
RB_FP_CONST(gamma1_2_c0, RB_GAMMA1_2_C0)
RB_FP_CONST(gamma1_2_c1, RB_GAMMA1_2_C1)
RB_FP2_CONST(gamma1_2)
RB_FP_CONST(gamma1_3_c0, RB_GAMMA1_3_C0)
RB_FP_CONST(gamma1_3_c1, RB_GAMMA1_3_C1)
RB_FP2_CONST(gamma1_3)
RB_FP_CONST(gamma1_4_c0, RB_GAMMA1_4_C0)
RB_FP_CONST(gamma1_4_c1, RB_GAMMA1_4_C1)
RB_FP2_CONST(gamma1_4)
RB_FP_CONST(gamma1_5_c0, RB_GAMMA1_5_C0)
RB_FP_CONST(gamma1_5_c1, RB_GAMMA1_5_C1)
RB_FP2_CONST(gamma1_5)
RB_FP_CONST(gamma3_1_c0, RB_GAMMA3_1_C0)
RB_FP_CONST(gamma3_1_c1, RB_GAMMA3_1_C1)
RB_FP2_CONST(gamma3_1)
RB_FP_CONST(gamma3_2_c0, RB_GAMMA3_2_C0)
RB_FP_CONST(gamma3_2_c1, RB_GAMMA3_2_C1)
RB_FP2_CONST(gamma3_2)
RB_FP_CONST(gamma3_3_c0, RB_GAMMA3_3_C0)
RB_FP_CONST(gamma3_3_c1, RB_GAMMA3_3_C1)
RB_FP2_CONST(gamma3_3)
RB_FP_CONST(gamma3_4_c0, RB_GAMMA3_4_C0)
RB_FP_CONST(gamma3_4_c1, RB_GAMMA3_4_C1)
RB_FP2_CONST(gamma3_4)
RB_FP_CONST(gamma3_5_c0, RB_GAMMA3_5_C0)
RB_FP_CONST(gamma3_5_c1, RB_GAMMA3_5_C1)
RB_FP2_CONST(gamma3_5)
RB_FP_CONST(gamma2_1, RB_GAMMA2_1_C0)
RB_FP_CONST(gamma2_2, RB_GAMMA2_2_C0)
RB_FP_CONST(gamma2_3, RB_GAMMA2_3_C0)
RB_FP_CONST(gamma2_4, RB_GAMMA2_4_C0)
RB_FP_CONST(gamma2_5, RB_GAMMA2_5_C0)
RB_FP_CONST(twist_b_c0, RB_TWIST_B_C0)
RB_FP_CONST(twist_b_c1, RB_TWIST_B_C1)
RB_FP2_CONST(twist_b)
RB_FP_CONST(fp_three, RB_FP_THREE)
RB_FP_CONST(fp_two_inv, RB_FP_TWO_INV)

// a^p: conjugate each Fp2 coefficient, multiply the coefficient of w^k by gamma1_k = xi^(k(p-1)/6).
// Coefficient order by w-power: a0:w^0, a1:w^2, a2:w^4 | b0:w^1, b1:w^3, b2:w^5.
RB_MID Fp12 fp12_frob1(const Fp12& a) {
  Fp12 r;
  r.c0.a0 = fp2_conj(a.c0.a0);
  r.c0.a1 = fp2_mul(fp2_conj(a.c0.a1), gamma1_2());
  r.c0.a2 = fp2_mul(fp2_conj(a.c0.a2), gamma1_4());
  r.c1.a0 = fp2_mul(fp2_conj(a.c1.a0), gamma1_1());
  r.c1.a1 = fp2_mul(fp2_conj(a.c1.a1), gamma1_3());
  r.c1.a2 = fp2_mul(fp2_conj(a.c1.a2), gamma1_5());
  return r;
}
// a^(p^2): no conjugation, gamma2_k = xi^(k(p^2-1)/6) lies in Fp.
RB_MID Fp12 fp12_frob2(const Fp12& a) {
  Fp12 r;
  r.c0.a0 = a.c0.a0;
  r.c0.a1 = fp2_mul_fp(a.c0.a1, gamma2_2());
  r.c0.a2 = fp2_mul_fp(a.c0.a2, gamma2_4());
  r.c1.a0 = fp2_mul_fp(a.c1.a0, gamma2_1());
  r.c1.a1 = fp2_mul_fp(a.c1.a1, gamma2_3());
  r.c1.a2 = fp2_mul_fp(a.c1.a2, gamma2_5());
  return r;
}
// a^(p^3): conjugation, gamma3_k = xi^(k(p^3-1)/6).
RB_MID Fp12 fp12_frob3(const Fp12& a) {
  Fp12 r;
  r.c0.a0 = fp2_conj(a.c0.a0);
  r.c0.a1 = fp2_mul(fp2_conj(a.c0.a1), gamma3_2());
  r.c0.a2 = fp2_mul(fp2_conj(a.c0.a2), gamma3_4());
  r.c1.a0 = fp2_mul(fp2_conj(a.c1.a0), gamma3_1());
  r.c1.a1 = fp2_mul(fp2_conj(a.c1.a1), gamma3_3());
  r.c1.a2 = fp2_mul(fp2_conj(a.c1.a2), gamma3_5());
  return r;
}

// ---------------------------------------------------------------------------- cyclotomic subgroup
// Granger-Scott squaring for elements of the cyclotomic subgroup G_{phi_12}(p) (everything after
// the easy part of the final exponentiation, and every Gt value).  Fp12 seen as three Fp4 = Fp2[s]/(s^2-xi)
// pairs: (a0=c0.a0, a1=c1.a1), (b0=c1.a0, b1=c0.a2), (c0=c0.a1, c1=c1.a2).
RB_HD void fp4_sqr(Fp2& r0, Fp2& r1, const Fp2& a, const Fp2& b) {
  Fp2 t0 = fp2_sqr(a);
  Fp2 t1 = fp2_sqr(b);
  r0 = fp2_add_mul_xi(t0, t1);                            // a^2 + xi b^2
  r1 = fp2_sub(fp2_sub(fp2_sqr(fp2_add(a, b)), t0), t1);  // 2ab
}
RB_MID Fp12 fp12_cyclotomic_sqr(const Fp12& f) {
  Fp2 z0 = f.c0.a0, z4 = f.c0.a1, z3 = f.c0.a2;
  Fp2 z2 = f.c1.a0, z1 = f.c1.a1, z5 = f.c1.a2;
  Fp2 t0, t1, t2, t3, t4, t5;
  fp4_sqr(t0, t1, z0, z1);
  fp4_sqr(t2, t3, z2, z3);
  fp4_sqr(t4, t5, z4, z5);
  Fp12 r;
  // z0 = 3 t0 - 2 z0 ; z1 = 3 t1 + 2 z1
  r.c0.a0 = fp2_add(fp2_dbl(fp2_sub(t0, z0)), t0);
  r.c1.a1 = fp2_add(fp2_dbl(fp2_add(t1, z1)), t1);
  // z2 = 3 xi t5 + 2 z2 ; z3 = 3 t4 - 2 z3
  Fp2 x5 = fp2_mul_xi(t5);
  r.c1.a0 = fp2_add(fp2_dbl(fp2_add(x5, z2)), x5);
  r.c0.a2 = fp2_add(fp2_dbl(fp2_sub(t4, z3)), t4);
  // z4 = 3 t2 - 2 z4 ; z5 = 3 t3 + 2 z5
  r.c0.a1 = fp2_add(fp2_dbl(fp2_sub(t2, z4)), t2);
  r.c1.a2 = fp2_add(fp2_dbl(fp2_add(t3, z5)), t3);
  return r;
}

}}  // namespace rabe::bn254
