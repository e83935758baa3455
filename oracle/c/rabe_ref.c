/*
 * rabe_ref.c -- CPU restatement of the reference's AC17 hot path in the REFERENCE'S OPERATION ORDER.
 *
 * TEST ORACLE / CPU BASELINE -- not product code.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (rabe_amd/) never does.  PARITY UNPINNED for group values
 * (see oracle/bn254.py header): the arithmetic crate `rabe-bn 0.4.23` is not in /root/reference; this
 * file is validated against the Python big-int oracle (tests/test_oracle_c.py), which is the ground truth.
 *
 * What is restated, statement by statement (paths relative to /root/reference):
 *   sha3_hash(g, label) = g * Fr::from_slice(SHA3-256(label))      src/utils/hash/mod.rs:10-20
 *   ac17::cp_encrypt group loops                                    src/schemes/ac17/mod.rs:289-369
 *   ac17::cp_decrypt group loops                                    src/schemes/ac17/mod.rs:398-418
 * with the cost profile of the reference's backend: every `G * Fr` is a plain left-to-right binary
 * double-and-add over Jacobian coordinates, every `pairing()` is a full Miller loop followed by its own
 * final exponentiation, hash-to-group is recomputed per row exactly where the Rust code recomputes it.
 * Plain C, 4 x 64-bit limbs, unsigned __int128; single-threaded like the reference.
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef struct { u64 l[4]; } fe;          /* Montgomery form, R = 2^256 */
typedef struct { fe a, b; } fe2;           /* a + b u */
typedef struct { fe2 c0, c1, c2; } fe6;
typedef struct { fe6 a, b; } fe12;         /* a + b w */
typedef struct { fe x, y, z; } g1j;        /* Jacobian, z = 0 infinity */
typedef struct { fe2 x, y, z; } g2j;

static const u64 P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const u64 RMOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const u64 BN_U = 4965661367192848881ull;

static u64 PINV, RINV;            /* -p^-1, -r^-1 mod 2^64 */
static fe FP_ONE, FP_R2, FR_ONE_, FR_R2;
static fe2 XI, TWIST_B, G1F[6], G3F[6];
static fe G2F[6];
static fe FP_TWO_INV;
static int inited = 0;

/* ------------------------------------------------------------------ generic 256-bit modular arithmetic */
static int geq(const u64 a[4], const u64 m[4]) {
  for (int i = 3; i >= 0; i--) { if (a[i] > m[i]) return 1; if (a[i] < m[i]) return 0; }
  return 1;
}
static void sub_n(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; r[i] = (u64)d; br = (u64)(d >> 64) & 1; }
}
static void mod_add(u64 r[4], const u64 a[4], const u64 b[4], const u64 m[4]) {
  u64 c = 0, t[4];
  for (int i = 0; i < 4; i++) { u128 s = (u128)a[i] + b[i] + c; t[i] = (u64)s; c = (u64)(s >> 64); }
  if (c || geq(t, m)) sub_n(r, t, m); else memcpy(r, t, 32);
}
static void mod_sub(u64 r[4], const u64 a[4], const u64 b[4], const u64 m[4]) {
  u64 br = 0, t[4];
  for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; t[i] = (u64)d; br = (u64)(d >> 64) & 1; }
  if (br) { u64 c = 0; for (int i = 0; i < 4; i++) { u128 s = (u128)t[i] + m[i] + c; t[i] = (u64)s; c = (u64)(s >> 64); } }
  memcpy(r, t, 32);
}
/* Montgomery product a*b/R mod m (a < m or b < m suffices for the bound; result < m) */
static void mont_mul(u64 r[4], const u64 a[4], const u64 b[4], const u64 m[4], u64 minv) {
  u64 t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u64 c = 0;
    for (int j = 0; j < 4; j++) { u128 x = (u128)a[j] * b[i] + t[j] + c; t[j] = (u64)x; c = (u64)(x >> 64); }
    u128 x = (u128)t[4] + c; t[4] = (u64)x; t[5] = (u64)(x >> 64);
    u64 q = t[0] * minv;
    x = (u128)q * m[0] + t[0]; c = (u64)(x >> 64);
    for (int j = 1; j < 4; j++) { x = (u128)q * m[j] + t[j] + c; t[j - 1] = (u64)x; c = (u64)(x >> 64); }
    x = (u128)t[4] + c; t[3] = (u64)x; t[4] = t[5] + (u64)(x >> 64);
  }
  if (t[4] || geq(t, m)) sub_n(r, t, m); else memcpy(r, t, 32);
}
static u64 neg_inv64(u64 m0) { u64 x = 1; for (int i = 0; i < 6; i++) x *= 2 - m0 * x; return (u64)0 - x; }

/* ------------------------------------------------------------------ Fp */
static void fp_add(fe* r, const fe* a, const fe* b) { mod_add(r->l, a->l, b->l, P); }
static void fp_sub(fe* r, const fe* a, const fe* b) { mod_sub(r->l, a->l, b->l, P); }
static void fp_mul(fe* r, const fe* a, const fe* b) { mont_mul(r->l, a->l, b->l, P, PINV); }
static void fp_sqr(fe* r, const fe* a) { mont_mul(r->l, a->l, a->l, P, PINV); }
static void fp_neg(fe* r, const fe* a) { fe z; memset(&z, 0, 32); mod_sub(r->l, z.l, a->l, P); }
static void fp_dbl(fe* r, const fe* a) { mod_add(r->l, a->l, a->l, P); }
static int fp_is_zero(const fe* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static int fp_eq(const fe* a, const fe* b) { return memcmp(a, b, 32) == 0; }
static void fp_pow(fe* r, const fe* a, const u64 e[4]) {
  fe acc = FP_ONE;
  for (int i = 255; i >= 0; i--) {
    fp_sqr(&acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, a);
  }
  *r = acc;
}
static void fp_inv(fe* r, const fe* a) { u64 e[4]; u64 two[4] = {2, 0, 0, 0}; sub_n(e, P, two); fp_pow(r, a, e); }
static void fp_from_u64(fe* r, u64 v) { u64 t[4] = {v, 0, 0, 0}; mont_mul(r->l, t, FP_R2.l, P, PINV); }
static void fp_from_le(fe* r, const uint8_t* b) { u64 t[4]; memcpy(t, b, 32); mont_mul(r->l, t, FP_R2.l, P, PINV); }
static void fp_to_le(uint8_t* b, const fe* a) { u64 o[4] = {1, 0, 0, 0}, t[4]; mont_mul(t, a->l, o, P, PINV); memcpy(b, t, 32); }

/* ------------------------------------------------------------------ Fr (scalars kept canonical except inside products) */
static void fr_from_be32_reduce(u64 out[4], const uint8_t d[32]) {
  /* Fr::from_slice: big-endian integer reduced mod r (mont_mul by R^2 then by 1) */
  u64 x[4], t[4], o[4] = {1, 0, 0, 0};
  for (int i = 0; i < 4; i++) { u64 w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | d[(3 - i) * 8 + j]; x[i] = w; }
  mont_mul(t, FR_R2.l, x, RMOD, RINV);
  mont_mul(out, t, o, RMOD, RINV);
}
static void fr_add(u64 r[4], const u64 a[4], const u64 b[4]) { mod_add(r, a, b, RMOD); }

/* ------------------------------------------------------------------ Fp2 */
static void f2_add(fe2* r, const fe2* a, const fe2* b) { fp_add(&r->a, &a->a, &b->a); fp_add(&r->b, &a->b, &b->b); }
static void f2_sub(fe2* r, const fe2* a, const fe2* b) { fp_sub(&r->a, &a->a, &b->a); fp_sub(&r->b, &a->b, &b->b); }
static void f2_neg(fe2* r, const fe2* a) { fp_neg(&r->a, &a->a); fp_neg(&r->b, &a->b); }
static void f2_dbl(fe2* r, const fe2* a) { fp_dbl(&r->a, &a->a); fp_dbl(&r->b, &a->b); }
static void f2_conj(fe2* r, const fe2* a) { r->a = a->a; fp_neg(&r->b, &a->b); }
static void f2_mul(fe2* r, const fe2* x, const fe2* y) {
  fe t0, t1, t2, s0, s1;
  fp_mul(&t0, &x->a, &y->a); fp_mul(&t1, &x->b, &y->b);
  fp_add(&s0, &x->a, &x->b); fp_add(&s1, &y->a, &y->b); fp_mul(&t2, &s0, &s1);
  fp_sub(&r->a, &t0, &t1); fp_sub(&t2, &t2, &t0); fp_sub(&r->b, &t2, &t1);
}
static void f2_sqr(fe2* r, const fe2* x) {
  fe s, d, t;
  fp_add(&s, &x->a, &x->b); fp_sub(&d, &x->a, &x->b); fp_mul(&t, &x->a, &x->b);
  fp_mul(&r->a, &s, &d); fp_dbl(&r->b, &t);
}
static void f2_mul_fp(fe2* r, const fe2* x, const fe* k) { fp_mul(&r->a, &x->a, k); fp_mul(&r->b, &x->b, k); }
static void f2_mul_xi(fe2* r, const fe2* x) { fe2 t; f2_mul(&t, x, &XI); *r = t; }
static void f2_inv(fe2* r, const fe2* x) {
  fe n, t, ni; fp_sqr(&n, &x->a); fp_sqr(&t, &x->b); fp_add(&n, &n, &t); fp_inv(&ni, &n);
  fp_mul(&r->a, &x->a, &ni); fp_mul(&t, &x->b, &ni); fp_neg(&r->b, &t);
}
static int f2_is_zero(const fe2* x) { return fp_is_zero(&x->a) && fp_is_zero(&x->b); }
static int f2_eq(const fe2* x, const fe2* y) { return fp_eq(&x->a, &y->a) && fp_eq(&x->b, &y->b); }
static void f2_pow(fe2* r, const fe2* x, const u64 e[4]) {
  fe2 acc; acc.a = FP_ONE; memset(&acc.b, 0, 32);
  for (int i = 255; i >= 0; i--) { f2_sqr(&acc, &acc); if ((e[i >> 6] >> (i & 63)) & 1) f2_mul(&acc, &acc, x); }
  *r = acc;
}

/* ------------------------------------------------------------------ Fp6 (schoolbook, v^3 = xi) */
static void f6_add(fe6* r, const fe6* a, const fe6* b) { f2_add(&r->c0, &a->c0, &b->c0); f2_add(&r->c1, &a->c1, &b->c1); f2_add(&r->c2, &a->c2, &b->c2); }
static void f6_sub(fe6* r, const fe6* a, const fe6* b) { f2_sub(&r->c0, &a->c0, &b->c0); f2_sub(&r->c1, &a->c1, &b->c1); f2_sub(&r->c2, &a->c2, &b->c2); }
static void f6_neg(fe6* r, const fe6* a) { f2_neg(&r->c0, &a->c0); f2_neg(&r->c1, &a->c1); f2_neg(&r->c2, &a->c2); }
static void f6_mul(fe6* r, const fe6* x, const fe6* y) {
  fe2 v0, v1, v2, t0, t1, t2, s0, s1;
  f2_mul(&v0, &x->c0, &y->c0); f2_mul(&v1, &x->c1, &y->c1); f2_mul(&v2, &x->c2, &y->c2);
  f2_add(&s0, &x->c1, &x->c2); f2_add(&s1, &y->c1, &y->c2); f2_mul(&t0, &s0, &s1); f2_sub(&t0, &t0, &v1); f2_sub(&t0, &t0, &v2);
  f2_add(&s0, &x->c0, &x->c1); f2_add(&s1, &y->c0, &y->c1); f2_mul(&t1, &s0, &s1); f2_sub(&t1, &t1, &v0); f2_sub(&t1, &t1, &v1);
  f2_add(&s0, &x->c0, &x->c2); f2_add(&s1, &y->c0, &y->c2); f2_mul(&t2, &s0, &s1); f2_sub(&t2, &t2, &v0); f2_sub(&t2, &t2, &v2);
  fe6 o;
  f2_mul_xi(&t0, &t0); f2_add(&o.c0, &v0, &t0);
  f2_mul_xi(&s0, &v2); f2_add(&o.c1, &t1, &s0);
  f2_add(&o.c2, &t2, &v1);
  *r = o;
}
static void f6_mul_v(fe6* r, const fe6* x) { fe6 o; f2_mul_xi(&o.c0, &x->c2); o.c1 = x->c0; o.c2 = x->c1; *r = o; }
static void f6_inv(fe6* r, const fe6* x) {
  fe2 c0, c1, c2, t, u;
  f2_sqr(&c0, &x->c0); f2_mul(&t, &x->c1, &x->c2); f2_mul_xi(&t, &t); f2_sub(&c0, &c0, &t);
  f2_sqr(&c1, &x->c2); f2_mul_xi(&c1, &c1); f2_mul(&t, &x->c0, &x->c1); f2_sub(&c1, &c1, &t);
  f2_sqr(&c2, &x->c1); f2_mul(&t, &x->c0, &x->c2); f2_sub(&c2, &c2, &t);
  f2_mul(&t, &x->c2, &c1); f2_mul(&u, &x->c1, &c2); f2_add(&t, &t, &u); f2_mul_xi(&t, &t);
  f2_mul(&u, &x->c0, &c0); f2_add(&t, &t, &u); f2_inv(&t, &t);
  f2_mul(&r->c0, &c0, &t); f2_mul(&r->c1, &c1, &t); f2_mul(&r->c2, &c2, &t);
}

/* ------------------------------------------------------------------ Fp12 */
static void f12_one(fe12* r) { memset(r, 0, sizeof *r); r->a.c0.a = FP_ONE; }
static void f12_mul(fe12* r, const fe12* x, const fe12* y) {
  fe6 t0, t1, t2, s0, s1;
  f6_mul(&t0, &x->a, &y->a); f6_mul(&t1, &x->b, &y->b);
  f6_add(&s0, &x->a, &x->b); f6_add(&s1, &y->a, &y->b); f6_mul(&t2, &s0, &s1);
  f6_sub(&t2, &t2, &t0); f6_sub(&t2, &t2, &t1);
  f6_mul_v(&s0, &t1); f6_add(&r->a, &t0, &s0); r->b = t2;
}
static void f12_sqr(fe12* r, const fe12* x) { fe12 t = *x; f12_mul(r, &t, &t); }
static void f12_conj(fe12* r, const fe12* x) { r->a = x->a; f6_neg(&r->b, &x->b); }
static void f12_inv(fe12* r, const fe12* x) {
  fe6 t0, t1; f6_mul(&t0, &x->a, &x->a); f6_mul(&t1, &x->b, &x->b); f6_mul_v(&t1, &t1); f6_sub(&t0, &t0, &t1); f6_inv(&t0, &t0);
  f6_mul(&r->a, &x->a, &t0); f6_mul(&t1, &x->b, &t0); f6_neg(&r->b, &t1);
}
static void f12_frob(fe12* r, const fe12* x, int k) {
  /* x^(p^k), k = 1..3: conjugate coefficients for odd k, multiply the w^j coefficient by gamma_k[j] */
  const fe2* c[6] = {&x->a.c0, &x->b.c0, &x->a.c1, &x->b.c1, &x->a.c2, &x->b.c2};   /* w^0..w^5 */
  fe2 o[6];
  for (int j = 0; j < 6; j++) {
    fe2 t = *c[j];
    if (k & 1) f2_conj(&t, &t);
    if (j == 0) o[j] = t;
    else if (k == 1) f2_mul(&o[j], &t, &G1F[j]);
    else if (k == 2) f2_mul_fp(&o[j], &t, &G2F[j]);
    else f2_mul(&o[j], &t, &G3F[j]);
  }
  r->a.c0 = o[0]; r->b.c0 = o[1]; r->a.c1 = o[2]; r->b.c1 = o[3]; r->a.c2 = o[4]; r->b.c2 = o[5];
}
static void f12_pow_u(fe12* r, const fe12* x) {   /* x^u */
  fe12 acc = *x;
  for (int i = 61; i >= 0; i--) { f12_sqr(&acc, &acc); if ((BN_U >> i) & 1) f12_mul(&acc, &acc, x); }
  *r = acc;
}
static void f12_pow(fe12* r, const fe12* x, const u64 e[4]) {   /* `Gt::pow(Fr)`: plain square-and-multiply */
  fe12 acc; f12_one(&acc);
  for (int i = 255; i >= 0; i--) { f12_sqr(&acc, &acc); if ((e[i >> 6] >> (i & 63)) & 1) f12_mul(&acc, &acc, x); }
  *r = acc;
}

/* ------------------------------------------------------------------ G1 / G2 Jacobian, a = 0 */
#define DEF_CURVE(PFX, F, PT, ADD, SUB, MUL, SQR, DBL, ISZ)                                            \
  static void PFX##_dbl(PT* r, const PT* p) {                                                          \
    if (ISZ(&p->z)) { *r = *p; return; }                                                               \
    F A, B, C, D, E, FF, t; SQR(&A, &p->x); SQR(&B, &p->y); SQR(&C, &B);                               \
    ADD(&t, &p->x, &B); SQR(&t, &t); SUB(&t, &t, &A); SUB(&t, &t, &C); DBL(&D, &t);                    \
    DBL(&E, &A); ADD(&E, &E, &A); SQR(&FF, &E);                                                        \
    PT o; DBL(&t, &D); SUB(&o.x, &FF, &t);                                                             \
    MUL(&o.z, &p->y, &p->z); DBL(&o.z, &o.z);                                                          \
    SUB(&t, &D, &o.x); MUL(&t, &E, &t); DBL(&C, &C); DBL(&C, &C); DBL(&C, &C); SUB(&o.y, &t, &C);      \
    *r = o;                                                                                            \
  }                                                                                                    \
  static void PFX##_add(PT* r, const PT* p, const PT* q) {                                             \
    if (ISZ(&p->z)) { *r = *q; return; }                                                               \
    if (ISZ(&q->z)) { *r = *p; return; }                                                               \
    F z1z1, z2z2, u1, u2, s1, s2, h, rr, i, j, v, t;                                                   \
    SQR(&z1z1, &p->z); SQR(&z2z2, &q->z); MUL(&u1, &p->x, &z2z2); MUL(&u2, &q->x, &z1z1);              \
    MUL(&s1, &p->y, &q->z); MUL(&s1, &s1, &z2z2); MUL(&s2, &q->y, &p->z); MUL(&s2, &s2, &z1z1);        \
    SUB(&h, &u2, &u1); SUB(&rr, &s2, &s1);                                                             \
    if (ISZ(&h)) { if (ISZ(&rr)) { PFX##_dbl(r, p); return; } memset(r, 0, sizeof *r); return; }       \
    DBL(&rr, &rr); DBL(&i, &h); SQR(&i, &i); MUL(&j, &h, &i); MUL(&v, &u1, &i);                        \
    PT o; SQR(&o.x, &rr); SUB(&o.x, &o.x, &j); DBL(&t, &v); SUB(&o.x, &o.x, &t);                       \
    SUB(&t, &v, &o.x); MUL(&t, &rr, &t); MUL(&s1, &s1, &j); DBL(&s1, &s1); SUB(&o.y, &t, &s1);         \
    ADD(&t, &p->z, &q->z); SQR(&t, &t); SUB(&t, &t, &z1z1); SUB(&t, &t, &z2z2); MUL(&o.z, &t, &h);     \
    *r = o;                                                                                            \
  }                                                                                                    \
  /* `G * Fr`: left-to-right binary double-and-add over the canonical scalar */                        \
  static void PFX##_mul(PT* r, const PT* p, const u64 k[4]) {                                          \
    PT acc; memset(&acc, 0, sizeof acc);                                                               \
    for (int i = 255; i >= 0; i--) {                                                                   \
      PFX##_dbl(&acc, &acc);                                                                           \
      if ((k[i >> 6] >> (i & 63)) & 1) PFX##_add(&acc, &acc, p);                                       \
    }                                                                                                  \
    *r = acc;                                                                                          \
  }
DEF_CURVE(g1, fe, g1j, fp_add, fp_sub, fp_mul, fp_sqr, fp_dbl, fp_is_zero)
DEF_CURVE(g2, fe2, g2j, f2_add, f2_sub, f2_mul, f2_sqr, f2_dbl, f2_is_zero)

static void g1_neg(g1j* r, const g1j* p) { *r = *p; fp_neg(&r->y, &p->y); }
static void g1_to_affine(fe* x, fe* y, const g1j* p) {
  if (fp_is_zero(&p->z)) { memset(x, 0, 32); memset(y, 0, 32); return; }
  fe zi, zi2; fp_inv(&zi, &p->z); fp_sqr(&zi2, &zi); fp_mul(x, &p->x, &zi2); fp_mul(&zi2, &zi2, &zi); fp_mul(y, &p->y, &zi2);
}
static void g2_to_affine(fe2* x, fe2* y, const g2j* p) {
  if (f2_is_zero(&p->z)) { memset(x, 0, 64); memset(y, 0, 64); return; }
  fe2 zi, zi2; f2_inv(&zi, &p->z); f2_sqr(&zi2, &zi); f2_mul(x, &p->x, &zi2); f2_mul(&zi2, &zi2, &zi); f2_mul(y, &p->y, &zi2);
}
static void g1_load(g1j* r, const uint8_t* b) {
  fp_from_le(&r->x, b); fp_from_le(&r->y, b + 32);
  if (fp_is_zero(&r->x) && fp_is_zero(&r->y)) memset(r, 0, sizeof *r); else r->z = FP_ONE;
}
static void g1_store(uint8_t* b, const g1j* p) { fe x, y; g1_to_affine(&x, &y, p); fp_to_le(b, &x); fp_to_le(b + 32, &y); }
static void g2_load(g2j* r, const uint8_t* b) {
  fp_from_le(&r->x.a, b); fp_from_le(&r->x.b, b + 32); fp_from_le(&r->y.a, b + 64); fp_from_le(&r->y.b, b + 96);
  if (f2_is_zero(&r->x) && f2_is_zero(&r->y)) memset(r, 0, sizeof *r); else { memset(&r->z, 0, 64); r->z.a = FP_ONE; }
}
static void g2_store(uint8_t* b, const g2j* p) {
  fe2 x, y; g2_to_affine(&x, &y, p); fp_to_le(b, &x.a); fp_to_le(b + 32, &x.b); fp_to_le(b + 64, &y.a); fp_to_le(b + 96, &y.b);
}
static void gt_load(fe12* r, const uint8_t* b) {
  fe2* c[6] = {&r->a.c0, &r->a.c1, &r->a.c2, &r->b.c0, &r->b.c1, &r->b.c2};
  for (int i = 0; i < 6; i++) { fp_from_le(&c[i]->a, b + 64 * i); fp_from_le(&c[i]->b, b + 64 * i + 32); }
}
static void gt_store(uint8_t* b, const fe12* x) {
  const fe2* c[6] = {&x->a.c0, &x->a.c1, &x->a.c2, &x->b.c0, &x->b.c1, &x->b.c2};
  for (int i = 0; i < 6; i++) { fp_to_le(b + 64 * i, &c[i]->a); fp_to_le(b + 64 * i + 32, &c[i]->b); }
}

/* ------------------------------------------------------------------ pairing: affine Miller loop + libff final exponentiation.
 * Lines in affine twist coordinates: slope lambda in Fp2 (one Fp2 inversion per step), line value
 *   l(P) = yP - lambda xP w + (lambda xT - yT) w^3    (w^3 = v w). */
static void line_mul(fe12* f, const fe2* lam, const fe2* xt, const fe2* yt, const fe* xp, const fe* yp) {
  fe12 l; memset(&l, 0, sizeof l);
  l.a.c0.a = *yp;                                  /* w^0 */
  fe2 t; f2_mul_fp(&t, lam, xp); f2_neg(&l.b.c0, &t);   /* w^1 */
  f2_mul(&t, lam, xt); f2_sub(&l.b.c1, &t, yt);        /* w^3 = v w */
  f12_mul(f, f, &l);
}
static void aff_double_step(fe12* f, fe2* x, fe2* y, const fe* xp, const fe* yp) {
  fe2 lam, t, x3, y3;
  f2_sqr(&t, x); f2_dbl(&lam, &t); f2_add(&lam, &lam, &t);       /* 3x^2 */
  f2_dbl(&t, y); f2_inv(&t, &t); f2_mul(&lam, &lam, &t);
  line_mul(f, &lam, x, y, xp, yp);
  f2_sqr(&x3, &lam); f2_sub(&x3, &x3, x); f2_sub(&x3, &x3, x);
  f2_sub(&t, x, &x3); f2_mul(&y3, &lam, &t); f2_sub(&y3, &y3, y);
  *x = x3; *y = y3;
}
static void aff_add_step(fe12* f, fe2* x, fe2* y, const fe2* qx, const fe2* qy, const fe* xp, const fe* yp) {
  fe2 lam, t, x3, y3;
  f2_sub(&lam, qy, y); f2_sub(&t, qx, x); f2_inv(&t, &t); f2_mul(&lam, &lam, &t);
  line_mul(f, &lam, x, y, xp, yp);
  f2_sqr(&x3, &lam); f2_sub(&x3, &x3, x); f2_sub(&x3, &x3, qx);
  f2_sub(&t, x, &x3); f2_mul(&y3, &lam, &t); f2_sub(&y3, &y3, y);
  *x = x3; *y = y3;
}
static void miller(fe12* f, const g1j* pj, const g2j* qj) {
  f12_one(f);
  if (fp_is_zero(&pj->z) || f2_is_zero(&qj->z)) return;
  fe xp, yp; fe2 qx, qy;
  g1_to_affine(&xp, &yp, pj); g2_to_affine(&qx, &qy, qj);
  fe2 x = qx, y = qy;
  u128 loop = (u128)6 * BN_U + 2;   /* 65 bits */
  for (int i = 63; i >= 0; i--) {
    f12_sqr(f, f);
    aff_double_step(f, &x, &y, &xp, &yp);
    if ((loop >> i) & 1) aff_add_step(f, &x, &y, &qx, &qy, &xp, &yp);
  }
  fe2 q1x, q1y, q2x, q2y, t;
  f2_conj(&t, &qx); f2_mul(&q1x, &t, &G1F[2]); f2_conj(&t, &qy); f2_mul(&q1y, &t, &G1F[3]);
  f2_mul_fp(&q2x, &qx, &G2F[2]); f2_mul_fp(&q2y, &qy, &G2F[3]); f2_neg(&q2y, &q2y);
  aff_add_step(f, &x, &y, &q1x, &q1y, &xp, &yp);
  aff_add_step(f, &x, &y, &q2x, &q2y, &xp, &yp);
}
static void final_exp(fe12* r, const fe12* fin) {
  fe12 f, t, a, b, c, d, e, ff, g, h, i, j, k, l, m, n, o, p, q, rr, s, uu;
  f12_conj(&t, fin); f12_inv(&f, fin); f12_mul(&f, &t, &f);       /* ^(p^6-1) */
  f12_frob(&t, &f, 2); f12_mul(&f, &t, &f);                          /* ^(p^2+1) */
  f12_pow_u(&a, &f); f12_conj(&a, &a);                               /* exp_by_neg_z */
  f12_sqr(&b, &a); f12_sqr(&c, &b); f12_mul(&d, &c, &b);
  f12_pow_u(&e, &d); f12_conj(&e, &e);
  f12_sqr(&ff, &e); f12_pow_u(&g, &ff); f12_conj(&g, &g);
  f12_conj(&h, &d); f12_conj(&i, &g);
  f12_mul(&j, &i, &e); f12_mul(&k, &j, &h); f12_mul(&l, &k, &b); f12_mul(&m, &k, &e); f12_mul(&n, &m, &f);
  f12_frob(&o, &l, 1); f12_mul(&p, &o, &n); f12_frob(&q, &k, 2); f12_mul(&rr, &q, &p);
  f12_conj(&s, &f); f12_mul(&t, &s, &l); f12_frob(&uu, &t, 3); f12_mul(r, &uu, &rr);
}
static void pairing(fe12* r, const g1j* p, const g2j* q) { fe12 m; miller(&m, p, q); final_exp(r, &m); }

/* ------------------------------------------------------------------ SHA3-256 (FIPS 202) */
static void keccakf(u64 s[25]) {
  static const u64 RC[24] = {1ull, 0x8082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x808bull, 0x80000001ull, 0x8000000080008081ull,
    0x8000000000008009ull, 0x8aull, 0x88ull, 0x80008009ull, 0x8000000aull, 0x8000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x800aull, 0x800000008000000aull, 0x8000000080008081ull,
    0x8000000000008080ull, 0x80000001ull, 0x8000000080008008ull};
  static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  for (int r = 0; r < 24; r++) {
    u64 bc[5], t;
    for (int i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
    for (int i = 0; i < 5; i++) { t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63)); for (int j = 0; j < 25; j += 5) s[j + i] ^= t; }
    t = s[1];
    for (int i = 0; i < 24; i++) { int j = PIL[i]; u64 b = s[j]; s[j] = (t << ROT[i]) | (t >> (64 - ROT[i])); t = b; }
    for (int j = 0; j < 25; j += 5) { for (int i = 0; i < 5; i++) bc[i] = s[j + i]; for (int i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5]; }
    s[0] ^= RC[r];
  }
}
static void sha3_256(uint8_t out[32], const uint8_t* in, size_t len) {
  u64 s[25]; memset(s, 0, sizeof s);
  uint8_t blk[136];
  while (len >= 136) { for (int i = 0; i < 17; i++) { u64 w; memcpy(&w, in + 8 * i, 8); s[i] ^= w; } keccakf(s); in += 136; len -= 136; }
  memset(blk, 0, 136); memcpy(blk, in, len); blk[len] ^= 0x06; blk[135] ^= 0x80;
  for (int i = 0; i < 17; i++) { u64 w; memcpy(&w, blk + 8 * i, 8); s[i] ^= w; }
  keccakf(s);
  memcpy(out, s, 32);
}
/* sha3_hash(g, label): g * Fr::from_slice(SHA3-256(label))   (src/utils/hash/mod.rs:10-20) */
static void sha3_hash_g1(g1j* r, const g1j* g, const char* label, size_t len) {
  uint8_t d[32]; u64 k[4];
  sha3_256(d, (const uint8_t*)label, len); fr_from_be32_reduce(k, d); g1_mul(r, g, k);
}

/* ------------------------------------------------------------------ init */
static void rabe_ref_init(void) {
  if (inited) return;
  PINV = neg_inv64(P[0]); RINV = neg_inv64(RMOD[0]);
  /* R mod m and R^2 mod m by repeated doubling */
  for (int which = 0; which < 2; which++) {
    const u64* m = which ? RMOD : P;
    u64 x[4] = {1, 0, 0, 0};
    fe one, r2;
    for (int i = 0; i < 512; i++) { mod_add(x, x, x, m); if (i == 255) memcpy(one.l, x, 32); }
    memcpy(r2.l, x, 32);
    if (which) { FR_ONE_ = one; FR_R2 = r2; } else { FP_ONE = one; FP_R2 = r2; }
  }
  fp_from_u64(&XI.a, 9); fp_from_u64(&XI.b, 1);
  fe three, two; fp_from_u64(&three, 3); fp_from_u64(&two, 2); fp_inv(&FP_TWO_INV, &two);
  fe2 xi_inv; f2_inv(&xi_inv, &XI); f2_mul_fp(&TWIST_B, &xi_inv, &three);
  /* (p-1)/6 */
  u64 e[4], pm1[4]; u64 o[4] = {1, 0, 0, 0}; sub_n(pm1, P, o);
  u128 rem = 0;
  for (int i = 3; i >= 0; i--) { u128 cur = (rem << 64) | pm1[i]; e[i] = (u64)(cur / 6); rem = cur % 6; }
  fe2 g1; f2_pow(&g1, &XI, e);
  fe2 g1c; f2_conj(&g1c, &g1);
  fe2 g2; f2_mul(&g2, &g1, &g1c);            /* xi^((p^2-1)/6) in Fp */
  fe2 g3; f2_mul(&g3, &g2, &g1);             /* xi^((p^3-1)/6) */
  memset(G1F, 0, sizeof G1F); memset(G2F, 0, sizeof G2F); memset(G3F, 0, sizeof G3F);
  G1F[0].a = FP_ONE; G2F[0] = FP_ONE; G3F[0].a = FP_ONE;
  for (int j = 1; j < 6; j++) { f2_mul(&G1F[j], &G1F[j - 1], &g1); fp_mul(&G2F[j], &G2F[j - 1], &g2.a); f2_mul(&G3F[j], &G3F[j - 1], &g3); }
  inited = 1;
}

/* ================================================================== exported API (canonical little-endian wire format) */
void rref_g1_mul(const uint8_t* p, const uint8_t* k, uint8_t* out) { rabe_ref_init(); g1j a, r; u64 kk[4]; g1_load(&a, p); memcpy(kk, k, 32); g1_mul(&r, &a, kk); g1_store(out, &r); }
void rref_g2_mul(const uint8_t* p, const uint8_t* k, uint8_t* out) { rabe_ref_init(); g2j a, r; u64 kk[4]; g2_load(&a, p); memcpy(kk, k, 32); g2_mul(&r, &a, kk); g2_store(out, &r); }
void rref_pairing(const uint8_t* p, const uint8_t* q, uint8_t* out) { rabe_ref_init(); g1j a; g2j b; fe12 r; g1_load(&a, p); g2_load(&b, q); pairing(&r, &a, &b); gt_store(out, &r); }
void rref_gt_pow(const uint8_t* a, const uint8_t* k, uint8_t* out) { rabe_ref_init(); fe12 x, r; u64 kk[4]; gt_load(&x, a); memcpy(kk, k, 32); f12_pow(&r, &x, kk); gt_store(out, &r); }
void rref_g1_add(const uint8_t* p, const uint8_t* q, uint8_t* out) { rabe_ref_init(); g1j a, b, r; g1_load(&a, p); g1_load(&b, q); g1_add(&r, &a, &b); g1_store(out, &r); }
void rref_g2_add(const uint8_t* p, const uint8_t* q, uint8_t* out) { rabe_ref_init(); g2j a, b, r; g2_load(&a, p); g2_load(&b, q); g2_add(&r, &a, &b); g2_store(out, &r); }
void rref_gt_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) { rabe_ref_init(); fe12 x, y, r; gt_load(&x, a); gt_load(&y, b); f12_mul(&r, &x, &y); gt_store(out, &r); }
void rref_gt_inv(const uint8_t* a, uint8_t* out) { rabe_ref_init(); fe12 x, r; gt_load(&x, a); f12_inv(&r, &x); gt_store(out, &r); }
void rref_sha3_256(const uint8_t* in, size_t len, uint8_t* out) { sha3_256(out, in, len); }
void rref_hash_fr(const char* label, size_t len, uint8_t* out) { rabe_ref_init(); uint8_t d[32]; u64 k[4]; sha3_256(d, (const uint8_t*)label, len); fr_from_be32_reduce(k, d); memcpy(out, k, 32); }

/* ac17::cp_encrypt group loops (ac17/mod.rs:289-369).  Inputs: pk (g, h_a[3], e_gh_ka[2]); the MSP as the
 * host computed it (n_rows x n_cols int8, row labels pi as NUL-terminated strings of stride label_stride);
 * explicit randomness s0, s1; the Gt msg.  Outputs c_0[3] (G2), c[n_rows][3] (G1), c_p (Gt). */
void rref_ac17_cp_encrypt(const uint8_t* g_, const uint8_t* h_a_, const uint8_t* e_gh_ka_, int n_rows, int n_cols, const int8_t* m,
                          const char* pi, int label_stride, const uint8_t* s_, const uint8_t* msg_, uint8_t* c0_out, uint8_t* c_out,
                          uint8_t* cp_out) {
  rabe_ref_init();
  g1j g; g1_load(&g, g_);
  u64 s[2][4], sum[4];
  memcpy(s[0], s_, 32); memcpy(s[1], s_ + 32, 32); fr_add(sum, s[0], s[1]);
  for (int i = 0; i < 3; i++) {                                              /* :297-302 */
    g2j h, r; g2_load(&h, h_a_ + 128 * i); g2_mul(&r, &h, i < 2 ? s[i] : sum); g2_store(c0_out + 128 * i, &r);
  }
  g1j* table = (g1j*)malloc(sizeof(g1j) * (size_t)n_cols * 6);               /* _hash_table :305-328 */
  for (int j = 0; j < n_cols; j++)
    for (int l = 0; l < 3; l++)
      for (int t = 0; t < 2; t++) {
        char lab[64]; int len = 0;
        lab[len++] = '0';
        { char num[16]; int nn = 0, v = j + 1; while (v) { num[nn++] = (char)('0' + v % 10); v /= 10; } while (nn) lab[len++] = num[--nn]; }
        lab[len++] = (char)('0' + l); lab[len++] = (char)('0' + t);
        sha3_hash_g1(&table[(j * 3 + l) * 2 + t], &g, lab, (size_t)len);
      }
  for (int i = 0; i < n_rows; i++) {                                          /* :330-356 */
    const char* name = pi + (size_t)i * label_stride;
    size_t nl = strlen(name);
    for (int l = 0; l < 3; l++) {
      g1j prod; memset(&prod, 0, sizeof prod);
      for (int t = 0; t < 2; t++) {
        char lab[300]; memcpy(lab, name, nl); lab[nl] = (char)('0' + l); lab[nl + 1] = (char)('0' + t);
        g1j hash; sha3_hash_g1(&hash, &g, lab, nl + 2);
        for (int j = 0; j < n_cols; j++) {
          int8_t mij = m[(size_t)i * n_cols + j];
          if (mij == 1) g1_add(&hash, &hash, &table[(j * 3 + l) * 2 + t]);
          else if (mij == -1) { g1j neg; g1_neg(&neg, &table[(j * 3 + l) * 2 + t]); g1_add(&hash, &hash, &neg); }
        }
        g1j term; g1_mul(&term, &hash, s[t]); g1_add(&prod, &prod, &term);
      }
      g1_store(c_out + ((size_t)i * 3 + l) * 64, &prod);
    }
  }
  free(table);
  fe12 cp, e, t; f12_one(&cp);                                                /* :357-368 */
  for (int i = 0; i < 2; i++) { gt_load(&e, e_gh_ka_ + 384 * i); f12_pow(&t, &e, s[i]); f12_mul(&cp, &cp, &t); }
  gt_load(&t, msg_); f12_mul(&cp, &cp, &t); gt_store(cp_out, &cp);
}

/* ac17::cp_decrypt group loops (ac17/mod.rs:398-418).  ct rows / sk rows selected by the host's
 * name matching (index lists), summed once per occurrence; 6 separate pairings, each with its own final exponentiation. */
void rref_ac17_cp_decrypt(const uint8_t* ct_c0, const uint8_t* ct_c, const uint8_t* ct_cp, const uint8_t* sk_k0, const uint8_t* sk_k,
                          const uint8_t* sk_kp, const uint32_t* ct_sel, int n_ct_sel, const uint32_t* sk_sel, int n_sk_sel, uint8_t* out) {
  rabe_ref_init();
  fe12 prod1, prod2, t; f12_one(&prod1); f12_one(&prod2);
  for (int i = 0; i < 3; i++) {
    g1j prod_h, prod_g, pt; memset(&prod_h, 0, sizeof prod_h); memset(&prod_g, 0, sizeof prod_g);
    for (int j = 0; j < n_ct_sel; j++) { g1_load(&pt, ct_c + ((size_t)ct_sel[j] * 3 + i) * 64); g1_add(&prod_g, &prod_g, &pt); }
    for (int j = 0; j < n_sk_sel; j++) { g1_load(&pt, sk_k + ((size_t)sk_sel[j] * 3 + i) * 64); g1_add(&prod_h, &prod_h, &pt); }
    g1j kp; g1_load(&kp, sk_kp + 64 * i); g1_add(&kp, &kp, &prod_h);
    g2j c0, k0; g2_load(&c0, ct_c0 + 128 * i); g2_load(&k0, sk_k0 + 128 * i);
    pairing(&t, &kp, &c0); f12_mul(&prod1, &prod1, &t);          /* :415 */
    pairing(&t, &prod_g, &k0); f12_mul(&prod2, &prod2, &t);      /* :416 */
  }
  fe12 cp, inv; gt_load(&cp, ct_cp); f12_inv(&inv, &prod1); f12_mul(&t, &prod2, &inv); f12_mul(&cp, &cp, &t);   /* :418 */
  gt_store(out, &cp);
}
