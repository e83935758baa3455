// TEST INFRASTRUCTURE: runs the engine's __host__ __device__ math (rabe_amd/csrc/bn254/*.h) on the CPU
// so the arithmetic can be checked against the Python oracle in a container without a GPU.
// Built with `hipcc --cuda-host-only`; never loaded by the product (rabe_amd/), which only ever
// launches the same functions as HIP kernels.
#include "../../rabe_amd/csrc/bn254/io.h"
#include "../../rabe_amd/csrc/bn254/coop3.h"
#include "../../rabe_amd/csrc/bn254/coop6.h"
#include "../../rabe_amd/csrc/bn254/selftest.h"
#include "../../rabe_amd/csrc/bn254/pairing29.h"
#include "../../rabe_amd/csrc/bn254/pairing29p.h"
#include <thread>
#include <pthread.h>
#include <string.h>

using namespace rabe::bn254;

extern "C" { unsigned long long rb_mul_counter = 0; unsigned long long rb_rr_mad_counter = 0; }

extern "C" {

unsigned long long hs_rr_mad_counter_reset() { unsigned long long v = ::rb_rr_mad_counter; ::rb_rr_mad_counter = 0; return v; }
unsigned long long hs_mul_counter_reset() { unsigned long long v = ::rb_mul_counter; ::rb_mul_counter = 0; return v; }

void hs_fp_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) { store_fp(out, mul(load_fp(a), load_fp(b))); }
void hs_fp_add(const uint32_t* a, const uint32_t* b, uint32_t* out) { store_fp(out, add(load_fp(a), load_fp(b))); }
void hs_fp_sub(const uint32_t* a, const uint32_t* b, uint32_t* out) { store_fp(out, sub(load_fp(a), load_fp(b))); }
void hs_fp_neg(const uint32_t* a, uint32_t* out) { store_fp(out, neg(load_fp(a))); }
void hs_fp_inv(const uint32_t* a, uint32_t* out) { store_fp(out, inv(load_fp(a))); }
void hs_fr_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) { store_fr(out, mul(load_fr(a), load_fr(b))); }
void hs_fr_inv(const uint32_t* a, uint32_t* out) { store_fr(out, inv(load_fr(a))); }
void hs_fr_reduce256(const uint32_t* a, uint32_t* out) { store_fr(out, to_mont_reduce256<FrParams>(a)); }

void hs_fp2_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) { store_fp2(out, fp2_mul(load_fp2(a), load_fp2(b))); }
void hs_fp2_sqr(const uint32_t* a, uint32_t* out) { store_fp2(out, fp2_sqr(load_fp2(a))); }
void hs_fp2_inv(const uint32_t* a, uint32_t* out) { store_fp2(out, fp2_inv(load_fp2(a))); }
void hs_fp2_mul_xi(const uint32_t* a, uint32_t* out) { store_fp2(out, fp2_mul_xi(load_fp2(a))); }

void hs_fp12_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) { store_gt(out, fp12_mul(load_gt(a), load_gt(b))); }
void hs_fp12_sqr(const uint32_t* a, uint32_t* out) { store_gt(out, fp12_sqr(load_gt(a))); }
void hs_fp12_inv(const uint32_t* a, uint32_t* out) { store_gt(out, fp12_inv(load_gt(a))); }
void hs_fp12_frob(const uint32_t* a, int k, uint32_t* out) {
  Fp12 x = load_gt(a);
  store_gt(out, k == 1 ? fp12_frob1(x) : (k == 2 ? fp12_frob2(x) : fp12_frob3(x)));
}
void hs_fp12_cyclotomic_sqr(const uint32_t* a, uint32_t* out) { store_gt(out, fp12_cyclotomic_sqr(load_gt(a))); }
void hs_fp12_mul_by_line(const uint32_t* f, const uint32_t* l0, const uint32_t* l1, const uint32_t* l3, uint32_t* out) {
  store_gt(out, fp12_mul_by_line(load_gt(f), load_fp2(l0), load_fp2(l1), load_fp2(l3)));
}

void hs_g1_add(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  store_g1(out, jac_to_aff(jac_add_aff(aff_to_jac(load_g1(a)), load_g1(b))));
}
void hs_g1_madd_inl(const uint32_t* a, const uint32_t* b, const uint32_t* zscale, uint32_t* out) {
  Fp z = load_fp(zscale);
  G1Aff pa = load_g1(a);
  Fp z2 = sqr(z);
  G1Jac ja = aff_is_inf(pa) ? jac_inf<Fp>() : G1Jac{mul(pa.x, z2), mul(pa.y, mul(z2, z)), z};
  store_g1(out, jac_to_aff(g1_madd_inl(ja, load_g1(b))));
}
void hs_g1_add_jac(const uint32_t* a, const uint32_t* b, const uint32_t* zscale, uint32_t* out) {
  // exercise the Jacobian+Jacobian path with non-trivial Z on both sides: scale (x,y,1) -> (x z^2, y z^3, z)
  Fp z = load_fp(zscale);
  G1Aff pa = load_g1(a), pb = load_g1(b);
  G1Jac ja = aff_to_jac(pa), jb = aff_to_jac(pb);
  if (!aff_is_inf(pa)) { Fp z2 = sqr(z); ja = G1Jac{mul(pa.x, z2), mul(pa.y, mul(z2, z)), z}; }
  if (!aff_is_inf(pb)) { Fp zz = add(z, one<FpParams>()); Fp z2 = sqr(zz); jb = G1Jac{mul(pb.x, z2), mul(pb.y, mul(z2, zz)), zz}; }
  store_g1(out, jac_to_aff(jac_add(ja, jb)));
}
void hs_g1_mul(const uint32_t* p, const uint32_t* k, uint32_t* out) { store_g1(out, jac_to_aff(jac_mul_binary(load_g1(p), k))); }
void hs_g2_add(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  store_g2(out, jac_to_aff(jac_add_aff(aff_to_jac(load_g2(a)), load_g2(b))));
}
void hs_g2_mul(const uint32_t* p, const uint32_t* k, uint32_t* out) { store_g2(out, jac_to_aff(jac_mul_binary(load_g2(p), k))); }
int hs_g1_on_curve(const uint32_t* p) { return aff_on_curve(load_g1(p)); }
int hs_g2_on_curve(const uint32_t* p) { return aff_on_curve(load_g2(p)); }

void hs_miller(const uint32_t* p, const uint32_t* q, uint32_t* out) {
  G1Aff P = load_g1(p);
  store_gt(out, miller_loop(miller_p_from_aff(P), aff_is_inf(P), load_g2(q)));
}
void hs_final_exp(const uint32_t* f, uint32_t* out) { store_gt(out, final_exponentiation(load_gt(f))); }
void hs_pairing(const uint32_t* p, const uint32_t* q, uint32_t* out) {
  G1Aff P = load_g1(p);
  store_gt(out, final_exponentiation(miller_loop(miller_p_from_aff(P), aff_is_inf(P), load_g2(q))));
}
// pairing with P handed over in Jacobian form scaled by z (checks the no-inversion path)
void hs_pairing_jac(const uint32_t* p, const uint32_t* zscale, const uint32_t* q, uint32_t* out) {
  G1Aff P = load_g1(p);
  Fp z = load_fp(zscale);
  Fp z2 = sqr(z);
  G1Jac J{mul(P.x, z2), mul(P.y, mul(z2, z)), z};
  store_gt(out, final_exponentiation(miller_loop(miller_p_from_jac(J), aff_is_inf(P), load_g2(q))));
}
struct HostLineLoad { const LineCoeffs* l; LineCoeffs operator()(int k) const { return l[k]; } };
void hs_pairing_prepared(const uint32_t* p, const uint32_t* q, uint32_t* out) {
  G1Aff P = load_g1(p);
  G2Aff Q = load_g2(q);
  LineCoeffs lines[RB_MILLER_LINES];
  g2_prepare_lines(Q, lines);
  store_gt(out, final_exponentiation(miller_loop_prepared(miller_p_from_aff(P), aff_is_inf(P), aff_is_inf(Q), HostLineLoad{lines})));
}
// the accumulator's home (LDS on the device): two halves + the parked Fq6
struct HostHome {
  Fp6* F;
  Fp6 ld_f6(int h) const { return F[h]; }
  void st_f6(int h, const Fp6& v) const { F[h] = v; }
  Fp6 ld_x() const { return F[2]; }
  void st_x(const Fp6& v) const { F[2] = v; }
  Fp2 ld_f2(int i) const { return (&F[i / 3].a0)[i % 3]; }
  void st_f2(int i, const Fp2& v) const { (&F[i / 3].a0)[i % 3] = v; }
  void fence() const {}
};
struct HostWs {
  Fp12* slots;
  Fp6* F;
  Fp12 ld(int i) const { return slots[i]; }
  void st(int i, const Fp12& v) const { slots[i] = v; }
  Fp6 ld6(int i, int h) const { return h ? slots[i].c1 : slots[i].c0; }
  void st6(int i, int h, const Fp6& v) const { (h ? slots[i].c1 : slots[i].c0) = v; }
  HostHome home() const { return HostHome{F}; }
};
void hs_final_exp_ws(const uint32_t* f, uint32_t* out) {
  Fp12 slots[FE_SLOTS];
  slots[FE_T0] = load_gt(f);
  Fp6 F[3];
  final_exponentiation_ws(HostWs{slots, F});
  store_gt(out, slots[FE_T1]);
}
void hs_g2_prepare(const uint32_t* q, uint32_t* out /* first line, 48 words */) {
  LineCoeffs lines[RB_MILLER_LINES];
  g2_prepare_lines(load_g2(q), lines);
  store_fp2(out, lines[0].cy); store_fp2(out + 16, lines[0].cx); store_fp2(out + 32, lines[0].c0);
}
// ---- what AFFINE walking steps would cost.  The 88 line triples of a G2 point by affine chord-and-tangent
// steps; the step's inversion is NOT computed here but its input collected (a device version would invert the denominators of all of a
// lane's walking pairs, and of the block, together): `hs_g2_prepare_affine` counts everything else -- the step formulas plus the three
// Fq2 products per element that Montgomery's simultaneous inversion adds -- and returns lines whose pairing value must equal the
// projective ones' (they differ by Fq2 factors, which the final exponentiation removes).
static int affine_lines(const G2Aff& q, LineCoeffs* out, int trick_products_per_step) {
  G2Aff t = q;
  const G2Aff qn = aff_neg(q);
  int n = 0;
  auto dbl = [&]() {
    const Fp2 x2 = fp2_sqr(t.x);
    const Fp2 num = fp2_add(fp2_dbl(x2), x2);                    // 3 x^2
    const Fp2 den = fp2_dbl(t.y);
    const unsigned long long before = ::rb_mul_counter;
    const Fp2 inv = fp2_inv(den);
    ::rb_mul_counter = before;                                   // the inversion itself is shared: not charged to the step
    for (int k = 0; k < trick_products_per_step; k++) (void)fp2_mul(num, den);          // Montgomery's trick: 3 products per element
    const Fp2 lam = fp2_mul(num, inv);
    const Fp2 x3 = fp2_sub(fp2_sqr(lam), fp2_dbl(t.x));
    const Fp2 lx = fp2_mul(lam, t.x);
    const Fp2 y3 = fp2_sub(fp2_sub(lx, fp2_mul(lam, x3)), t.y);
    // line through T with slope lam, evaluated as cy * yP + cx * xP w + c0 w^3: cy = 1, cx = -lam, c0 = lam x - y
    out[n++] = LineCoeffs{fp2_one(), fp2_neg(lam), fp2_sub(lx, t.y)};
    t = G2Aff{x3, y3};
  };
  auto add = [&](const G2Aff& r) {
    const Fp2 num = fp2_sub(r.y, t.y), den = fp2_sub(r.x, t.x);
    const unsigned long long before = ::rb_mul_counter;
    const Fp2 inv = fp2_inv(den);
    ::rb_mul_counter = before;
    for (int k = 0; k < trick_products_per_step; k++) (void)fp2_mul(num, den);
    const Fp2 lam = fp2_mul(num, inv);
    const Fp2 x3 = fp2_sub(fp2_sub(fp2_sqr(lam), t.x), r.x);
    const Fp2 lx = fp2_mul(lam, t.x);
    const Fp2 y3 = fp2_sub(fp2_sub(lx, fp2_mul(lam, x3)), t.y);
    out[n++] = LineCoeffs{fp2_one(), fp2_neg(lam), fp2_sub(lx, t.y)};
    t = G2Aff{x3, y3};
  };
  for (int i = RB_ATE_NAF_LEN - 2; i >= 0; i--) {
    dbl();
    const bool pos = (i < 64) && ((RB_ATE_NAF_POS >> i) & 1ull);
    const bool ngt = (i < 64) && ((RB_ATE_NAF_NEG >> i) & 1ull);
    if (pos | ngt) add(pos ? q : qn);
  }
  add(g2_frob1(q));
  add(aff_neg(g2_frob2(q)));
  return n;
}
void hs_g2_prepare_affine(const uint32_t* q, uint32_t* out /* first line, 48 words */) {
  LineCoeffs lines[RB_MILLER_LINES];
  affine_lines(load_g2(q), lines, 3);
  store_fp2(out, lines[0].cy); store_fp2(out + 16, lines[0].cx); store_fp2(out + 32, lines[0].c0);
}
void hs_pairing_affine_lines(const uint32_t* p, const uint32_t* q, uint32_t* out) {
  G1Aff P = load_g1(p);
  G2Aff Q = load_g2(q);
  LineCoeffs lines[RB_MILLER_LINES];
  affine_lines(Q, lines, 0);
  store_gt(out, final_exponentiation(miller_loop_prepared(miller_p_from_aff(P), aff_is_inf(P), aff_is_inf(Q), HostLineLoad{lines})));
}
// FE( miller_pair(pa, prepared qa; pb, qb) ) = e(pa, qa) * e(pb, qb)
void hs_pairing_pair(const uint32_t* pa, const uint32_t* qa, const uint32_t* pb, const uint32_t* qb, uint32_t* out) {
  G1Aff PA = load_g1(pa), PB = load_g1(pb);
  G2Aff QA = load_g2(qa), QB = load_g2(qb);
  LineCoeffs lines[RB_MILLER_LINES];
  g2_prepare_lines(QA, lines);
  // B enters Jacobian-scaled like the decrypt kernel's row sums: (x z^2, y z^3, z) with z = 3
  Fp z = add(add(one<FpParams>(), one<FpParams>()), one<FpParams>());
  Fp z2 = sqr(z);
  G1Jac JB{mul(PB.x, z2), mul(PB.y, mul(z2, z)), aff_is_inf(PB) ? zero<FpParams>() : z};
  store_gt(out, final_exponentiation(miller_loop_pair(miller_p_from_aff(PA), aff_is_inf(PA) || aff_is_inf(QA), HostLineLoad{lines},
                                                      miller_p_from_jac(JB), jac_is_inf(JB), QB)));
}
struct HostPark {
  Fp* a;
  Fp ld(int i) const { return a[i]; }
  void st(int i, const Fp& v) const { a[i] = v; }
};
// the parked form of hs_pairing_pair (both P's enter Jacobian-scaled)
void hs_pairing_pair_parked(const uint32_t* pa, const uint32_t* qa, const uint32_t* pb, const uint32_t* qb, uint32_t* out) {
  G1Aff PA = load_g1(pa), PB = load_g1(pb);
  G2Aff QA = load_g2(qa), QB = load_g2(qb);
  LineCoeffs lines[RB_MILLER_LINES];
  g2_prepare_lines(QA, lines);
  Fp z = add(add(one<FpParams>(), one<FpParams>()), one<FpParams>());
  Fp z2 = sqr(z);
  G1Jac JA{mul(PA.x, z2), mul(PA.y, mul(z2, z)), aff_is_inf(PA) ? zero<FpParams>() : z};
  G1Jac JB{mul(PB.x, z2), mul(PB.y, mul(z2, z)), aff_is_inf(PB) ? zero<FpParams>() : z};
  Fp park[PK_FPS];
  HostPark pk{park};
  pk_st_p(pk, PK_PA, miller_p_from_jac(JA));
  pk_st_p(pk, PK_PB, miller_p_from_jac(JB));
  pk_st2(pk, PK_QB, QB.x);
  pk_st2(pk, PK_QB + 2, QB.y);
  store_gt(out, final_exponentiation(miller_loop_pair_parked(pk, jac_is_inf(JA) || aff_is_inf(QA), HostLineLoad{lines},
                                                             jac_is_inf(JB) || aff_is_inf(QB))));
}

void hs_g1_mul_naf(const uint32_t* p, const uint32_t* k, uint32_t* out) { store_g1(out, jac_to_aff(jac_mul_naf(load_g1(p), k))); }
void hs_g1_mul_glv(const uint32_t* p, const uint32_t* k, uint32_t* out) { store_g1(out, jac_to_aff(jac_mul_glv_g1(load_g1(p), k))); }
void hs_g2_mul_naf(const uint32_t* p, const uint32_t* k, uint32_t* out) { store_g2(out, jac_to_aff(jac_mul_naf(load_g2(p), k))); }
}
// GLV decomposition (bn254/curve.h): out = |k1| (8 words), |k2| (8 words), sign of k1, sign of k2
extern "C" void hs_glv_decompose(const uint32_t* k, uint32_t* out) {
  bool n1, n2;
  glv_decompose(k, out, n1, out + 8, n2);
  out[16] = n1 ? 1u : 0u;
  out[17] = n2 ? 1u : 0u;
}
extern "C" void hs_g1_mul_naf_plain(const uint32_t* p, const uint32_t* k, uint32_t* out) { store_g1(out, jac_to_aff(jac_mul_naf_plain(load_g1(p), k))); }
// the carry-capture plan the device multiplication uses (bn254/fp.h: ColumnPlan): field 0 = Fp, 1 = Fr; kind 0 = a bare
// reduction (redc2), 1 = a full product of two reduced operands.  out: safe[16], then last_safe.
extern "C" void hs_column_plan(int field, int kind, uint16_t* out) {
  auto emit = [&](auto plan) { for (int k = 0; k < 16; k++) out[k] = plan.safe[k]; out[16] = plan.last_safe; };
  if (field == 0) {
    if (kind == 0) emit(ColumnPlan<FpParams>(false, 0, 0));
    else emit(ColumnPlan<FpParams>(true, FpParams::mod(7) + 1, FpParams::mod(7) + 1));
  } else {
    if (kind == 0) emit(ColumnPlan<FrParams>(false, 0, 0));
    else emit(ColumnPlan<FrParams>(true, FrParams::mod(7) + 1, FrParams::mod(7) + 1));
  }
}
// host accessors of the multi-pairing loop / the shared-doubling MSM (the device ones live in engine_jobs.hip)
struct HostMultiAcc {
  int n;
  const int* kinds;
  const G1Aff* P;
  const G2Aff* Q;
  const LineCoeffs* lines;     // [n][RB_MILLER_LINES]
  G2Hom* T;
  Fp6* F;                      // [3]: the accumulator's two halves + the parked value
  int count() const { return n; }
  Fp6 ld_f6(int h) const { return F[h]; }
  void st_f6(int h, const Fp6& v) const { F[h] = v; }
  Fp6 ld_x() const { return F[2]; }
  void st_x(const Fp6& v) const { F[2] = v; }
  Fp2 ld_f2(int i) const { return (&F[i / 3].a0)[i % 3]; }
  void st_f2(int i, const Fp2& v) const { (&F[i / 3].a0)[i % 3] = v; }
  void fence() const {}
  int kind(int j) const { return kinds[j]; }
  MillerP p(int j) const { return miller_p_from_aff(P[j]); }
  G2Aff q(int j) const { return Q[j]; }
  LineCoeffs line(int j, int k) const { return lines[j * RB_MILLER_LINES + k]; }
  G2Hom ld_t(int j) const { return T[j]; }
  void st_t(int j, const G2Hom& t) const { T[j] = t; }
  Fp2 ld_tc(int j, int c) const { return c == 0 ? T[j].x : c == 1 ? T[j].y : T[j].z; }
  void st_tc(int j, int c, const Fp2& v) const { (c == 0 ? T[j].x : c == 1 ? T[j].y : T[j].z) = v; }
};
template <class F>
struct HostTerms {
  int n;
  const Aff<F>* b;
  const uint32_t* pos;
  const uint32_t* neg;
  int count() const { return n; }
  Aff<F> base(int j) const { return b[j]; }
  uint32_t pos_word(int j, int w) const { return pos[8 * j + w]; }
  uint32_t neg_word(int j, int w) const { return neg[8 * j + w]; }
};
extern "C" {
// FE( miller_loop_multi ) over n pairs; kinds[j]: 0 walk, 1 prepared lines, 2 skip.  p: n x 16 words, q: n x 32 words.
void hs_pairing_multi(int n, const int* kinds, const uint32_t* p, const uint32_t* q, uint32_t* out) {
  G1Aff* P = new G1Aff[n];
  G2Aff* Q = new G2Aff[n];
  LineCoeffs* lines = new LineCoeffs[(size_t)n * RB_MILLER_LINES];
  G2Hom* T = new G2Hom[n];
  int* kk = new int[n];
  for (int j = 0; j < n; j++) {
    P[j] = load_g1(p + 16 * j);
    Q[j] = load_g2(q + 32 * j);
    kk[j] = kinds[j];
    if (aff_is_inf(P[j]) || aff_is_inf(Q[j])) kk[j] = MP_SKIP;
    if (kk[j] == MP_LINES) g2_prepare_lines(Q[j], lines + (size_t)j * RB_MILLER_LINES);
  }
  Fp6 F[3];
  store_gt(out, final_exponentiation(miller_loop_multi(HostMultiAcc{n, kk, P, Q, lines, T, F})));
  delete[] P; delete[] Q; delete[] lines; delete[] T; delete[] kk;
}
void hs_g1_msm(int n, const uint32_t* p, const uint32_t* k, uint32_t* out) {
  G1Aff* P = new G1Aff[n];
  uint32_t* pos = new uint32_t[8 * n];
  uint32_t* neg = new uint32_t[8 * n];
  for (int j = 0; j < n; j++) { P[j] = load_g1(p + 16 * j); naf_masks(k + 8 * j, pos + 8 * j, neg + 8 * j); }
  store_g1(out, jac_to_aff(jac_msm_naf<Fp>(HostTerms<Fp>{n, P, pos, neg})));
  delete[] P; delete[] pos; delete[] neg;
}
void hs_g2_msm(int n, const uint32_t* p, const uint32_t* k, uint32_t* out) {
  G2Aff* P = new G2Aff[n];
  uint32_t* pos = new uint32_t[8 * n];
  uint32_t* neg = new uint32_t[8 * n];
  for (int j = 0; j < n; j++) { P[j] = load_g2(p + 32 * j); naf_masks(k + 8 * j, pos + 8 * j, neg + 8 * j); }
  store_g2(out, jac_to_aff(jac_msm_naf<Fp2>(HostTerms<Fp2>{n, P, pos, neg})));
  delete[] P; delete[] pos; delete[] neg;
}
void hs_gt_pow(const uint32_t* a, const uint32_t* k, uint32_t* out) { store_gt(out, gt_pow_binary(load_gt(a), k)); }
void hs_gt_pow_window(const uint32_t* a, const uint32_t* k, uint32_t* out) { store_gt(out, gt_pow_window(load_gt(a), k)); }


}  // extern "C"

// ---- three-lane cooperative pairing (coop3.h) emulated with three host threads: the all-gather is a shared
// buffer between two barriers, everything else is the exact code the device lanes run.
struct HostShared { pthread_barrier_t bar; unsigned char slot[3][1024]; };
struct HostComm {
  int L;
  HostShared* sh;
  template <class T>
  void gather(const T& mine, T* out) {
    static_assert(sizeof(T) <= 1024, "slot too small");
    memcpy(sh->slot[L], &mine, sizeof(T));
    pthread_barrier_wait(&sh->bar);
    for (int r = 0; r < 3; r++) memcpy(&out[r], sh->slot[r], sizeof(T));
    pthread_barrier_wait(&sh->bar);
  }
};
struct C3Job { int L; HostShared* sh; const uint32_t *p, *zs, *q; int mode; Fp12 out; };
static void* c3_worker(void* arg) {
  C3Job* j = (C3Job*)arg;
  HostComm cm{j->L, j->sh};
  G1Aff P = load_g1(j->p);
  MillerP mp = miller_p_from_aff(P);
  if (j->zs) {
    Fp z = load_fp(j->zs);
    Fp z2 = sqr(z);
    G1Jac J{mul(P.x, z2), mul(P.y, mul(z2, z)), z};
    mp = miller_p_from_jac(J);
  }
  Fp12 f = c3_miller_loop(cm, mp, aff_is_inf(P), load_g2(j->q));
  if (j->mode == 1) f = c3_final_exponentiation(cm, f);
  j->out = f;
  return nullptr;
}
extern "C" {
// mode 0: Miller value, mode 1: full pairing.  Returns 1 if the three lanes agree; writes lane 0's value.
int hs_c3_pairing(const uint32_t* p, const uint32_t* zscale, const uint32_t* q, int mode, uint32_t* out) {
  HostShared sh;
  pthread_barrier_init(&sh.bar, nullptr, 3);
  C3Job jobs[3];
  pthread_t th[3];
  for (int L = 0; L < 3; L++) { jobs[L] = C3Job{L, &sh, p, zscale, q, mode, Fp12{}}; pthread_create(&th[L], nullptr, c3_worker, &jobs[L]); }
  for (int L = 0; L < 3; L++) pthread_join(th[L], nullptr);
  pthread_barrier_destroy(&sh.bar);
  store_gt(out, jobs[0].out);
  return fp12_eq(jobs[0].out, jobs[1].out) && fp12_eq(jobs[0].out, jobs[2].out);
}

}  // extern "C"

// ---- six-lane cooperative Fq12 arithmetic (coop6.h) emulated with six host threads per group: the group's slots are a shared
// array, sync() is a thread barrier, everything else is the exact code the device lanes run.
struct C6Shared { pthread_barrier_t bar; Fp2 rows[C6_ROWS][6]; int flag[6]; };
struct HostCX6 {
  C6Shared* sh;
  int k;
  int role() const { return k; }
  Fp2 ld(int row, int lane) const { return sh->rows[row][lane]; }
  void st(int row, const Fp2& v) const { sh->rows[row][k] = v; }
  void sync() const { pthread_barrier_wait(&sh->bar); }
  bool all(bool v) const {
    sh->flag[k] = v;
    pthread_barrier_wait(&sh->bar);
    bool r = true;
    for (int i = 0; i < 6; i++) r = r && sh->flag[i];
    pthread_barrier_wait(&sh->bar);
    return r;
  }
};
// op: 0 a*b, 1 a^2, 2 cyclotomic a^2, 3 a * line (l0, l1, l3 in b's first three Fq2), 4 final exponentiation of a, 5 a^u,
//     6 frob1, 7 frob2, 8 frob3, 9 Miller loop multi + final exponentiation (a, b unused)
struct C6Job { HostCX6 cx; int op; Fp12 a, b; Fp2 out; HostMultiAcc acc; };
static void* c6_worker(void* arg) {
  C6Job* j = (C6Job*)arg;
  const HostCX6 cx = j->cx;
  const int k = cx.k;
  const Fp2 ak = c6_coeff(j->a, k), bk = c6_coeff(j->b, k);
  switch (j->op) {
    case 0: j->out = c6_mul(cx, ak, bk); break;
    case 1: c6_put_f(cx, ak); c6_sqr(cx); j->out = c6_mine(cx); break;
    case 2: c6_put_f(cx, ak); c6_csqr(cx); j->out = c6_mine(cx); break;
    case 3:
      c6_put_f(cx, ak);
      cx.sync();
      if (k == 3) { cx.st(C6_L0, j->b.c0.a0); cx.st(C6_L1, j->b.c0.a1); cx.st(C6_L3, j->b.c0.a2); }
      cx.sync();
      j->out = c6_dot(cx, C6_OP_LINE, 3);
      break;
    case 4: j->out = c6_final_exponentiation(cx, ak); break;
    case 5: j->out = c6_exp_u(cx, ak); break;
    case 6: case 7: case 8: j->out = c6_frob(cx, ak, j->op - 5); break;
    case 10: j->out = c6_gt_is_member(cx, ak, true) ? fp2_one() : fp2_zero(); break;
    default: {
      const Fp2 m = c6_miller_loop_multi(cx, j->acc, j->acc.count());
      j->out = c6_final_exponentiation(cx, m);
    }
  }
  return nullptr;
}
static void c6_run(int op, const Fp12& a, const Fp12& b, const HostMultiAcc* acc, uint32_t* out) {
  C6Shared sh;
  pthread_barrier_init(&sh.bar, nullptr, 6);
  C6Job jobs[6];
  pthread_t th[6];
  for (int k = 0; k < 6; k++) {
    jobs[k] = C6Job{HostCX6{&sh, k}, op, a, b, Fp2{}, acc ? *acc : HostMultiAcc{}};
    pthread_create(&th[k], nullptr, c6_worker, &jobs[k]);
  }
  for (int k = 0; k < 6; k++) pthread_join(th[k], nullptr);
  pthread_barrier_destroy(&sh.bar);
  for (int k = 0; k < 6; k++) store_fp2(out + 16 * c6_tower_index(k), jobs[k].out);
}
extern "C" {
void hs_c6_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) { c6_run(op, load_gt(a), b ? load_gt(b) : fp12_one(), nullptr, out); }
// 1 when the six lanes agree that a is a member of Gt (c6_gt_is_member), 0 when they agree that it is not, -1 when they disagree
int hs_c6_gt_is_member(const uint32_t* a) {
  uint32_t out[96];
  c6_run(10, load_gt(a), fp12_one(), nullptr, out);
  const Fp12 v = load_gt(out);
  int ones = 0;
  for (int k = 0; k < 6; k++) ones += fp2_eq(c6_coeff(v, k), fp2_one()) ? 1 : 0;
  return ones == 6 ? 1 : (ones == 0 ? 0 : -1);
}
// FE( c6_miller_loop_multi ) over n pairs of one group; same arguments as hs_pairing_multi
void hs_c6_pairing_multi(int n, const int* kinds, const uint32_t* p, const uint32_t* q, uint32_t* out) {
  G1Aff* P = new G1Aff[n];
  G2Aff* Q = new G2Aff[n];
  LineCoeffs* lines = new LineCoeffs[(size_t)n * RB_MILLER_LINES];
  G2Hom* T = new G2Hom[n];
  int* kk = new int[n];
  for (int j = 0; j < n; j++) {
    P[j] = load_g1(p + 16 * j);
    Q[j] = load_g2(q + 32 * j);
    kk[j] = kinds[j];
    if (aff_is_inf(P[j]) || aff_is_inf(Q[j])) kk[j] = MP_SKIP;
    if (kk[j] == MP_LINES) g2_prepare_lines(Q[j], lines + (size_t)j * RB_MILLER_LINES);
  }
  const HostMultiAcc acc{n, kk, P, Q, lines, T, nullptr};
  c6_run(9, fp12_one(), fp12_one(), &acc, out);
  delete[] P; delete[] Q; delete[] lines; delete[] T; delete[] kk;
}
}  // extern "C"

// the context-creation self-test's per-lane digest (bn254/selftest.h) on the CPU, and the compiled-in expectation
extern "C" unsigned hs_selftest_digest(int lane) { return selftest_digest(lane); }
extern "C" unsigned hs_selftest_expected(int lane) { constexpr uint32_t e[64] = RB_SELFTEST_EXPECT; return e[lane]; }

// ------------------------------------------------------------------------------------------------ reduced-radix core (bn254/fp29.h, pairing29.h)
struct HostMultiAcc29 {
  int n;
  const int* kinds;
  const G1Aff* P;
  const G2Aff* Q;
  const LineCoeffs* lines;     // [n][RB_MILLER_LINES], 8 x 32-bit form: converted where they are fetched
  rr::G2Hom29* T;
  rr::F6* F;                   // [3]
  int count() const { return n; }
  rr::F6 ld_f6(int h) const { return F[h]; }
  void st_f6(int h, const rr::F6& v) const { F[h] = v; }
  rr::F6 ld_x() const { return F[2]; }
  void st_x(const rr::F6& v) const { F[2] = v; }
  void fence() const {}
  rr::F2* Y;                   // [3]: the right-hand operands of the dot products (slots 1, 2)
  rr::F2& f2(int i) const { return (&F[i / 3].a0)[i % 3]; }
  void st_f2(int i, const rr::F2& v) const { f2(i) = v; }
  void set_y(int s, const rr::F2& v) const { Y[s] = v; }
  rr::F2 dot3(const rr::F2& y0, int ia, int ib, int ic) const { return rr::dot3(f2(ia), y0, f2(ib), Y[1], f2(ic), Y[2]); }
  int kind(int j) const { return kinds[j]; }
  rr::MillerP29 p(int j) const { return rr::MillerP29{rr::from_fp(P[j].x), rr::from_fp(P[j].y)}; }
  rr::G2Aff29 q(int j) const { return rr::G2Aff29{rr::from_fp2(Q[j].x), rr::from_fp2(Q[j].y)}; }
  rr::F2 dot3s(const rr::F& s, int ia, int ib, int ic) const { return rr::dot3s(f2(ia), s, f2(ib), Y[1], f2(ic), Y[2]); }
  rr::LineU29 line_u(int j, int k) const {          // what engine_rr.hip's k_lines_to_rr stores: the line divided by its y-coefficient
    const LineCoeffs& l = lines[j * RB_MILLER_LINES + k];
    const Fp2 iy = fp2_inv(l.cy);
    rr::LineU29 r;
    r.cx = rr::from_fp2(fp2_mul(l.cx, iy)); r.c0 = rr::from_fp2(fp2_mul(l.c0, iy));
    return r;
  }
  rr::G2Hom29 ld_t(int j) const { return T[j]; }
  void st_t(int j, const rr::G2Hom29& t) const { T[j] = t; }
  void begin() const {
    for (int j = 0; j < n; j++)
      if (kinds[j] == MP_WALK) { const rr::G2Aff29 a = q(j); T[j] = rr::G2Hom29{a.x, a.y, rr::one2()}; }
  }
};
extern "C" {
void hs_rr_roundtrip(const uint32_t* a, uint32_t* out) { store_fp(out, rr::to_fp(rr::from_fp(load_fp(a)))); }
void hs_rr_fp2_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) { store_fp2(out, rr::to_fp2(rr::mul2(rr::from_fp2(load_fp2(a)), rr::from_fp2(load_fp2(b))))); }
void hs_rr_fp2_sqr(const uint32_t* a, uint32_t* out) { store_fp2(out, rr::to_fp2(rr::sqr2(rr::from_fp2(load_fp2(a))))); }
void hs_rr_fp2_mul_xi(const uint32_t* a, uint32_t* out) { store_fp2(out, rr::to_fp2(rr::mul_xi2(rr::from_fp2(load_fp2(a))))); }
void hs_rr_fp_half(const uint32_t* a, uint32_t* out) { store_fp(out, rr::to_fp(rr::normf(rr::half(rr::from_fp(load_fp(a)))))); }
// chains of lazily reduced operations: ((a + b)(a - b) - 3 a b) with sums of sums normalised on the way -- exercises norm / norm_lin9 / half
void hs_rr_fp2_mix(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  const rr::F2 x = rr::from_fp2(load_fp2(a)), y = rr::from_fp2(load_fp2(b));
  const rr::F2 m = rr::mul2(rr::add2(x, y), rr::sub2(x, y));
  const rr::F2 n = rr::mul2(x, y);
  const rr::F2 r = rr::normf2(rr::half2(rr::sub2(m, rr::tpl2(n))));
  store_fp2(out, rr::to_fp2(rr::add_mul_xi2(r, rr::sub2(rr::dbl2(n), m))));
}
// the Miller value itself (no final exponentiation), converted back to the canonical 8 x 32-bit form; and the running points
void hs_rr_miller_multi(int n, const int* kinds, const uint32_t* p, const uint32_t* q, uint32_t* out, uint32_t* t_out /* n x 96 words or NULL */) {
  G1Aff* P = new G1Aff[n];
  G2Aff* Q = new G2Aff[n];
  LineCoeffs* lines = new LineCoeffs[(size_t)n * RB_MILLER_LINES];
  rr::G2Hom29* T = new rr::G2Hom29[n];
  int* kk = new int[n];
  for (int j = 0; j < n; j++) {
    P[j] = load_g1(p + 16 * j);
    Q[j] = load_g2(q + 32 * j);
    kk[j] = kinds[j];
    if (aff_is_inf(P[j]) || aff_is_inf(Q[j])) kk[j] = MP_SKIP;
    if (kk[j] == MP_LINES) g2_prepare_lines(Q[j], lines + (size_t)j * RB_MILLER_LINES);
  }
  rr::F6 F[3];
  rr::F2 Y[3];
  rr::miller_loop_multi(HostMultiAcc29{n, kk, P, Q, lines, T, F, Y});
  Fp12 f;
  f.c0 = Fp6{rr::to_fp2(F[0].a0), rr::to_fp2(F[0].a1), rr::to_fp2(F[0].a2)};
  f.c1 = Fp6{rr::to_fp2(F[1].a0), rr::to_fp2(F[1].a1), rr::to_fp2(F[1].a2)};
  store_gt(out, f);
  if (t_out)
    for (int j = 0; j < n; j++)
      if (kk[j] == MP_WALK) { store_fp2(t_out + 96 * j, rr::to_fp2(T[j].x)); store_fp2(t_out + 96 * j + 32, rr::to_fp2(T[j].y)); store_fp2(t_out + 96 * j + 64, rr::to_fp2(T[j].z)); }
  delete[] P; delete[] Q; delete[] lines; delete[] T; delete[] kk;
}
}  // extern "C"
// ---- bn254/pairing29p.h: ONE unit on TWO lanes.  The two lanes of a pair are two host THREADS that run miller_loop_pair side by side;
// every accessor that touches what the lanes share (the LDS home, the operand slots, the DPP moves) is bracketed by barriers, which is what
// a wave's lock-step execution and its in-order LDS traffic give the device code: all reads of a step see the state before any write of it.
struct PairShared {
  rr::F2 home[6];
  rr::F2 slot[2];
  rr::F slot0_fp;
  rr::F2 xch[2];
  pthread_barrier_t bar;
};
struct HostPairAcc29 {
  PairShared* s;
  int me;                      // 0: owns c0, 1: owns c1
  int n;
  const int* kinds;
  const G1Aff* P;
  const G2Aff* Q;
  const LineCoeffs* lines;
  rr::G2Hom29* T;
  void sync() const { pthread_barrier_wait(&s->bar); }
  bool hi() const { return me != 0; }
  int count() const { return n; }
  int kind(int j) const { return kinds[j]; }
  void fence() const {}
  rr::F2 ld_co(int i) const { sync(); const rr::F2 v = s->home[i]; sync(); return v; }
  void st_own(int i, const rr::F2& v) const { sync(); s->home[3 * me + i] = v; sync(); }
  void set_slot(int k, const rr::F2& v, bool w) const { sync(); if (w) s->slot[k] = v; sync(); }
  void set_slot0_fp(const rr::F& v, bool w) const { sync(); if (w) s->slot0_fp = v; sync(); }
  rr::F2 dotp(const rr::F2& yr, int is0, int ir, int is1) const {
    sync();
    const rr::F2 r = rr::dot3(s->home[is0], s->slot[0], s->home[ir], yr, s->home[is1], s->slot[1]);
    sync();
    return r;
  }
  rr::F2 dotps(const rr::F2& yr, int is0, int ir, int is1) const {
    sync();
    const rr::F2 r = rr::dot3s(s->home[is0], s->slot0_fp, s->home[ir], yr, s->home[is1], s->slot[1]);
    sync();
    return r;
  }
  rr::F2 other2(const rr::F2& v) const { sync(); s->xch[me] = v; sync(); const rr::F2 r = s->xch[1 - me]; sync(); return r; }
  template <int O> rr::F2 from2(const rr::F2& v) const { sync(); if (me == O) s->xch[0] = v; sync(); const rr::F2 r = s->xch[0]; sync(); return r; }
  rr::MillerP29 p(int j) const { return rr::MillerP29{rr::from_fp(P[j].x), rr::from_fp(P[j].y)}; }
  rr::G2Aff29 q(int j) const { return rr::G2Aff29{rr::from_fp2(Q[j].x), rr::from_fp2(Q[j].y)}; }
  rr::LineU29 line_u(int j, int k) const {
    const LineCoeffs& l = lines[j * RB_MILLER_LINES + k];
    const Fp2 iy = fp2_inv(l.cy);
    rr::LineU29 r;
    r.cx = rr::from_fp2(fp2_mul(l.cx, iy)); r.c0 = rr::from_fp2(fp2_mul(l.c0, iy));
    return r;
  }
  rr::G2Hom29 ld_t(int j) const { return T[j]; }          // a walking pair is walked by ONE lane: nothing to synchronise
  void st_t(int j, const rr::G2Hom29& t) const { T[j] = t; }
  void begin() const {
    sync();
    if (me == 0)
      for (int j = 0; j < n; j++)
        if (kinds[j] == MP_WALK) { const rr::G2Aff29 a = q(j); T[j] = rr::G2Hom29{a.x, a.y, rr::one2()}; }
    sync();
  }
};
extern "C" void hs_rr_miller_pair(int n, const int* kinds, const uint32_t* p, const uint32_t* q, uint32_t* out, uint32_t* t_out) {
  G1Aff* P = new G1Aff[n];
  G2Aff* Q = new G2Aff[n];
  LineCoeffs* lines = new LineCoeffs[(size_t)n * RB_MILLER_LINES];
  rr::G2Hom29* T = new rr::G2Hom29[n];
  int* kk = new int[n];
  for (int j = 0; j < n; j++) {
    P[j] = load_g1(p + 16 * j);
    Q[j] = load_g2(q + 32 * j);
    kk[j] = kinds[j];
    if (aff_is_inf(P[j]) || aff_is_inf(Q[j])) kk[j] = MP_SKIP;
    if (kk[j] == MP_LINES) g2_prepare_lines(Q[j], lines + (size_t)j * RB_MILLER_LINES);
  }
  PairShared sh;
  pthread_barrier_init(&sh.bar, nullptr, 2);
  std::thread lane1([&] { rr::miller_loop_pair(HostPairAcc29{&sh, 1, n, kk, P, Q, lines, T}); });
  rr::miller_loop_pair(HostPairAcc29{&sh, 0, n, kk, P, Q, lines, T});
  lane1.join();
  pthread_barrier_destroy(&sh.bar);
  Fp12 f;
  f.c0 = Fp6{rr::to_fp2(sh.home[0]), rr::to_fp2(sh.home[1]), rr::to_fp2(sh.home[2])};
  f.c1 = Fp6{rr::to_fp2(sh.home[3]), rr::to_fp2(sh.home[4]), rr::to_fp2(sh.home[5])};
  store_gt(out, f);
  if (t_out)
    for (int j = 0; j < n; j++)
      if (kk[j] == MP_WALK) { store_fp2(t_out + 96 * j, rr::to_fp2(T[j].x)); store_fp2(t_out + 96 * j + 32, rr::to_fp2(T[j].y)); store_fp2(t_out + 96 * j + 64, rr::to_fp2(T[j].z)); }
  delete[] P; delete[] Q; delete[] lines; delete[] T; delete[] kk;
}
extern "C" {
// the same value from pairing.h's loop, for the comparison
void hs_miller_multi(int n, const int* kinds, const uint32_t* p, const uint32_t* q, uint32_t* out, uint32_t* t_out) {
  G1Aff* P = new G1Aff[n];
  G2Aff* Q = new G2Aff[n];
  LineCoeffs* lines = new LineCoeffs[(size_t)n * RB_MILLER_LINES];
  G2Hom* T = new G2Hom[n];
  int* kk = new int[n];
  for (int j = 0; j < n; j++) {
    P[j] = load_g1(p + 16 * j);
    Q[j] = load_g2(q + 32 * j);
    kk[j] = kinds[j];
    if (aff_is_inf(P[j]) || aff_is_inf(Q[j])) kk[j] = MP_SKIP;
    if (kk[j] == MP_LINES) g2_prepare_lines(Q[j], lines + (size_t)j * RB_MILLER_LINES);
  }
  Fp6 F[3];
  store_gt(out, miller_loop_multi(HostMultiAcc{n, kk, P, Q, lines, T, F}));
  if (t_out)
    for (int j = 0; j < n; j++)
      if (kk[j] == MP_WALK) { store_fp2(t_out + 96 * j, T[j].x); store_fp2(t_out + 96 * j + 32, T[j].y); store_fp2(t_out + 96 * j + 64, T[j].z); }
  delete[] P; delete[] Q; delete[] lines; delete[] T; delete[] kk;
}
}

struct HostHome29 {
  rr::F6* F;
  rr::F6 ld_f6(int h) const { return F[h]; }
  void st_f6(int h, const rr::F6& v) const { F[h] = v; }
  rr::F6 ld_x() const { return F[2]; }
  void st_x(const rr::F6& v) const { F[2] = v; }
  void fence() const {}
};
struct HostWs29 {
  rr::F12* slots;
  rr::F6* F;
  rr::F12 ld(int i) const { return slots[i]; }
  void st(int i, const rr::F12& v) const { slots[i] = v; }
  rr::F6 ld6(int i, int h) const { return h ? slots[i].c1 : slots[i].c0; }
  void st6(int i, int h, const rr::F6& v) const { (h ? slots[i].c1 : slots[i].c0) = v; }
  HostHome29 home() const { return HostHome29{F}; }
};
extern "C" {
void hs_rr_final_exp(const uint32_t* f, uint32_t* out) {
  rr::F12 slots[FE_SLOTS];
  slots[FE_T0] = rr::from_fp12(load_gt(f));
  rr::F6 F[3];
  rr::final_exponentiation_ws(HostWs29{slots, F});
  store_gt(out, rr::to_fp12(slots[FE_T1]));
}
void hs_rr_fp12_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  rr::F12 slots[2];
  slots[0] = rr::from_fp12(load_gt(a));
  slots[1] = rr::from_fp12(load_gt(b));
  rr::F6 F[3];
  rr::wsx_mul(HostWs29{slots, F}, 0, 0, false, 1, true);
  store_gt(out, rr::to_fp12(slots[0]));
}
void hs_rr_cyclotomic_sqr(const uint32_t* a, uint32_t* out) { store_gt(out, rr::to_fp12(rr::cyclotomic_sqr(rr::from_fp12(load_gt(a))))); }
void hs_rr_fp12_frob(const uint32_t* a, int k, uint32_t* out) { store_gt(out, rr::to_fp12(rr::frob(rr::from_fp12(load_gt(a)), k))); }
}
