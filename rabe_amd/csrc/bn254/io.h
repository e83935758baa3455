// Canonical wire format <-> Montgomery (see include/rabe_hip.h for the layouts):
//   Fp/Fr  32 B  little-endian canonical integer (8 x uint32 limbs)
//   G1     64 B  x || y             (infinity: all zero)
//   G2    128 B  x.c0 || x.c1 || y.c0 || y.c1
//   Gt    384 B  12 Fp in tower order c0.a0.c0, c0.a0.c1, c0.a1.c0, ..., c1.a2.c1
#pragma once
#include "pairing.h"

namespace rabe { namespace bn254 {

RB_HD Fp load_fp(const uint32_t* p) {
  uint32_t t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = p[i];
  return to_mont<FpParams>(t);
}
RB_HD void store_fp(uint32_t* p, const Fp& a) {
  uint32_t t[8];
  from_mont<FpParams>(t, a);
#pragma unroll
  for (int i = 0; i < 8; i++) p[i] = t[i];
}
RB_HD Fr load_fr(const uint32_t* p) {
  uint32_t t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = p[i];
  return to_mont<FrParams>(t);
}
RB_HD void store_fr(uint32_t* p, const Fr& a) {
  uint32_t t[8];
  from_mont<FrParams>(t, a);
#pragma unroll
  for (int i = 0; i < 8; i++) p[i] = t[i];
}
// decoding check of the membership entry points: each of the n little-endian 256-bit words at p is a canonical field element
// (< p).  load_fp itself reduces any 256-bit word (to_mont is exact for every input), so a non-canonical coordinate would
// otherwise be a second accepted encoding of the same element; rabe-bn's decoding rejects it (FieldError::NotMember).
RB_HD bool wire_words_canonical(const uint32_t* p, int n) {
  bool ok = true;
  for (int e = 0; e < n; e++) {
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) (void)subb32(p[8 * e + i], FpParams::mod(i), borrow);
    ok = ok && (borrow != 0);          // x - p borrows  <=>  x < p
  }
  return ok;
}
RB_HD Fp2 load_fp2(const uint32_t* p) { return Fp2{load_fp(p), load_fp(p + 8)}; }
RB_HD void store_fp2(uint32_t* p, const Fp2& a) { store_fp(p, a.c0); store_fp(p + 8, a.c1); }
RB_HD G1Aff load_g1(const uint32_t* p) { return G1Aff{load_fp(p), load_fp(p + 8)}; }
RB_HD void store_g1(uint32_t* p, const G1Aff& a) { store_fp(p, a.x); store_fp(p + 8, a.y); }
RB_HD G2Aff load_g2(const uint32_t* p) { return G2Aff{load_fp2(p), load_fp2(p + 16)}; }
RB_HD void store_g2(uint32_t* p, const G2Aff& a) { store_fp2(p, a.x); store_fp2(p + 16, a.y); }
RB_HD Fp12 load_gt(const uint32_t* p) {
  Fp12 r;
  r.c0.a0 = load_fp2(p);      r.c0.a1 = load_fp2(p + 16); r.c0.a2 = load_fp2(p + 32);
  r.c1.a0 = load_fp2(p + 48); r.c1.a1 = load_fp2(p + 64); r.c1.a2 = load_fp2(p + 80);
  return r;
}
RB_HD void store_gt(uint32_t* p, const Fp12& a) {
  store_fp2(p, a.c0.a0);      store_fp2(p + 16, a.c0.a1); store_fp2(p + 32, a.c0.a2);
  store_fp2(p + 48, a.c1.a0); store_fp2(p + 64, a.c1.a1); store_fp2(p + 80, a.c1.a2);
}

}}  // namespace rabe::bn254
