// Host-side scalar field Fr (BN254 group order r): the integer work the reference does on the CPU around
// its group loops -- label hashes (src/utils/hash/mod.rs:23-31), MSP row combination, Shamir shares and
// Lagrange coefficients (src/utils/secretsharing/mod.rs).  Canonical integers, 4 x 64-bit limbs,
// little-endian; byte layout identical to rhip_fr (include/rabe_hip.h).
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>
#include "sha3.h"

namespace rabe { namespace host {

struct Fr {
  uint64_t l[4];
  bool operator==(const Fr& o) const { return memcmp(l, o.l, 32) == 0; }
  bool operator!=(const Fr& o) const { return !(*this == o); }
};

namespace frdetail {
typedef unsigned __int128 u128;
static const uint64_t MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
inline bool geq(const uint64_t a[4], const uint64_t b[4]) {
  for (int i = 3; i >= 0; i--) { if (a[i] > b[i]) return true; if (a[i] < b[i]) return false; }
  return true;
}
inline void sub_raw(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; r[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
// reduce an 8-limb (512-bit) value mod r by binary long division (host side, not hot)
inline void reduce512(uint64_t out[4], const uint64_t in[8]) {
  uint64_t rem[5] = {0, 0, 0, 0, 0};
  for (int bit = 511; bit >= 0; bit--) {
    // rem = rem*2 + bit
    for (int i = 4; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
    rem[0] = (rem[0] << 1) | ((in[bit >> 6] >> (bit & 63)) & 1);
    if (rem[4] || geq(rem, MOD)) {
      uint64_t br = 0;
      for (int i = 0; i < 4; i++) { u128 d = (u128)rem[i] - MOD[i] - br; rem[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
      rem[4] -= br;
    }
  }
  memcpy(out, rem, 32);
}
}  // namespace frdetail

inline Fr fr_zero() { Fr r; memset(r.l, 0, 32); return r; }
inline Fr fr_from_u64(uint64_t v) { Fr r = fr_zero(); r.l[0] = v; return r; }
inline Fr fr_one() { return fr_from_u64(1); }
inline bool fr_is_zero(const Fr& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
inline Fr fr_add(const Fr& a, const Fr& b) {
  using namespace frdetail;
  Fr r;
  uint64_t c = 0;
  for (int i = 0; i < 4; i++) { u128 s = (u128)a.l[i] + b.l[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
  if (c || geq(r.l, MOD)) sub_raw(r.l, r.l, MOD);
  return r;
}
inline Fr fr_sub(const Fr& a, const Fr& b) {
  using namespace frdetail;
  Fr r;
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
  if (br) { uint64_t c = 0; for (int i = 0; i < 4; i++) { u128 s = (u128)r.l[i] + MOD[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
  return r;
}
inline Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }
namespace frdetail {
static const uint64_t R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull};   // 2^512 mod r
inline uint64_t neg_inv64() {
  uint64_t x = 1;
  for (int i = 0; i < 6; i++) x *= 2 - MOD[0] * x;
  return (uint64_t)0 - x;
}
// a*b/2^256 mod r (CIOS)
inline void mont_mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  static const uint64_t INV = neg_inv64();
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    uint64_t c = 0;
    for (int j = 0; j < 4; j++) { u128 x = (u128)a[j] * b[i] + t[j] + c; t[j] = (uint64_t)x; c = (uint64_t)(x >> 64); }
    u128 x = (u128)t[4] + c; t[4] = (uint64_t)x; t[5] = (uint64_t)(x >> 64);
    uint64_t q = t[0] * INV;
    x = (u128)q * MOD[0] + t[0]; c = (uint64_t)(x >> 64);
    for (int j = 1; j < 4; j++) { x = (u128)q * MOD[j] + t[j] + c; t[j - 1] = (uint64_t)x; c = (uint64_t)(x >> 64); }
    x = (u128)t[4] + c; t[3] = (uint64_t)x; t[4] = t[5] + (uint64_t)(x >> 64);
  }
  if (t[4] || geq(t, MOD)) sub_raw(r, t, MOD); else memcpy(r, t, 32);
}
}  // namespace frdetail
namespace frdetail {
// x mod r for a 512-bit x = hi 2^256 + lo: both halves are brought below r by subtraction (2^256 < 6 r), then
// hi * 2^256 mod r is one Montgomery multiplication by R^2.  (reduce512 above is the bit-by-bit long division it replaces
// on the hot paths -- every hash-to-Fr and every random Fr goes through here; tests compare the two.)
inline void reduce256(uint64_t x[4]) { while (geq(x, MOD)) sub_raw(x, x, MOD); }
inline void reduce512_fast(uint64_t out[4], const uint64_t in[8]) {
  uint64_t lo[4], hi[4], t[4];
  memcpy(lo, in, 32);
  memcpy(hi, in + 4, 32);
  reduce256(lo);
  reduce256(hi);
  mont_mul(t, hi, R2);                     // hi * R^2 / R = hi * 2^256 mod r
  uint64_t c = 0;
  for (int i = 0; i < 4; i++) { u128 s = (u128)t[i] + lo[i] + c; out[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
  if (c || geq(out, MOD)) sub_raw(out, out, MOD);
}
}  // namespace frdetail
inline Fr fr_mul(const Fr& a, const Fr& b) {
  Fr t, r;
  frdetail::mont_mul(t.l, a.l, b.l);            // a*b/R
  frdetail::mont_mul(r.l, t.l, frdetail::R2);   // * R^2 / R = a*b
  return r;
}
inline Fr fr_pow(const Fr& a, const uint64_t e[4]) {
  Fr acc = fr_one();
  for (int i = 255; i >= 0; i--) {
    acc = fr_mul(acc, acc);
    if ((e[i >> 6] >> (i & 63)) & 1) acc = fr_mul(acc, a);
  }
  return acc;
}
// `Fr::inverse()`: None for zero (the reference unwraps it, src/utils/secretsharing/mod.rs:66)
inline bool fr_inv(const Fr& a, Fr* out) {
  if (fr_is_zero(a)) return false;
  uint64_t e[4];
  uint64_t two[4] = {2, 0, 0, 0};
  frdetail::sub_raw(e, frdetail::MOD, two);
  *out = fr_pow(a, e);
  return true;
}
// `Fr::from_slice(&digest)`: 32 big-endian bytes, reduced mod r (SURVEY.md 8c assumption (i))
inline Fr fr_from_be32_reduce(const uint8_t d[32]) {
  uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) { uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | d[(3 - i) * 8 + j]; t[i] = w; }
  Fr r;
  frdetail::reduce512_fast(r.l, t);
  return r;
}
// 64 uniformly random bytes -> Fr (`Fr::random`: 512 random bits mod r, SURVEY.md 8c assumption (iv))
inline Fr fr_from_le64_reduce(const uint8_t b[64]) {
  uint64_t t[8];
  memcpy(t, b, 64);
  Fr r;
  frdetail::reduce512_fast(r.l, t);
  return r;
}
// sha3_hash_fr (src/utils/hash/mod.rs:23-31)
inline Fr sha3_hash_fr(const std::string& label) {
  uint8_t d[32];
  sha3_256((const uint8_t*)label.data(), label.size(), d);
  return fr_from_be32_reduce(d);
}
inline void fr_to_bytes(uint8_t out[32], const Fr& a) { memcpy(out, a.l, 32); }
inline Fr fr_from_bytes(const uint8_t in[32]) { Fr r; memcpy(r.l, in, 32); return r; }

}}  // namespace rabe::host
