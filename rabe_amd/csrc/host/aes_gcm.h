// KDF + AES-256-GCM of the reference's KEM/DEM step (src/utils/aes/mod.rs:10-55):
//   key = SHA3-256(bytes(Gt)); output = nonce(12) || ciphertext || tag(16).
// The packed entry points do this step on the DEVICE (engine_sym.hip); this is the host's form for the one-call object API and the
// checker of the device path.  On x86-64 with AES-NI + PCLMULQDQ (every host this runs on) the block cipher is `aesenc` and GHASH a
// carry-less multiplication: no table indexed by key-dependent bytes (constant time, ~1 cycle per byte).  The portable table form
// below stays as the fallback and as its cross-check (RABE_AES_PORTABLE=1 forces it; tests/test_host_kats.py runs both).  `bytes(Gt)` is rabe-bn's `Into<Vec<u8>> for Gt`, whose layout is not visible in
// /root/reference: gt_kdf_bytes() below is the single place that choice is made
// (12 Fp coefficients in tower order, each 32-byte big-endian).
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "sha3.h"

namespace rabe { namespace host {

namespace aesdetail {
static const uint8_t SBOX[256] = {
    0x63, 0x7c, 0x77, 0x7b, 0xf2, 0x6b, 0x6f, 0xc5, 0x30, 0x01, 0x67, 0x2b, 0xfe, 0xd7, 0xab, 0x76, 0xca, 0x82, 0xc9, 0x7d, 0xfa, 0x59, 0x47, 0xf0,
    0xad, 0xd4, 0xa2, 0xaf, 0x9c, 0xa4, 0x72, 0xc0, 0xb7, 0xfd, 0x93, 0x26, 0x36, 0x3f, 0xf7, 0xcc, 0x34, 0xa5, 0xe5, 0xf1, 0x71, 0xd8, 0x31, 0x15,
    0x04, 0xc7, 0x23, 0xc3, 0x18, 0x96, 0x05, 0x9a, 0x07, 0x12, 0x80, 0xe2, 0xeb, 0x27, 0xb2, 0x75, 0x09, 0x83, 0x2c, 0x1a, 0x1b, 0x6e, 0x5a, 0xa0,
    0x52, 0x3b, 0xd6, 0xb3, 0x29, 0xe3, 0x2f, 0x84, 0x53, 0xd1, 0x00, 0xed, 0x20, 0xfc, 0xb1, 0x5b, 0x6a, 0xcb, 0xbe, 0x39, 0x4a, 0x4c, 0x58, 0xcf,
    0xd0, 0xef, 0xaa, 0xfb, 0x43, 0x4d, 0x33, 0x85, 0x45, 0xf9, 0x02, 0x7f, 0x50, 0x3c, 0x9f, 0xa8, 0x51, 0xa3, 0x40, 0x8f, 0x92, 0x9d, 0x38, 0xf5,
    0xbc, 0xb6, 0xda, 0x21, 0x10, 0xff, 0xf3, 0xd2, 0xcd, 0x0c, 0x13, 0xec, 0x5f, 0x97, 0x44, 0x17, 0xc4, 0xa7, 0x7e, 0x3d, 0x64, 0x5d, 0x19, 0x73,
    0x60, 0x81, 0x4f, 0xdc, 0x22, 0x2a, 0x90, 0x88, 0x46, 0xee, 0xb8, 0x14, 0xde, 0x5e, 0x0b, 0xdb, 0xe0, 0x32, 0x3a, 0x0a, 0x49, 0x06, 0x24, 0x5c,
    0xc2, 0xd3, 0xac, 0x62, 0x91, 0x95, 0xe4, 0x79, 0xe7, 0xc8, 0x37, 0x6d, 0x8d, 0xd5, 0x4e, 0xa9, 0x6c, 0x56, 0xf4, 0xea, 0x65, 0x7a, 0xae, 0x08,
    0xba, 0x78, 0x25, 0x2e, 0x1c, 0xa6, 0xb4, 0xc6, 0xe8, 0xdd, 0x74, 0x1f, 0x4b, 0xbd, 0x8b, 0x8a, 0x70, 0x3e, 0xb5, 0x66, 0x48, 0x03, 0xf6, 0x0e,
    0x61, 0x35, 0x57, 0xb9, 0x86, 0xc1, 0x1d, 0x9e, 0xe1, 0xf8, 0x98, 0x11, 0x69, 0xd9, 0x8e, 0x94, 0x9b, 0x1e, 0x87, 0xe9, 0xce, 0x55, 0x28, 0xdf,
    0x8c, 0xa1, 0x89, 0x0d, 0xbf, 0xe6, 0x42, 0x68, 0x41, 0x99, 0x2d, 0x0f, 0xb0, 0x54, 0xbb, 0x16};
inline uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }
struct Aes256 {
  uint8_t rk[15][16];
  explicit Aes256(const uint8_t key[32]) {
    uint8_t w[240];
    memcpy(w, key, 32);
    uint8_t rcon = 1;
    for (int i = 32; i < 240; i += 4) {
      uint8_t t[4] = {w[i - 4], w[i - 3], w[i - 2], w[i - 1]};
      if (i % 32 == 0) {
        uint8_t t0 = t[0];
        t[0] = SBOX[t[1]] ^ rcon; t[1] = SBOX[t[2]]; t[2] = SBOX[t[3]]; t[3] = SBOX[t0];
        rcon = xtime(rcon);
      } else if (i % 32 == 16) {
        for (int k = 0; k < 4; k++) t[k] = SBOX[t[k]];
      }
      for (int k = 0; k < 4; k++) w[i + k] = w[i - 32 + k] ^ t[k];
    }
    memcpy(rk, w, 240);
  }
  void encrypt_block(const uint8_t in[16], uint8_t out[16]) const {
    uint8_t s[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[0][i];
    for (int r = 1; r <= 14; r++) {
      uint8_t t[16];
      for (int i = 0; i < 16; i++) t[i] = SBOX[s[i]];
      // shift rows (state is column-major: byte index = 4*col + row)
      uint8_t u[16];
      for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++) u[4 * c + rr] = t[4 * ((c + rr) % 4) + rr];
      if (r != 14) {
        for (int c = 0; c < 4; c++) {
          uint8_t* col = u + 4 * c;
          uint8_t a0 = col[0], a1 = col[1], a2 = col[2], a3 = col[3];
          uint8_t x = a0 ^ a1 ^ a2 ^ a3;
          col[0] = a0 ^ x ^ xtime(a0 ^ a1);
          col[1] = a1 ^ x ^ xtime(a1 ^ a2);
          col[2] = a2 ^ x ^ xtime(a2 ^ a3);
          col[3] = a3 ^ x ^ xtime(a3 ^ a0);
        }
      }
      for (int i = 0; i < 16; i++) s[i] = u[i] ^ rk[r][i];
    }
    memcpy(out, s, 16);
  }
};
// GF(2^128) multiplication, GCM bit order
inline void gf_mul(uint8_t x[16], const uint8_t y[16]) {
  uint8_t z[16] = {0}, v[16];
  memcpy(v, y, 16);
  for (int i = 0; i < 128; i++) {
    if ((x[i >> 3] >> (7 - (i & 7))) & 1)
      for (int k = 0; k < 16; k++) z[k] ^= v[k];
    uint8_t lsb = v[15] & 1;
    for (int k = 15; k > 0; k--) v[k] = (uint8_t)((v[k] >> 1) | (v[k - 1] << 7));
    v[0] >>= 1;
    if (lsb) v[0] ^= 0xe1;
  }
  memcpy(x, z, 16);
}
inline void ghash(const uint8_t h[16], const uint8_t* ct, size_t len, uint8_t out[16]) {
  uint8_t y[16] = {0};
  size_t off = 0;
  while (off < len) {
    size_t n = len - off < 16 ? len - off : 16;
    for (size_t k = 0; k < n; k++) y[k] ^= ct[off + k];
    gf_mul(y, h);
    off += n;
  }
  uint8_t lens[16] = {0};     // len(AAD) = 0 bits || len(C) bits, big-endian
  uint64_t bits = (uint64_t)len * 8;
  for (int k = 0; k < 8; k++) lens[15 - k] = (uint8_t)(bits >> (8 * k));
  for (int k = 0; k < 16; k++) y[k] ^= lens[k];
  gf_mul(y, h);
  memcpy(out, y, 16);
}
inline void inc32(uint8_t ctr[16]) {
  for (int k = 15; k >= 12; k--) if (++ctr[k]) break;
}
}  // namespace aesdetail

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
}}  // namespace rabe::host
#include <immintrin.h>
#include <stdlib.h>
namespace rabe { namespace host {
namespace aeshw {
#define RABE_AESNI __attribute__((target("aes,pclmul,ssse3,sse4.1")))
inline bool available() {
  static const bool ok = [] {
    const char* e = getenv("RABE_AES_PORTABLE");
    if (e && *e && *e != '0') return false;
    __builtin_cpu_init();
    return __builtin_cpu_supports("aes") && __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("ssse3") && __builtin_cpu_supports("sse4.1");
  }();
  return ok;
}
struct Key { __m128i rk[15]; };
RABE_AESNI inline __m128i expand_a(__m128i k, __m128i assist) {          // even round keys (FIPS 197 key expansion with RotWord / Rcon)
  assist = _mm_shuffle_epi32(assist, 0xff);
  k = _mm_xor_si128(k, _mm_slli_si128(k, 4));
  k = _mm_xor_si128(k, _mm_slli_si128(k, 4));
  k = _mm_xor_si128(k, _mm_slli_si128(k, 4));
  return _mm_xor_si128(k, assist);
}
RABE_AESNI inline __m128i expand_b(__m128i k, __m128i prev) {            // odd round keys (SubWord only)
  __m128i assist = _mm_shuffle_epi32(_mm_aeskeygenassist_si128(prev, 0), 0xaa);
  k = _mm_xor_si128(k, _mm_slli_si128(k, 4));
  k = _mm_xor_si128(k, _mm_slli_si128(k, 4));
  k = _mm_xor_si128(k, _mm_slli_si128(k, 4));
  return _mm_xor_si128(k, assist);
}
RABE_AESNI inline void expand(const uint8_t key[32], Key* k) {
  __m128i a = _mm_loadu_si128((const __m128i*)key), b = _mm_loadu_si128((const __m128i*)(key + 16));
  k->rk[0] = a; k->rk[1] = b;
#define RABE_AES_STEP(i, rcon) a = expand_a(a, _mm_aeskeygenassist_si128(b, rcon)); k->rk[i] = a; b = expand_b(b, a); k->rk[i + 1] = b;
  RABE_AES_STEP(2, 0x01) RABE_AES_STEP(4, 0x02) RABE_AES_STEP(6, 0x04) RABE_AES_STEP(8, 0x08) RABE_AES_STEP(10, 0x10) RABE_AES_STEP(12, 0x20)
#undef RABE_AES_STEP
  k->rk[14] = expand_a(a, _mm_aeskeygenassist_si128(b, 0x40));
}
RABE_AESNI inline __m128i encrypt(const Key& k, __m128i x) {
  x = _mm_xor_si128(x, k.rk[0]);
  for (int r = 1; r < 14; r++) x = _mm_aesenc_si128(x, k.rk[r]);
  return _mm_aesenclast_si128(x, k.rk[14]);
}
// GF(2^128) product in GCM's bit order on byte-reversed operands (Gueron / Kounavis: carry-less multiply, shift left by one, reduce)
RABE_AESNI inline __m128i gfmul(__m128i a, __m128i b) {
  __m128i t3 = _mm_clmulepi64_si128(a, b, 0x00), t4 = _mm_clmulepi64_si128(a, b, 0x10), t5 = _mm_clmulepi64_si128(a, b, 0x01),
          t6 = _mm_clmulepi64_si128(a, b, 0x11);
  t4 = _mm_xor_si128(t4, t5);
  t5 = _mm_slli_si128(t4, 8);
  t4 = _mm_srli_si128(t4, 8);
  t3 = _mm_xor_si128(t3, t5);
  t6 = _mm_xor_si128(t6, t4);
  __m128i t7 = _mm_srli_epi32(t3, 31), t8 = _mm_srli_epi32(t6, 31);
  t3 = _mm_slli_epi32(t3, 1);
  t6 = _mm_slli_epi32(t6, 1);
  __m128i t9 = _mm_srli_si128(t7, 12);
  t8 = _mm_slli_si128(t8, 4);
  t7 = _mm_slli_si128(t7, 4);
  t3 = _mm_or_si128(t3, t7);
  t6 = _mm_or_si128(t6, t8);
  t6 = _mm_or_si128(t6, t9);
  t7 = _mm_slli_epi32(t3, 31);
  t8 = _mm_slli_epi32(t3, 30);
  t9 = _mm_slli_epi32(t3, 25);
  t7 = _mm_xor_si128(t7, t8);
  t7 = _mm_xor_si128(t7, t9);
  t8 = _mm_srli_si128(t7, 4);
  t7 = _mm_slli_si128(t7, 12);
  t3 = _mm_xor_si128(t3, t7);
  __m128i t2 = _mm_srli_epi32(t3, 1);
  t4 = _mm_srli_epi32(t3, 2);
  t5 = _mm_srli_epi32(t3, 7);
  t2 = _mm_xor_si128(t2, t4);
  t2 = _mm_xor_si128(t2, t5);
  t2 = _mm_xor_si128(t2, t8);
  t3 = _mm_xor_si128(t3, t2);
  return _mm_xor_si128(t6, t3);
}
RABE_AESNI inline __m128i bswap(__m128i x) { return _mm_shuffle_epi8(x, _mm_set_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)); }
RABE_AESNI inline __m128i load_partial(const uint8_t* p, size_t n) {
  uint8_t b[16] = {0};
  memcpy(b, p, n);
  return _mm_loadu_si128((const __m128i*)b);
}
// ciphertext bytes in `ct` (len), H and E_K(J0) from the key: the tag
RABE_AESNI inline void tag_of(const Key& k, const uint8_t nonce[12], const uint8_t* ct, size_t len, uint8_t tag[16]) {
  const __m128i h = bswap(encrypt(k, _mm_setzero_si128()));
  __m128i y = _mm_setzero_si128();
  for (size_t off = 0; off < len; off += 16) {
    const size_t n = len - off < 16 ? len - off : 16;
    y = gfmul(_mm_xor_si128(y, bswap(n == 16 ? _mm_loadu_si128((const __m128i*)(ct + off)) : load_partial(ct + off, n))), h);
  }
  uint8_t lens[16] = {0};
  const uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) lens[15 - i] = (uint8_t)(bits >> (8 * i));
  y = gfmul(_mm_xor_si128(y, bswap(_mm_loadu_si128((const __m128i*)lens))), h);
  uint8_t j0[16];
  memcpy(j0, nonce, 12); j0[12] = j0[13] = j0[14] = 0; j0[15] = 1;
  _mm_storeu_si128((__m128i*)tag, _mm_xor_si128(bswap(y), encrypt(k, _mm_loadu_si128((const __m128i*)j0))));
}
RABE_AESNI inline void ctr(const Key& k, const uint8_t nonce[12], const uint8_t* in, size_t len, uint8_t* out) {
  uint8_t c[16];
  memcpy(c, nonce, 12);
  uint32_t n32 = 2;
  for (size_t off = 0; off < len; off += 16, n32++) {
    c[12] = (uint8_t)(n32 >> 24); c[13] = (uint8_t)(n32 >> 16); c[14] = (uint8_t)(n32 >> 8); c[15] = (uint8_t)n32;
    const __m128i ks = encrypt(k, _mm_loadu_si128((const __m128i*)c));
    const size_t n = len - off < 16 ? len - off : 16;
    if (n == 16) {
      _mm_storeu_si128((__m128i*)(out + off), _mm_xor_si128(ks, _mm_loadu_si128((const __m128i*)(in + off))));
    } else {
      uint8_t b[16];
      _mm_storeu_si128((__m128i*)b, _mm_xor_si128(ks, load_partial(in + off, n)));
      memcpy(out + off, b, n);
    }
  }
}
RABE_AESNI inline std::vector<uint8_t> gcm_encrypt(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* pt, size_t len) {
  Key k;
  expand(key, &k);
  std::vector<uint8_t> out(len + 16);
  ctr(k, nonce, pt, len, out.data());
  tag_of(k, nonce, out.data(), len, out.data() + len);
  return out;
}
RABE_AESNI inline bool gcm_decrypt(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* ct_tag, size_t len_with_tag, std::vector<uint8_t>* pt) {
  if (len_with_tag < 16) return false;
  const size_t len = len_with_tag - 16;
  Key k;
  expand(key, &k);
  uint8_t tag[16];
  tag_of(k, nonce, ct_tag, len, tag);
  uint8_t diff = 0;
  for (int i = 0; i < 16; i++) diff |= (uint8_t)(tag[i] ^ ct_tag[len + i]);
  if (diff) return false;
  pt->assign(len, 0);
  ctr(k, nonce, ct_tag, len, pt->data());
  return true;
}
// raw keystream E_K(iv + 0), E_K(iv + 1), ... (64-bit little-endian counter in the low half of the block): the batch randomness source
RABE_AESNI inline void keystream(const Key& k, uint64_t iv_hi, uint64_t* counter, uint8_t* out, size_t n_blocks) {
  for (size_t b = 0; b < n_blocks; b++) {
    const __m128i x = _mm_set_epi64x((long long)iv_hi, (long long)(*counter)++);
    _mm_storeu_si128((__m128i*)(out + 16 * b), encrypt(k, x));
  }
}
#undef RABE_AESNI
}  // namespace aeshw
#endif

// AES-256-GCM, 96-bit nonce, no AAD.  out = ciphertext || tag.
inline std::vector<uint8_t> aes256_gcm_encrypt(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* pt, size_t len) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  if (aeshw::available()) return aeshw::gcm_encrypt(key, nonce, pt, len);
#endif
  using namespace aesdetail;
  Aes256 aes(key);
  uint8_t h[16], zero[16] = {0}, j0[16], ctr[16], ks[16];
  aes.encrypt_block(zero, h);
  memcpy(j0, nonce, 12); j0[12] = j0[13] = j0[14] = 0; j0[15] = 1;
  memcpy(ctr, j0, 16);
  std::vector<uint8_t> out(len + 16);
  for (size_t off = 0; off < len; off += 16) {
    inc32(ctr);
    aes.encrypt_block(ctr, ks);
    size_t n = len - off < 16 ? len - off : 16;
    for (size_t k = 0; k < n; k++) out[off + k] = pt[off + k] ^ ks[k];
  }
  uint8_t s[16];
  ghash(h, out.data(), len, s);
  aes.encrypt_block(j0, ks);
  for (int k = 0; k < 16; k++) out[len + k] = s[k] ^ ks[k];
  return out;
}
// returns false on authentication failure
inline bool aes256_gcm_decrypt(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* ct_tag, size_t len_with_tag, std::vector<uint8_t>* pt) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  if (aeshw::available()) return aeshw::gcm_decrypt(key, nonce, ct_tag, len_with_tag, pt);
#endif
  using namespace aesdetail;
  if (len_with_tag < 16) return false;
  size_t len = len_with_tag - 16;
  Aes256 aes(key);
  uint8_t h[16], zero[16] = {0}, j0[16], ctr[16], ks[16], s[16];
  aes.encrypt_block(zero, h);
  memcpy(j0, nonce, 12); j0[12] = j0[13] = j0[14] = 0; j0[15] = 1;
  ghash(h, ct_tag, len, s);
  aes.encrypt_block(j0, ks);
  uint8_t diff = 0;
  for (int k = 0; k < 16; k++) diff |= (uint8_t)((s[k] ^ ks[k]) ^ ct_tag[len + k]);
  if (diff) return false;
  memcpy(ctr, j0, 16);
  pt->assign(len, 0);
  for (size_t off = 0; off < len; off += 16) {
    inc32(ctr);
    aes.encrypt_block(ctr, ks);
    size_t n = len - off < 16 ? len - off : 16;
    for (size_t k = 0; k < n; k++) (*pt)[off + k] = ct_tag[off + k] ^ ks[k];
  }
  return true;
}

// bytes(Gt) for the KDF: wire format is 12 little-endian Fp coefficients; emit each as 32-byte big-endian.
inline std::vector<uint8_t> gt_kdf_bytes(const uint8_t gt_wire[384]) {
  std::vector<uint8_t> o(384);
  for (int c = 0; c < 12; c++)
    for (int k = 0; k < 32; k++) o[32 * c + k] = gt_wire[32 * c + 31 - k];
  return o;
}
inline void kdf(const uint8_t gt_wire[384], uint8_t key[32]) {      // aes/mod.rs:47-55
  std::vector<uint8_t> b = gt_kdf_bytes(gt_wire);
  sha3_256(b.data(), b.size(), key);
}
// encrypt_symmetric (aes/mod.rs:10-26): nonce || ct || tag
inline std::vector<uint8_t> encrypt_symmetric(const uint8_t gt_wire[384], const uint8_t* data, size_t len, const uint8_t nonce[12]) {
  uint8_t key[32];
  kdf(gt_wire, key);
  std::vector<uint8_t> body = aes256_gcm_encrypt(key, nonce, data, len);
  std::vector<uint8_t> out(nonce, nonce + 12);
  out.insert(out.end(), body.begin(), body.end());
  return out;
}
// decrypt_symmetric (aes/mod.rs:29-44)
inline bool decrypt_symmetric(const uint8_t gt_wire[384], const uint8_t* nonce_ct, size_t len, std::vector<uint8_t>* out) {
  if (len < 12 + 16) return false;
  uint8_t key[32];
  kdf(gt_wire, key);
  return aes256_gcm_decrypt(key, nonce_ct, nonce_ct + 12, len - 12, out);
}

}}  // namespace rabe::host
