// engine_sym.hip -- the KEM -> DEM step and label hashing on the device.
//
// Reference: src/utils/aes/mod.rs:10-55 (`encrypt_symmetric` / `decrypt_symmetric`: key = SHA3-256(bytes(Gt)), AES-256-GCM with a
// 12-byte nonce, output = nonce || ciphertext || tag) and src/utils/hash/mod.rs:10-31 (`sha3_hash*`: Fr::from_slice(SHA3-256(label))).
// The packed entry points of the host layer call these so that the Gt of an encrypt / decrypt never leaves HBM: plaintext bytes and
// record bytes are the only payload that crosses PCIe.
//
// Everything here is byte / bit work, one lane per independent unit, no field arithmetic:
//   * Keccak-f[1600] in 25 x 64-bit registers per lane;
//   * AES-256 without a memory table: the 256-byte S-box is ONE register of the wave (lane j holds bytes 4j .. 4j+3) and a byte is
//     looked up with ds_bpermute_b32 -- the LDS crossbar routes a dword from lane (b >> 2), no address that depends on a secret ever
//     reaches a memory bank or a cache (the host's aes_gcm.h indexes a table in memory with key-dependent bytes);
//   * GHASH by the shift-and-add multiplication in GF(2^128) with masks instead of branches;
//   * CTR blocks are independent: one lane per 16-byte block of the whole batch; GHASH runs per 64-block segment and the segment
//     values are folded with powers of H, so a long plaintext is not one lane's serial chain.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "engine_internal.h"
// the S-box lookup below reads one register of the WAVE through ds_bpermute with `threadIdx.x & 63`: 64-wide wavefronts only
#if defined(__HIP_DEVICE_COMPILE__) && defined(__AMDGCN_WAVEFRONT_SIZE) && __AMDGCN_WAVEFRONT_SIZE != 64
#error "engine_sym.hip is written for 64-lane wavefronts (gfx950)"
#endif

namespace {

// ------------------------------------------------------------------------------------------------ Keccak-f[1600] / SHA3-256
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }
__device__ void keccak_f1600(uint64_t s[25]) {
  const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
                           0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
                           0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
                           0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
                           0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
#pragma unroll 1
  for (int round = 0; round < 24; round++) {
    uint64_t c0 = s[0] ^ s[5] ^ s[10] ^ s[15] ^ s[20], c1 = s[1] ^ s[6] ^ s[11] ^ s[16] ^ s[21], c2 = s[2] ^ s[7] ^ s[12] ^ s[17] ^ s[22],
             c3 = s[3] ^ s[8] ^ s[13] ^ s[18] ^ s[23], c4 = s[4] ^ s[9] ^ s[14] ^ s[19] ^ s[24];
    const uint64_t d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1), d3 = c2 ^ rotl64(c4, 1), d4 = c3 ^ rotl64(c0, 1);
#pragma unroll
    for (int y = 0; y < 25; y += 5) { s[y] ^= d0; s[y + 1] ^= d1; s[y + 2] ^= d2; s[y + 3] ^= d3; s[y + 4] ^= d4; }
    // rho + pi (fully unrolled: every index and rotation is a constant, nothing is runtime-indexed)
    uint64_t b[25];
    b[0] = s[0];
    b[10] = rotl64(s[1], 1);   b[7] = rotl64(s[10], 3);   b[11] = rotl64(s[7], 6);   b[17] = rotl64(s[11], 10);  b[18] = rotl64(s[17], 15);
    b[3] = rotl64(s[18], 21);  b[5] = rotl64(s[3], 28);   b[16] = rotl64(s[5], 36);  b[8] = rotl64(s[16], 45);   b[21] = rotl64(s[8], 55);
    b[24] = rotl64(s[21], 2);  b[4] = rotl64(s[24], 14);  b[15] = rotl64(s[4], 27);  b[23] = rotl64(s[15], 41);  b[19] = rotl64(s[23], 56);
    b[13] = rotl64(s[19], 8);  b[12] = rotl64(s[13], 25); b[2] = rotl64(s[12], 43);  b[20] = rotl64(s[2], 62);   b[14] = rotl64(s[20], 18);
    b[22] = rotl64(s[14], 39); b[9] = rotl64(s[22], 61);  b[6] = rotl64(s[9], 20);   b[1] = rotl64(s[6], 44);
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
      s[y] = b[y] ^ (~b[y + 1] & b[y + 2]);
      s[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]);
      s[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]);
      s[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y]);
      s[y + 4] = b[y + 4] ^ (~b[y] & b[y + 1]);
    }
    s[0] ^= RC[round];
  }
}
__device__ __forceinline__ uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }
__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// SHA3-256 of bytes(Gt): 12 coefficients, each as 32 big-endian bytes (aes_gcm.h: gt_kdf_bytes -- SURVEY.md 8c convention (v)).  The wire
// form is 12 x 32 little-endian bytes, so message lane 4c + q (8 bytes) is the byte-swapped 64-bit word 3 - q of coefficient c.
__device__ void kdf_of_gt(const uint32_t* gt /*96 words*/, uint32_t key[8]) {
  uint64_t s[25];
#pragma unroll
  for (int i = 0; i < 25; i++) s[i] = 0;
  auto lane = [&](int j) -> uint64_t {
    const int c = j >> 2, q = j & 3;
    const uint32_t lo = gt[8 * c + 2 * (3 - q)], hi = gt[8 * c + 2 * (3 - q) + 1];
    return bswap64(((uint64_t)hi << 32) | lo);
  };
#pragma unroll
  for (int i = 0; i < 17; i++) s[i] ^= lane(i);
  keccak_f1600(s);
#pragma unroll
  for (int i = 0; i < 17; i++) s[i] ^= lane(17 + i);
  keccak_f1600(s);
#pragma unroll
  for (int i = 0; i < 14; i++) s[i] ^= lane(34 + i);
  s[14] ^= 0x06ull;                          // 384 = 2 * 136 + 112: the padding starts at byte 112 of the third block
  s[16] ^= 0x8000000000000000ull;
  keccak_f1600(s);
#pragma unroll
  for (int i = 0; i < 4; i++) { key[2 * i] = (uint32_t)s[i]; key[2 * i + 1] = (uint32_t)(s[i] >> 32); }
}
// SHA3-256 of an arbitrary byte string (labels)
__device__ void sha3_256_bytes(const uint8_t* data, size_t len, uint32_t out[8]) {
  uint64_t s[25];
#pragma unroll
  for (int i = 0; i < 25; i++) s[i] = 0;
  size_t off = 0;
  for (;;) {
    const size_t left = len - off;
    const bool last = left < 136;
    const size_t take = last ? left : 136;
#pragma unroll
    for (int w = 0; w < 17; w++) {          // unrolled: the state stays in registers (nothing runtime-indexed)
      uint64_t v = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const size_t p = (size_t)w * 8 + k;
        uint64_t byte = p < take ? data[off + p] : 0;
        if (last && p == take) byte ^= 0x06;
        if (last && p == 135) byte ^= 0x80;
        v |= byte << (8 * k);
      }
      s[w] ^= v;
    }
    keccak_f1600(s);
    if (last) break;
    off += 136;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) { out[2 * i] = (uint32_t)s[i]; out[2 * i + 1] = (uint32_t)(s[i] >> 32); }
}

// ------------------------------------------------------------------------------------------------ AES-256 (FIPS 197), table in a register
__constant__ uint32_t SBOX_WORDS[64] = {
    0x7b777c63u, 0xc56f6bf2u, 0x2b670130u, 0x76abd7feu, 0x7dc982cau, 0xf04759fau, 0xafa2d4adu, 0xc072a49cu, 0x2693fdb7u, 0xccf73f36u, 0xf1e5a534u,
    0x1531d871u, 0xc323c704u, 0x9a059618u, 0xe2801207u, 0x75b227ebu, 0x1a2c8309u, 0xa05a6e1bu, 0xb3d63b52u, 0x842fe329u, 0xed00d153u, 0x5bb1fc20u,
    0x39becb6au, 0xcf584c4au, 0xfbaaefd0u, 0x85334d43u, 0x7f02f945u, 0xa89f3c50u, 0x8f40a351u, 0xf5389d92u, 0x21dab6bcu, 0xd2f3ff10u, 0xec130ccdu,
    0x1744975fu, 0x3d7ea7c4u, 0x73195d64u, 0xdc4f8160u, 0x88902a22u, 0x14b8ee46u, 0xdb0b5edeu, 0x0a3a32e0u, 0x5c240649u, 0x62acd3c2u, 0x79e49591u,
    0x6d37c8e7u, 0xa94ed58du, 0xeaf4566cu, 0x08ae7a65u, 0x2e2578bau, 0xc6b4a61cu, 0x1f74dde8u, 0x8a8bbd4bu, 0x66b53e70u, 0x0ef60348u, 0xb9573561u,
    0x9e1dc186u, 0x1198f8e1u, 0x948ed969u, 0xe9871e9bu, 0xdf2855ceu, 0x0d89a18cu, 0x6842e6bfu, 0x0f2d9941u, 0x16bb54b0u};
// S-box of one byte (bits 8k .. 8k+7 of `w`), result in the low byte.  All 64 lanes of the wave must be active.
__device__ __forceinline__ uint32_t sbox_byte(uint32_t tab, uint32_t w, int k) {
  const uint32_t b = (w >> (8 * k)) & 0xFFu;
  const uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(b & 0xFCu), (int)tab);     // address = 4 * source lane
  return (word >> ((b & 3u) * 8)) & 0xFFu;
}
__device__ __forceinline__ uint32_t sub_word(uint32_t tab, uint32_t w) {
  return sbox_byte(tab, w, 0) | (sbox_byte(tab, w, 1) << 8) | (sbox_byte(tab, w, 2) << 16) | (sbox_byte(tab, w, 3) << 24);
}
__device__ __forceinline__ uint32_t xtime4(uint32_t a) {          // multiplication by x in GF(2^8) on four packed bytes
  const uint32_t hi = a & 0x80808080u;
  return ((a & 0x7f7f7f7fu) << 1) ^ ((hi >> 7) * 0x1bu);
}
__device__ __forceinline__ uint32_t rotr8(uint32_t a) { return (a >> 8) | (a << 24); }
__device__ __forceinline__ uint32_t rotr16(uint32_t a) { return (a >> 16) | (a << 16); }
// state: 4 column words, byte `row` of column c at bits 8 * row (the byte order of the block in memory, little-endian words)
struct AesKey { uint32_t w[60]; };
__device__ void aes256_expand(uint32_t tab, const uint32_t key[8], AesKey* rk) {
#pragma unroll
  for (int i = 0; i < 8; i++) rk->w[i] = key[i];
  uint32_t rcon = 1;
#pragma unroll
  for (int i = 8; i < 60; i++) {
    uint32_t t = rk->w[i - 1];
    if (i % 8 == 0) {
      t = sub_word(tab, rotr8(t)) ^ rcon;
      rcon = xtime4(rcon) & 0xFFu;
    } else if (i % 8 == 4) {
      t = sub_word(tab, t);
    }
    rk->w[i] = rk->w[i - 8] ^ t;
  }
}
__device__ void aes256_encrypt(uint32_t tab, const AesKey& rk, const uint32_t in[4], uint32_t out[4]) {
  uint32_t s0 = in[0] ^ rk.w[0], s1 = in[1] ^ rk.w[1], s2 = in[2] ^ rk.w[2], s3 = in[3] ^ rk.w[3];
#pragma unroll
  for (int r = 1; r <= 14; r++) {
    // SubBytes + ShiftRows: new column c, row k = S(old column (c + k) mod 4, row k)
    const uint32_t t0 = sbox_byte(tab, s0, 0) | (sbox_byte(tab, s1, 1) << 8) | (sbox_byte(tab, s2, 2) << 16) | (sbox_byte(tab, s3, 3) << 24);
    const uint32_t t1 = sbox_byte(tab, s1, 0) | (sbox_byte(tab, s2, 1) << 8) | (sbox_byte(tab, s3, 2) << 16) | (sbox_byte(tab, s0, 3) << 24);
    const uint32_t t2 = sbox_byte(tab, s2, 0) | (sbox_byte(tab, s3, 1) << 8) | (sbox_byte(tab, s0, 2) << 16) | (sbox_byte(tab, s1, 3) << 24);
    const uint32_t t3 = sbox_byte(tab, s3, 0) | (sbox_byte(tab, s0, 1) << 8) | (sbox_byte(tab, s1, 2) << 16) | (sbox_byte(tab, s2, 3) << 24);
    if (r != 14) {
      // MixColumns on packed bytes: out_k = a_k ^ x ^ xtime(a_k ^ a_{k+1}), x = a_0 ^ a_1 ^ a_2 ^ a_3
      auto mix = [](uint32_t a) -> uint32_t {
        const uint32_t u = a ^ rotr8(a);
        return a ^ (u ^ rotr16(u)) ^ xtime4(u);
      };
      s0 = mix(t0) ^ rk.w[4 * r]; s1 = mix(t1) ^ rk.w[4 * r + 1]; s2 = mix(t2) ^ rk.w[4 * r + 2]; s3 = mix(t3) ^ rk.w[4 * r + 3];
    } else {
      s0 = t0 ^ rk.w[56]; s1 = t1 ^ rk.w[57]; s2 = t2 ^ rk.w[58]; s3 = t3 ^ rk.w[59];
    }
  }
  out[0] = s0; out[1] = s1; out[2] = s2; out[3] = s3;
}

// ------------------------------------------------------------------------------------------------ GHASH (SP 800-38D), GF(2^128)
// a block is four big-endian words: bit 0 of the field element is the most significant bit of x[0]
struct Gf { uint32_t x[4]; };
__device__ Gf gf_mul(const Gf& a, const Gf& b) {
  uint32_t z0 = 0, z1 = 0, z2 = 0, z3 = 0, v0 = b.x[0], v1 = b.x[1], v2 = b.x[2], v3 = b.x[3];
#pragma unroll
  for (int w = 0; w < 4; w++) {
    uint32_t aw = a.x[w];
#pragma unroll 1
    for (int i = 0; i < 32; i++) {
      const uint32_t m = 0u - (aw >> 31);
      aw <<= 1;
      z0 ^= v0 & m; z1 ^= v1 & m; z2 ^= v2 & m; z3 ^= v3 & m;
      const uint32_t red = (0u - (v3 & 1u)) & 0xe1000000u;
      v3 = (v3 >> 1) | (v2 << 31); v2 = (v2 >> 1) | (v1 << 31); v1 = (v1 >> 1) | (v0 << 31); v0 = (v0 >> 1) ^ red;
    }
  }
  return Gf{{z0, z1, z2, z3}};
}

// ------------------------------------------------------------------------------------------------ byte access at any alignment
__device__ __forceinline__ void load_block(const uint8_t* p, uint32_t nbytes /*1..16*/, uint32_t w[4]) {      // little-endian words, zero padded
  w[0] = w[1] = w[2] = w[3] = 0;
  if (nbytes == 16 && (((uintptr_t)p) & 3) == 0) {
    const uint32_t* q = (const uint32_t*)p;
    w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; w[3] = q[3];
    return;
  }
#pragma unroll
  for (int k = 0; k < 16; k++)
    if ((uint32_t)k < nbytes) w[k >> 2] |= (uint32_t)p[k] << (8 * (k & 3));
}
__device__ __forceinline__ void store_block(uint8_t* p, uint32_t nbytes, const uint32_t w[4]) {
  if (nbytes == 16 && (((uintptr_t)p) & 3) == 0) {
    uint32_t* q = (uint32_t*)p;
    q[0] = w[0]; q[1] = w[1]; q[2] = w[2]; q[3] = w[3];
    return;
  }
#pragma unroll
  for (int k = 0; k < 16; k++)
    if ((uint32_t)k < nbytes) p[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
}
// the item a flattened unit (block, segment) belongs to: largest i with off[i] <= u
__device__ __forceinline__ uint32_t owner_of(const uint32_t* off, uint32_t n, uint32_t u) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= u) lo = mid; else hi = mid;
  }
  return lo;
}

// per-item context in HBM, word-major ([word][item]: coalesced): 60 round-key words, H (4, big-endian words), E_K(J0) (4, memory order)
constexpr int SYM_CTX_WORDS = 68;
constexpr uint32_t SEG_BLOCKS = 64;

// ------------------------------------------------------------------------------------------------ kernels
__global__ void __launch_bounds__(256) k_sym_kdf(uint32_t n, const rhip_gt* gt, const uint32_t* gt_idx, uint32_t* keys /*[n][8]*/) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t key[8];
  kdf_of_gt(gt[gt_idx ? gt_idx[i] : i].l, key);
#pragma unroll
  for (int k = 0; k < 8; k++) keys[(size_t)i * 8 + k] = key[k];
}
__global__ void __launch_bounds__(256) k_sha3_256(uint32_t n, const uint8_t* data, const uint64_t* off, uint32_t* digest /*[n][8]*/) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t d[8];
  sha3_256_bytes(data + off[i], (size_t)(off[i + 1] - off[i]), d);
#pragma unroll
  for (int k = 0; k < 8; k++) digest[(size_t)i * 8 + k] = d[k];
}
// Fr::from_slice(digest): the 32 bytes as a big-endian integer, reduced mod r (hash/mod.rs:16; convention (i)).  2^256 < 6 r.
__global__ void __launch_bounds__(256) k_digest_to_fr(uint32_t n, const uint32_t* digest, rhip_fr* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t R[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  uint32_t v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = bswap32(digest[(size_t)i * 8 + 7 - k]);          // little-endian limbs of the big-endian integer
#pragma unroll 1
  for (int rep = 0; rep < 5; rep++) {
    uint32_t d[8];
    uint64_t borrow = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint64_t t = (uint64_t)v[k] - R[k] - borrow;
      d[k] = (uint32_t)t;
      borrow = (t >> 63) & 1u;
    }
    const uint32_t keep = 0u - (uint32_t)borrow;          // borrow: v < r, keep v
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = (v[k] & keep) | (d[k] & ~keep);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) out[i].l[k] = v[k];
}

// round keys, H = E_K(0), E_K(J0) per item.  Every lane of a wave stays active (the S-box lives in the wave's registers).
__global__ void __launch_bounds__(256) k_sym_setup(uint32_t n, const uint32_t* keys, const uint8_t* nonce, const uint64_t* nonce_off, uint32_t* ctxw,
                                                   uint32_t stride) {
  const uint32_t tab = SBOX_WORDS[threadIdx.x & 63];
  const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = gi < n;
  const uint32_t i = live ? gi : n - 1;
  uint32_t key[8];
#pragma unroll
  for (int k = 0; k < 8; k++) key[k] = keys[(size_t)i * 8 + k];
  AesKey rk;
  aes256_expand(tab, key, &rk);
  uint32_t zero[4] = {0, 0, 0, 0}, h[4], j0[4], ej0[4];
  aes256_encrypt(tab, rk, zero, h);
  const uint8_t* np = nonce + (nonce_off ? nonce_off[i] : (uint64_t)i * 12);
  j0[0] = j0[1] = j0[2] = 0;
#pragma unroll
  for (int k = 0; k < 12; k++) j0[k >> 2] |= (uint32_t)np[k] << (8 * (k & 3));
  j0[3] = 0x01000000u;                        // bytes 12..15 = 00 00 00 01
  aes256_encrypt(tab, rk, j0, ej0);
  if (!live) return;
#pragma unroll
  for (int k = 0; k < 60; k++) ctxw[(size_t)k * stride + i] = rk.w[k];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    ctxw[(size_t)(60 + k) * stride + i] = bswap32(h[k]);
    ctxw[(size_t)(64 + k) * stride + i] = ej0[k];
  }
}
// one lane per 16-byte block of the batch: out = in ^ E_K(nonce || be32(2 + block))
__global__ void __launch_bounds__(256) k_sym_ctr(uint32_t n, uint32_t total_blocks, const uint32_t* blk_off, const uint32_t* ctxw, uint32_t stride,
                                                 const uint8_t* nonce, const uint64_t* nonce_off, const uint8_t* in, const uint64_t* in_off,
                                                 uint8_t* out, const uint64_t* out_off, const uint32_t* len, const uint32_t* ok) {
  const uint32_t tab = SBOX_WORDS[threadIdx.x & 63];
  const uint32_t gu = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = gu < total_blocks;
  const uint32_t u = live ? gu : total_blocks - 1;
  const uint32_t i = owner_of(blk_off, n, u);
  const uint32_t b = u - blk_off[i];
  AesKey rk;
#pragma unroll
  for (int k = 0; k < 60; k++) rk.w[k] = ctxw[(size_t)k * stride + i];
  const uint8_t* np = nonce + (nonce_off ? nonce_off[i] : (uint64_t)i * 12);
  uint32_t ctr[4] = {0, 0, 0, 0}, ks[4], x[4];
#pragma unroll
  for (int k = 0; k < 12; k++) ctr[k >> 2] |= (uint32_t)np[k] << (8 * (k & 3));
  ctr[3] = bswap32(b + 2u);
  aes256_encrypt(tab, rk, ctr, ks);
  if (!live) return;
  const uint32_t left = len[i] - 16u * b;
  const uint32_t nb = left < 16u ? left : 16u;
  load_block(in + in_off[i] + 16ull * b, nb, x);
  const uint32_t keep = (ok && !ok[i]) ? 0u : 0xFFFFFFFFu;          // a failed tag releases no plaintext (zeros, as the host's memset)
#pragma unroll
  for (int k = 0; k < 4; k++) x[k] = (x[k] ^ ks[k]) & keep;
  store_block(out + out_off[i] + 16ull * b, nb, x);
}
// GHASH of one segment (<= 64 blocks) of one item's ciphertext: Horner with H from zero.  Segment 0 of an item is the short one.
__global__ void __launch_bounds__(256) k_sym_ghash_seg(uint32_t n, uint32_t total_segs, const uint32_t* seg_off, const uint32_t* ctxw, uint32_t stride,
                                                       const uint8_t* ct, const uint64_t* ct_off, const uint32_t* len, uint32_t* part /*[segs][4]*/) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= total_segs) return;
  const uint32_t i = owner_of(seg_off, n, u);
  const uint32_t sg = u - seg_off[i], nseg = seg_off[i + 1] - seg_off[i];
  const uint32_t nblk = (len[i] + 15u) / 16u;
  const uint32_t first = nblk - SEG_BLOCKS * (nseg - 1);            // blocks in segment 0 (1 .. 64)
  const uint32_t b0 = sg == 0 ? 0 : first + SEG_BLOCKS * (sg - 1), cnt = sg == 0 ? first : SEG_BLOCKS;
  Gf h, y = {{0, 0, 0, 0}};
#pragma unroll
  for (int k = 0; k < 4; k++) h.x[k] = ctxw[(size_t)(60 + k) * stride + i];
  const uint8_t* base = ct + ct_off[i];
#pragma unroll 1
  for (uint32_t b = b0; b < b0 + cnt; b++) {
    const uint32_t left = len[i] - 16u * b;
    uint32_t x[4];
    load_block(base + 16ull * b, left < 16u ? left : 16u, x);
#pragma unroll
    for (int k = 0; k < 4; k++) y.x[k] ^= bswap32(x[k]);
    y = gf_mul(y, h);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) part[(size_t)u * 4 + k] = y.x[k];
}
// fold the segments, the length block, E_K(J0).  seal != 0: write nonce and tag (and, with len_prefix, the u32 length of the sealed bytes
// in front of them); seal == 0: compare with the tag that follows the ciphertext -> ok[i].
__global__ void __launch_bounds__(256) k_sym_tag(uint32_t n, const uint32_t* seg_off, const uint32_t* ctxw, uint32_t stride, const uint32_t* part,
                                                 const uint8_t* nonce, const uint64_t* nonce_off, uint8_t* blob, const uint64_t* sealed_off,
                                                 const uint32_t* len, int seal, int len_prefix, uint32_t* ok) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Gf h, t = {{0, 0, 0, 0}};
#pragma unroll
  for (int k = 0; k < 4; k++) h.x[k] = ctxw[(size_t)(60 + k) * stride + i];
  const uint32_t s0 = seg_off[i], nseg = seg_off[i + 1] - s0;
  if (nseg) {
#pragma unroll
    for (int k = 0; k < 4; k++) t.x[k] = part[(size_t)s0 * 4 + k];
    if (nseg > 1) {
      Gf hs = h;                                // H^64
#pragma unroll 1
      for (int q = 0; q < 6; q++) hs = gf_mul(hs, hs);
#pragma unroll 1
      for (uint32_t sg = 1; sg < nseg; sg++) {
        t = gf_mul(t, hs);
#pragma unroll
        for (int k = 0; k < 4; k++) t.x[k] ^= part[(size_t)(s0 + sg) * 4 + k];
      }
    }
  }
  const uint64_t bits = (uint64_t)len[i] * 8;          // len(A) = 0 || len(C), both 64-bit big-endian
  t.x[2] ^= (uint32_t)(bits >> 32);
  t.x[3] ^= (uint32_t)bits;
  t = gf_mul(t, h);
  uint32_t tag[4];
#pragma unroll
  for (int k = 0; k < 4; k++) tag[k] = bswap32(t.x[k]) ^ ctxw[(size_t)(64 + k) * stride + i];
  uint8_t* p = blob + sealed_off[i];
  if (seal) {
    const uint8_t* np = nonce + (nonce_off ? nonce_off[i] : (uint64_t)i * 12);
    if (len_prefix) {
      const uint32_t total = len[i] + 28u;
      for (int k = 0; k < 4; k++) p[k - 4] = (uint8_t)(total >> (8 * k));
    }
    for (int k = 0; k < 12; k++) p[k] = np[k];
    store_block(p + 12 + len[i], 16, tag);
  } else {
    uint32_t have[4];
    load_block(p + 12 + len[i], 16, have);
    const uint32_t diff = (have[0] ^ tag[0]) | (have[1] ^ tag[1]) | (have[2] ^ tag[2]) | (have[3] ^ tag[3]);
    ok[i] = diff == 0 ? 1u : 0u;
  }
}
// plain AES-256 of n blocks under n keys (the FIPS-197 known-answer test goes through here)
__global__ void __launch_bounds__(256) k_aes256_blocks(uint32_t n, const uint32_t* keys, const uint32_t* in, uint32_t* out) {
  const uint32_t tab = SBOX_WORDS[threadIdx.x & 63];
  const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = gi < n ? gi : n - 1;
  uint32_t key[8], x[4], y[4];
#pragma unroll
  for (int k = 0; k < 8; k++) key[k] = keys[(size_t)i * 8 + k];
#pragma unroll
  for (int k = 0; k < 4; k++) x[k] = in[(size_t)i * 4 + k];
  AesKey rk;
  aes256_expand(tab, key, &rk);
  aes256_encrypt(tab, rk, x, y);
  if (gi >= n) return;
#pragma unroll
  for (int k = 0; k < 4; k++) out[(size_t)i * 4 + k] = y[k];
}

// records of a batch from their parts: out byte b of item i = map[layout_off[layout[i]] + b]:
//   0xFF0000vv        the literal byte vv (policy text, names, counts: the per-policy template)
//   k << 24 | o       byte o of item i's part in source k: src[k] + src_item_off[k * n + i]
__global__ void __launch_bounds__(256) k_assemble_records(uint32_t n, uint8_t* out, const uint64_t* out_off, const uint32_t* layout,
                                                          const uint32_t* layout_off, const uint32_t* map, uint32_t n_src,
                                                          const uint8_t* const* src, const uint64_t* src_item_off) {
  const uint32_t i = blockIdx.x;
  const uint32_t l = layout[i];
  const uint32_t m0 = layout_off[l], bytes = layout_off[l + 1] - m0;
  uint8_t* o = out + out_off[i];
  for (uint32_t b = threadIdx.x; b < bytes; b += blockDim.x) {
    const uint32_t v = map[m0 + b];
    const uint32_t k = v >> 24;
    uint8_t byte;
    if (k == 0xFFu) byte = (uint8_t)v;
    else byte = src[k][src_item_off[(size_t)k * n + i] + (v & 0xFFFFFFu)];
    o[b] = byte;
  }
}
// the reverse for a decrypt: the parts of n records gathered into dense arrays.  Part j of item i: `bytes[j]` bytes from
// blob + rec_off[i] + rel[j-th entry of the item's layout] to dst[k] + dst_item_off[k * n + i] + dst_rel
__global__ void __launch_bounds__(256) k_gather_parts(uint32_t n, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* layout,
                                                      const uint32_t* layout_off, const uint32_t* part_src /*[parts]*/, const uint32_t* part_dst,
                                                      const uint32_t* part_len, const uint32_t* part_k, uint8_t* const* dst,
                                                      const uint64_t* dst_item_off) {
  const uint32_t i = blockIdx.x;
  const uint32_t l = layout[i];
  const uint8_t* r = blob + rec_off[i];
  for (uint32_t p = layout_off[l]; p < layout_off[l + 1]; p++) {
    const uint32_t k = part_k[p], len = part_len[p];
    uint8_t* d = dst[k] + dst_item_off[(size_t)k * n + i] + part_dst[p];
    const uint8_t* s = r + part_src[p];
    for (uint32_t b = threadIdx.x; b < len; b += blockDim.x) d[b] = s[b];
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int32_t rhip_sha3_256_batch(rhip_ctx* ctx, size_t n, const uint8_t* dev_data, const uint64_t* dev_off, uint8_t* dev_digest) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  if (n > 0xFFFFFFFFull || !dev_data || !dev_off || !dev_digest) return RHIP_ERR_ARG;
  KLAUNCH(ctx, "k_sha3_256", k_sha3_256, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, (uint32_t)n, dev_data, dev_off, (uint32_t*)dev_digest);
  return RHIP_OK;
}
extern "C" int32_t rhip_sha3_fr_batch(rhip_ctx* ctx, size_t n, const uint8_t* dev_data, const uint64_t* dev_off, rhip_fr* dev_out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  if (n > 0xFFFFFFFFull || !dev_data || !dev_off || !dev_out) return RHIP_ERR_ARG;
  // the digest is written into the output slot and reduced in place (32 bytes either way)
  KLAUNCH(ctx, "k_sha3_256", k_sha3_256, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, (uint32_t)n, dev_data, dev_off, (uint32_t*)dev_out);
  KLAUNCH(ctx, "k_digest_to_fr", k_digest_to_fr, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, (uint32_t)n, (const uint32_t*)dev_out, dev_out);
  return RHIP_OK;
}
extern "C" int32_t rhip_gt_kdf_batch(rhip_ctx* ctx, size_t n, const rhip_gt* dev_gt, const uint32_t* dev_gt_idx, uint8_t* dev_keys) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  if (n > 0xFFFFFFFFull || !dev_gt || !dev_keys) return RHIP_ERR_ARG;
  KLAUNCH(ctx, "k_sym_kdf", k_sym_kdf, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, (uint32_t)n, dev_gt, dev_gt_idx, (uint32_t*)dev_keys);
  return RHIP_OK;
}
extern "C" int32_t rhip_aes256_encrypt_blocks(rhip_ctx* ctx, size_t n, const uint8_t* dev_keys, const uint8_t* dev_in, uint8_t* dev_out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  if (n > 0xFFFFFFFFull || !dev_keys || !dev_in || !dev_out) return RHIP_ERR_ARG;
  KLAUNCH(ctx, "k_aes256_blocks", k_aes256_blocks, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, (uint32_t)n, (const uint32_t*)dev_keys,
          (const uint32_t*)dev_in, (uint32_t*)dev_out);
  return RHIP_OK;
}
static inline size_t pad64(size_t v) { return (v + 63) / 64 * 64; }
static size_t rhip_gcm_workspace_bytes(size_t n, size_t total_segments) {
  return pad64(n) * SYM_CTX_WORDS * 4 + (total_segments + 1) * 16 + 256;
}
// common tail of seal / open: keys are in `dev_keys`
static int32_t gcm_run(rhip_ctx* ctx, int seal, size_t n, const uint8_t* dev_keys, const uint8_t* dev_nonce, const uint64_t* dev_nonce_off,
                       const uint8_t* dev_in, const uint64_t* dev_in_off, uint8_t* dev_out, const uint64_t* dev_out_off, const uint32_t* dev_len,
                       const uint32_t* dev_blk_off, size_t total_blocks, const uint32_t* dev_seg_off, size_t total_segs, int len_prefix,
                       uint32_t* dev_ok, void* dev_ws) {
  const uint32_t stride = (uint32_t)pad64(n);
  uint32_t* ctxw = (uint32_t*)dev_ws;
  uint32_t* part = ctxw + (size_t)stride * SYM_CTX_WORDS;
  const uint32_t nn = (uint32_t)n;
  if (seal) {
    // in = plaintext, out = the sealed area (nonce || ciphertext || tag) of every item
    KLAUNCH(ctx, "k_sym_setup", k_sym_setup, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, nn, (const uint32_t*)dev_keys, dev_nonce, dev_nonce_off,
            ctxw, stride);
    if (total_blocks) {
      // ciphertext bytes go 12 bytes into the sealed area: the kernels take that as a shifted base pointer
      KLAUNCH(ctx, "k_sym_ctr", k_sym_ctr, dim3(blocks_for(total_blocks, 256)), dim3(256), 0, ctx->stream, nn, (uint32_t)total_blocks, dev_blk_off, ctxw,
              stride, dev_nonce, dev_nonce_off, dev_in, dev_in_off, dev_out + 12, dev_out_off, dev_len, (const uint32_t*)nullptr);
      KLAUNCH(ctx, "k_sym_ghash_seg", k_sym_ghash_seg, dim3(blocks_for(total_segs, 256)), dim3(256), 0, ctx->stream, nn, (uint32_t)total_segs,
              dev_seg_off, ctxw, stride, dev_out + 12, dev_out_off, dev_len, part);
    }
    KLAUNCH(ctx, "k_sym_tag", k_sym_tag, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, nn, dev_seg_off, ctxw, stride, part, dev_nonce,
            dev_nonce_off, dev_out, dev_out_off, dev_len, 1, len_prefix, (uint32_t*)nullptr);
  } else {
    // in = the sealed area; its first 12 bytes are the nonce
    KLAUNCH(ctx, "k_sym_setup", k_sym_setup, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, nn, (const uint32_t*)dev_keys, dev_in, dev_in_off, ctxw,
            stride);
    if (total_blocks)
      KLAUNCH(ctx, "k_sym_ghash_seg", k_sym_ghash_seg, dim3(blocks_for(total_segs, 256)), dim3(256), 0, ctx->stream, nn, (uint32_t)total_segs,
              dev_seg_off, ctxw, stride, dev_in + 12, dev_in_off, dev_len, part);
    KLAUNCH(ctx, "k_sym_tag", k_sym_tag, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, nn, dev_seg_off, ctxw, stride, part,
            (const uint8_t*)nullptr, (const uint64_t*)nullptr, (uint8_t*)dev_in, dev_in_off, dev_len, 0, 0, dev_ok);
    if (total_blocks)
      KLAUNCH(ctx, "k_sym_ctr", k_sym_ctr, dim3(blocks_for(total_blocks, 256)), dim3(256), 0, ctx->stream, nn, (uint32_t)total_blocks, dev_blk_off, ctxw,
              stride, dev_in, dev_in_off, dev_in + 12, dev_in_off, dev_out, dev_out_off, dev_len, dev_ok);
  }
  return RHIP_OK;
}
extern "C" int32_t rhip_aes256_gcm_batch(rhip_ctx* ctx, int32_t seal, size_t n, const uint8_t* dev_keys, const uint8_t* dev_nonce,
                                         const uint8_t* dev_in, const uint64_t* dev_in_off, uint8_t* dev_out, const uint64_t* dev_out_off,
                                         const uint32_t* dev_len, const uint32_t* dev_blk_off, size_t total_blocks, const uint32_t* dev_seg_off,
                                         size_t total_segments, int32_t len_prefix, uint32_t* dev_ok, void* dev_ws) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  if (n > 0x7FFFFFFFull || total_blocks > 0xFFFFFFF0ull || !dev_keys || !dev_in || !dev_in_off || !dev_out || !dev_out_off || !dev_len ||
      !dev_blk_off || !dev_seg_off || !dev_ws || (seal && !dev_nonce) || (!seal && !dev_ok))
    return RHIP_ERR_ARG;
  return gcm_run(ctx, seal, n, dev_keys, dev_nonce, nullptr, dev_in, dev_in_off, dev_out, dev_out_off, dev_len, dev_blk_off, total_blocks, dev_seg_off,
                 total_segments, len_prefix, dev_ok, dev_ws);
}
extern "C" int32_t rhip_seal_batch(rhip_ctx* ctx, size_t n, const rhip_gt* dev_gt, const uint8_t* dev_nonce, const uint8_t* dev_pt,
                                   const uint64_t* dev_pt_off, uint8_t* dev_out, const uint64_t* dev_sealed_off, const uint32_t* dev_len,
                                   const uint32_t* dev_blk_off, size_t total_blocks, const uint32_t* dev_seg_off, size_t total_segments,
                                   int32_t len_prefix, void* dev_ws) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  if (n > 0x7FFFFFFFull || total_blocks > 0xFFFFFFF0ull || !dev_gt || !dev_nonce || !dev_pt || !dev_pt_off || !dev_out || !dev_sealed_off ||
      !dev_len || !dev_blk_off || !dev_seg_off || !dev_ws)
    return RHIP_ERR_ARG;
  uint8_t* keys = (uint8_t*)dev_ws + rhip_gcm_workspace_bytes(n, total_segments);
  int32_t rc = rhip_gt_kdf_batch(ctx, n, dev_gt, nullptr, keys);
  if (rc) return rc;
  return gcm_run(ctx, 1, n, keys, dev_nonce, nullptr, dev_pt, dev_pt_off, dev_out, dev_sealed_off, dev_len, dev_blk_off, total_blocks, dev_seg_off,
                 total_segments, len_prefix, nullptr, dev_ws);
}
extern "C" int32_t rhip_open_batch(rhip_ctx* ctx, size_t n, const rhip_gt* dev_gt, const uint32_t* dev_gt_idx, const uint8_t* dev_blob,
                                   const uint64_t* dev_sealed_off, uint8_t* dev_pt, const uint64_t* dev_pt_off, const uint32_t* dev_len,
                                   const uint32_t* dev_blk_off, size_t total_blocks, const uint32_t* dev_seg_off, size_t total_segments,
                                   uint32_t* dev_ok, void* dev_ws) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  if (n > 0x7FFFFFFFull || total_blocks > 0xFFFFFFF0ull || !dev_gt || !dev_blob || !dev_sealed_off || !dev_pt || !dev_pt_off || !dev_len ||
      !dev_blk_off || !dev_seg_off || !dev_ok || !dev_ws)
    return RHIP_ERR_ARG;
  uint8_t* keys = (uint8_t*)dev_ws + rhip_gcm_workspace_bytes(n, total_segments);
  int32_t rc = rhip_gt_kdf_batch(ctx, n, dev_gt, dev_gt_idx, keys);
  if (rc) return rc;
  return gcm_run(ctx, 0, n, keys, nullptr, nullptr, dev_blob, dev_sealed_off, dev_pt, dev_pt_off, dev_len, dev_blk_off, total_blocks, dev_seg_off,
                 total_segments, 0, dev_ok, dev_ws);
}
extern "C" size_t rhip_seal_workspace_bytes(size_t n, size_t total_segments) { return rhip_gcm_workspace_bytes(n, total_segments) + n * 32 + 64; }
extern "C" int32_t rhip_assemble_records(rhip_ctx* ctx, size_t n, uint8_t* dev_out, const uint64_t* dev_out_off, const uint32_t* dev_layout,
                                         const uint32_t* dev_layout_off, const uint32_t* dev_map, uint32_t n_src, const uint8_t* const* dev_src,
                                         const uint64_t* dev_src_item_off) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  if (n > 0x7FFFFFFFull || n_src > 254 || !dev_out || !dev_out_off || !dev_layout || !dev_layout_off || !dev_map || !dev_src || !dev_src_item_off)
    return RHIP_ERR_ARG;
  KLAUNCH(ctx, "k_assemble_records", k_assemble_records, dim3((unsigned)n), dim3(256), 0, ctx->stream, (uint32_t)n, dev_out, dev_out_off, dev_layout,
          dev_layout_off, dev_map, n_src, dev_src, dev_src_item_off);
  return RHIP_OK;
}
extern "C" int32_t rhip_gather_parts(rhip_ctx* ctx, size_t n, const uint8_t* dev_blob, const uint64_t* dev_rec_off, const uint32_t* dev_layout,
                                     const uint32_t* dev_layout_off, const uint32_t* dev_part_src, const uint32_t* dev_part_dst,
                                     const uint32_t* dev_part_len, const uint32_t* dev_part_k, uint8_t* const* dev_dst,
                                     const uint64_t* dev_dst_item_off) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  if (n > 0x7FFFFFFFull || !dev_blob || !dev_rec_off || !dev_layout || !dev_layout_off || !dev_part_src || !dev_part_dst || !dev_part_len ||
      !dev_part_k || !dev_dst || !dev_dst_item_off)
    return RHIP_ERR_ARG;
  KLAUNCH(ctx, "k_gather_parts", k_gather_parts, dim3((unsigned)n), dim3(256), 0, ctx->stream, (uint32_t)n, dev_blob, dev_rec_off, dev_layout,
          dev_layout_off, dev_part_src, dev_part_dst, dev_part_len, dev_part_k, dev_dst, dev_dst_item_off);
  return RHIP_OK;
}
