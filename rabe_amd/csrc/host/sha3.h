// SHA3-256 (FIPS 202) for the host layer: label hashing (src/utils/hash/mod.rs:10-31) and the
// KDF in front of AES (src/utils/aes/mod.rs:47-55).  The reference uses the `sha3 0.10` crate.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

namespace rabe { namespace host {

inline void keccak_f1600(uint64_t s[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
      0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  static const int RHO[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int PI[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  for (int round = 0; round < 24; round++) {
    uint64_t c[5];
    for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
    for (int x = 0; x < 5; x++) {
      uint64_t d = c[(x + 4) % 5] ^ ((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63));
      for (int y = 0; y < 25; y += 5) s[y + x] ^= d;
    }
    uint64_t cur = s[1];
    for (int i = 0; i < 24; i++) {
      int j = PI[i];
      uint64_t nxt = s[j];
      s[j] = (cur << RHO[i]) | (cur >> (64 - RHO[i]));
      cur = nxt;
    }
    for (int y = 0; y < 25; y += 5) {
      uint64_t row[5];
      for (int x = 0; x < 5; x++) row[x] = s[y + x];
      for (int x = 0; x < 5; x++) s[y + x] = row[x] ^ ((~row[(x + 1) % 5]) & row[(x + 2) % 5]);
    }
    s[0] ^= RC[round];
  }
}

inline void sha3_256(const uint8_t* data, size_t len, uint8_t out[32]) {
  const size_t rate = 136;
  uint64_t s[25];
  memset(s, 0, sizeof s);
  while (len >= rate) {
    for (size_t i = 0; i < rate / 8; i++) { uint64_t w; memcpy(&w, data + 8 * i, 8); s[i] ^= w; }
    keccak_f1600(s);
    data += rate;
    len -= rate;
  }
  uint8_t blk[136];
  memset(blk, 0, rate);
  memcpy(blk, data, len);
  blk[len] ^= 0x06;
  blk[rate - 1] ^= 0x80;
  for (size_t i = 0; i < rate / 8; i++) { uint64_t w; memcpy(&w, blk + 8 * i, 8); s[i] ^= w; }
  keccak_f1600(s);
  memcpy(out, s, 32);
}
inline std::vector<uint8_t> sha3_256(const std::string& sdata) {
  std::vector<uint8_t> o(32);
  sha3_256((const uint8_t*)sdata.data(), sdata.size(), o.data());
  return o;
}

}}  // namespace rabe::host
