// sha256.h -- FIPS 180-4 SHA-256 and RFC 9380 section 5.3.1 expand_message_xmd, one instance per lane.
// Replaces elliptic_curve::hash2curve::ExpandMsgXmd<sha2::Sha256> as called from Fp::hash / Fp2::hash
// (fp.rs:433-458, fp2.rs:463-487).  32-bit rotates and adds only: native VALU work.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "fp29.h"

namespace bn {

struct Sha256 {
  uint32_t h[8];
  uint32_t w[16];      // current block, big-endian words (complete words only)
  uint32_t cur;        // the word being assembled
  uint32_t fill;       // bytes taken into the current block
  uint64_t total;
};

BN_INL uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// A real function on the device: the byte-wise update below has a compression behind every "block full" test, and inlined
// the hash kernel carried 65 unrolled copies (143 k of its 190 k instructions, 1.1 MB of code for nine compressions executed).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BN_SHA_INLINE)
#define BN_SHA_FUNC BN_HD __attribute__((noinline))
#else
#define BN_SHA_FUNC BN_FUNC
#endif
BN_SHA_FUNC void sha256_compress(uint32_t* h, const uint32_t* blk) {
  const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
      0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
      0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
      0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
      0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
      0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t w[16];
  BN_UNROLL for (int i = 0; i < 16; ++i) w[i] = blk[i];
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  BN_UNROLL for (int i = 0; i < 64; ++i) {
    if (i >= 16) {
      uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      uint32_t s0 = ror32(w15, 7) ^ ror32(w15, 18) ^ (w15 >> 3);
      uint32_t s1 = ror32(w2, 17) ^ ror32(w2, 19) ^ (w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
    }
    uint32_t S1 = ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25), ch = (e & f) ^ (~e & g);
    uint32_t t1 = hh + S1 + ch + K[i] + w[i & 15];
    uint32_t S0 = ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
BN_FUNC void sha256_init(Sha256& s) {
  s.h[0] = 0x6a09e667; s.h[1] = 0xbb67ae85; s.h[2] = 0x3c6ef372; s.h[3] = 0xa54ff53a;
  s.h[4] = 0x510e527f; s.h[5] = 0x9b05688c; s.h[6] = 0x1f83d9ab; s.h[7] = 0x5be0cd19;
  for (int i = 0; i < 16; ++i) s.w[i] = 0;
  s.cur = 0; s.fill = 0; s.total = 0;
}
// One byte: assembled in a register, the block array is written once per completed word (it is indexed by a
// run-time value, so it lives in scratch: a read-modify-write per byte was an exposed scratch round trip per byte).
BN_FUNC void sha256_byte(Sha256& s, uint8_t b) {
  s.cur = (s.cur << 8) | b;
  ++s.fill; ++s.total;
  if ((s.fill & 3) == 0) {
    s.w[(s.fill >> 2) - 1] = s.cur;
    s.cur = 0;
    if (s.fill == 64) { sha256_compress(s.h, s.w); s.fill = 0; }
  }
}
BN_FUNC void sha256_update(Sha256& s, const uint8_t* p, size_t n) { for (size_t i = 0; i < n; ++i) sha256_byte(s, p[i]); }
BN_FUNC void sha256_final(Sha256& s, uint8_t* out) {
  uint64_t bits = s.total * 8;
  sha256_byte(s, 0x80);
  while (s.fill != 56) sha256_byte(s, 0);
  for (int i = 0; i < 8; ++i) sha256_byte(s, (uint8_t)(bits >> (8 * (7 - i))));
  for (int i = 0; i < 8; ++i) store_be32(out + 4 * i, s.h[i]);
}

// out[0..n) = expand_message_xmd(msg, dst, n), n <= 192 here (ell <= 6).  dst_len <= 255 (longer
// DSTs are pre-hashed on the host, RFC 9380 5.3.3).
BN_FUNC void expand_message_xmd(uint8_t* out, uint32_t n, const uint8_t* msg, size_t msg_len,
                                     const uint8_t* dst, uint32_t dst_len) {
  uint8_t b0[32], bi[32];
  Sha256 s;
  sha256_init(s);
  for (int i = 0; i < 64; ++i) sha256_byte(s, 0);                 // Z_pad
  sha256_update(s, msg, msg_len);
  sha256_byte(s, (uint8_t)(n >> 8)); sha256_byte(s, (uint8_t)n); sha256_byte(s, 0);
  sha256_update(s, dst, dst_len); sha256_byte(s, (uint8_t)dst_len);
  sha256_final(s, b0);
  sha256_init(s);
  sha256_update(s, b0, 32); sha256_byte(s, 1);
  sha256_update(s, dst, dst_len); sha256_byte(s, (uint8_t)dst_len);
  sha256_final(s, bi);
  uint32_t ell = (n + 31) / 32, done = 0;
  for (uint32_t i = 1; i <= ell; ++i) {
    for (uint32_t k = 0; k < 32 && done < n; ++k) out[done++] = bi[k];
    if (i == ell) break;
    sha256_init(s);
    for (int k = 0; k < 32; ++k) sha256_byte(s, b0[k] ^ bi[k]);
    sha256_byte(s, (uint8_t)(i + 1));
    sha256_update(s, dst, dst_len); sha256_byte(s, (uint8_t)dst_len);
    sha256_final(s, bi);
  }
}

}  // namespace bn
