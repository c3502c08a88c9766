// glv.h -- scalar decomposition for the G1 endomorphism phi(x, y) = (beta x, y) = [lambda](x, y) of BN254
// (Gallant-Lambert-Vanstone).  Used by the threshold path's multi-scalar multiplication
//     sum_i lambda_i * sigma_i     (Mul<Scalar> g1.rs:518-534 + Sum g1.rs:561-565; the reference multiplies by the full
// 255-bit scalar, g1.rs:821-841): k = k1 + k2 * lambda (mod r) with |k1|, |k2| < 2^127, so the doubling chain on the
// critical path is 127 long instead of 254.  The group element is the same, hence the same output bytes.
//
// k1 = k - c1 a1 - c2 a2,  k2 = -c1 b1 - c2 b2  with  c1 = floor(k G1 / 2^256) ~ b2 k / r,  c2 = floor(k G2 / 2^256) ~ -b1 k / r
// for the short basis (a1, b1), (a2, b2) of {(a, b): a + b lambda = 0 mod r}; all of it in 256-bit two's-complement word
// arithmetic.  Constants and the bound |k1|, |k2| <= 2^127 (checked on 20 000 random and the edge scalars with this exact
// word algorithm) come from tools/gen_constants.py.
#pragma once
#include "fp29.h"

namespace bn {

struct GlvSplit { uint32_t k1[4], k2[4]; bool neg1, neg2; };     // magnitudes (128 bits, little-endian words) and signs

// low 256 bits of a * b (8 x 32-bit words each)
BN_INL void mul_lo_256(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint64_t carry = 0;
  BN_UNROLL for (int k = 0; k < 8; ++k) {
    uint64_t lo = carry, hi = 0;
    BN_UNROLL for (int i = 0; i <= k; ++i) {
      uint64_t p = (uint64_t)a[i] * b[k - i];
      lo += p & 0xffffffffu; hi += p >> 32;
    }
    r[k] = (uint32_t)lo;
    carry = (lo >> 32) + hi;
  }
}
// words 8..15 of the 512-bit product a * b, i.e. floor(a b / 2^256)
BN_INL void mul_hi_256(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint64_t carry = 0;
  BN_UNROLL for (int k = 0; k < 16; ++k) {
    uint64_t lo = carry, hi = 0;
    BN_UNROLL for (int i = 0; i < 8; ++i) {
      int j = k - i;
      if (j < 0 || j >= 8) continue;
      uint64_t p = (uint64_t)a[i] * b[j];
      lo += p & 0xffffffffu; hi += p >> 32;
    }
    if (k >= 8) r[k - 8] = (uint32_t)lo;
    carry = (lo >> 32) + hi;
  }
}
BN_INL void neg_256(uint32_t* a) {
  uint64_t c = 1;
  BN_UNROLL for (int i = 0; i < 8; ++i) { c += (uint32_t)~a[i]; a[i] = (uint32_t)c; c >>= 32; }
}
BN_INL void sub_256(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  int64_t c = 0;
  BN_UNROLL for (int i = 0; i < 8; ++i) { c += (int64_t)a[i] - (int64_t)b[i]; r[i] = (uint32_t)c; c >>= 32; }
}
BN_INL void add_256(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint64_t c = 0;
  BN_UNROLL for (int i = 0; i < 8; ++i) { c += (uint64_t)a[i] + b[i]; r[i] = (uint32_t)c; c >>= 32; }
}
// k: canonical scalar (< r), 8 little-endian words
BN_INL GlvSplit glv_split(const uint32_t* k) {
  uint32_t c1[8], c2[8], t[8], u[8], k1[8], k2[8];
  mul_hi_256(c1, k, bnc::GLV_G1);
  mul_hi_256(c2, k, bnc::GLV_G2);
  if (bnc::GLV_C1_NEG) neg_256(c1);
  if (bnc::GLV_C2_NEG) neg_256(c2);
  mul_lo_256(t, c1, bnc::GLV_A1); mul_lo_256(u, c2, bnc::GLV_A2);
  sub_256(k1, k, t); sub_256(k1, k1, u);
  mul_lo_256(t, c1, bnc::GLV_B1); mul_lo_256(u, c2, bnc::GLV_B2);
  add_256(k2, t, u); neg_256(k2);
  GlvSplit s;
  s.neg1 = (k1[7] >> 31) != 0; s.neg2 = (k2[7] >> 31) != 0;
  if (s.neg1) neg_256(k1);
  if (s.neg2) neg_256(k2);
#ifdef BN_CHECK
  // hostsim: the halves must fit the 128 bits the window kernels read (analytic bound asserted in tools/gen_constants.py)
  if (k1[4] | k1[5] | k1[6] | k1[7] | k2[4] | k2[5] | k2[6] | k2[7]) check_fail("glv_split half exceeds 128 bits", 0);
#endif
  BN_UNROLL for (int i = 0; i < 4; ++i) { s.k1[i] = k1[i]; s.k2[i] = k2[i]; }
  return s;
}

}  // namespace bn
