// fr29.h -- scalar field Fr of BN254 on the device, for the threshold path's Lagrange coefficients
// lambda_i = prod_{j != i} x_j / (x_j - x_i)  (Scalar add/sub/mul/invert, scalar.rs:523-548, :216-219).
// Same radix-2^29 / R = 2^261 Montgomery layout as fp29.h but always fully normalised (this path is
// O(t) multiplies per lane and nowhere near the hot loop, so no lazy-limb tricks here).
// Wire format: 32 bytes big-endian in both directions (the reference's little-endian to_repr, E11,
// is not reproduced).
#pragma once
#include "fp29.h"

namespace bn {

struct Fr { int32_t l[NL]; };      // strict limbs, value in [0, r)

BN_INL Fr fr_reduce_once(const int32_t* a) {          // strict-limb input in (-r, 2r) -> [0, r)
  Fr u, t, r;
  int32_t cu = 0, ct = 0;
  BN_UNROLL for (int i = 0; i < NL - 1; ++i) {
    int32_t du = a[i] + bnc::FR_MOD[i] + cu; u.l[i] = du & MASK; cu = du >> RB;
    int32_t dt = a[i] - bnc::FR_MOD[i] + ct; t.l[i] = dt & MASK; ct = dt >> RB;
  }
  u.l[NL - 1] = a[NL - 1] + bnc::FR_MOD[NL - 1] + cu;
  t.l[NL - 1] = a[NL - 1] - bnc::FR_MOD[NL - 1] + ct;
  bool neg = a[NL - 1] < 0, ge = t.l[NL - 1] >= 0;
  BN_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = neg ? u.l[i] : (ge ? t.l[i] : a[i]);
  return r;
}
BN_INL Fr fr_mul(const Fr& a, const Fr& b) {          // Montgomery product, canonical output
  int32_t m[NL], r[NL];
  int64_t acc = 0;
  BN_UNROLL for (int k = 0; k < NL; ++k) {
    BN_UNROLL for (int i = 0; i <= k; ++i) acc += (int64_t)a.l[i] * b.l[k - i];
    BN_UNROLL for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * bnc::FR_MOD[k - i];
    m[k] = (int32_t)(((uint32_t)acc * (uint32_t)bnc::FR_PINV) & (uint32_t)MASK);
    acc += (int64_t)m[k] * bnc::FR_MOD[0];
    acc >>= RB;
  }
  BN_UNROLL for (int k = NL; k < 2 * NL - 1; ++k) {
    BN_UNROLL for (int i = k - NL + 1; i < NL; ++i) acc += (int64_t)a.l[i] * b.l[k - i] + (int64_t)m[i] * bnc::FR_MOD[k - i];
    r[k - NL] = (int32_t)((uint32_t)acc & (uint32_t)MASK);
    acc >>= RB;
  }
  r[NL - 1] = (int32_t)acc;
  return fr_reduce_once(r);
}
BN_INL Fr fr_sub(const Fr& a, const Fr& b) {
  int32_t d[NL]; int32_t c = 0;
  BN_UNROLL for (int i = 0; i < NL - 1; ++i) { int32_t v = a.l[i] - b.l[i] + c; d[i] = v & MASK; c = v >> RB; }
  d[NL - 1] = a.l[NL - 1] - b.l[NL - 1] + c;
  return fr_reduce_once(d);
}
BN_INL Fr fr_add(const Fr& a, const Fr& b) {
  int32_t d[NL]; int32_t c = 0;
  BN_UNROLL for (int i = 0; i < NL - 1; ++i) { int32_t v = a.l[i] + b.l[i] + c; d[i] = v & MASK; c = v >> RB; }
  d[NL - 1] = a.l[NL - 1] + b.l[NL - 1] + c;
  return fr_reduce_once(d);
}
BN_INL Fr fr_const(const int32_t (&c)[NL]) { Fr r; BN_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = c[i]; return r; }
BN_INL bool fr_is_zero(const Fr& a) { int32_t o = 0; BN_UNROLL for (int i = 0; i < NL; ++i) o |= a.l[i]; return o == 0; }
BN_INL Fr fr_select(bool c, const Fr& a, const Fr& b) { Fr r; BN_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = c ? a.l[i] : b.l[i]; return r; }
// 32 big-endian bytes -> Montgomery form; ok = value < r (Scalar::from_repr, scalar.rs:229-239)
BN_INL Fr fr_from_be(const uint8_t* b, bool& ok) {
  uint32_t w[8];
  BN_UNROLL for (int j = 0; j < 8; ++j) w[j] = load_be32(b + 4 * (7 - j));
  Fr x;
  words_to_limbs(x.l, w);
  int32_t c = 0;
  BN_UNROLL for (int i = 0; i < NL; ++i) { int32_t d = x.l[i] - bnc::FR_MOD[i] + c; c = d >> RB; }
  ok = c < 0;
  // values >= r are rejected by the caller; reduce anyway so the arithmetic stays in range
  Fr red = fr_reduce_once(x.l);
  return fr_mul(red, fr_const(bnc::FR_R2));
}
BN_INL void fr_to_be(uint8_t* b, const Fr& a) {
  Fr one; BN_UNROLL for (int i = 0; i < NL; ++i) one.l[i] = i == 0;
  Fr c = fr_mul(a, one);
  uint32_t w[8];
  limbs_to_words(w, c.l);
  BN_UNROLL for (int j = 0; j < 8; ++j) store_be32(b + 4 * (7 - j), w[j]);
}
// 48 big-endian bytes -> OS2IP(okm) mod r in Montgomery form (RFC 9380 hash_to_field with L = 48; the
// result Scalar::from_okm is meant to produce, scalar.rs:346-352 -- its Reduce<U384>, :393-402, subtracts
// r only once and truncates, which is not a reduction for 384-bit inputs and is not reproduced).
// value = lo + hi * 2^261  =>  lo*R + hi*R^2 = mont(lo, R^2) + mont(hi, R^3).
BN_INL Fr fr_from_okm(const uint8_t* okm) {
  uint32_t w[12];
  BN_UNROLL for (int j = 0; j < 12; ++j) w[j] = load_be32(okm + 4 * (11 - j));
  Fr lo, hi;
  BN_UNROLL for (int i = 0; i < 2 * NL; ++i) {
    int bit = RB * i, j = bit >> 5, s = bit & 31;
    uint32_t v = 0;
    if (j < 12) { v = w[j] >> s; if (s > 32 - RB && j + 1 < 12) v |= w[j + 1] << (32 - s); }
    (i < NL ? lo.l[i] : hi.l[i - NL]) = (int32_t)(v & (uint32_t)MASK);
  }
  return fr_add(fr_mul(lo, fr_const(bnc::FR_R2)), fr_mul(hi, fr_const(bnc::FR_R3)));
}
BN_FUNC Fr fr_inv(const Fr& a) {                 // a^(r-2)
  Fr r = fr_const(bnc::FR_ONE);
  for (int i = 255; i >= 0; --i) {
    r = fr_mul(r, r);
    Fr m = fr_mul(r, a);
    uint64_t w = i >= 192 ? bnc::EXP_RM2[3] : i >= 128 ? bnc::EXP_RM2[2] : i >= 64 ? bnc::EXP_RM2[1] : bnc::EXP_RM2[0];
    r = fr_select((w >> (i & 63)) & 1, m, r);
  }
  return r;
}
// lambda_i for lane i over the t ids (32 B big-endian each).  ok = all ids decode, are non-zero and
// id_i differs from every other id.
BN_FUNC Fr lagrange_at_zero(const uint8_t* ids, size_t t, size_t i, bool& ok) {
  bool oki;
  Fr xi = fr_from_be(ids + 32 * i, oki);
  ok = oki & !fr_is_zero(xi);
  Fr num = fr_const(bnc::FR_ONE), den = num, one = num;
  for (size_t j = 0; j < t; ++j) {
    bool okj;
    Fr xj = fr_from_be(ids + 32 * j, okj);
    Fr d = fr_sub(xj, xi);
    bool self = j == i;
    ok &= okj & (self | !fr_is_zero(d));
    num = fr_mul(num, fr_select(self, one, xj));
    den = fr_mul(den, fr_select(self, one, d));
  }
  return fr_mul(num, fr_inv(den));
}

}  // namespace bn
