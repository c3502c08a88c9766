// lazy_tower.h -- the lazily reduced 10 x 26-bit Fp6 product measured by lazy_tower.hip (see there); host + device.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../bls-bn254_amd/csrc/tower.h"

namespace lz {
constexpr int N = 10, RB26 = 26;
constexpr int32_t M26 = (1 << RB26) - 1;
// p = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47 in 26-bit limbs, and -p^-1 mod 2^26
BN_INL int32_t P26(int i) {
  constexpr int32_t P[N] = {0x07cfd47, 0x02305b6, 0x0a8d3c2, 0x245a1c7, 0x197816a, 0x0605617, 0x1045b68, 0x280a6e1, 0x272e131, 0x00c1913};
  return P[i];
}
constexpr uint32_t PINV26 = 0x0866389;      // -p^-1 mod 2^26 (asserted on the host)
struct F { int32_t l[N]; };                 // limbs in [0, 2^26] after a reduction (top limb small)
struct D { int64_t c[2 * N - 1]; };         // raw column sums of limb products, no carries between columns
struct F2 { F c0, c1; };
struct D2 { D c0, c1; };
struct F6 { F2 c0, c1, c2; };

BN_INL F f_sub(const F& a, const F& b) { F r; BN_UNROLL for (int i = 0; i < N; ++i) r.l[i] = a.l[i] - b.l[i]; return r; }
BN_INL F f_neg(const F& a) { F r; BN_UNROLL for (int i = 0; i < N; ++i) r.l[i] = -a.l[i]; return r; }
BN_INL F2 f2_sub(const F2& a, const F2& b) { return {f_sub(a.c0, b.c0), f_sub(a.c1, b.c1)}; }
BN_INL void d_zero(D& d) { BN_UNROLL for (int k = 0; k < 2 * N - 1; ++k) d.c[k] = 0; }
// acc += a * b, column-wise (100 MADs, every column its own chain: no serial dependency at all)
BN_INL void d_mac(D& acc, const F& a, const F& b) {
  BN_UNROLL for (int i = 0; i < N; ++i) BN_UNROLL for (int j = 0; j < N; ++j) acc.c[i + j] += (int64_t)a.l[i] * b.l[j];
}
// acc += x * y in Fp2: c0 += x0 y0 - x1 y1, c1 += x0 y1 + x1 y0  (400 MADs)
BN_INL void d2_mac(D2& acc, const F2& x, const F2& y) {
  const F nx1 = f_neg(x.c1);
  d_mac(acc.c0, x.c0, y.c0); d_mac(acc.c0, nx1, y.c1);
  d_mac(acc.c1, x.c0, y.c1); d_mac(acc.c1, x.c1, y.c0);
}
BN_INL void d_add(D& r, const D& a) { BN_UNROLL for (int k = 0; k < 2 * N - 1; ++k) r.c[k] += a.c[k]; }
BN_INL void d2_add(D2& r, const D2& a) { d_add(r.c0, a.c0); d_add(r.c1, a.c1); }
// r = xi * a + b on double-width values, xi = 9 + u: (9 a0 - a1 + b0, a0 + 9 a1 + b1), per column
BN_INL D2 d2_mul_xi_add(const D2& a, const D2& b) {
  D2 r;
  BN_UNROLL for (int k = 0; k < 2 * N - 1; ++k) {
    r.c0.c[k] = (a.c0.c[k] << 3) + a.c0.c[k] - a.c1.c[k] + b.c0.c[k];
    r.c1.c[k] = (a.c1.c[k] << 3) + a.c1.c[k] + a.c0.c[k] + b.c1.c[k];
  }
  return r;
}
// Montgomery reduction of 19 raw columns: T / 2^260 mod p, limbs 0..8 in [0, 2^26), top limb small and possibly negative
BN_INL F d_reduce(const D& t) {
  F r;
  int32_t m[N];
  int64_t acc = 0;
  BN_UNROLL for (int k = 0; k < N; ++k) {
    acc += t.c[k];
    BN_UNROLL for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * P26(k - i);
    m[k] = (int32_t)(((uint32_t)acc * PINV26) & (uint32_t)M26);
    acc += (int64_t)m[k] * P26(0);
    acc >>= RB26;
  }
  BN_UNROLL for (int k = N; k < 2 * N - 1; ++k) {
    acc += t.c[k];
    BN_UNROLL for (int i = k - N + 1; i < N; ++i) acc += (int64_t)m[i] * P26(k - i);
    r.l[k - N] = (int32_t)((uint32_t)acc & (uint32_t)M26);
    acc >>= RB26;
  }
  r.l[N - 1] = (int32_t)acc;
  return r;
}
BN_INL F2 d2_reduce(const D2& t) { return {d_reduce(t.c0), d_reduce(t.c1)}; }
// limbs back to [0, 2^26) with the carry pushed up (the reduction's top limb may be slightly negative / outputs feed the next product)
BN_INL F f_norm(const F& a) {
  F r; int32_t c = 0;
  BN_UNROLL for (int i = 0; i < N - 1; ++i) { int32_t v = a.l[i] + c; r.l[i] = v & M26; c = v >> RB26; }
  r.l[N - 1] = a.l[N - 1] + c;
  return r;
}
// a * b in Fp6 = Fp2[v] / (v^3 - xi), Karatsuba over Fp2 with everything before the six reductions kept double-width:
//   c0 = V0 + xi (V1 + V2 - W0), c1 = V0 + V1 - W1 + xi V2, c2 = V0 + V1 + V2 - W2,  Wk = the three cross products of differences
// Operand differences of normalised limbs stay within one limb width, so the products need no normalisation; the cross
// products are accumulated with one operand negated, straight into the column sums of their output.
BN_FUNC F6 fp6_mul_lazy(const F6& a, const F6& b) {
  D2 V0, V1, V2;
  d_zero(V0.c0); d_zero(V0.c1); d_zero(V1.c0); d_zero(V1.c1); d_zero(V2.c0); d_zero(V2.c1);
  d2_mac(V0, a.c0, b.c0); d2_mac(V1, a.c1, b.c1); d2_mac(V2, a.c2, b.c2);
  F6 r;
  {   // c2 = V0 + V1 + V2 + (a2 - a0)(b0 - b2)
    D2 t = V0; d2_add(t, V1); d2_add(t, V2);
    d2_mac(t, f2_sub(a.c2, a.c0), f2_sub(b.c0, b.c2));
    r.c2 = d2_reduce(t);
  }
  {   // c1 = xi V2 + (V0 + V1) + (a1 - a0)(b0 - b1)
    D2 s = V0; d2_add(s, V1);
    D2 t = d2_mul_xi_add(V2, s);
    d2_mac(t, f2_sub(a.c1, a.c0), f2_sub(b.c0, b.c1));
    r.c1 = d2_reduce(t);
  }
  {   // c0 = xi (V1 + V2 + (a2 - a1)(b1 - b2)) + V0
    d2_add(V1, V2);
    d2_mac(V1, f2_sub(a.c2, a.c1), f2_sub(b.c1, b.c2));
    r.c0 = d2_reduce(d2_mul_xi_add(V1, V0));
  }
  return r;
}
BN_INL F6 f6_norm(const F6& a) {
  return {{f_norm(a.c0.c0), f_norm(a.c0.c1)}, {f_norm(a.c1.c0), f_norm(a.c1.c1)}, {f_norm(a.c2.c0), f_norm(a.c2.c1)}};
}
}  // namespace lz
