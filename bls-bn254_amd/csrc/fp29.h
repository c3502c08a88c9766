// fp29.h -- BN254 base field for gfx950: 9 signed limbs, radix 2^29, Montgomery form (R = 2^261).
//
// Why this shape (measured on MI355X, profiles/valu_peak_r01.json): v_mad_u64_u32 / v_mad_i64_i32
// issue at ~1.2-1.3x the cost of a plain VALU op, so a carry instruction costs as much as a
// multiply.  A saturated 8x32-bit Montgomery multiply needs one add-with-carry per MAD (~300 VALU
// ops); with 29-bit limbs every column sum of 9 products plus the 9 reduction products fits a
// signed 64-bit accumulator, so a modular multiply is 162 v_mad_i64_i32 + 43 shift/mask ops and
// NO carry chain, and field additions/subtractions are 9 independent 32-bit ops with no reduction
// ("lazy").  Replaces the reference's canonical-form wide-multiply + bit-serial division
// (Fp::multiply, fp.rs:404-407; add/sub fp.rs:388-402).
//
// Representation contract ("interval discipline"), in units of L = 2^29:
//   * fp_mul / fp_sqr / fp_dot2 outputs: limbs 0..7 in [0, L), limb 8 small and possibly negative,
//     value in (-eps*p, (1+eps)*p).  Not canonical; congruent mod p.
//   * fp_add / fp_sub / fp_neg are limb-wise and grow the limb interval.
//   * fp_mul(a,b) requires mag(a)*mag(b) < 2.97 (mag = max |limb 0..7| / L), fp_dot2 requires
//     mag(a)mag(b)+mag(c)mag(d) < 2.97: then every column (<= 8 full operand products + 8
//     reduction products, the top limbs being small) stays below 2^63.
//   * fp_norm / fp_lc* bring limbs back to ~[0, L] without changing the value.
// The discipline is machine-checked: compiled for the host with -DBN_CHECK every Fp carries its
// proven limb interval and value bound and every multiply asserts its precondition
// (tests/hostsim; data-independent interval arithmetic, so one run covers all inputs).
#pragma once
#include <stdint.h>
#include "bn254_consts.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BN_HD __host__ __device__
#else
#define BN_HD
#endif
#define BN_INL BN_HD inline __attribute__((always_inline))
// BN_FUNC marks the large building blocks (tower multiplications, line steps, exponentiations).  By
// default the compiler decides whether to outline them; -DBN_FORCE_INLINE inlines them all.
#ifdef BN_FORCE_INLINE
#define BN_FUNC BN_INL
#else
#define BN_FUNC BN_HD inline
#endif

#define BN_UNROLL _Pragma("unroll")
// nothing is scheduled across this point (device only): used to keep a prefetch above the work that hides it
#if defined(__HIP_DEVICE_COMPILE__)
#define BN_SCHED_BARRIER __builtin_amdgcn_sched_barrier(0)
#else
#define BN_SCHED_BARRIER ((void)0)
#endif

#ifdef BN_CHECK
#include <cmath>
#include <cstdio>
#include <cstdlib>
#endif

namespace bn {

constexpr int NL = 9;
constexpr int RB = 29;
constexpr int32_t MASK = (1 << RB) - 1;

struct Fp {
  int32_t l[NL];
#ifdef BN_CHECK
  double lo = 0, hi = 0;    // proven interval of limbs 0..7, units of 2^29
  double tlo = 0, thi = 0;  // proven interval of the top limb (limb 8)
  double vb = 0;            // proven bound on |value| / p
#endif
};

// ------------------------------------------------------------------ check-mode instrumentation
#ifdef BN_CHECK
struct CheckStats { double worst_mul = 0, worst_dot = 0, worst_vb = 0; long muls = 0, sqrs = 0, dots = 0, norms = 0, lcs = 0, lc_terms = 0; };
inline CheckStats& check_stats() { static thread_local CheckStats s; return s; }    // per thread: tests/hostsim runs the lanes of a quad (tri.h) as threads
inline double mag(const Fp& a) { return std::fmax(std::fabs(a.lo), std::fabs(a.hi)); }
inline double tmag(const Fp& a) { return std::fmax(std::fabs(a.tlo), std::fabs(a.thi)); }
struct CtxStack { const char* name[64]; int line[64]; int n = 0; };
inline CtxStack& ctx_stack() { static thread_local CtxStack s; return s; }
struct CtxGuard {
  CtxGuard(const char* f, int l) { CtxStack& s = ctx_stack(); if (s.n < 64) { s.name[s.n] = f; s.line[s.n] = l; } ++s.n; }
  ~CtxGuard() { --ctx_stack().n; }
};
inline void check_fail(const char* what, double v) {
  std::fprintf(stderr, "BN_CHECK violation: %s (%g)\n  context:", what, v);
  CtxStack& s = ctx_stack();
  for (int i = 0; i < s.n && i < 64; ++i) std::fprintf(stderr, " > %s:%d", s.name[i], s.line[i]);
  std::fprintf(stderr, "\n");
  std::abort();
}
#define BN_CTX ::bn::CtxGuard _bn_ctx_guard(__func__, __LINE__)
inline void check_actual(const Fp& a, const char* where) {
  const double L = 536870912.0;
  for (int i = 0; i < NL; ++i) {
    double v = a.l[i] / L, lo = i == NL - 1 ? a.tlo : a.lo, hi = i == NL - 1 ? a.thi : a.hi;
    if (v < lo - 1e-6 || v > hi + 1e-6) { std::fprintf(stderr, "limb %d = %g outside [%g,%g] at %s\n", i, v, lo, hi, where); check_fail("actual limb outside proven interval", v); }
  }
}
inline void set_trk(Fp& r, double lo, double hi, double tlo, double thi, double vb) {
  r.lo = lo; r.hi = hi; r.tlo = tlo; r.thi = thi; r.vb = vb;
  if (vb > check_stats().worst_vb) check_stats().worst_vb = vb;
}
#define BN_TRK(...) __VA_ARGS__
#else
#define BN_TRK(...)
#define BN_CTX
#endif

constexpr double P_OVER_R = 0.0059605;   // p / 2^261 (rounded up)

// ------------------------------------------------------------------ lazy linear ops
BN_INL Fp fp_add(const Fp& a, const Fp& b) {
  Fp r;
  BN_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + b.l[i];
  BN_TRK(set_trk(r, a.lo + b.lo, a.hi + b.hi, a.tlo + b.tlo, a.thi + b.thi, a.vb + b.vb); if (mag(r) >= 3.99 || tmag(r) >= 3.99) check_fail("fp_add limb overflow", mag(r));)
  return r;
}
BN_INL Fp fp_sub(const Fp& a, const Fp& b) {
  Fp r;
  BN_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] - b.l[i];
  BN_TRK(set_trk(r, a.lo - b.hi, a.hi - b.lo, a.tlo - b.thi, a.thi - b.tlo, a.vb + b.vb); if (mag(r) >= 3.99 || tmag(r) >= 3.99) check_fail("fp_sub limb overflow", mag(r));)
  return r;
}
BN_INL Fp fp_neg(const Fp& a) {
  Fp r;
  BN_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = -a.l[i];
  BN_TRK(set_trk(r, -a.hi, -a.lo, -a.thi, -a.tlo, a.vb);)
  return r;
}
BN_INL Fp fp_dbl(const Fp& a) { return fp_add(a, a); }

BN_INL Fp fp_zero() {
  Fp r;
  BN_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = 0;
  BN_TRK(set_trk(r, 0, 0, 0, 0, 0);)
  return r;
}
BN_INL Fp fp_const(const int32_t (&c)[NL]) {          // a generated constant (Montgomery form, < p)
  Fp r;
  BN_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = c[i];
  BN_TRK(set_trk(r, 0, 1, 0, 0.006, 1);)
  return r;
}
BN_INL Fp fp_from_limbs(const int32_t* c) {           // strict limbs loaded from memory, value < 2p
  Fp r;
  BN_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = c[i];
  BN_TRK(set_trk(r, 0, 1, 0, 0.012, 2);)
  return r;
}
BN_INL Fp fp_one() { return fp_const(bnc::ONE); }

// r = cond ? a : b   (limb-wise v_cndmask; branch free like the reference's conditional_select)
BN_INL Fp fp_select(bool cond, const Fp& a, const Fp& b) {
  Fp r;
  BN_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = cond ? a.l[i] : b.l[i];
  BN_TRK(set_trk(r, std::fmin(a.lo, b.lo), std::fmax(a.hi, b.hi), std::fmin(a.tlo, b.tlo), std::fmax(a.thi, b.thi), std::fmax(a.vb, b.vb));)
  return r;
}

// Parallel carry: limbs -> [-4, L+3] (top limb keeps the sign), value unchanged.  27 VALU ops.
BN_INL Fp fp_norm(const Fp& a) {
  Fp r;
  int32_t c[NL];
  BN_UNROLL for (int i = 0; i < NL - 1; ++i) c[i] = a.l[i] >> RB;
  r.l[0] = a.l[0] & MASK;
  BN_UNROLL for (int i = 1; i < NL - 1; ++i) r.l[i] = (a.l[i] & MASK) + c[i - 1];
  r.l[NL - 1] = a.l[NL - 1] + c[NL - 2];
  BN_TRK(++check_stats().norms;
         // top limb after normalisation is bounded by the value: |value| / 2^232 / L = vb * p/2^261
         double top = a.vb * P_OVER_R + 1e-7;
         if (top > 1) check_fail("fp_norm value too large for the top limb", a.vb);
         set_trk(r, -1e-8, 1.0 + 1e-8, -top, top, a.vb); check_actual(r, "fp_norm");)
  return r;
}

// r = K1*x1 + K2*x2 (+ K3*x3 + K4*x4), normalised like fp_norm.  |Ki| small compile-time integers.
// 64-bit per-limb sums (v_mad_i64_i32 with an inline constant), then one parallel carry.
// When the coefficients are large (REDUCE) the value is also brought back to (-eps*p, (1+eps)*p):
// q = floor(value / p) is estimated from the top limbs (value / 2^232 = top + O(sum|Ki|)), and -q*p is
// folded into the same per-limb sums.  Without it the xi-multiplications (x10) would outgrow the
// Montgomery contraction (p / 2^261 = 1/168) and the top limb would eventually overflow.
constexpr int lc_abs(int k) { return k < 0 ? -k : k; }
template <int K1, int K2, int K3, int K4, bool REDUCE = (lc_abs(K1) + lc_abs(K2) + lc_abs(K3) + lc_abs(K4) > 4)>
BN_INL Fp fp_lc4(const Fp& x1, const Fp& x2, const Fp& x3, const Fp& x4) {
  Fp r;
  int32_t lo[NL], c[NL];
  int32_t q = 0;
#if defined(__HIP_DEVICE_COMPILE__) && defined(BN_LC_MAD)
  // Small constants hidden in scalar registers: every term becomes one v_mad_i64_i32.  Left visible, the compiler
  // strength-reduces x * -2 or x * 1 into sign extension + 64-bit shift + subtract with borrow (5 instructions),
  // and zero-extends known-non-negative limbs with whatever register "is known to hold 0" (a spilled one, in the
  // Miller kernel: hundreds of scratch re-loads per iteration).
  int32_t k1 = K1, k2 = K2, k3 = K3, k4 = K4;
  asm("; lc k1" : "+s"(k1)); asm("; lc k2" : "+s"(k2)); asm("; lc k3" : "+s"(k3)); asm("; lc k4" : "+s"(k4));   // distinct texts: equal constants must not be merged, or x1*k + x2*k is refactored into a 64-bit (x1 + x2) * k
#else
  const int32_t k1 = K1, k2 = K2, k3 = K3, k4 = K4;
#endif
  if (REDUCE) {
    int64_t te = (int64_t)x1.l[NL - 1] * k1 + (int64_t)x2.l[NL - 1] * k2;
    if (K3 != 0) te += (int64_t)x3.l[NL - 1] * k3;
    if (K4 != 0) te += (int64_t)x4.l[NL - 1] * k4;
    q = (int32_t)((te * bnc::LC_QINV) >> 52);
  }
  BN_UNROLL for (int i = 0; i < NL; ++i) {
    const int32_t a1 = x1.l[i], a2 = x2.l[i], a3 = x3.l[i], a4 = x4.l[i];
#if defined(__HIP_DEVICE_COMPILE__) && defined(BN_LC_MAD)
    int64_t t = REDUCE ? -(int64_t)q * bnc::P[i] : 0;
    if (K4 != 0) t += (int64_t)a4 * k4;
    if (K3 != 0) t += (int64_t)a3 * k3;
    t += (int64_t)a2 * k2;
    t += (int64_t)a1 * k1;
#else
    int64_t t = (int64_t)a1 * K1 + (int64_t)a2 * K2;
    if (K3 != 0) t += (int64_t)a3 * K3;
    if (K4 != 0) t += (int64_t)a4 * K4;
    if (REDUCE) t -= (int64_t)q * bnc::P[i];
#endif
    lo[i] = (int32_t)((uint32_t)t & (uint32_t)MASK);
    c[i] = (int32_t)(t >> RB);
  }
  r.l[0] = lo[0];
  BN_UNROLL for (int i = 1; i < NL - 1; ++i) r.l[i] = lo[i] + c[i - 1];
  r.l[NL - 1] = lo[NL - 1] + c[NL - 2] + (c[NL - 1] << RB);     // top limb keeps everything above
  BN_TRK(++check_stats().lcs; check_stats().lc_terms += (K1 != 0) + (K2 != 0) + (K3 != 0) + (K4 != 0) + (REDUCE ? 1 : 0);
         double a1 = lc_abs(K1), a2 = lc_abs(K2), a3 = lc_abs(K3), a4 = lc_abs(K4);
         double m = a1 * mag(x1) + a2 * mag(x2) + a3 * mag(x3) + a4 * mag(x4);            // |t| / L before the q*p term
         double vb = a1 * x1.vb + a2 * x2.vb + a3 * x3.vb + a4 * x4.vb;
         double tin = a1 * tmag(x1) + a2 * tmag(x2) + a3 * tmag(x3) + a4 * tmag(x4);       // |te| / L
         if (tin >= 7.9) check_fail("fp_lc top-limb estimate: te * LC_QINV must fit 64 bits", tin);
         double mq = 0;
         if (REDUCE) { mq = vb + 1; m += mq; vb = 1.0 + (m + 4) / 3171406.0; }   // |q| <= vb + 1, |q p_i| <= mq L
         if (m >= 1.0e9) check_fail("fp_lc 64-bit limb sum", m);
         double top = vb * P_OVER_R + 1e-7;
         if (top > 1) check_fail("fp_lc value too large for the top limb", vb);
         set_trk(r, -(m + 1) / 536870912.0, 1.0 + (m + 1) / 536870912.0, -top, top, vb); check_actual(r, "fp_lc");)
  return r;
}
template <int K1, int K2, int K3>
BN_INL Fp fp_lc3(const Fp& x1, const Fp& x2, const Fp& x3) { return fp_lc4<K1, K2, K3, 0>(x1, x2, x3, x1); }
template <int K1, int K2>
BN_INL Fp fp_lc2(const Fp& x1, const Fp& x2) { return fp_lc3<K1, K2, 0>(x1, x2, x1); }

// r = sum_j k[j] * x[j] with RUN-TIME small integer coefficients (one per lane: the wave-per-tuple sums of wide.h, where the
// lane's output index decides which partial products enter directly, which times xi = 9 + u): one v_mad_i64_i32 per term and limb
// with the coefficient in a register, one parallel carry, value brought back to (-eps*p, (1+eps)*p) like fp_lc4's REDUCE form.
// Replaces select-then-add trees (a select costs as much as the MAD that makes it unnecessary).  sum |k[j]| mag(x[j]) < 2^30.
template <int N>
BN_INL Fp fp_lc_rt(const Fp* x, const int32_t* k) {
  Fp r;
  int32_t lo[NL], c[NL];
  int64_t te = 0;
  BN_UNROLL for (int j = 0; j < N; ++j) te += (int64_t)x[j].l[NL - 1] * k[j];
  const int32_t q = (int32_t)((te * bnc::LC_QINV) >> 52);
  BN_UNROLL for (int i = 0; i < NL; ++i) {
    int64_t t = -(int64_t)q * bnc::P[i];
    BN_UNROLL for (int j = 0; j < N; ++j) t += (int64_t)x[j].l[i] * k[j];
    lo[i] = (int32_t)((uint32_t)t & (uint32_t)MASK);
    c[i] = (int32_t)(t >> RB);
  }
  r.l[0] = lo[0];
  BN_UNROLL for (int i = 1; i < NL - 1; ++i) r.l[i] = lo[i] + c[i - 1];
  r.l[NL - 1] = lo[NL - 1] + c[NL - 2] + (c[NL - 1] << RB);
  BN_TRK(++check_stats().lcs; check_stats().lc_terms += N + 1;
         double m = 0, vb = 0, tin = 0;
         for (int j = 0; j < N; ++j) { double a = std::fabs((double)k[j]); m += a * mag(x[j]); vb += a * x[j].vb; tin += a * tmag(x[j]); }
         if (tin >= 7.9) check_fail("fp_lc_rt top-limb estimate: te * LC_QINV must fit 64 bits", tin);
         double mq = vb + 1; m += mq; vb = 1.0 + (m + 4) / 3171406.0;
         if (m >= 1.0e9) check_fail("fp_lc_rt 64-bit limb sum", m);
         double top = vb * P_OVER_R + 1e-7;
         set_trk(r, -(m + 1) / 536870912.0, 1.0 + (m + 1) / 536870912.0, -top, top, vb); check_actual(r, "fp_lc_rt");)
  return r;
}

// ------------------------------------------------------------------ Montgomery products
// r = (a*b + c*d) / R  (USE_CD = false: r = a*b/R).  Product scanning, one signed 64-bit column
// accumulator, reduction digits m_k interleaved; the compiler maps every `acc += (int64)x*y` to one
// v_mad_i64_i32.
template <bool USE_CD>
BN_INL Fp fp_mont_core(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {
  Fp r;
  int32_t m[NL];
  int64_t acc = 0;
  BN_UNROLL for (int k = 0; k < NL; ++k) {
    BN_UNROLL for (int i = 0; i <= k; ++i) {
      acc += (int64_t)a.l[i] * b.l[k - i];
      if (USE_CD) acc += (int64_t)c.l[i] * d.l[k - i];
    }
    BN_UNROLL for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * bnc::P[k - i];
    m[k] = (int32_t)(((uint32_t)acc * (uint32_t)bnc::PINV) & (uint32_t)MASK);
    acc += (int64_t)m[k] * bnc::P[0];
    acc >>= RB;
  }
  BN_UNROLL for (int k = NL; k < 2 * NL - 1; ++k) {
    BN_UNROLL for (int i = k - NL + 1; i < NL; ++i) {
      acc += (int64_t)a.l[i] * b.l[k - i];
      if (USE_CD) acc += (int64_t)c.l[i] * d.l[k - i];
    }
    BN_UNROLL for (int i = k - NL + 1; i < NL; ++i) acc += (int64_t)m[i] * bnc::P[k - i];
    r.l[k - NL] = (int32_t)((uint32_t)acc & (uint32_t)MASK);
    acc >>= RB;
  }
  r.l[NL - 1] = (int32_t)acc;
  return r;
}

#ifdef BN_CHECK
// Column budget of the product scanning loop, in units of L^2 (signed 64-bit holds 32 L^2):
//   column 7: 8 operand products (none with a top limb) + 8 reduction products
//   column 8: 7 operand products + a0*b8 + a8*b0 + 9 reduction products (m0*p8 is tiny)
// mm = sum over the (one or two) operand pairs of mag*mag, tt = sum of tmag(a)*mag(b) + mag(a)*tmag(b).
inline void trk_product(Fp& r, double mm, double tt, double vprod, const char* what) {
  if (8.0 * mm + 8.1 >= 31.9) check_fail(what, mm);
  if (7.0 * mm + tt + 8.1 >= 31.9) check_fail(what, 7.0 * mm + tt);
  if (vprod > 16384.0) check_fail("product of value bounds too large", vprod);
  double vout = vprod * P_OVER_R + 1.0;
  set_trk(r, 0, 1.0, -vprod * P_OVER_R * P_OVER_R - 1e-9, vout * P_OVER_R + 1e-9, vout);
  check_actual(r, what);
}
#endif

BN_INL Fp fp_mul(const Fp& a, const Fp& b) {
  Fp r = fp_mont_core<false>(a, b, a, b);
  BN_TRK(++check_stats().muls; double lb = mag(a) * mag(b); if (lb > check_stats().worst_mul) check_stats().worst_mul = lb;
         trk_product(r, lb, tmag(a) * mag(b) + mag(a) * tmag(b), a.vb * b.vb, "fp_mul precondition mag(a)*mag(b) < 2.97");)
  return r;
}
BN_INL Fp fp_dot2(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {     // (a*b + c*d)/R
  Fp r = fp_mont_core<true>(a, b, c, d);
  BN_TRK(++check_stats().dots; double lb = mag(a) * mag(b) + mag(c) * mag(d); if (lb > check_stats().worst_dot) check_stats().worst_dot = lb;
         trk_product(r, lb, tmag(a) * mag(b) + mag(a) * tmag(b) + tmag(c) * mag(d) + mag(c) * tmag(d), a.vb * b.vb + c.vb * d.vb,
                     "fp_dot2 precondition sum of mag products < 2.97");)
  return r;
}
// Squaring: the 45 distinct limb products, off-diagonal ones taken with a doubled limb.
BN_INL Fp fp_sqr(const Fp& a) {
  Fp r;
  int32_t m[NL], a2[NL];
  BN_UNROLL for (int i = 0; i < NL; ++i) a2[i] = a.l[i] * 2;
  int64_t acc = 0;
  BN_UNROLL for (int k = 0; k < 2 * NL - 1; ++k) {
    BN_UNROLL for (int i = 0; i < NL; ++i) {
      int j = k - i;
      if (j < 0 || j >= NL || i > j) continue;
      if (i == j) acc += (int64_t)a.l[i] * a.l[i]; else acc += (int64_t)a2[i] * a.l[j];
    }
    if (k < NL) {
      BN_UNROLL for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * bnc::P[k - i];
      m[k] = (int32_t)(((uint32_t)acc * (uint32_t)bnc::PINV) & (uint32_t)MASK);
      acc += (int64_t)m[k] * bnc::P[0];
    } else {
      BN_UNROLL for (int i = k - NL + 1; i < NL; ++i) acc += (int64_t)m[i] * bnc::P[k - i];
      r.l[k - NL] = (int32_t)((uint32_t)acc & (uint32_t)MASK);
    }
    acc >>= RB;
  }
  r.l[NL - 1] = (int32_t)acc;
  BN_TRK(++check_stats().sqrs; double lb = mag(a) * mag(a); if (lb > check_stats().worst_mul) check_stats().worst_mul = lb;
         if (mag(a) >= 1.99) check_fail("fp_sqr doubled limb overflow", mag(a));
         trk_product(r, lb, 2 * tmag(a) * mag(a), a.vb * a.vb, "fp_sqr precondition mag(a)^2 < 2.97");)
  return r;
}

// ------------------------------------------------------------------ canonical form, comparisons, I/O
// Input: limbs 0..7 in [0, L), value in (-p, 2p) (any fp_mul output).  Output: the canonical
// representative in [0, p) with strict limbs.
BN_INL Fp fp_canon_strict(const Fp& a) {
  Fp u, t;
  int32_t cu = 0, ct = 0;
  BN_UNROLL for (int i = 0; i < NL - 1; ++i) {
    int32_t du = a.l[i] + bnc::P[i] + cu; u.l[i] = du & MASK; cu = du >> RB;
    int32_t dt = a.l[i] - bnc::P[i] + ct; t.l[i] = dt & MASK; ct = dt >> RB;
  }
  u.l[NL - 1] = a.l[NL - 1] + bnc::P[NL - 1] + cu;
  t.l[NL - 1] = a.l[NL - 1] - bnc::P[NL - 1] + ct;
  bool neg = a.l[NL - 1] < 0;
  bool ge = t.l[NL - 1] >= 0;
  Fp r;
  BN_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = neg ? u.l[i] : (ge ? t.l[i] : a.l[i]);
  BN_TRK(set_trk(r, 0, 1, 0, 0.006, 1);)
  return r;
}
// Any lazy value -> canonical Montgomery representative (one multiply by R^2/R = R, i.e. by "one").
BN_INL Fp fp_canon(const Fp& a) {
  Fp one_plain_r2;   // plain limbs of R^2 mod p: (a * R^2) / R = a * R ... see below
  (void)one_plain_r2;
  // a is in Montgomery form (a = x*R).  fp_mul(a, ONE) = a * (1*R) / R = a: same residue, reduced.
  return fp_canon_strict(fp_mul(a, fp_one()));
}
BN_INL bool fp_is_zero(const Fp& a) {
  Fp c = fp_canon(a);
  int32_t o = 0;
  BN_UNROLL for (int i = 0; i < NL; ++i) o |= c.l[i];
  return o == 0;
}
BN_INL bool fp_eq(const Fp& a, const Fp& b) { return fp_is_zero(fp_sub(a, b)); }

// plain (non-Montgomery) canonical value of a Montgomery-form element: a * 1 / R
BN_INL Fp fp_from_mont(const Fp& a) {
  Fp one;
  BN_UNROLL for (int i = 0; i < NL; ++i) one.l[i] = i == 0 ? 1 : 0;
  BN_TRK(set_trk(one, 0, 1e-8, 0, 0, 1e-60);)
  return fp_canon_strict(fp_mul(a, one));
}
BN_INL Fp fp_to_mont(const Fp& plain) {              // plain strict limbs (value < 2^261) -> Montgomery
  Fp r2 = fp_const(bnc::R2);
  return fp_mul(plain, r2);
}
BN_INL int fp_sgn0(const Fp& a) { return fp_from_mont(a).l[0] & 1; }        // parity, fp.rs:164-168

// 32 big-endian bytes <-> limbs.  w[0] = least significant 32-bit word.
BN_INL void words_to_limbs(int32_t* l, const uint32_t* w) {
  BN_UNROLL for (int i = 0; i < NL; ++i) {
    int bit = RB * i, j = bit >> 5, s = bit & 31;
    uint32_t v = w[j] >> s;
    if (s > 32 - RB && j + 1 < 8) v |= w[j + 1] << (32 - s);
    l[i] = (int32_t)(v & (uint32_t)MASK);
  }
}
BN_INL void limbs_to_words(uint32_t* w, const int32_t* l) {   // strict non-negative limbs
  BN_UNROLL for (int j = 0; j < 8; ++j) {
    uint32_t v = 0;
    BN_UNROLL for (int i = 0; i < NL; ++i) {
      int bit = RB * i - 32 * j;                 // position of limb i's bit 0 inside word j
      if (bit > -RB && bit < 32) v |= bit >= 0 ? ((uint32_t)l[i] << bit) : ((uint32_t)l[i] >> (-bit));
    }
    w[j] = v;
  }
}
BN_INL uint32_t load_be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
BN_INL void store_be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

// Decode 32 big-endian bytes; ok = value < p (PrimeField::from_repr, fp.rs:249-254).  Result Montgomery.
BN_INL Fp fp_from_be(const uint8_t* b, bool& ok) {
  uint32_t w[8];
  BN_UNROLL for (int j = 0; j < 8; ++j) w[j] = load_be32(b + 4 * (7 - j));
  Fp x;
  words_to_limbs(x.l, w);
  BN_TRK(set_trk(x, 0, 1, 0, 0.04, 6);)
  // x < p  <=>  x - p < 0
  int32_t c = 0;
  BN_UNROLL for (int i = 0; i < NL; ++i) { int32_t d = x.l[i] - bnc::P[i] + c; c = d >> RB; }
  ok = c < 0;
  return fp_to_mont(x);
}
BN_INL void fp_to_be(uint8_t* b, const Fp& a) {
  Fp c = fp_from_mont(a);
  uint32_t w[8];
  limbs_to_words(w, c.l);
  BN_UNROLL for (int j = 0; j < 8; ++j) store_be32(b + 4 * (7 - j), w[j]);
}
// 48 big-endian bytes mod p (FromOkm, fp.rs:115-122): low 261 bits * R^2/R + high 123 bits * R^3/R
BN_INL Fp fp_from_okm(const uint8_t* okm) {
  uint32_t w[12];
  BN_UNROLL for (int j = 0; j < 12; ++j) w[j] = load_be32(okm + 4 * (11 - j));
  Fp lo, hi;
  BN_UNROLL for (int i = 0; i < NL; ++i) {
    int bit = RB * i, j = bit >> 5, s = bit & 31;
    uint32_t v = w[j] >> s;
    if (s > 32 - RB) v |= w[j + 1] << (32 - s);
    lo.l[i] = (int32_t)(v & (uint32_t)MASK);
  }
  BN_UNROLL for (int i = 0; i < NL; ++i) {
    int bit = RB * (i + NL), j = bit >> 5, s = bit & 31;
    uint32_t v = 0;
    if (j < 12) { v = w[j] >> s; if (s > 32 - RB && j + 1 < 12) v |= w[j + 1] << (32 - s); }
    hi.l[i] = (int32_t)(v & (uint32_t)MASK);
  }
  BN_TRK(set_trk(lo, 0, 1, 0, 1, 170); set_trk(hi, 0, 1, 0, 0, 1e-30);)
  return fp_dot2(lo, fp_const(bnc::R2), hi, fp_const(bnc::R3));
}

// ------------------------------------------------------------------ exponentiation helpers
// a^e for a fixed public 256-bit exponent (uniform control flow: every lane runs the same bits).
// 4-bit fixed windows: 252 squarings + 64 multiplies + 14 table products.
struct Exp256 { uint64_t w[4]; };
#define BN_EXP(name) (::bn::Exp256{{bnc::name[0], bnc::name[1], bnc::name[2], bnc::name[3]}})
BN_FUNC Fp fp_pow(const Fp& a, Exp256 e) {
  Fp tab[16];
  tab[0] = fp_one();
  tab[1] = fp_norm(a);
  for (int i = 2; i < 16; ++i) tab[i] = fp_mul(tab[i - 1], tab[1]);
  const uint64_t e0 = e.w[0], e1 = e.w[1], e2 = e.w[2], e3 = e.w[3];   // wave-uniform: stay in scalar registers
  Fp r = tab[(int)(e3 >> 60)];
  for (int w = 62; w >= 0; --w) {
    uint64_t word = w >= 48 ? e3 : w >= 32 ? e2 : w >= 16 ? e1 : e0;
    int d = (int)((word >> ((w & 15) * 4)) & 15);
    // The table lives in scratch.  Its entry is fetched before the four squarings, and the barrier keeps the
    // loads there, so their latency is hidden; multiplying unconditionally (tab[0] = 1) keeps the use in this block.
    Fp t = tab[d];
    BN_SCHED_BARRIER;
    r = fp_sqr(r); r = fp_sqr(r); r = fp_sqr(r); r = fp_sqr(r);
    r = fp_mul(r, t);
  }
  return r;
}
BN_FUNC Fp fp_inv_pow(const Fp& a) { return fp_pow(a, BN_EXP(EXP_PM2)); }             // Fermat: the reference's Fp::invert (fp.rs:207-210); inv0(0) = 0 (E15)

// Inversion by the Bernstein-Yang divstep recurrence ("safegcd", 2019/266) in the half-delta form, 29 steps per batch so that a
// batch shifts the operands by exactly one limb of this file's radix.  Same result as Fermat's a^(p-2) (the inverse is unique;
// 0 -> 0), uniform control flow (every lane runs the same instructions whatever its data: nothing to leak on the signing side,
// nothing to diverge on), ~15 k instructions against ~65 k for the 254-bit power.
//   f = p, g = x (canonical, as an integer), d = 0, e = 1, invariant f = d x, g = e x (mod p); one batch runs 29 divsteps on the
//   low words of (f, g) and yields a 2x2 integer matrix t with (f, g) <- t (f, g) / 2^29 exactly and (d, e) <- t (d, e) / 2^29
//   mod p (the division made exact by adding a multiple of p, the Montgomery way).  590 divsteps bring g to 0 for any 256-bit
//   input (the half-delta bound); 21 batches = 609.  Then f = +-1 and x^-1 = +-d.
// |u| + |v| <= 2^29 for a row (u, v) of t, so |d|, |e| grow by at most p per batch: below 22 p, inside the lazy range.
BN_INL void sg_divsteps29(int32_t& zeta, uint32_t f, uint32_t g, int32_t& u, int32_t& v, int32_t& q, int32_t& r) {
  u = 1; v = 0; q = 0; r = 1;
  BN_UNROLL for (int i = 0; i < RB; ++i) {
    int32_t c1 = zeta >> 31;                               // all ones: delta > 0
    const int32_t c2 = -(int32_t)(g & 1u);                 // all ones: g odd
    const uint32_t x = (f ^ (uint32_t)c1) - (uint32_t)c1;  // +-f
    const int32_t y = (u ^ c1) - c1, z = (v ^ c1) - c1;
    g += x & (uint32_t)c2; q += y & c2; r += z & c2;       // g odd: g <- g +- f
    c1 &= c2;                                              // swap: delta > 0 and g odd
    zeta = (zeta ^ c1) - 1;
    f += g & (uint32_t)c1; u += q & c1; v += r & c1;       // swap: f <- old g
    g >>= 1; u *= 2; v *= 2;
  }
}
// (a, b) <- (u a + v b, q a + r b) / 2^29; MODP: mod p (a multiple of p makes the division exact), else exact as it stands
template <bool MODP>
BN_INL void sg_update(int32_t* a, int32_t* b, int32_t u, int32_t v, int32_t q, int32_t r) {
  int64_t ca = (int64_t)u * a[0] + (int64_t)v * b[0], cb = (int64_t)q * a[0] + (int64_t)r * b[0];
  int32_t ka = 0, kb = 0;
  if (MODP) {
    ka = (int32_t)(((uint32_t)ca * (uint32_t)bnc::PINV) & (uint32_t)MASK); kb = (int32_t)(((uint32_t)cb * (uint32_t)bnc::PINV) & (uint32_t)MASK);
    ca += (int64_t)ka * bnc::P[0]; cb += (int64_t)kb * bnc::P[0];
  }
  ca >>= RB; cb >>= RB;
  BN_UNROLL for (int i = 1; i < NL; ++i) {
    const int32_t ai = a[i], bi = b[i];
    ca += (int64_t)u * ai + (int64_t)v * bi; cb += (int64_t)q * ai + (int64_t)r * bi;
    if (MODP) { ca += (int64_t)ka * bnc::P[i]; cb += (int64_t)kb * bnc::P[i]; }
    a[i - 1] = (int32_t)((uint32_t)ca & (uint32_t)MASK); b[i - 1] = (int32_t)((uint32_t)cb & (uint32_t)MASK);
    ca >>= RB; cb >>= RB;
  }
  a[NL - 1] = (int32_t)ca; b[NL - 1] = (int32_t)cb;
}
constexpr int SG_BATCHES = 21;
BN_FUNC Fp fp_inv(const Fp& a) {
  const Fp x = fp_from_mont(a);                       // the plain value as a canonical integer in [0, p), strict limbs
  int32_t f[NL], g[NL], d[NL], e[NL];
  BN_UNROLL for (int i = 0; i < NL; ++i) { f[i] = bnc::P[i]; g[i] = x.l[i]; d[i] = 0; e[i] = i == 0 ? 1 : 0; }
  int32_t zeta = -1;
  for (int b = 0; b < SG_BATCHES; ++b) {
    int32_t u, v, q, r;
    sg_divsteps29(zeta, (uint32_t)f[0] | ((uint32_t)f[1] << RB), (uint32_t)g[0] | ((uint32_t)g[1] << RB), u, v, q, r);
    sg_update<false>(f, g, u, v, q, r);
    sg_update<true>(d, e, u, v, q, r);
  }
  int32_t gnz = 0;
  BN_UNROLL for (int i = 0; i < NL; ++i) gnz |= g[i];
  // g = 0 is guaranteed by the 590-step bound; should a lane ever miss it the whole wave takes Fermat's route instead
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_amdgcn_ballot_w64(gnz != 0) != 0) return fp_inv_pow(a);
#else
  if (gnz != 0) return fp_inv_pow(a);
#endif
  const bool neg = f[NL - 1] < 0;                     // f = -1: limbs all ones below a negative top limb
  Fp di;
  BN_UNROLL for (int i = 0; i < NL; ++i) di.l[i] = neg ? -d[i] : d[i];
  BN_TRK(set_trk(di, -1, 1, -0.14, 0.14, 22);)
  return fp_mul(di, fp_const(bnc::R2));               // di = x^-1 as a plain integer: times R^2 / R = its Montgomery form
}
// y = a^((p+1)/4); is_sq = (y^2 == a).  One exponentiation gives Euler's criterion (fp.rs:428-431)
// and the square root (sqrt_ratio with v = 1, fp.rs:212-243) together.
BN_FUNC Fp fp_sqrt_cand(const Fp& a, bool& is_sq) {
  Fp y = fp_pow(a, BN_EXP(EXP_PP1_4));
  is_sq = fp_eq(fp_sqr(y), a);
  return y;
}

// is_square (Euler's criterion in the reference, fp.rs:428-431: a^((p-1)/2) is 0 or 1) without an exponentiation:
// the Jacobi symbol (a / p) by the binary algorithm on 8 x 32-bit words, branch-free per step.  One step: strip ALL trailing
// zeros of a (up to 31 at once: count-trailing-zeros + eight funnel shifts; (2 / n) = -1 iff n = 3, 5 mod 8 enters once per odd
// count), then, a being odd, swap (a, n) if a < n (reciprocity: both = 3 mod 4 flips the sign) and subtract: a <- |a - n| with
// n <- min(a, n).  bitlen(a) + bitlen(n) shrinks by the zeros stripped, about 2.6 bits per step: ~180 steps for 254-bit inputs
// (194 for the slowest of a wave's 64 lanes, measured over 20 000 random inputs; the one-zero-per-step form needed ~400), ~65
// plain 32-bit instructions each, against ~70 k MAD-heavy instructions for a 254-bit power.  A wave leaves the loop when all
// its lanes have reached a = 0.  a is a Montgomery representative: (a R / p) = (a / p) because R = 2^261 and (2 / p) = +1
// for p = 7 mod 8.
BN_INL uint32_t bn_funnel_shr(uint32_t hi, uint32_t lo, uint32_t k) {       // low word of {hi, lo} >> k, 0 <= k <= 31
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, k);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> k);
#endif
}
BN_FUNC bool fp_is_square(const Fp& x) {
  Fp c = fp_canon(x);
  uint32_t a[8], n[8];
  limbs_to_words(a, c.l);
  limbs_to_words(n, bnc::P);
  uint32_t t = 0;                                   // parity of the accumulated sign
  for (int it = 0; it < 520; ++it) {                // every step strips at least one bit of a unless a is 0: 510 is the worst case
    // strip the trailing zeros of a (a zero low word: 31 now, the rest next time round; a == 0 stays 0)
    const uint32_t k = (uint32_t)__builtin_ctz(a[0] | 0x80000000u);
    BN_UNROLL for (int j = 0; j < 7; ++j) a[j] = bn_funnel_shr(a[j + 1], a[j], k);
    a[7] >>= k;
    t ^= k & ((n[0] >> 1) ^ (n[0] >> 2));           // (2 / n)^k
    const uint32_t odd = a[0] & 1u;                  // 0 only while a low word is being skipped, or at a == 0
    uint32_t d[8], borrow = 0;                      // d = a - n, borrow = a < n
    BN_UNROLL for (int j = 0; j < 8; ++j) {
      const uint64_t v = (uint64_t)a[j] - n[j] - borrow;
      d[j] = (uint32_t)v; borrow = (uint32_t)(v >> 63);
    }
    const uint32_t swap = odd & borrow;
    t ^= swap & ((a[0] & n[0]) >> 1);               // reciprocity: both = 3 mod 4
    const uint32_t sm = 0u - swap, om = 0u - odd;
    uint32_t carry = swap, nz = 0;                  // a <- odd ? |d| : a  (|d| = (d ^ sm) - sm),  n <- swap ? a : n
    BN_UNROLL for (int j = 0; j < 8; ++j) {
      const uint64_t v = (uint64_t)(d[j] ^ sm) + carry;
      carry = (uint32_t)(v >> 32);
      n[j] = swap ? a[j] : n[j];
      a[j] = (a[j] & ~om) | ((uint32_t)v & om);
      nz |= a[j];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_ballot_w64(nz != 0) == 0) break;         // every lane of the wave has reached a == 0
#else
    if (nz == 0) break;
#endif
  }
  // a == 0 now and n = gcd(x, p): p itself for x == 0 (counted as a square, like the reference), else 1
  return (t & 1u) == 0;
}

}  // namespace bn
