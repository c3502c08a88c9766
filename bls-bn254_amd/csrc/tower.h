// tower.h -- Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v), xi = 9+u,
// on the lazy radix-2^29 base field of fp29.h.
//
// Reference operators replaced (values identical, formulas chosen for gfx950):
//   Fp2  mul/square/invert/conjugate      fp2.rs:377-437, :161-166
//   Fp6  multiply/mul_by_01/mul_by_1/invert  fp6.rs:123-152, :225-287   (xi fixed to 9+u, E1)
//   Fp12 multiply/square/conjugate/invert/frobenius, sparse multiply  fp12.rs:120-219 (E3, E7)
//   fp4_square / cyclotomic_square        pairings.rs:52-115
//
// Design notes:
//  * Fp2 multiplication is "schoolbook-lazy": c0 = a0*b0 - a1*b1 and c1 = a0*b1 + a1*b0 are each ONE
//    double product accumulated in the 64-bit columns and reduced once (fp_dot2): 4*81 MADs + 2
//    reductions = 572 VALU ops, cheaper than Karatsuba (3 full multiplies + 5 additions = 660)
//    because on gfx950 a MAD costs about the same as an add.
//  * Karatsuba at the Fp6/Fp12 level is SUBTRACTIVE ((a0-a1)(b1-b0) + a0b0 + a1b1): differences of
//    normalised non-negative limbs stay within one limb width, so the inner products need no
//    normalisation (see the interval discipline in fp29.h).
//  * Every function documents the interval it needs; tests/hostsim proves them with -DBN_CHECK.
#pragma once
#include "fp29.h"

namespace bn {

struct Fp2 { Fp c0, c1; };
struct Fp6 { Fp2 c0, c1, c2; };
struct Fp12 { Fp6 c0, c1; };

// ------------------------------------------------------------------ Fp2
BN_INL Fp2 fp2_add(const Fp2& a, const Fp2& b) { return {fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
BN_INL Fp2 fp2_sub(const Fp2& a, const Fp2& b) { return {fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
BN_INL Fp2 fp2_neg(const Fp2& a) { return {fp_neg(a.c0), fp_neg(a.c1)}; }
BN_INL Fp2 fp2_dbl(const Fp2& a) { return {fp_dbl(a.c0), fp_dbl(a.c1)}; }
BN_INL Fp2 fp2_conj(const Fp2& a) { return {a.c0, fp_neg(a.c1)}; }
BN_INL Fp2 fp2_norm(const Fp2& a) { return {fp_norm(a.c0), fp_norm(a.c1)}; }
BN_INL Fp2 fp2_zero() { return {fp_zero(), fp_zero()}; }
BN_INL Fp2 fp2_one() { return {fp_one(), fp_zero()}; }
BN_INL Fp2 fp2_const(const int32_t (&c)[2 * NL]) {
  Fp2 r;
  BN_UNROLL for (int i = 0; i < NL; ++i) { r.c0.l[i] = c[i]; r.c1.l[i] = c[NL + i]; }
  BN_TRK(set_trk(r.c0, 0, 1, 0, 0.006, 1); set_trk(r.c1, 0, 1, 0, 0.006, 1);)
  return r;
}
BN_INL Fp2 fp2_from_limbs(const int32_t* c) { return {fp_from_limbs(c), fp_from_limbs(c + NL)}; }
BN_INL Fp2 fp2_select(bool cond, const Fp2& a, const Fp2& b) { return {fp_select(cond, a.c0, b.c0), fp_select(cond, a.c1, b.c1)}; }
// needs mag(a)*mag(b) <= 1.27 per component pair
BN_INL Fp2 fp2_mul(const Fp2& a, const Fp2& b) {
  BN_CTX;
  Fp na1 = fp_neg(a.c1);
  return {fp_dot2(a.c0, b.c0, na1, b.c1), fp_dot2(a.c0, b.c1, a.c1, b.c0)};
}
// complex squaring; needs a in [0, 1.2] limb-wise (non-negative), i.e. a multiply output or fp_norm
BN_INL Fp2 fp2_sqr(const Fp2& a) {
  BN_CTX;
  return {fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1)), fp_mul(fp_dbl(a.c0), a.c1)};
}
BN_INL Fp2 fp2_mul_fp(const Fp2& a, const Fp& s) { return {fp_mul(a.c0, s), fp_mul(a.c1, s)}; }
// (9+u)(a0 + a1 u) = (9 a0 - a1) + (a0 + 9 a1) u ; normalised output
BN_INL Fp2 fp2_mul_xi(const Fp2& a) { return {fp_lc2<9, -1>(a.c0, a.c1), fp_lc2<1, 9>(a.c0, a.c1)}; }
// x + xi*y, normalised
BN_INL Fp2 fp2_add_mul_xi(const Fp2& x, const Fp2& y) {
  return {fp_lc3<1, 9, -1>(x.c0, y.c0, y.c1), fp_lc3<1, 1, 9>(x.c1, y.c0, y.c1)};
}
// x - xi*y, normalised
BN_INL Fp2 fp2_sub_mul_xi(const Fp2& x, const Fp2& y) {
  return {fp_lc3<1, -9, 1>(x.c0, y.c0, y.c1), fp_lc3<1, -1, -9>(x.c1, y.c0, y.c1)};
}
BN_FUNC Fp2 fp2_inv(const Fp2& a) {                                  // fp2.rs:161-166
  Fp2 n = fp2_norm(a);
  Fp t = fp_inv(fp_dot2(n.c0, n.c0, n.c1, n.c1));
  return {fp_mul(n.c0, t), fp_mul(fp_neg(n.c1), t)};
}
BN_INL bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) & fp_is_zero(a.c1); }
BN_INL bool fp2_eq(const Fp2& a, const Fp2& b) { return fp2_is_zero(fp2_sub(a, b)); }
BN_INL int fp2_sgn0(const Fp2& a) {                                       // fp2.rs:95-99
  Fp c0 = fp_from_mont(a.c0), c1 = fp_from_mont(a.c1);
  int32_t o = 0;
  BN_UNROLL for (int i = 0; i < NL; ++i) o |= c0.l[i];
  return (c0.l[0] & 1) | ((o == 0) & (c1.l[0] & 1));
}

// ------------------------------------------------------------------ Fp6
BN_INL Fp6 fp6_add(const Fp6& a, const Fp6& b) { return {fp2_add(a.c0, b.c0), fp2_add(a.c1, b.c1), fp2_add(a.c2, b.c2)}; }
BN_INL Fp6 fp6_sub(const Fp6& a, const Fp6& b) { return {fp2_sub(a.c0, b.c0), fp2_sub(a.c1, b.c1), fp2_sub(a.c2, b.c2)}; }
BN_INL Fp6 fp6_neg(const Fp6& a) { return {fp2_neg(a.c0), fp2_neg(a.c1), fp2_neg(a.c2)}; }
BN_INL Fp6 fp6_norm(const Fp6& a) { return {fp2_norm(a.c0), fp2_norm(a.c1), fp2_norm(a.c2)}; }
BN_INL Fp6 fp6_zero() { return {fp2_zero(), fp2_zero(), fp2_zero()}; }
BN_INL Fp6 fp6_one() { return {fp2_one(), fp2_zero(), fp2_zero()}; }
// inputs normalised ([~0, ~1] limbs); output normalised.  6 Fp2 products (fp6.rs:225-242 value).
BN_FUNC Fp6 fp6_mul(const Fp6& a, const Fp6& b) {
  BN_CTX;
  Fp2 v0 = fp2_mul(a.c0, b.c0), v1 = fp2_mul(a.c1, b.c1), v2 = fp2_mul(a.c2, b.c2);
  Fp2 w0 = fp2_mul(fp2_sub(a.c1, a.c2), fp2_sub(b.c1, b.c2));
  Fp2 w1 = fp2_mul(fp2_sub(a.c0, a.c1), fp2_sub(b.c0, b.c1));
  Fp2 w2 = fp2_mul(fp2_sub(a.c0, a.c2), fp2_sub(b.c0, b.c2));
  Fp2 t0 = fp2_sub(fp2_add(v1, v2), w0);        // a1 b2 + a2 b1
  Fp2 t1 = fp2_sub(fp2_add(v0, v1), w1);        // a0 b1 + a1 b0
  Fp2 t2 = fp2_sub(fp2_add(v0, v2), w2);        // a0 b2 + a2 b0
  return {fp2_add_mul_xi(v0, t0), fp2_add_mul_xi(t1, v2), fp2_norm(fp2_add(t2, v1))};
}
BN_FUNC Fp6 fp6_sqr(const Fp6& a) { return fp6_mul(a, a); }
BN_INL Fp6 fp6_mul_v(const Fp6& a) { return {fp2_mul_xi(a.c2), a.c0, a.c1}; }      // fp6.rs:146-152
BN_INL Fp6 fp6_mul_fp2(const Fp6& a, const Fp2& s) { return {fp2_mul(a.c0, s), fp2_mul(a.c1, s), fp2_mul(a.c2, s)}; }
// a * (c0 + c1 v), fp6.rs:131-144 value; inputs and output normalised
BN_FUNC Fp6 fp6_mul_by_01(const Fp6& a, const Fp2& c0, const Fp2& c1) {
  BN_CTX;
  Fp2 aa = fp2_mul(a.c0, c0), bb = fp2_mul(a.c1, c1);
  Fp2 cc = fp2_mul(a.c2, c1), dd = fp2_mul(a.c2, c0);
  Fp2 w = fp2_mul(fp2_sub(a.c0, a.c1), fp2_sub(c0, c1));     // a0c0 + a1c1 - a0c1 - a1c0
  Fp2 t2 = fp2_sub(fp2_add(aa, bb), w);                       // a0 c1 + a1 c0
  return {fp2_add_mul_xi(aa, cc), fp2_norm(t2), fp2_norm(fp2_add(dd, bb))};
}
BN_FUNC Fp6 fp6_inv(const Fp6& a) {                       // fp6.rs:261-287 (denominator corrected)
  Fp2 s0 = fp2_sqr(a.c0), s1 = fp2_sqr(a.c1), s2 = fp2_sqr(a.c2);
  Fp2 m01 = fp2_mul(a.c0, a.c1), m02 = fp2_mul(a.c0, a.c2), m12 = fp2_mul(a.c1, a.c2);
  Fp2 c0 = fp2_sub_mul_xi(s0, m12);
  Fp2 c1 = fp2_norm(fp2_sub(fp2_mul_xi(s2), m01));
  Fp2 c2 = fp2_norm(fp2_sub(s1, m02));
  Fp2 u = fp2_add(fp2_mul(a.c2, c1), fp2_mul(a.c1, c2));
  Fp2 t = fp2_inv(fp2_add_mul_xi(fp2_mul(a.c0, c0), u));
  return {fp2_mul(c0, t), fp2_mul(c1, t), fp2_mul(c2, t)};
}

// ------------------------------------------------------------------ Fp12
BN_INL Fp12 fp12_one() { return {fp6_one(), fp6_zero()}; }
BN_INL Fp12 fp12_conj(const Fp12& a) { return {a.c0, fp6_norm(fp6_neg(a.c1))}; }            // fp12.rs:131-137
// x + v*y for Fp6 x, y ; normalised
BN_INL Fp6 fp6_add_mul_v(const Fp6& x, const Fp6& y) {
  BN_CTX;
  return {fp2_add_mul_xi(x.c0, y.c2), fp2_norm(fp2_add(x.c1, y.c0)), fp2_norm(fp2_add(x.c2, y.c1))};
}
BN_FUNC Fp12 fp12_mul(const Fp12& a, const Fp12& b) {     // fp12.rs:203-210 value
  Fp6 v0 = fp6_mul(a.c0, b.c0), v1 = fp6_mul(a.c1, b.c1);
  Fp6 w = fp6_mul(fp6_norm(fp6_sub(a.c0, a.c1)), fp6_norm(fp6_sub(b.c1, b.c0)));   // a0b1 + a1b0 - v0 - v1
  return {fp6_add_mul_v(v0, v1), fp6_norm(fp6_add(fp6_add(w, v0), v1))};
}
// ---- operands parked in memory (LDS or an HBM workspace), limb-major.  A Ws names one lane's column of a
// workspace: limb k of it is base[k * stride + lane].  `base` and `stride` are wave-uniform (scalar registers) and
// the lane enters as one 32-bit byte offset, so a load is `global_load_dword v, v_lane4, s[row]` (or a
// ds_read_b32 with an immediate) -- no 64-bit per-lane pointer that would itself be spilled and re-loaded in
// front of every access.  With stride >= the wave's lane count, the 64 lanes hit 64 consecutive dwords.
// BN_MEM_FENCE stops the compiler from keeping a loaded operand alive in registers across phases (that is the
// whole point of parking it); BN_OPAQUE makes the lane offset unknown to the optimiser at that point, so loads
// through it are neither hoisted out of a loop nor merged with earlier ones.
// `buf` selects buffer addressing for HBM workspaces on the device: the row base goes into a scalar buffer
// descriptor and k * stride into the scalar offset, so no per-limb 64-bit address pair is ever formed in (or
// spilled from) vector registers.  k * stride * 4 must stay below 4 GB (108 limbs x 4 M lanes: 1.8 GB).
struct Ws { int32_t* base; size_t stride; uint32_t lane4; bool buf; };
BN_INL Ws ws_at(const Ws& w, size_t limbs) { return {w.base + limbs * w.stride, w.stride, w.lane4, w.buf}; }
BN_INL int32_t* ws_addr(const Ws& w, int k) { return (int32_t*)((char*)(w.base + (size_t)k * w.stride) + w.lane4); }
// The same reference with its base marked wave-uniform.  Buffer addressing keeps the base in scalar registers; a base that
// reaches the load through the arguments of a REAL function (wide.h's noinline primitives) is not known to be uniform, and every
// load would be wrapped in a serialising loop over the distinct bases of the wave.  Only for bases that ARE the same in every lane.
BN_INL Ws ws_uniform(const Ws& w) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint64_t b = (uint64_t)w.base;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return {(int32_t*)(((uint64_t)hi << 32) | lo), w.stride, w.lane4, w.buf};
#else
  return w;
#endif
}
BN_INL int32_t ws_load(const Ws& w, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (w.buf)
    return __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc((void*)w.base, 0, -1, 0x00020000), (int)w.lane4,
                                                (int)(uint32_t)((size_t)k * w.stride * 4), 0);
#endif
  return *ws_addr(w, k);
}
BN_INL void ws_store(const Ws& w, int k, int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (w.buf) {
    __builtin_amdgcn_raw_buffer_store_b32(v, __builtin_amdgcn_make_buffer_rsrc((void*)w.base, 0, -1, 0x00020000), (int)w.lane4,
                                          (int)(uint32_t)((size_t)k * w.stride * 4), 0);
    return;
  }
#endif
  *ws_addr(w, k) = v;
}
#if defined(__HIP_DEVICE_COMPILE__)
#define BN_MEM_FENCE asm volatile("" ::: "memory")
#define BN_OPAQUE(w) asm volatile("" : "+v"((w).lane4))
// ... and its (scalar) stride as well: the k * stride * 4 scalar offsets of a strided buffer reference are loop invariants, and a loop
// body that touches many named values gets all of them hoisted -- more than there are scalar registers; the compiler then parks
// them in VECTOR registers and wraps every load and store that uses one in a serialising read-first-lane loop (k_fe_tri_hard:
// 1377 such loops).  Re-deriving them after this point costs one scalar multiply each.
#define BN_OPAQUE_S(w) asm volatile("" : "+v"((w).lane4), "+s"((w).stride))
#else
#define BN_MEM_FENCE do { } while (0)
#define BN_OPAQUE(w) do { } while (0)
#define BN_OPAQUE_S(w) do { } while (0)
#endif
#ifdef BN_CHECK
}  // namespace bn
#include <map>
#include <mutex>
namespace bn {
// check mode: the proven bounds of a parked value travel with its address (one table for all threads: the lanes of a quad,
// run as threads by tests/hostsim, read each other's parked values)
struct ParkTrk { double lo, hi, tlo, thi, vb; };
inline std::map<const int32_t*, ParkTrk>& park_trk() { static std::map<const int32_t*, ParkTrk> m; return m; }
inline std::mutex& park_mutex() { static std::mutex m; return m; }
#endif
BN_INL Fp fp_load_mem(const Ws& w) {
  Fp r;
  BN_UNROLL for (int k = 0; k < NL; ++k) r.l[k] = ws_load(w, k);
  BN_TRK(ParkTrk pt; { std::lock_guard<std::mutex> g(park_mutex()); auto it = park_trk().find(ws_addr(w, 0));
           if (it == park_trk().end()) check_fail("fp_load_mem of an address never stored", 0); pt = it->second; }
         set_trk(r, pt.lo, pt.hi, pt.tlo, pt.thi, pt.vb);)
  return r;
}
BN_INL void fp_store_mem(const Ws& w, const Fp& a) {
  BN_TRK(if (a.lo < -4e-6 || a.hi > 1.0 + 4e-6) check_fail("fp_store_mem needs a normalised value", mag(a));      // a carry pass leaves limbs within a few thousand units of [0, 2^29]; the proven interval travels with the value (park_trk)
         { std::lock_guard<std::mutex> g(park_mutex()); park_trk()[ws_addr(w, 0)] = ParkTrk{a.lo, a.hi, a.tlo, a.thi, a.vb}; })
  BN_UNROLL for (int k = 0; k < NL; ++k) ws_store(w, k, a.l[k]);
}
BN_INL Fp2 fp2_load_mem(const Ws& w) { return {fp_load_mem(w), fp_load_mem(ws_at(w, 9))}; }
BN_INL Fp6 fp6_load_mem(const Ws& w) { return {fp2_load_mem(w), fp2_load_mem(ws_at(w, 18)), fp2_load_mem(ws_at(w, 36))}; }
BN_INL void fp2_store_mem(const Ws& w, const Fp2& a) { fp_store_mem(w, a.c0); fp_store_mem(ws_at(w, 9), a.c1); }
BN_INL void fp6_store_mem(const Ws& w, const Fp6& a) {
  fp2_store_mem(w, a.c0); fp2_store_mem(ws_at(w, 18), a.c1); fp2_store_mem(ws_at(w, 36), a.c2);
}
BN_INL void fp12_store_mem(const Ws& w, const Fp12& a) { fp6_store_mem(w, a.c0); fp6_store_mem(ws_at(w, 54), a.c1); }
BN_INL Fp12 fp12_load_mem(const Ws& w) { return {fp6_load_mem(w), fp6_load_mem(ws_at(w, 54))}; }
// a * b with b parked in memory.  Without `park`, b's Fp6 halves are loaded when needed and never held across
// phases (three loads of b).  With `park` naming an LDS column of 108 limbs, b is read once: b1 - b0 and the first
// partial product wait in LDS while the others are computed.
BN_FUNC Fp12 fp12_mul_mem(const Fp12& a, const Ws& b, const Ws* park = nullptr) {
  Fp6 v0, v1, w;
  if (park) {
    Fp6 b0 = fp6_load_mem(b), b1 = fp6_load_mem(ws_at(b, 54));
    fp6_store_mem(ws_at(*park, 54), fp6_norm(fp6_sub(b1, b0)));
    v0 = fp6_mul(a.c0, b0);
    fp6_store_mem(*park, v0);
    BN_MEM_FENCE;
    v1 = fp6_mul(a.c1, b1);
    BN_MEM_FENCE;
    w = fp6_mul(fp6_norm(fp6_sub(a.c0, a.c1)), fp6_load_mem(ws_at(*park, 54)));
    v0 = fp6_load_mem(*park);
  } else {
    { Fp6 b0 = fp6_load_mem(b); v0 = fp6_mul(a.c0, b0); }
    BN_MEM_FENCE;
    { Fp6 b1 = fp6_load_mem(ws_at(b, 54)); v1 = fp6_mul(a.c1, b1); }
    BN_MEM_FENCE;
    { Fp6 b0 = fp6_load_mem(b), b1 = fp6_load_mem(ws_at(b, 54));
      w = fp6_mul(fp6_norm(fp6_sub(a.c0, a.c1)), fp6_norm(fp6_sub(b1, b0))); }
  }
  return {fp6_add_mul_v(v0, v1), fp6_norm(fp6_add(fp6_add(w, v0), v1))};
}
BN_FUNC Fp12 fp12_sqr(const Fp12& a) {                    // complex squaring, fp12.rs:170-180 value
  Fp6 ab = fp6_mul(a.c0, a.c1);
  Fp6 s1 = fp6_norm(fp6_add(a.c0, a.c1));
  Fp6 s2 = fp6_add_mul_v(a.c0, a.c1);
  Fp6 t = fp6_mul(s1, s2);                                     // a0^2 + v a1^2 + (1+v) ab
  Fp6 tm = fp6_sub(t, ab);
  Fp12 r;
  r.c0 = {fp2_sub_mul_xi(tm.c0, ab.c2), fp2_norm(fp2_sub(tm.c1, ab.c0)), fp2_norm(fp2_sub(tm.c2, ab.c1))};
  r.c1 = fp6_norm(fp6_add(ab, ab));
  return r;
}
BN_FUNC Fp12 fp12_inv(const Fp12& a) {                    // fp12.rs:212-219
  Fp6 s0 = fp6_sqr(a.c0), s1 = fp6_sqr(a.c1);
  Fp6 d = {fp2_sub_mul_xi(s0.c0, s1.c2), fp2_norm(fp2_sub(s0.c1, s1.c0)), fp2_norm(fp2_sub(s0.c2, s1.c1))};
  Fp6 t = fp6_inv(d);
  return {fp6_mul(a.c0, t), fp6_norm(fp6_neg(fp6_mul(a.c1, t)))};
}
// f * (o0 + o3 w + o4 w^3): sparse multiply for D-type twist lines (E7).  13 Fp2 products.
BN_FUNC Fp12 fp12_mul_by_034(const Fp12& f, const Fp2& o0, const Fp2& o3, const Fp2& o4) {
  BN_CTX;
  Fp6 a = fp6_mul_fp2(f.c0, o0);
  Fp6 b = fp6_mul_by_01(f.c1, o3, o4);
  // (f0 + f1)(o0 + o3, o4) - a - b, computed subtractively: e = (f0 - f1)*(o3 - o0, o4) ... keep it
  // simple: normalise the two sums (their limbs are non-negative) and use mul_by_01.
  Fp6 e = fp6_mul_by_01(fp6_norm(fp6_add(f.c0, f.c1)), fp2_norm(fp2_add(o0, o3)), o4);
  return {fp6_add_mul_v(a, b), fp6_norm(fp6_sub(fp6_sub(e, a), b))};
}
// f * la * lb for two sparse lines la = a0 + a3 w + a4 v w, lb likewise: first the product of the lines
//   (a0 b0 + xi a4 b4) + a3 b3 v + (a3 b4 + a4 b3) v^2 + (a0 b3 + a3 b0) w + (a0 b4 + a4 b0) v w      (6 Fp2 products),
// then one Fp12 product that uses the missing v^2 w coefficient (17 Fp2 products): 23 instead of 2 x 13.
BN_FUNC Fp12 fp12_mul_by_two_lines(const Fp12& f, const Fp2& a0, const Fp2& a3, const Fp2& a4, const Fp2& b0, const Fp2& b3, const Fp2& b4) {
  BN_CTX;
  Fp2 m00 = fp2_mul(a0, b0), m33 = fp2_mul(a3, b3), m44 = fp2_mul(a4, b4);
  Fp2 w34 = fp2_mul(fp2_sub(a3, a4), fp2_sub(b3, b4));
  Fp2 w03 = fp2_mul(fp2_sub(a0, a3), fp2_sub(b0, b3));
  Fp2 w04 = fp2_mul(fp2_sub(a0, a4), fp2_sub(b0, b4));
  Fp6 l0 = {fp2_add_mul_xi(m00, m44), m33, fp2_norm(fp2_sub(fp2_add(m33, m44), w34))};
  Fp2 l10 = fp2_norm(fp2_sub(fp2_add(m00, m33), w03)), l11 = fp2_norm(fp2_sub(fp2_add(m00, m44), w04));
  Fp6 v0 = fp6_mul(f.c0, l0);
  Fp6 v1 = fp6_mul_by_01(f.c1, l10, l11);
  Fp6 dl = {fp2_norm(fp2_sub(l10, l0.c0)), fp2_norm(fp2_sub(l11, l0.c1)), fp2_norm(fp2_neg(l0.c2))};       // l1 - l0
  Fp6 w = fp6_mul(fp6_norm(fp6_sub(f.c0, f.c1)), dl);
  return {fp6_add_mul_v(v0, v1), fp6_norm(fp6_add(fp6_add(w, v0), v1))};
}
// Frobenius^k, k = 1..3: coefficient of w^i -> conj^k(.) * xi^(i (p^k-1)/6)   (E3 fixed)
template <int K, int I>
BN_INL Fp2 fp12_frob_coeff(const Fp2& in) {                     // coefficient of w^I
  Fp2 c = in;
  if (K & 1) c = fp2_norm(fp2_conj(c));
  if (I > 0) {
    if (K == 1) c = fp2_mul(c, fp2_const(bnc::GAMMA1[I - 1]));
    if (K == 2) c = fp2_mul_fp(c, fp_const(bnc::GAMMA2[I - 1]));
    if (K == 3) c = fp2_mul(c, fp2_const(bnc::GAMMA3[I - 1]));
  }
  return c;
}
template <int K>
BN_FUNC Fp12 fp12_frob(const Fp12& a) {
  BN_CTX;
  // tower slot -> w index: c0.c0=0 c1.c0=1 c0.c1=2 c1.c1=3 c0.c2=4 c1.c2=5
  return {{fp12_frob_coeff<K, 0>(a.c0.c0), fp12_frob_coeff<K, 2>(a.c0.c1), fp12_frob_coeff<K, 4>(a.c0.c2)},
          {fp12_frob_coeff<K, 1>(a.c1.c0), fp12_frob_coeff<K, 3>(a.c1.c1), fp12_frob_coeff<K, 5>(a.c1.c2)}};
}
// Granger-Scott squaring in the cyclotomic subgroup, pairings.rs:68-115 (valid with xi = 9+u).  Same values as
//   fp4_square(t0, t1, z0, z1) ... ; z0' = 3 t0 - 2 z0 ; z1' = 3 t1 + 2 z1 ; ...
// but the combination 3 (ta + xi tb) -+ 2 z of each output is ONE four-term pass over the raw squarings ta = a^2,
// tb = b^2, s = (a + b)^2 -- three squarings (2 x 162 MADs each) instead of two squarings and a product (486 MADs) -- (fp_lc4) instead of fp4_square's (pairings.rs:52-63) own pass followed by a second one: 4 passes per Fp4 square
// instead of 8 (a pass is ~60 instructions per Fp, about a quarter of a multiplication).
struct Fp4Sq { Fp2 ta, tb, s; };
BN_INL Fp4Sq fp4_sq_raw(const Fp2& a, const Fp2& b) { return {fp2_sqr(a), fp2_sqr(b), fp2_sqr(fp2_norm(fp2_add(a, b)))}; }
// 3 (ta + xi tb) - 2 z
BN_INL Fp2 cyc_c0(const Fp4Sq& q, const Fp2& z) {
  return {fp_lc4<3, 27, -3, -2>(q.ta.c0, q.tb.c0, q.tb.c1, z.c0), fp_lc4<3, 3, 27, -2>(q.ta.c1, q.tb.c0, q.tb.c1, z.c1)};
}
// 3 (s - ta - tb) + 2 z
BN_INL Fp2 cyc_c1(const Fp4Sq& q, const Fp2& z) {
  return {fp_lc4<3, -3, -3, 2>(q.s.c0, q.ta.c0, q.tb.c0, z.c0), fp_lc4<3, -3, -3, 2>(q.s.c1, q.ta.c1, q.tb.c1, z.c1)};
}
BN_FUNC Fp12 fp12_cyclotomic_sqr(const Fp12& f) {
  BN_CTX;
  Fp2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
  Fp4Sq q01 = fp4_sq_raw(z0, z1);
  Fp2 r0 = cyc_c0(q01, z0), r1 = cyc_c1(q01, z1);               // z0 = 3 t0 - 2 z0 ; z1 = 3 t1 + 2 z1
  Fp4Sq q23 = fp4_sq_raw(z2, z3);
  Fp2 r4 = cyc_c0(q23, z4), r5 = cyc_c1(q23, z5);               // z4 = 3 t2 - 2 z4 ; z5 = 3 t3 + 2 z5
  Fp4Sq q45 = fp4_sq_raw(z4, z5);
  Fp2 r3 = cyc_c0(q45, z3);                                     // z3 = 3 t4 - 2 z3
  // t5 = s - ta - tb with its value brought back below ~p (a reducing pass): it is scaled by 27 next, and the three
  // squarings of a lazily combined input can each be worth several p
  Fp2 t5 = {fp_lc4<1, -1, -1, 0, true>(q45.s.c0, q45.ta.c0, q45.tb.c0, q45.s.c0), fp_lc4<1, -1, -1, 0, true>(q45.s.c1, q45.ta.c1, q45.tb.c1, q45.s.c1)};
  Fp2 r2 = {fp_lc3<27, -3, 2>(t5.c0, t5.c1, z2.c0), fp_lc3<3, 27, 2>(t5.c0, t5.c1, z2.c1)};   // z2 = 3 xi t5 + 2 z2
  return {{r0, r4, r3}, {r2, r1, r5}};
}
BN_INL bool fp12_is_one(const Fp12& a) {
  bool z = fp_is_zero(fp_sub(a.c0.c0.c0, fp_one()));
  z &= fp_is_zero(a.c0.c0.c1);
  z &= fp2_is_zero(a.c0.c1) & fp2_is_zero(a.c0.c2);
  z &= fp2_is_zero(a.c1.c0) & fp2_is_zero(a.c1.c1) & fp2_is_zero(a.c1.c2);
  return z;
}

}  // namespace bn
