// tri.h -- THREE LANES PER TUPLE: Fp12 arithmetic for launches between "a few thousand" and "a round of waves" of tuples.
//
// One lane per tuple is right for throughput, but a launch that does not fill the chip costs the latency of ONE lane's serial
// chain whatever its size: 4.4 ms for the table-only Miller loop and 5.4 ms for the final exponentiation for 5 000 tuples as for
// 65 000 (the chunk and fallback rounds of the RLC path, any mid-size verify_batch).  One WAVE per tuple (wide.h) shortens the
// chain ~7 x but spends ~14 x the SIMD time per tuple: right up to ~4 000 tuples only.  In between, a QUAD of lanes shares a
// tuple: lane 0 holds c0 and lane 1 holds c1 of f = c0 + c1 w (Fp6 each, in registers), and the three Fp6 products of an Fp12
// Karatsuba product (fp12.rs:203-210) run side by side on lanes 0, 1, 2 -- every lane executes the SAME fp6_mul (tower.h) on
// its own operands; operands and results cross lanes by DPP quad permutes (one v_mov_dpp per limb, no LDS, no barrier).
// Lane 3 of the quad idles.  Chains get ~2.4 x shorter for ~1.7 x the lane-instructions per tuple, and 16 384 tuples are
// exactly one round of waves.  Same field values as the serial code, hence the same canonical limbs and bytes.
//   tri_mul        a b           v0 = a0 b0 | v1 = a1 b1 | (a0 - a1)(b1 - b0);   c0 = v0 + v v1,  c1 = w + v0 + v1
//   tri_sqr        a^2           a0^2 | a1^2 | (a0 + a1)^2;                      c0 = s0 + v s1,  c1 = s2 - s0 - s1   (fp12.rs:170-180 value)
//   tri_cyc_sqr    Granger-Scott (pairings.rs:68-115): the Fp4 squarings of (z0, z1) | (z2, z3) | (z4, z5), one per lane
//   tri_conj / tri_frob<K>       per lane (its own three Fp2 coefficients, its own constants)
// On the host (tests/hostsim, -DBN_CHECK) the four lanes of a quad run as four threads and a fetch is a rendezvous: the same
// source, under the interval checker.
#pragma once
#include "pairing.h"

namespace bn {

#if defined(__HIP_DEVICE_COMPILE__)
BN_INL uint32_t tri_role() { return threadIdx.x & 3u; }
template <int P0, int P1, int P2, int P3> BN_INL Fp tri_fetch(const Fp& x) {     // lane r of every quad receives x of lane P_r
  Fp r;
  BN_UNROLL for (int k = 0; k < NL; ++k) r.l[k] = __builtin_amdgcn_update_dpp(0, x.l[k], P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, false);
  return r;
}
#else
uint32_t tri_host_role();
Fp tri_host_fetch(const Fp& x, int p0, int p1, int p2, int p3);                   // hostsim: rendezvous of the quad's four threads
BN_INL uint32_t tri_role() { return tri_host_role(); }
template <int P0, int P1, int P2, int P3> BN_INL Fp tri_fetch(const Fp& x) { return tri_host_fetch(x, P0, P1, P2, P3); }
#endif
template <int P0, int P1, int P2, int P3> BN_INL Fp2 tri_fetch2(const Fp2& x) { return {tri_fetch<P0, P1, P2, P3>(x.c0), tri_fetch<P0, P1, P2, P3>(x.c1)}; }
template <int P0, int P1, int P2, int P3> BN_INL Fp6 tri_fetch6(const Fp6& x) {
  return {tri_fetch2<P0, P1, P2, P3>(x.c0), tri_fetch2<P0, P1, P2, P3>(x.c1), tri_fetch2<P0, P1, P2, P3>(x.c2)};
}
// Selection by the lane's ROLE (never by data): a v_cndmask per limb on the device.  Under the interval checker the bounds of
// the branch this lane takes are carried over as they are -- the untaken branch holds another role's intermediate, and
// inheriting its bounds (as the data-dependent fp_select must) would make every lane pay for values it never computes with.
BN_INL Fp fp_pick(bool c, const Fp& a, const Fp& b) {
#ifdef BN_CHECK
  return c ? a : b;
#else
  return fp_select(c, a, b);
#endif
}
BN_INL Fp2 fp2_pick(bool c, const Fp2& a, const Fp2& b) { return {fp_pick(c, a.c0, b.c0), fp_pick(c, a.c1, b.c1)}; }
BN_INL Fp6 fp6_select(bool c, const Fp6& a, const Fp6& b) { return {fp2_select(c, a.c0, b.c0), fp2_select(c, a.c1, b.c1), fp2_select(c, a.c2, b.c2)}; }   // data-dependent
BN_INL Fp6 fp6_pick(bool c, const Fp6& a, const Fp6& b) { return {fp2_pick(c, a.c0, b.c0), fp2_pick(c, a.c1, b.c1), fp2_pick(c, a.c2, b.c2)}; }

// The third Karatsuba operand: lanes 0 and 1 keep their own half x (normalised); lanes 2 (and 3) get x0 + x1 (ADD) or
// x_hi - x_lo with (hi, lo) = (0, 1) or (1, 0), normalised.
template <bool ADD, bool SWAP> BN_INL Fp6 tri_third(const Fp6& x, uint32_t role) {
  BN_CTX;
  const Fp6 u = SWAP ? tri_fetch6<0, 1, 1, 1>(x) : tri_fetch6<0, 1, 0, 0>(x);      // lanes 2, 3: the minuend's half
  const Fp6 v = SWAP ? tri_fetch6<0, 1, 0, 0>(x) : tri_fetch6<0, 1, 1, 1>(x);
  const Fp6 d = fp6_norm(ADD ? fp6_add(u, v) : fp6_sub(u, v));
  return fp6_pick(role >= 2u, d, x);
}
// c0 = p + v q on lane 0, c1 as given on lane 1: the recombination shared by the product and the square
BN_INL Fp6 tri_recombine(const Fp6& p, const Fp6& x1, const Fp6& c1, uint32_t role) {
  BN_CTX;
  return fp6_pick(role == 0u, fp6_add_mul_v(p, x1), c1);
}
// a * b: each lane holds its half of a and of b (normalised); returns its half of the product (normalised).  Lanes 2, 3: don't care.
BN_FUNC Fp6 tri_mul(const Fp6& a, const Fp6& b, uint32_t role) {
  BN_CTX;
  const Fp6 p = fp6_mul(tri_third<false, false>(a, role), tri_third<false, true>(b, role));     // v0 | v1 | (a0 - a1)(b1 - b0)
  const Fp6 x1 = tri_fetch6<1, 0, 0, 0>(p);                      // lane 0: v1, lane 1: v0
  const Fp6 x2 = tri_fetch6<2, 2, 2, 2>(p);                      // the cross product
  return tri_recombine(p, x1, fp6_norm(fp6_add(fp6_add(p, x1), x2)), role);
}
// a^2 for any a (the Miller loop's squaring)
BN_FUNC Fp6 tri_sqr(const Fp6& a, uint32_t role) {
  BN_CTX;
  const Fp6 s = tri_third<true, false>(a, role);
  const Fp6 p = fp6_mul(s, s);                                   // a0^2 | a1^2 | (a0 + a1)^2
  const Fp6 x1 = tri_fetch6<1, 0, 0, 0>(p);
  const Fp6 x2 = tri_fetch6<2, 2, 2, 2>(p);
  return tri_recombine(p, x1, fp6_norm(fp6_sub(fp6_sub(x2, x1), p)), role);
}
// (c0, -c1): lane 1 negates
BN_INL Fp6 tri_conj(const Fp6& a, uint32_t role) { return fp6_pick(role == 1u, fp6_norm(fp6_neg(a)), a); }
// Frobenius^K: lane 0 holds the coefficients of w^0, w^2, w^4, lane 1 those of w^1, w^3, w^5 (tower.h fp12_frob)
template <int K> BN_FUNC Fp6 tri_frob(const Fp6& a, uint32_t role) {
  BN_CTX;
  const bool odd = role == 1u;
  const Fp6 e = {fp12_frob_coeff<K, 0>(a.c0), fp12_frob_coeff<K, 2>(a.c1), fp12_frob_coeff<K, 4>(a.c2)};
  const Fp6 o = {fp12_frob_coeff<K, 1>(a.c0), fp12_frob_coeff<K, 3>(a.c1), fp12_frob_coeff<K, 5>(a.c2)};
  return fp6_pick(odd, o, e);
}
// Granger-Scott squaring in the cyclotomic subgroup (tower.h fp12_cyclotomic_sqr: same values).  Lane p < 3 squares the Fp4
// pair p -- (z0, z1) | (z2, z3) | (z4, z5) with z0 = c0.c0, z4 = c0.c1, z3 = c0.c2 on lane 0 and z2 = c1.c0, z1 = c1.c1,
// z5 = c1.c2 on lane 1 -- and forms the two outputs that pair feeds: (r0, r1) | (r4, r5) | (r3, r2), which need the old
// z of their OWN slot.  The outputs then go back to their slots: c0 = (r0, r4, r3), c1 = (r2, r1, r5).
BN_FUNC Fp6 tri_cyc_sqr(const Fp6& h, uint32_t role) {
  BN_CTX;
  const Fp6 f0 = tri_fetch6<0, 0, 0, 0>(h), f1 = tri_fetch6<1, 1, 1, 1>(h);       // (z0, z4, z3), (z2, z1, z5) on every lane
  const bool r0 = role == 0u, r1 = role == 1u;
  const Fp2 za = fp2_pick(r0, f0.c0, fp2_pick(r1, f1.c0, f0.c1));            // z0 | z2 | z4
  const Fp2 zb = fp2_pick(r0, f1.c1, fp2_pick(r1, f0.c2, f1.c2));            // z1 | z3 | z5
  const Fp2 oa = fp2_pick(r0, f0.c0, fp2_pick(r1, f0.c1, f0.c2));            // old z0 | z4 | z3
  const Fp2 ob = fp2_pick(r0, f1.c1, fp2_pick(r1, f1.c2, f1.c0));            // old z1 | z5 | z2
  const Fp4Sq q = fp4_sq_raw(za, zb);
  const Fp2 ra = cyc_c0(q, oa);                                                   // 3 (ta + xi tb) - 2 z
  // second output: 3 (s - ta - tb) + 2 z on lanes 0, 1; 3 xi (s - ta - tb) + 2 z on lane 2 (its 2ab term sits one v higher).
  // t5 = s - ta - tb is brought back below ~p first in both forms (it is scaled by up to 27 next).
  const Fp2 t5 = {fp_lc4<1, -1, -1, 0, true>(q.s.c0, q.ta.c0, q.tb.c0, q.s.c0), fp_lc4<1, -1, -1, 0, true>(q.s.c1, q.ta.c1, q.tb.c1, q.s.c1)};
  const Fp2 rb_plain = {fp_lc2<3, 2>(t5.c0, ob.c0), fp_lc2<3, 2>(t5.c1, ob.c1)};
  const Fp2 rb_xi = {fp_lc3<27, -3, 2>(t5.c0, t5.c1, ob.c0), fp_lc3<3, 27, 2>(t5.c0, t5.c1, ob.c1)};
  const Fp2 rb = fp2_pick(role >= 2u, rb_xi, rb_plain);
  const Fp2 a0 = tri_fetch2<0, 0, 0, 0>(ra), a1 = tri_fetch2<1, 1, 1, 1>(ra), a2 = tri_fetch2<2, 2, 2, 2>(ra);      // r0, r4, r3
  const Fp2 b0 = tri_fetch2<0, 0, 0, 0>(rb), b1 = tri_fetch2<1, 1, 1, 1>(rb), b2 = tri_fetch2<2, 2, 2, 2>(rb);      // r1, r5, r2
  const Fp6 e = {a0, a1, a2}, o = {b2, b0, b1};
  return fp6_pick(role == 1u, o, e);
}

// ---- values parked in memory: an Fp12 value of tuple i is 108 limbs of a limb-major workspace, lane `role` owns limbs
// 54 role .. 54 role + 53 (so the layout IS the one-lane-per-tuple layout of fp12_store_limbs: c0 first).  Lanes 2, 3 of a
// quad read lane 1's half (any valid address will do) and never store.
BN_INL uint32_t tri_arole(uint32_t role) { return role < 1u ? 0u : 1u; }
// A workspace reference moved by a LANE-DEPENDENT number of limbs: the offset goes into the per-lane byte offset.  (ws_at moves
// the base, which buffer addressing keeps in scalar registers: a base that differs between the lanes of a wave would have to be
// serialised lane by lane.)  limbs * stride * 4 must fit 32 bits together with the lane's own offset.
BN_INL Ws ws_at_lane(const Ws& w, uint32_t limbs) { return {w.base, w.stride, w.lane4 + limbs * (uint32_t)w.stride * 4u, w.buf}; }
// canonical limbs: the exchange format with the one-lane-per-tuple kernels (f_ws, the easy part's output)
BN_INL Fp6 tri_load_canon(const Ws& w, uint32_t role) {
  const Ws h = ws_at_lane(w, 54u * tri_arole(role));
  return {fp2_load_limbs(h), fp2_load_limbs(ws_at(h, 18)), fp2_load_limbs(ws_at(h, 36))};
}
BN_INL void tri_store_canon(const Ws& w, const Fp6& a, uint32_t role) {
  if (role < 2u) {
    const Ws h = ws_at_lane(w, 54u * role);
    fp2_store_limbs(h, a.c0); fp2_store_limbs(ws_at(h, 18), a.c1); fp2_store_limbs(ws_at(h, 36), a.c2);
  }
}
// normalised limbs as they are (no canonicalisation): the named values of the hard part
// (all 54 limbs through ONE reference: fp6_load_mem would move the base once per Fp -- six buffer descriptors of four scalar
// registers each per value, and with two or three values in flight the scalar file overflows into vector registers, whose
// contents can only come back through a read-first-lane loop around every access)
BN_INL Fp fp_load_mem_at(const Ws& w, int j) {                   // the j-th Fp (limbs 9 j .. 9 j + 8) behind w
  Fp r;
#if defined(__HIP_DEVICE_COMPILE__)
  if (w.buf) {                                                   // running VECTOR offset: no per-limb scalar offsets to keep (54 per value)
    const uint32_t step = (uint32_t)w.stride * 4u;
    uint32_t off = w.lane4 + 9u * (uint32_t)j * step;
    BN_UNROLL for (int k = 0; k < NL; ++k) {
      r.l[k] = __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc((void*)w.base, 0, -1, 0x00020000), (int)off, 0, 0);
      off += step;
    }
    return r;
  }
#endif
  BN_UNROLL for (int k = 0; k < NL; ++k) r.l[k] = ws_load(w, 9 * j + k);
  BN_TRK(ParkTrk pt; { std::lock_guard<std::mutex> g(park_mutex()); auto it = park_trk().find(ws_addr(w, 9 * j));
           if (it == park_trk().end()) check_fail("fp_load_mem_at of an address never stored", 0); pt = it->second; }
         set_trk(r, pt.lo, pt.hi, pt.tlo, pt.thi, pt.vb);)
  return r;
}
BN_INL void fp_store_mem_at(const Ws& w, int j, const Fp& a) {
  BN_TRK(if (a.lo < -4e-6 || a.hi > 1.0 + 4e-6) check_fail("fp_store_mem_at needs a normalised value", mag(a));
         { std::lock_guard<std::mutex> g(park_mutex()); park_trk()[ws_addr(w, 9 * j)] = ParkTrk{a.lo, a.hi, a.tlo, a.thi, a.vb}; })
#if defined(__HIP_DEVICE_COMPILE__)
  if (w.buf) {
    const uint32_t step = (uint32_t)w.stride * 4u;
    uint32_t off = w.lane4 + 9u * (uint32_t)j * step;
    BN_UNROLL for (int k = 0; k < NL; ++k) {
      __builtin_amdgcn_raw_buffer_store_b32(a.l[k], __builtin_amdgcn_make_buffer_rsrc((void*)w.base, 0, -1, 0x00020000), (int)off, 0, 0);
      off += step;
    }
    return;
  }
#endif
  BN_UNROLL for (int k = 0; k < NL; ++k) ws_store(w, 9 * j + k, a.l[k]);
}
BN_INL Fp6 tri_load(const Ws& w0, uint32_t role) {
  const Ws w = ws_at_lane(w0, 54u * tri_arole(role));
  return {{fp_load_mem_at(w, 0), fp_load_mem_at(w, 1)}, {fp_load_mem_at(w, 2), fp_load_mem_at(w, 3)}, {fp_load_mem_at(w, 4), fp_load_mem_at(w, 5)}};
}
BN_INL void tri_store(const Ws& w0, const Fp6& a, uint32_t role) {
  if (role < 2u) {
    const Ws w = ws_at_lane(w0, 54u * role);
    fp_store_mem_at(w, 0, a.c0.c0); fp_store_mem_at(w, 1, a.c0.c1); fp_store_mem_at(w, 2, a.c1.c0); fp_store_mem_at(w, 3, a.c1.c1);
    fp_store_mem_at(w, 4, a.c2.c0); fp_store_mem_at(w, 5, a.c2.c1);
  }
}
// lane 0: c0 == 1, lane 1: c1 == 0; other lanes: true
BN_INL bool tri_half_is_one(const Fp6& a, uint32_t role) {
  const bool z = fp_is_zero(fp_sub(a.c0.c0, fp_pick(role == 0u, fp_one(), fp_zero()))) & fp_is_zero(a.c0.c1) & fp2_is_zero(a.c1) & fp2_is_zero(a.c2);
  return z | (role >= 2u);
}

// ---- the table-only Miller loop of the verify equation (miller_loop_prepared, pairing.h) on a quad.
// Line pair of a step from the key's expanded table e (T0 .. T8, 162 limbs) and the tuple's nine coordinate values cw (LDS,
// shared by the quad):  l0 = (T0 ysY + T1 Z, T2 xsX, T3 xsZ + T4 X) on lane 0,  l1 = (T5 ysX + T6 xsY, T7 ysZ + T8 Y, 0) on lane 1,
// every lane evaluating its three coefficients by the same three double products (a missing term multiplies by zero), and
// l1 - l0 on lane 2.  Then f <- f * (l0 + l1 w) as a tri_mul whose second operand is already distributed.
BN_FUNC Fp6 tri_line_pair(const Ws& e, const Ws& cw, uint32_t role) {
  BN_CTX;
  const bool odd = role == 1u;
  // table entries / coordinate slots of the three coefficients: (ta, ca, tb, cb); cb < 0: no second term
  const uint32_t ta0 = odd ? 5u : 0u, ca0 = odd ? 7u : 4u, tb0 = odd ? 6u : 1u, cb0 = odd ? 8u : 2u;
  const uint32_t ta1 = odd ? 7u : 2u, ca1 = odd ? 6u : 3u, tb1 = odd ? 8u : 2u, cb1 = odd ? 1u : 0u;      // lane 0: T2 xsX alone (second term zeroed)
  const uint32_t ta2 = odd ? 0u : 3u, ca2 = odd ? 0u : 5u, tb2 = odd ? 0u : 4u, cb2 = odd ? 0u : 0u;      // lane 1: no third coefficient
  const Fp zero = fp_zero();
  const Fp s1b = fp_pick(odd, fp_load_mem(ws_at_lane(cw, 9u * cb1)), zero);
  const Fp s2a = fp_pick(odd, zero, fp_load_mem(ws_at_lane(cw, 9u * ca2))), s2b = fp_pick(odd, zero, fp_load_mem(ws_at_lane(cw, 9u * cb2)));
  Fp6 l;
  l.c0 = fp2_dot_fp(fp2_load_limbs_lazy(ws_at_lane(e, 18u * ta0)), fp_load_mem(ws_at_lane(cw, 9u * ca0)), fp2_load_limbs_lazy(ws_at_lane(e, 18u * tb0)), fp_load_mem(ws_at_lane(cw, 9u * cb0)));
  l.c1 = fp2_dot_fp(fp2_load_limbs_lazy(ws_at_lane(e, 18u * ta1)), fp_load_mem(ws_at_lane(cw, 9u * ca1)), fp2_load_limbs_lazy(ws_at_lane(e, 18u * tb1)), s1b);
  l.c2 = fp2_dot_fp(fp2_load_limbs_lazy(ws_at_lane(e, 18u * ta2)), s2a, fp2_load_limbs_lazy(ws_at_lane(e, 18u * tb2)), s2b);
  return l;                                                       // lane 0: l0, lane 1: l1 (c2 = 0); lanes 2, 3: l0's formulas (unused)
}
BN_FUNC Fp6 tri_miller_prepared(const Ws& cw, const Ws& ktab_in, uint32_t role) {
  BN_CTX;
  Fp6 f = fp6_pick(role == 0u, fp6_one(), fp6_zero());
  f = fp6_norm(f);
  Ws kt = ktab_in, p = cw;
  int ti = 0;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= -2; --j) {               // j = -1, -2: the two final lines (no squaring)
    if (j >= 0) f = tri_sqr(f, role);
    const int lines = j >= 0 ? (ate_naf_digit(j) != 0 ? 2 : 1) : 1;
    for (int q = 0; q < lines; ++q) {
      BN_OPAQUE(kt); BN_OPAQUE(p);
      const Fp6 l = tri_line_pair(ws_at(kt, 162 * (size_t)ti), p, role);
      ++ti;
      // second operand already per lane: lanes 0, 1 own halves; lane 2 needs l1 - l0
      const Fp6 u = tri_fetch6<0, 1, 1, 1>(l), v = tri_fetch6<0, 1, 0, 0>(l);
      const Fp6 lb = fp6_pick(role >= 2u, fp6_norm(fp6_sub(u, v)), l);
      const Fp6 pr = fp6_mul(tri_third<false, false>(f, role), lb);
      const Fp6 x1 = tri_fetch6<1, 0, 0, 0>(pr);
      const Fp6 x2 = tri_fetch6<2, 2, 2, 2>(pr);
      f = tri_recombine(pr, x1, fp6_norm(fp6_add(fp6_add(pr, x1), x2)), role);
    }
  }
  return f;
}

// ---- hard part of the final exponentiation on a quad: the op sequence of wide_fe_hard (wide.h) = fe_h1 / fe_h2 / fe_h3 around
// three t -> t^x chains (pairing.h), with the accumulator R in registers and the named values parked in `vals`
// (TRI_VALUES x 108 limbs per tuple, limb-major).
enum : uint32_t { TV_T = 0, TV_A = 1, TV_B = 2, TV_C = 3, TV_B2 = 4, TV_D2 = 5, TV_X = 6, TV_E = 7, TV_D = 8, TV_TMP = 9, TV_SLOT0 = 10, TRI_VALUES = 20 };
// (ws_uniform: v comes from the chain's program table, and the 64-bit base arithmetic on it may be done in vector registers -- the
// base would then pass for lane-dependent and every access be wrapped in a read-first-lane loop: 1377 of them in k_fe_tri_hard)
BN_INL Ws tri_val(const Ws& vals, uint32_t v) { return ws_uniform(ws_at(vals, 108u * (size_t)v)); }
BN_FUNC Fp6 tri_exp_x(const Fp6& in, const Ws& vals, uint32_t role) {
  const ChainOp prog[BN_X_CHAIN_LEN] = BN_X_CHAIN;
  Ws vw = vals;
  tri_store(tri_val(vw, TV_SLOT0), in, role);
  BN_MEM_FENCE;
  Fp6 r = in;
  for (int k = 0; k < BN_X_CHAIN_LEN; ++k) {
    const ChainOp op = prog[k];
    BN_OPAQUE_S(vw);
    if (op.load >= 0) r = tri_load(tri_val(vw, TV_SLOT0 + (uint32_t)op.load), role);
    for (int q = 0; q < op.sq; ++q) r = tri_cyc_sqr(r, role);
    if (op.mul >= 0) r = tri_mul(r, tri_load(tri_val(vw, TV_SLOT0 + (uint32_t)op.mul), role), role);
    if (op.store >= 0) tri_store(tri_val(vw, TV_SLOT0 + (uint32_t)op.store), r, role);
    if (op.cstore >= 0) tri_store(tri_val(vw, TV_SLOT0 + (uint32_t)op.cstore), tri_conj(r, role), role);
    if (op.store >= 0 || op.cstore >= 0) BN_MEM_FENCE;
  }
  return r;
}
// t = f^((p^6-1)(p^2+1)) in -> f^((p^12-1)/r * 2x(6x^2+3x+1)) out (each lane its half)
BN_FUNC Fp6 tri_fe_hard(const Fp6& t, const Ws& vals_in, uint32_t role) {
  Ws vals = vals_in;
  tri_store(tri_val(vals, TV_T), t, role);
  Fp6 x = tri_exp_x(t, vals, role);                                                          // t^x
  const Fp6 a = tri_cyc_sqr(tri_conj(x, role), role);                                        // a = t^-2x
  tri_store(tri_val(vals, TV_A), a, role);
  const Fp6 b = tri_mul(tri_cyc_sqr(a, role), a, role);                                      // b = t^-6x
  tri_store(tri_val(vals, TV_B), b, role);
  BN_MEM_FENCE;
  x = tri_exp_x(b, vals, role);                                                              // t^(-6x^2)
  BN_OPAQUE_S(vals);
  const Fp6 c = tri_conj(x, role);                                                           // c = t^(6x^2)
  tri_store(tri_val(vals, TV_C), c, role);
  const Fp6 b2 = tri_mul(c, tri_conj(tri_load(tri_val(vals, TV_B), role), role), role);      // b2 = c conj(b)
  tri_store(tri_val(vals, TV_B2), b2, role);
  BN_MEM_FENCE;
  x = tri_exp_x(tri_cyc_sqr(c, role), vals, role);                                           // (c^2)^x = t^(12x^3)
  BN_OPAQUE_S(vals);
  const Fp6 e = tri_mul(tri_load(tri_val(vals, TV_B2), role), x, role);                      // e
  tri_store(tri_val(vals, TV_E), e, role);
  const Fp6 d = tri_mul(tri_load(tri_val(vals, TV_A), role), e, role);                       // d
  tri_store(tri_val(vals, TV_D), d, role);
  BN_MEM_FENCE;
  Fp6 r = tri_mul(tri_load(tri_val(vals, TV_C), role), e, role);
  BN_OPAQUE_S(vals);
  r = tri_mul(tri_load(tri_val(vals, TV_T), role), r, role);                                 // l0 = t (c e)
  r = tri_mul(r, tri_frob<1>(tri_load(tri_val(vals, TV_D), role), role), role);
  BN_OPAQUE_S(vals);
  r = tri_mul(r, tri_frob<2>(tri_load(tri_val(vals, TV_E), role), role), role);
  BN_OPAQUE_S(vals);
  const Fp6 l3 = tri_mul(tri_conj(tri_load(tri_val(vals, TV_T), role), role), tri_load(tri_val(vals, TV_D), role), role);   // conj(t) d
  return tri_mul(r, tri_frob<3>(l3, role), role);
}

}  // namespace bn
