// pairing.h -- optimal ate pairing on BN254: line steps, multi-pair Miller loop, final exponentiation.
//
// Reference operators replaced:
//   doubling_step / addition_step / ell   pairings.rs:888-962  (same Jacobian formulas and the same
//                                         coefficient scaling, so Miller-loop outputs are bit-identical
//                                         to the CPU oracle's, not just equal after final exponentiation)
//   miller_loop / pairing / multi_miller_loop  pairings.rs:760-886  (E4: BN optimal ate over NAF(6x+2)
//                                         plus the two Frobenius line additions; one shared f^2 per digit)
//   MillerLoopResult::final_exponentiation  pairings.rs:50-178 (E5: BN hard part, exponent pinned by
//                                         Gt::generator(): (p^12-1)/r * 2x(6x^2+3x+1))
//   G2Prepared / PairingCoefficients      pairings.rs:609-660,:726-757 (E6): the fixed -G2gen line table
//                                         is generated offline (bn254_consts.h), 88 entries
#pragma once
#include "curve.h"

namespace bn {

struct G2J { Fp2 x, y, z; };     // Jacobian running point T of the Miller loop
struct Line { Fp2 c0, c1, c2; };  // l = c0*yP + c1*xP*w + c2*w^3

// T <- 2T, returns the tangent line coefficients (pairings.rs:901-930).  T normalised in and out.
BN_FUNC Line doubling_step(G2J& r) {
  BN_CTX;
  Fp2 tmp0 = fp2_sqr(r.x);
  Fp2 tmp1 = fp2_sqr(r.y);
  Fp2 tmp2 = fp2_sqr(tmp1);
  Fp2 s3 = fp2_sqr(fp2_norm(fp2_add(tmp1, r.x)));
  Fp2 tmp3 = f_lc3<2, -2, -2>(s3, tmp0, tmp2);                   // 2((tmp1+x)^2 - tmp0 - tmp2)
  Fp2 tmp4 = f_lc2<3, 0>(tmp0, tmp0);                            // 3 x^2
  Fp2 tmp6 = fp2_norm(fp2_add(r.x, tmp4));
  Fp2 tmp5 = fp2_sqr(tmp4);
  Fp2 zsq = fp2_sqr(r.z);
  Fp2 nx = f_lc2<1, -2>(tmp5, tmp3);
  Fp2 nz = fp2_norm(fp2_sub(fp2_sub(fp2_sqr(fp2_norm(fp2_add(r.z, r.y))), tmp1), zsq));
  Fp2 ny = f_lc2<1, -8>(fp2_mul(fp2_sub(tmp3, nx), tmp4), tmp2);
  Line l;
  l.c1 = f_lc2<-2, 0>(fp2_mul(tmp4, zsq), zsq);                  // -2 * 3x^2 * z^2
  Fp2 s6 = fp2_sub(fp2_sub(fp2_sqr(tmp6), tmp0), tmp5);
  l.c2 = f_lc2<1, -4>(s6, tmp1);
  l.c0 = f_lc2<2, 0>(fp2_mul(nz, zsq), zsq);                     // 2 * z3 * z^2
  r.x = nx; r.y = ny; r.z = nz;
  return l;
}
// T <- T + Q (Q affine), returns the chord line coefficients (pairings.rs:932-962)
BN_FUNC Line addition_step(G2J& r, const Fp2& qx, const Fp2& qy) {
  BN_CTX;
  Fp2 zsq = fp2_sqr(r.z);
  Fp2 ysq = fp2_sqr(qy);
  Fp2 t0 = fp2_mul(zsq, qx);
  Fp2 sy = fp2_sqr(fp2_norm(fp2_add(qy, r.z)));
  Fp2 t1 = fp2_mul(fp2_norm(fp2_sub(fp2_sub(sy, ysq), zsq)), zsq);
  Fp2 t2 = fp2_norm(fp2_sub(t0, r.x));
  Fp2 t3 = fp2_sqr(t2);
  Fp2 t4 = f_lc2<4, 0>(t3, t3);
  Fp2 t5 = fp2_mul(t4, t2);
  Fp2 t6 = f_lc2<1, -2>(t1, r.y);
  Fp2 t9 = fp2_mul(t6, qx);
  Fp2 t7 = fp2_mul(t4, r.x);
  Fp2 nx = f_lc3<1, -1, -2>(fp2_sqr(t6), t5, t7);
  Fp2 nz = fp2_norm(fp2_sub(fp2_sub(fp2_sqr(fp2_norm(fp2_add(r.z, t2))), zsq), t3));
  Fp2 t8 = fp2_mul(fp2_sub(t7, nx), t6);
  Fp2 ny = f_lc2<1, -2>(t8, fp2_mul(r.y, t5));
  Fp2 s10 = fp2_sqr(fp2_norm(fp2_add(qy, nz)));
  Fp2 t10 = fp2_sub(fp2_sub(s10, ysq), fp2_sqr(nz));
  Line l;
  l.c2 = f_lc2<2, -1>(t9, t10);
  l.c0 = f_lc2<2, 0>(nz, nz);
  l.c1 = f_lc2<-2, 0>(t6, t6);
  r.x = nx; r.y = ny; r.z = nz;
  return l;
}
// f * l(P): scale the line by the G1 point and multiply sparsely (pairings.rs:888-899, slots 0/3/4)
BN_FUNC Fp12 ell(const Fp12& f, const Line& l, const Fp& px, const Fp& py) {
  BN_CTX;
  return fp12_mul_by_034(f, fp2_mul_fp(l.c0, py), fp2_mul_fp(l.c1, px), l.c2);
}
// f * la(Pa) * lb(Pb): both lines scaled by their G1 points, multiplied together first (fp12_mul_by_two_lines)
BN_FUNC Fp12 ell2(const Fp12& f, const Line& la, const Fp& pax, const Fp& pay, const Line& lb, const Fp& pbx, const Fp& pby) {
  BN_CTX;
  return fp12_mul_by_two_lines(f, fp2_mul_fp(la.c0, pay), fp2_mul_fp(la.c1, pax), la.c2, fp2_mul_fp(lb.c0, pby), fp2_mul_fp(lb.c1, pbx), lb.c2);
}
BN_INL Line fp2_norm_line(const Line& l) { return {fp2_norm(l.c0), fp2_norm(l.c1), fp2_norm(l.c2)}; }
BN_INL Line line_from_table(const int32_t* t) {                  // 54 strict limbs from the generated table
  return {fp2_from_limbs(t), fp2_from_limbs(t + 18), fp2_from_limbs(t + 36)};
}

// Digit j of NAF(6x+2), from two compile-time bit masks (non-zero, negative): scalar ALU only.  Read from the digit
// array in memory, each loop iteration ended in an exposed global load (~1.5 us per iteration at one wave per SIMD).
constexpr uint64_t ate_naf_mask(int word, bool neg) {
  uint64_t m = 0;
  for (int i = 0; i < 64; ++i) {
    int j = 64 * word + i;
    if (j < bnc::ATE_NAF_LEN && (neg ? bnc::ATE_NAF[j] < 0 : bnc::ATE_NAF[j] != 0)) m |= (uint64_t)1 << i;
  }
  return m;
}
BN_INL int ate_naf_digit(int j) {
  constexpr uint64_t nz0 = ate_naf_mask(0, false), nz1 = ate_naf_mask(1, false), ng0 = ate_naf_mask(0, true), ng1 = ate_naf_mask(1, true);
  const uint64_t nz = j < 64 ? nz0 : nz1, ng = j < 64 ? ng0 : ng1;
  const int b = j & 63;
  return ((nz >> b) & 1) ? (((ng >> b) & 1) ? -1 : 1) : 0;
}
// One-pair Miller loop f_{6x+2,Q}(P) * l_{T,pi(Q)}(P) * l_{T+pi(Q),-pi^2(Q)}(P)
BN_FUNC Fp12 miller_loop_1(const G1A& p, const G2A& q) {
  BN_CTX;
  Fp12 f = fp12_one();
  G2J T = {q.x, q.y, fp2_one()};
  Fp2 nqy = fp2_norm(fp2_neg(q.y));
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    f = ell(f, doubling_step(T), p.x, p.y);
    int d = ate_naf_digit(j);
    if (d != 0) f = ell(f, addition_step(T, q.x, d > 0 ? q.y : nqy), p.x, p.y);
  }
  Fp2 g2 = fp2_const(bnc::GAMMA1[1]), g3 = fp2_const(bnc::GAMMA1[2]);
  Fp2 q1x = fp2_mul(fp2_norm(fp2_conj(q.x)), g2), q1y = fp2_mul(fp2_norm(fp2_conj(q.y)), g3);        // pi(Q)
  Fp2 q2x = fp2_mul(fp2_norm(fp2_conj(q1x)), g2);
  Fp2 q2y = fp2_norm(fp2_neg(fp2_mul(fp2_norm(fp2_conj(q1y)), g3)));                                  // -pi^2(Q)
  f = ell(f, addition_step(T, q1x, q1y), p.x, p.y);
  f = ell(f, addition_step(T, q2x, q2y), p.x, p.y);
  return f;
}

// Two-pair loop of the verify equation: e(sig, -G2gen) * e(H, pk), the first pair's lines read from
// the precomputed table (uniform address: every lane of the wave reads the same entry).
BN_FUNC Fp12 miller_loop_verify(const G1A& sig, const G1A& h, const G2A& pk,
                                     const int32_t (*table)[54]) {
  BN_CTX;
  Fp12 f = fp12_one();
  G2J T = {pk.x, pk.y, fp2_one()};
  Fp2 nqy = fp2_norm(fp2_neg(pk.y));
  int ti = 0;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    f = ell(f, line_from_table(table[ti++]), sig.x, sig.y);
    f = ell(f, doubling_step(T), h.x, h.y);
    int d = ate_naf_digit(j);
    if (d != 0) {
      f = ell(f, line_from_table(table[ti++]), sig.x, sig.y);
      f = ell(f, addition_step(T, pk.x, d > 0 ? pk.y : nqy), h.x, h.y);
    }
  }
  Fp2 g2 = fp2_const(bnc::GAMMA1[1]), g3 = fp2_const(bnc::GAMMA1[2]);
  Fp2 q1x = fp2_mul(fp2_norm(fp2_conj(pk.x)), g2), q1y = fp2_mul(fp2_norm(fp2_conj(pk.y)), g3);
  Fp2 q2x = fp2_mul(fp2_norm(fp2_conj(q1x)), g2);
  Fp2 q2y = fp2_norm(fp2_neg(fp2_mul(fp2_norm(fp2_conj(q1y)), g3)));
  f = ell(f, line_from_table(table[ti++]), sig.x, sig.y);
  f = ell(f, addition_step(T, q1x, q1y), h.x, h.y);
  f = ell(f, line_from_table(table[ti++]), sig.x, sig.y);
  f = ell(f, addition_step(T, q2x, q2y), h.x, h.y);
  return f;
}

// Variant for k_miller_verify: the loop invariants are NOT kept in registers.  They sit in a limb-major
// workspace `inv` (slots of 9 limbs: 0 sig.x, 1 sig.y, 2 h.x, 3 h.y, 4-5 pk.x, 6-7 pk.y; LDS in the kernel) and are
// re-loaded where they are used (once per line); BN_OPAQUE hides the lane offset from the optimiser in every
// iteration so the loads are not hoisted back out of the loop.  Frees 90 of the 256 architectural VGPRs for f, T
// and the temporaries of the current product.
BN_FUNC Fp12 miller_loop_verify_ws(const Ws& inv, const int32_t (*table)[54]) {
  Fp12 f = fp12_one();
  G2J T = {fp2_load_mem(ws_at(inv, 36)), fp2_load_mem(ws_at(inv, 54)), fp2_one()};
  int ti = 0;
  Ws p = inv;
  // per step: the variable line first (T is the register-hungry part), then both lines are scaled and folded into f
  // as one product (ell2): 23 Fp2 products instead of 26
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    BN_OPAQUE(p);
    Fp sx = fp_load_mem(p), sy = fp_load_mem(ws_at(p, 9)), hx = fp_load_mem(ws_at(p, 18)), hy = fp_load_mem(ws_at(p, 27));
    BN_SCHED_BARRIER;                 // the four LDS reads are issued here; the doubling step hides their latency
    Line l = doubling_step(T);
    f = ell2(f, line_from_table(table[ti++]), sx, sy, l, hx, hy);
    int d = ate_naf_digit(j);
    if (d != 0) {
      BN_OPAQUE(p);
      Fp2 qy = fp2_load_mem(ws_at(p, 54));
      Fp2 nqy = fp2_norm(fp2_neg(qy));
      Fp2 qx = fp2_load_mem(ws_at(p, 36));
      BN_OPAQUE(p);
      sx = fp_load_mem(p); sy = fp_load_mem(ws_at(p, 9)); hx = fp_load_mem(ws_at(p, 18)); hy = fp_load_mem(ws_at(p, 27));
      BN_SCHED_BARRIER;
      l = addition_step(T, qx, fp2_select(d > 0, qy, nqy));
      f = ell2(f, line_from_table(table[ti++]), sx, sy, l, hx, hy);
    }
  }
  BN_OPAQUE(p);
  Fp2 g2 = fp2_const(bnc::GAMMA1[1]), g3 = fp2_const(bnc::GAMMA1[2]);
  Fp2 q1x = fp2_mul(fp2_norm(fp2_conj(fp2_load_mem(ws_at(p, 36)))), g2);
  Fp2 q1y = fp2_mul(fp2_norm(fp2_conj(fp2_load_mem(ws_at(p, 54)))), g3);
  Fp2 q2x = fp2_mul(fp2_norm(fp2_conj(q1x)), g2);
  Fp2 q2y = fp2_norm(fp2_neg(fp2_mul(fp2_norm(fp2_conj(q1y)), g3)));
  Line l = addition_step(T, q1x, q1y);
  BN_OPAQUE(p);
  f = ell2(f, line_from_table(table[ti++]), fp_load_mem(p), fp_load_mem(ws_at(p, 9)), l, fp_load_mem(ws_at(p, 18)), fp_load_mem(ws_at(p, 27)));
  l = addition_step(T, q2x, q2y);
  BN_OPAQUE(p);
  f = ell2(f, line_from_table(table[ti++]), fp_load_mem(p), fp_load_mem(ws_at(p, 9)), l, fp_load_mem(ws_at(p, 18)), fp_load_mem(ws_at(p, 27)));
  return f;
}

BN_INL void g2j_store(const Ws& w, const G2J& t) { fp2_store_mem(w, t.x); fp2_store_mem(ws_at(w, 18), t.y); fp2_store_mem(ws_at(w, 36), t.z); }
BN_INL G2J g2j_load(const Ws& w) { return {fp2_load_mem(w), fp2_load_mem(ws_at(w, 18)), fp2_load_mem(ws_at(w, 36))}; }
// The verify loop with the running point T ALSO parked: T sits in `park` (LDS, 54 limbs) except while its line step runs,
// so that neither the squaring of f nor the two-line product has to hold it (the same scheme as the two-pair loop below).
BN_FUNC Fp12 miller_loop_verify_ws2(const Ws& inv, const Ws& park_in, const int32_t (*table)[54]) {
  Fp12 f = fp12_one();
  Ws p = inv, park = park_in;
  fp2_store_mem(park, fp2_load_mem(ws_at(inv, 36))); fp2_store_mem(ws_at(park, 18), fp2_load_mem(ws_at(inv, 54))); fp2_store_mem(ws_at(park, 36), fp2_one());
  BN_MEM_FENCE;
  int ti = 0;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    BN_OPAQUE(p); BN_OPAQUE(park);
    Fp sx = fp_load_mem(p), sy = fp_load_mem(ws_at(p, 9)), hx = fp_load_mem(ws_at(p, 18)), hy = fp_load_mem(ws_at(p, 27));
    G2J T = g2j_load(park);
    BN_SCHED_BARRIER;
    Line l = doubling_step(T);
    g2j_store(park, T);
    BN_MEM_FENCE;
    f = ell2(f, line_from_table(table[ti++]), sx, sy, l, hx, hy);
    int d = ate_naf_digit(j);
    if (d != 0) {
      BN_OPAQUE(p); BN_OPAQUE(park);
      Fp2 qy = fp2_load_mem(ws_at(p, 54));
      Fp2 nqy = fp2_norm(fp2_neg(qy));
      Fp2 qx = fp2_load_mem(ws_at(p, 36));
      sx = fp_load_mem(p); sy = fp_load_mem(ws_at(p, 9)); hx = fp_load_mem(ws_at(p, 18)); hy = fp_load_mem(ws_at(p, 27));
      T = g2j_load(park);
      BN_SCHED_BARRIER;
      l = addition_step(T, qx, fp2_select(d > 0, qy, nqy));
      g2j_store(park, T);
      BN_MEM_FENCE;
      f = ell2(f, line_from_table(table[ti++]), sx, sy, l, hx, hy);
    }
  }
  BN_OPAQUE(p); BN_OPAQUE(park);
  Fp2 g2 = fp2_const(bnc::GAMMA1[1]), g3 = fp2_const(bnc::GAMMA1[2]);
  Fp2 q1x = fp2_mul(fp2_norm(fp2_conj(fp2_load_mem(ws_at(p, 36)))), g2);
  Fp2 q1y = fp2_mul(fp2_norm(fp2_conj(fp2_load_mem(ws_at(p, 54)))), g3);
  Fp2 q2x = fp2_mul(fp2_norm(fp2_conj(q1x)), g2);
  Fp2 q2y = fp2_norm(fp2_neg(fp2_mul(fp2_norm(fp2_conj(q1y)), g3)));
  G2J T = g2j_load(park);
  Line l = addition_step(T, q1x, q1y);
  BN_OPAQUE(p);
  f = ell2(f, line_from_table(table[ti++]), fp_load_mem(p), fp_load_mem(ws_at(p, 9)), l, fp_load_mem(ws_at(p, 18)), fp_load_mem(ws_at(p, 27)));
  l = addition_step(T, q2x, q2y);
  BN_OPAQUE(p);
  f = ell2(f, line_from_table(table[ti++]), fp_load_mem(p), fp_load_mem(ws_at(p, 9)), l, fp_load_mem(ws_at(p, 18)), fp_load_mem(ws_at(p, 27)));
  return f;
}

// The one-pair loop with its invariants and the running point parked (the scheme of the loops above): `inv` (LDS, 54 limbs)
// holds P.x, P.y (9 each) and Q.x, Q.y (18 each), `park` (LDS, 54 limbs) holds T except while its line step runs.  Same
// values as miller_loop_1.
BN_FUNC Fp12 miller_loop_1_ws(const Ws& inv_in, const Ws& park_in) {
  Fp12 f = fp12_one();
  Ws p = inv_in, park = park_in;
  fp2_store_mem(park, fp2_load_mem(ws_at(p, 18))); fp2_store_mem(ws_at(park, 18), fp2_load_mem(ws_at(p, 36))); fp2_store_mem(ws_at(park, 36), fp2_one());
  BN_MEM_FENCE;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    BN_OPAQUE(p); BN_OPAQUE(park);
    Fp px = fp_load_mem(p), py = fp_load_mem(ws_at(p, 9));
    G2J T = g2j_load(park);
    BN_SCHED_BARRIER;
    Line l = doubling_step(T);
    g2j_store(park, T);
    BN_MEM_FENCE;
    f = ell(f, l, px, py);
    int d = ate_naf_digit(j);
    if (d != 0) {
      BN_OPAQUE(p); BN_OPAQUE(park);
      Fp2 qy = fp2_load_mem(ws_at(p, 36));
      Fp2 nqy = fp2_norm(fp2_neg(qy));
      Fp2 qx = fp2_load_mem(ws_at(p, 18));
      px = fp_load_mem(p); py = fp_load_mem(ws_at(p, 9));
      T = g2j_load(park);
      BN_SCHED_BARRIER;
      l = addition_step(T, qx, fp2_select(d > 0, qy, nqy));
      g2j_store(park, T);
      BN_MEM_FENCE;
      f = ell(f, l, px, py);
    }
  }
  BN_OPAQUE(p); BN_OPAQUE(park);
  Fp2 g2 = fp2_const(bnc::GAMMA1[1]), g3 = fp2_const(bnc::GAMMA1[2]);
  Fp2 q1x = fp2_mul(fp2_norm(fp2_conj(fp2_load_mem(ws_at(p, 18)))), g2);
  Fp2 q1y = fp2_mul(fp2_norm(fp2_conj(fp2_load_mem(ws_at(p, 36)))), g3);
  Fp2 q2x = fp2_mul(fp2_norm(fp2_conj(q1x)), g2);
  Fp2 q2y = fp2_norm(fp2_neg(fp2_mul(fp2_norm(fp2_conj(q1y)), g3)));
  G2J T = g2j_load(park);
  Line l = addition_step(T, q1x, q1y);
  BN_OPAQUE(p);
  f = ell(f, l, fp_load_mem(p), fp_load_mem(ws_at(p, 9)));
  l = addition_step(T, q2x, q2y);
  BN_OPAQUE(p);
  f = ell(f, l, fp_load_mem(p), fp_load_mem(ws_at(p, 9)));
  return f;
}

// Two VARIABLE pairs per lane sharing one f^2 per digit (multi_miller_loop, pairings.rs:808-857: "one shared f.square()
// per bit for all terms"): f = ML(Ha, Qa) * ML(Hb, Qb), bit-identical to the product of the two one-pair loops because the
// arithmetic is exact.  Used by aggregate verify (k_miller_hpk2.hip).  Register budget = that of the verify loop: only ONE
// running point T is in registers at a time; the other waits in `park` (LDS, 54 limbs), and the first line of a step waits
// in `lpark` (LDS, 54 limbs) while the second is computed.  `hh` (LDS) holds Ha.x, Ha.y, Hb.x, Hb.y (9 limbs each); `qw`
// (HBM workspace, limb-major) holds Qa.x, Qa.y, Qb.x, Qb.y (18 limbs each), read in the addition steps only.
// live_a / live_b: a pair that is padding (odd count) or failed validation contributes the line 1, i.e. nothing.
BN_INL void line_store(const Ws& w, const Line& l) { fp2_store_mem(w, l.c0); fp2_store_mem(ws_at(w, 18), l.c1); fp2_store_mem(ws_at(w, 36), l.c2); }
BN_INL Line line_load(const Ws& w) { return {fp2_load_mem(w), fp2_load_mem(ws_at(w, 18)), fp2_load_mem(ws_at(w, 36))}; }
BN_INL Line line_mask(bool live, const Line& l) {               // live ? l : the constant line 1
  return {fp2_select(live, l.c0, fp2_one()), fp2_select(live, l.c1, fp2_zero()), fp2_select(live, l.c2, fp2_zero())};
}
// one line step for both pairs: kind 0 doubling, 1 addition with (qx, +-qy) read from qw, 2 / 3 the two Frobenius additions
template <int KIND>
BN_FUNC Fp12 miller2_step(const Fp12& f_in, G2J& T, const Ws& park, const Ws& lpark, const Ws& hh_in, const Ws& qw_in, bool negq, bool live_a, bool live_b) {
  Ws hh = hh_in, qw = qw_in, pk = park, lp = lpark;
  BN_OPAQUE(hh); BN_OPAQUE(qw); BN_OPAQUE(pk); BN_OPAQUE(lp);
  const Fp2 g2 = fp2_const(bnc::GAMMA1[1]), g3 = fp2_const(bnc::GAMMA1[2]);
  auto q_of = [&](int pair, Fp2& qx, Fp2& qy) {
    qx = fp2_load_mem(ws_at(qw, 36 * pair)); qy = fp2_load_mem(ws_at(qw, 36 * pair + 18));
    if (KIND == 1) qy = fp2_select(negq, fp2_norm(fp2_neg(qy)), qy);
    if (KIND >= 2) { qx = fp2_mul(fp2_norm(fp2_conj(qx)), g2); qy = fp2_mul(fp2_norm(fp2_conj(qy)), g3); }                        // pi(Q)
    if (KIND == 3) { qx = fp2_mul(fp2_norm(fp2_conj(qx)), g2); qy = fp2_norm(fp2_neg(fp2_mul(fp2_norm(fp2_conj(qy)), g3))); }     // -pi^2(Q)
  };
  Line la;
  if (KIND == 0) la = doubling_step(T);
  else { Fp2 qx, qy; q_of(0, qx, qy); la = addition_step(T, qx, qy); }
  line_store(lp, fp2_norm_line(line_mask(live_a, la)));
  // swap the running points: Ta -> park, Tb <- park
  G2J Tb = g2j_load(pk);
  BN_MEM_FENCE;
  g2j_store(pk, T);
  BN_MEM_FENCE;
  Line lb;
  if (KIND == 0) lb = doubling_step(Tb);
  else { Fp2 qx, qy; q_of(1, qx, qy); lb = addition_step(Tb, qx, qy); }
  lb = line_mask(live_b, lb);
  T = g2j_load(pk);
  BN_MEM_FENCE;
  g2j_store(pk, Tb);
  BN_MEM_FENCE;
  la = line_load(lp);
  Fp hax = fp_load_mem(hh), hay = fp_load_mem(ws_at(hh, 9)), hbx = fp_load_mem(ws_at(hh, 18)), hby = fp_load_mem(ws_at(hh, 27));
  return ell2(f_in, la, hax, hay, lb, hbx, hby);
}
BN_FUNC Fp12 miller_loop_2var_ws(const Ws& hh, const Ws& qw, const Ws& park, const Ws& lpark, bool live_a, bool live_b) {
  Fp12 f = fp12_one();
  G2J T = {fp2_load_mem(qw), fp2_load_mem(ws_at(qw, 18)), fp2_one()};                       // Ta = Qa
  g2j_store(park, G2J{fp2_load_mem(ws_at(qw, 36)), fp2_load_mem(ws_at(qw, 54)), fp2_one()});   // Tb = Qb
  BN_MEM_FENCE;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    f = miller2_step<0>(f, T, park, lpark, hh, qw, false, live_a, live_b);
    int d = ate_naf_digit(j);
    if (d != 0) f = miller2_step<1>(f, T, park, lpark, hh, qw, d < 0, live_a, live_b);
  }
  f = miller2_step<2>(f, T, park, lpark, hh, qw, false, live_a, live_b);
  f = miller2_step<3>(f, T, park, lpark, hh, qw, false, live_a, live_b);
  return f;
}

// f^x for the BN parameter x (63 bits, x > 0), f in the cyclotomic subgroup
BN_FUNC Fp12 cyclotomic_exp_x(const Fp12& f) {
  BN_CTX;
  Fp12 r = f;
  for (int i = 61; i >= 0; --i) {
    r = fp12_cyclotomic_sqr(r);
    if ((bnc::BN_X >> i) & 1) r = fp12_mul(r, f);
  }
  return r;
}
// Final exponentiation f -> f^((p^12-1)/r * 2x(6x^2+3x+1)), split into phases so that each phase can be
// its own register-resident kernel (k_finalexp.hip); final_exponentiation() is their composition.
//   easy:  t  = f^((p^6-1)(p^2+1))
//   hard:  t^(l0 + l1 p + l2 p^2 + l3 p^3), Fuentes-Castaneda et al. arrangement with three t -> t^x
//          exponentiations (cyclotomic_exp_x) separated by the small steps h1, h2, h3.
// f^x by the signed chain BN_X_CHAIN (curve.h) = 62 cyclotomic squarings + 13 multiplications (binary: 27): f^-17 and f^-35 are
// the conjugates of f^17 and f^35.  Run as a small uniform interpreter (one inlined squaring and one inlined
// multiply-by-memory-operand in the loop body): the five named powers live in `slots` (limb-major memory, 10 x 108 limbs per
// lane reserved) and are read back one Fp6 half at a time.
BN_FUNC Fp12 cyclotomic_exp_x_chain(const Fp12& f, const Ws& slots, const Ws* park = nullptr) {
  const ChainOp prog[BN_X_CHAIN_LEN] = BN_X_CHAIN;
  fp12_store_mem(slots, f);
  BN_MEM_FENCE;
  Fp12 r = f;
  for (int k = 0; k < BN_X_CHAIN_LEN; ++k) {
    const ChainOp op = prog[k];
    if (op.load >= 0) r = fp12_load_mem(ws_at(slots, 108 * op.load));
    for (int q = 0; q < op.sq; ++q) r = fp12_cyclotomic_sqr(r);
    if (op.mul >= 0) r = fp12_mul_mem(r, ws_at(slots, 108 * op.mul), park);
    if (op.store >= 0) fp12_store_mem(ws_at(slots, 108 * op.store), r);
    if (op.cstore >= 0) { fp6_store_mem(ws_at(slots, 108 * op.cstore), r.c0); fp6_store_mem(ws_at(slots, 108 * op.cstore + 54), fp6_norm(fp6_neg(r.c1))); }
    if (op.store >= 0 || op.cstore >= 0) BN_MEM_FENCE;
  }
  return r;
}
BN_FUNC Fp12 fe_easy(const Fp12& f) {
  Fp12 t = fp12_mul(fp12_conj(f), fp12_inv(f));
  return fp12_mul(fp12_frob<2>(t), t);
}
// The easy part with its ONE Fp inversion taken out (k_fe_easy.hip, three launches): the inversion is ~60 % of fe_easy's
// instructions, and with one tuple per lane every lane pays it.  Split at the inversion, the middle launch inverts FOUR
// tuples' norms per lane by Montgomery's trick (3 + 6 products around one a^(p-2)), a quarter of the lanes.
//   fe_easy_head:  everything of fp12_inv (fp12.rs:212-219 -> fp6.rs:261-287 -> fp2.rs:161-166) before the inversion:
//                  the three Fp2 cofactors c0, c1, c2 of the Fp6 inverse, the Fp2 norm N, and nu = N.c0^2 + N.c1^2 in Fp
//   fp_inv4:       nu_i^-1 for four values with one inversion (a zero stays zero: inv0, E15)
//   fe_easy_tail:  the rest: f^-1 from the parked pieces and nu^-1, then t = conj(f) f^-1, t^(p^2) t
// Same field values as fe_easy, hence the same canonical limbs.
struct FeEasyHead { Fp2 c0, c1, c2, nrm; Fp nu; };
BN_FUNC FeEasyHead fe_easy_head(const Fp12& a) {
  BN_CTX;
  Fp6 s0 = fp6_sqr(a.c0), s1 = fp6_sqr(a.c1);
  Fp6 d = {fp2_sub_mul_xi(s0.c0, s1.c2), fp2_norm(fp2_sub(s0.c1, s1.c0)), fp2_norm(fp2_sub(s0.c2, s1.c1))};
  Fp2 q0 = fp2_sqr(d.c0), q1 = fp2_sqr(d.c1), q2 = fp2_sqr(d.c2);
  Fp2 m01 = fp2_mul(d.c0, d.c1), m02 = fp2_mul(d.c0, d.c2), m12 = fp2_mul(d.c1, d.c2);
  FeEasyHead h;
  h.c0 = fp2_sub_mul_xi(q0, m12);
  h.c1 = fp2_norm(fp2_sub(fp2_mul_xi(q2), m01));
  h.c2 = fp2_norm(fp2_sub(q1, m02));
  Fp2 u = fp2_add(fp2_mul(d.c2, h.c1), fp2_mul(d.c1, h.c2));
  h.nrm = fp2_add_mul_xi(fp2_mul(d.c0, h.c0), u);
  h.nu = fp_dot2(h.nrm.c0, h.nrm.c0, h.nrm.c1, h.nrm.c1);
  return h;
}
BN_FUNC Fp12 fe_easy_tail(const Fp12& a, const FeEasyHead& h, const Fp& nu_inv) {
  BN_CTX;
  Fp2 ti = {fp_mul(h.nrm.c0, nu_inv), fp_mul(fp_neg(h.nrm.c1), nu_inv)};      // N^-1 = conj(N) / nu
  Fp6 t6 = {fp2_mul(h.c0, ti), fp2_mul(h.c1, ti), fp2_mul(h.c2, ti)};         // d^-1
  Fp12 inv = {fp6_mul(a.c0, t6), fp6_norm(fp6_neg(fp6_mul(a.c1, t6)))};
  Fp12 t = fp12_mul(fp12_conj(a), inv);
  return fp12_mul(fp12_frob<2>(t), t);
}
// x_i^-1, i < 4, with one inversion; x_i must be multiplication outputs (or normalised)
BN_FUNC void fp_inv4(Fp x[4]) {
  BN_CTX;
  bool z[4];
  Fp v[4], pre[4];
  for (int i = 0; i < 4; ++i) { z[i] = fp_is_zero(x[i]); v[i] = fp_select(z[i], fp_one(), fp_norm(x[i])); }
  pre[0] = v[0];
  for (int i = 1; i < 4; ++i) pre[i] = fp_mul(pre[i - 1], v[i]);
  Fp inv = fp_inv(pre[3]);
  for (int i = 3; i >= 1; --i) {
    Fp r = fp_mul(inv, pre[i - 1]);
    inv = fp_mul(inv, v[i]);
    x[i] = fp_select(z[i], fp_zero(), r);
  }
  x[0] = fp_select(z[0], fp_zero(), inv);
}
// x0 = t^x  ->  a = t^-2x, b = t^-6x
BN_FUNC void fe_h1(const Fp12& x0, Fp12& a, Fp12& b) {
  a = fp12_cyclotomic_sqr(fp12_conj(x0));
  Fp12 a2 = fp12_cyclotomic_sqr(a);                        // t^-4x
  b = fp12_mul(a, a2);
}
// x0 = b^x = t^(-6x^2)  ->  c = t^(6x^2), b2 = t^(6x^2+6x), d2 = t^(12x^2)
BN_FUNC void fe_h2(const Fp12& x0, const Fp12& b, Fp12& c, Fp12& b2, Fp12& d2) {
  c = fp12_conj(x0);
  b2 = fp12_mul(c, fp12_conj(b));
  d2 = fp12_cyclotomic_sqr(c);
}
// x0 = d2^x = t^(12x^3)  ->  result
BN_FUNC Fp12 fe_h3(const Fp12& t, const Fp12& a, const Fp12& c, const Fp12& b2, const Fp12& x0) {
  Fp12 e = fp12_mul(b2, x0);                               // l2 = 12x^3+6x^2+6x
  Fp12 d = fp12_mul(a, e);                                 // l1 = 12x^3+6x^2+4x
  Fp12 l0 = fp12_mul(t, fp12_mul(c, e));                   // l0 = 12x^3+12x^2+6x+1
  Fp12 r = fp12_mul(l0, fp12_frob<1>(d));
  r = fp12_mul(r, fp12_frob<2>(e));
  Fp12 l3 = fp12_mul(fp12_conj(t), d);                     // l3 = 12x^3+6x^2+4x-1
  return fp12_mul(r, fp12_frob<3>(l3));
}
BN_FUNC Fp12 final_exponentiation(const Fp12& f) {
  Fp12 t = fe_easy(f), a, b, c, b2, d2;
  fe_h1(cyclotomic_exp_x(t), a, b);
  fe_h2(cyclotomic_exp_x(b), b, c, b2, d2);
  return fe_h3(t, a, c, b2, cyclotomic_exp_x(d2));
}

// Gt byte layout (Gt::to_repr / from_repr, pairings.rs:499-579): c0.c0.c0, c0.c0.c1, c0.c1.c0, ...
// The six Fp2 components are named one by one below, never reached through an array of pointers: an index
// the compiler cannot resolve would force the whole Fp12 -- the accumulator of the caller's main loop -- to
// live in scratch memory instead of registers.
BN_INL void fp2_to_be(uint8_t* out, const Fp2& c) { fp_to_be(out, c.c0); fp_to_be(out + 32, c.c1); }
BN_FUNC void fp12_to_be(uint8_t* out, const Fp12& a) {
  BN_CTX;
  fp2_to_be(out, a.c0.c0); fp2_to_be(out + 64, a.c0.c1); fp2_to_be(out + 128, a.c0.c2);
  fp2_to_be(out + 192, a.c1.c0); fp2_to_be(out + 256, a.c1.c1); fp2_to_be(out + 320, a.c1.c2);
}
BN_INL Fp2 fp2_from_be(const uint8_t* in, bool& ok) {
  bool o0, o1;
  Fp2 c = {fp_from_be(in, o0), fp_from_be(in + 32, o1)};
  ok &= o0 & o1;
  return c;
}
BN_FUNC Fp12 fp12_from_be(const uint8_t* in, bool& ok) {
  BN_CTX;
  ok = true;
  Fp12 a;
  a.c0.c0 = fp2_from_be(in, ok); a.c0.c1 = fp2_from_be(in + 64, ok); a.c0.c2 = fp2_from_be(in + 128, ok);
  a.c1.c0 = fp2_from_be(in + 192, ok); a.c1.c1 = fp2_from_be(in + 256, ok); a.c1.c2 = fp2_from_be(in + 320, ok);
  return a;
}
// Device-side workspace form of an Fp12: 108 strict limbs (canonical Montgomery), c0.c0.c0 first.
BN_INL void fp2_store_limbs(int32_t* out, size_t stride, const Fp2& c) {
  Fp c0 = fp_canon(c.c0), c1 = fp_canon(c.c1);
  BN_UNROLL for (int k = 0; k < NL; ++k) { out[(size_t)k * stride] = c0.l[k]; out[(size_t)(9 + k) * stride] = c1.l[k]; }
}
// the same through a workspace reference (buffer addressing on the device)
BN_INL void fp2_store_limbs(const Ws& w, const Fp2& c) {
  Fp c0 = fp_canon(c.c0), c1 = fp_canon(c.c1);
  BN_UNROLL for (int k = 0; k < NL; ++k) { ws_store(w, k, c0.l[k]); ws_store(w, 9 + k, c1.l[k]); }
}
BN_FUNC void fp12_store_limbs(const Ws& w, const Fp12& a) {
  BN_CTX;
  fp2_store_limbs(w, a.c0.c0); fp2_store_limbs(ws_at(w, 18), a.c0.c1); fp2_store_limbs(ws_at(w, 36), a.c0.c2);
  fp2_store_limbs(ws_at(w, 54), a.c1.c0); fp2_store_limbs(ws_at(w, 72), a.c1.c1); fp2_store_limbs(ws_at(w, 90), a.c1.c2);
}
BN_INL Fp2 fp2_load_limbs(const Ws& w) {
  Fp2 c;
  BN_UNROLL for (int k = 0; k < NL; ++k) { c.c0.l[k] = ws_load(w, k); c.c1.l[k] = ws_load(w, 9 + k); }
  BN_TRK(set_trk(c.c0, 0, 1, 0, 0.006, 1); set_trk(c.c1, 0, 1, 0, 0.006, 1);)
  return c;
}
BN_FUNC Fp12 fp12_load_limbs(const Ws& w) {
  BN_CTX;
  return {{fp2_load_limbs(w), fp2_load_limbs(ws_at(w, 18)), fp2_load_limbs(ws_at(w, 36))},
          {fp2_load_limbs(ws_at(w, 54)), fp2_load_limbs(ws_at(w, 72)), fp2_load_limbs(ws_at(w, 90))}};
}
BN_FUNC void fp12_store_limbs(int32_t* out, size_t stride, const Fp12& a) {
  BN_CTX;
  fp2_store_limbs(out, stride, a.c0.c0); fp2_store_limbs(out + 18 * stride, stride, a.c0.c1);
  fp2_store_limbs(out + 36 * stride, stride, a.c0.c2); fp2_store_limbs(out + 54 * stride, stride, a.c1.c0);
  fp2_store_limbs(out + 72 * stride, stride, a.c1.c1); fp2_store_limbs(out + 90 * stride, stride, a.c1.c2);
}
BN_INL Fp2 fp2_load_limbs(const int32_t* in, size_t stride) {
  Fp2 c;
  BN_UNROLL for (int k = 0; k < NL; ++k) { c.c0.l[k] = in[(size_t)k * stride]; c.c1.l[k] = in[(size_t)(9 + k) * stride]; }
  BN_TRK(set_trk(c.c0, 0, 1, 0, 0.006, 1); set_trk(c.c1, 0, 1, 0, 0.006, 1);)
  return c;
}
BN_FUNC Fp12 fp12_load_limbs(const int32_t* in, size_t stride) {
  BN_CTX;
  return {{fp2_load_limbs(in, stride), fp2_load_limbs(in + 18 * stride, stride), fp2_load_limbs(in + 36 * stride, stride)},
          {fp2_load_limbs(in + 54 * stride, stride), fp2_load_limbs(in + 72 * stride, stride), fp2_load_limbs(in + 90 * stride, stride)}};
}

// fe_h3 as a small uniform interpreter for the phase kernel (k_fe_h3.hip): ONE inlined multiply-by-memory-operand
// in the loop body instead of eight inlined products, the running value r in registers, every other factor read from a
// limb-major workspace by fp12_mul_mem.  src[0..4] = t, a, c, b2, x0 (canonical limbs, as the phase kernels store
// them); tmp = 4 x 108 limbs for e, e^(p^2), d^p, l3^(p^3).  Same value as fe_h3:
//   e = b2 x0, d = a e, l3 = conj(t) d = conj(t conj(d)), result = (c e t) d^p e^(p^2) l3^(p^3).
struct H3Op { int8_t pre, mul, post; };      // pre: 1 r = b2, 2 r = conj(r), 3 r = e;  mul: operand index (5.. = tmp slots);  post: see below
BN_FUNC Fp12 fe_h3_loop(const Ws* src, const Ws& tmp, const Ws* park) {
  const H3Op prog[8] = {{1, 4, 1}, {0, 1, 2}, {2, 0, 3}, {3, 2, 0}, {0, 0, 0}, {0, 7, 0}, {0, 6, 0}, {0, 8, 0}};
  Fp12 r = fp12_one();
  for (int k = 0; k < 8; ++k) {
    const H3Op op = prog[k];
    if (op.pre == 1) r = fp12_load_mem(src[3]);
    else if (op.pre == 2) r = fp12_conj(r);
    else if (op.pre == 3) r = fp12_load_mem(tmp);
    const int m = op.mul;
    const Ws b = m == 0 ? src[0] : m == 1 ? src[1] : m == 2 ? src[2] : m == 4 ? src[4] : ws_at(tmp, 108 * (size_t)(m - 5));
    r = fp12_mul_mem(r, b, park);
    if (op.post == 1) { fp12_store_mem(tmp, r); fp12_store_mem(ws_at(tmp, 108), fp12_frob<2>(r)); }      // e, e^(p^2)
    else if (op.post == 2) fp12_store_mem(ws_at(tmp, 216), fp12_frob<1>(r));                              // d^p
    else if (op.post == 3) fp12_store_mem(ws_at(tmp, 324), fp12_frob<3>(fp12_conj(r)));                   // l3^(p^3)
    BN_MEM_FENCE;
  }
  return r;
}

// fe_h1 / fe_h2 as tails of the t^x phase kernels (k_fe_expx_tail.hip): computed right after the chain from the value
// still in registers, results written to the limb-major phase buffers (normalised limbs).  Same values as fe_h1 / fe_h2.
BN_FUNC void fe_h1_tail(const Fp12& x0, const Ws& a_out, const Ws& b_out, const Ws* park) {
  Fp12 a = fp12_cyclotomic_sqr(fp12_conj(x0));
  fp12_store_mem(a_out, a);
  BN_MEM_FENCE;
  fp12_store_mem(b_out, fp12_mul_mem(fp12_cyclotomic_sqr(a), a_out, park));            // b = a^2 * a
}
BN_FUNC void fe_h2_tail(const Fp12& x0, const Ws& b_in, const Ws& c_out, const Ws& b2_out, const Ws& d2_out, const Ws* park) {
  fp12_store_mem(c_out, fp12_conj(x0));                                                   // c
  BN_MEM_FENCE;
  fp12_store_mem(b2_out, fp12_conj(fp12_mul_mem(x0, b_in, park)));                        // b2 = c conj(b) = conj(x0 b)
  BN_MEM_FENCE;
  fp12_store_mem(d2_out, fp12_cyclotomic_sqr(fp12_load_mem(c_out)));                      // d2 = c^2
}

// G2Prepared (pairings.rs:609-660, :726-757; E6: the loop pushes 88 coefficients, the reference's array holds 68): the 88
// line coefficient triples of the optimal ate loop for ONE public key, in evaluation order, as canonical Montgomery limbs
// (54 per line).  g2_prepare_lines writes them; miller_loop_prepared consumes them next to the fixed -G2gen table, so a
// verify for a known key runs no point arithmetic at all: f <- f^2 * l_sig * l_key per step.
BN_INL void line_store_limbs(const Ws& w, const Line& l) { fp2_store_limbs(w, l.c0); fp2_store_limbs(ws_at(w, 18), l.c1); fp2_store_limbs(ws_at(w, 36), l.c2); }
BN_INL Line line_load_limbs(const Ws& w) { return {fp2_load_limbs(w), fp2_load_limbs(ws_at(w, 18)), fp2_load_limbs(ws_at(w, 36))}; }
BN_FUNC void g2_prepare_lines(const G2A& q, const Ws& out) {
  BN_CTX;
  G2J T = {q.x, q.y, fp2_one()};
  Fp2 nqy = fp2_norm(fp2_neg(q.y));
  int ti = 0;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    line_store_limbs(ws_at(out, 54 * (size_t)ti++), doubling_step(T));
    int d = ate_naf_digit(j);
    if (d != 0) line_store_limbs(ws_at(out, 54 * (size_t)ti++), addition_step(T, q.x, d > 0 ? q.y : nqy));
  }
  Fp2 g2 = fp2_const(bnc::GAMMA1[1]), g3 = fp2_const(bnc::GAMMA1[2]);
  Fp2 q1x = fp2_mul(fp2_norm(fp2_conj(q.x)), g2), q1y = fp2_mul(fp2_norm(fp2_conj(q.y)), g3);
  Fp2 q2x = fp2_mul(fp2_norm(fp2_conj(q1x)), g2);
  Fp2 q2y = fp2_norm(fp2_neg(fp2_mul(fp2_norm(fp2_conj(q1y)), g3)));
  line_store_limbs(ws_at(out, 54 * (size_t)ti++), addition_step(T, q1x, q1y));
  line_store_limbs(ws_at(out, 54 * (size_t)ti++), addition_step(T, q2x, q2y));
}
// Both pairs from tables, one step further: the first pair's table (-G2gen) is the same for every tuple, so the PRODUCT of
// the two lines of a step can be prepared per key as well.  With la = a0 ys + a3 xs w + a4 w^3 (a: -G2gen line, (xs, ys) the
// signature) and lb = b0 yh + b3 xh w + b4 w^3 (b: the key's line, (xh, yh) = H(msg)):
//   la lb = (a0b0 ys yh + xi a4b4) + a3b3 xs xh v + (a3b4 xs + a4b3 xh) v^2 + [(a0b3 ys xh + a3b0 xs yh) + (a0b4 ys + a4b0 yh) v] w
// The nine Fp2 products a_i b_j depend on the key only (k_g2_expand, 162 limbs per line: T0 = a0b0, T1 = xi a4b4, T2 = a3b3,
// T3 = a3b4, T4 = a4b3, T5 = a0b3, T6 = a3b0, T7 = a0b4, T8 = a4b0); per tuple four products of coordinates are formed once.
// A step then spends 1 scaling + 4 double products (2268 MADs) on the line pair instead of 4 scalings + 6 Fp2 products
// (4212), and the 17-product sparse multiplication of f is unchanged.  The loop's value differs from the textbook product of
// the two Miller values only by powers of Z (an element of Fp), which the final exponentiation removes: same bitmap.
// Entries of the pair table are stored as the products come out of the multiplier (limbs 0..7 in [0, 2^29], a small signed top
// limb, value within (1 + eps) p) -- NOT canonical: canonicalising 18 field elements per (key, step) was 45 % of k_g2_expand,
// and the only readers are the double products of the line evaluation below (and of tri.h / wide.h), which take lazy operands.
BN_INL void fp_store_limbs_lazy(const Ws& w, const Fp& c) {
  BN_TRK(if (c.lo < -4e-6 || c.hi > 1.0 + 4e-6 || c.tlo < -0.02 || c.thi > 0.02 || c.vb > 1.2) check_fail("pair table entry outside its declared range", c.vb);)
  BN_UNROLL for (int k = 0; k < NL; ++k) ws_store(w, k, c.l[k]);
}
BN_INL Fp fp_load_limbs_lazy(const Ws& w) {
  Fp c;
  BN_UNROLL for (int k = 0; k < NL; ++k) c.l[k] = ws_load(w, k);
  BN_TRK(set_trk(c, -4e-6, 1.0 + 4e-6, -0.02, 0.02, 1.2); check_actual(c, "pair table entry");)
  return c;
}
BN_INL void fp2_store_limbs_lazy(const Ws& w, const Fp2& c) { fp_store_limbs_lazy(w, c.c0); fp_store_limbs_lazy(ws_at(w, 9), c.c1); }
BN_INL Fp2 fp2_load_limbs_lazy(const Ws& w) {                    // both components through ONE reference, like fp2_load_limbs (a moved base is another buffer descriptor)
  Fp2 c;
  BN_UNROLL for (int k = 0; k < NL; ++k) { c.c0.l[k] = ws_load(w, k); c.c1.l[k] = ws_load(w, 9 + k); }
  BN_TRK(set_trk(c.c0, -4e-6, 1.0 + 4e-6, -0.02, 0.02, 1.2); set_trk(c.c1, -4e-6, 1.0 + 4e-6, -0.02, 0.02, 1.2);
         check_actual(c.c0, "pair table entry"); check_actual(c.c1, "pair table entry");)
  return c;
}
BN_FUNC void line_pair_expand(const Line& a, const Line& b, const Ws& out) {
  BN_CTX;
  fp2_store_limbs_lazy(out, fp2_mul(a.c0, b.c0));
  fp2_store_limbs_lazy(ws_at(out, 18), fp2_mul_xi(fp2_mul(a.c2, b.c2)));
  fp2_store_limbs_lazy(ws_at(out, 36), fp2_mul(a.c1, b.c1));
  fp2_store_limbs_lazy(ws_at(out, 54), fp2_mul(a.c1, b.c2));
  fp2_store_limbs_lazy(ws_at(out, 72), fp2_mul(a.c2, b.c1));
  fp2_store_limbs_lazy(ws_at(out, 90), fp2_mul(a.c0, b.c1));
  fp2_store_limbs_lazy(ws_at(out, 108), fp2_mul(a.c1, b.c0));
  fp2_store_limbs_lazy(ws_at(out, 126), fp2_mul(a.c0, b.c2));
  fp2_store_limbs_lazy(ws_at(out, 144), fp2_mul(a.c2, b.c0));
}
BN_INL Fp2 fp2_dot_fp(const Fp2& t, const Fp& s, const Fp2& u, const Fp& r) {      // t s + u r, one reduction per component
  return {fp_dot2(t.c0, s, u.c0, r), fp_dot2(t.c1, s, u.c1, r)};
}
// f * (la lb Z) from the expanded pair `e` (this step's 162 limbs) and the tuple's coordinates in `cw` (LDS, 9 limbs each):
// the signature (xs, ys) affine, H(msg) = (X : Y : Z) homogeneous as the hash kernel leaves it (no inversion), and their products
//   cw: 0 X, 1 Y, 2 Z, 3 xs X, 4 ys Y, 5 xs Z, 6 ys Z, 7 ys X, 8 xs Y
// la lb Z = (T0 ys Y + T1 Z) + T2 xs X v + (T3 xs Z + T4 X) v^2 + [(T5 ys X + T6 xs Y) + (T7 ys Z + T8 Y) v] w
BN_FUNC Fp12 ell_pair_expanded(const Fp12& f, const Ws& e, const Ws& cw) {
  BN_CTX;
  Fp X = fp_load_mem(cw), Y = fp_load_mem(ws_at(cw, 9)), Z = fp_load_mem(ws_at(cw, 18));
  Fp xsX = fp_load_mem(ws_at(cw, 27)), ysY = fp_load_mem(ws_at(cw, 36)), xsZ = fp_load_mem(ws_at(cw, 45));
  Fp ysZ = fp_load_mem(ws_at(cw, 54)), ysX = fp_load_mem(ws_at(cw, 63)), xsY = fp_load_mem(ws_at(cw, 72));
  Fp6 l0 = {fp2_dot_fp(fp2_load_limbs_lazy(e), ysY, fp2_load_limbs_lazy(ws_at(e, 18)), Z),
            fp2_mul_fp(fp2_load_limbs_lazy(ws_at(e, 36)), xsX),
            fp2_dot_fp(fp2_load_limbs_lazy(ws_at(e, 54)), xsZ, fp2_load_limbs_lazy(ws_at(e, 72)), X)};
  Fp2 l10 = fp2_dot_fp(fp2_load_limbs_lazy(ws_at(e, 90)), ysX, fp2_load_limbs_lazy(ws_at(e, 108)), xsY);
  Fp2 l11 = fp2_dot_fp(fp2_load_limbs_lazy(ws_at(e, 126)), ysZ, fp2_load_limbs_lazy(ws_at(e, 144)), Y);
  Fp6 v0 = fp6_mul(f.c0, l0);
  Fp6 v1 = fp6_mul_by_01(f.c1, l10, l11);
  Fp6 dl = {fp2_norm(fp2_sub(l10, l0.c0)), fp2_norm(fp2_sub(l11, l0.c1)), fp2_norm(fp2_neg(l0.c2))};       // l1 - l0
  Fp6 w = fp6_mul(fp6_norm(fp6_sub(f.c0, f.c1)), dl);
  return {fp6_add_mul_v(v0, v1), fp6_norm(fp6_add(fp6_add(w, v0), v1))};
}
// inv (LDS, 81 limbs): the nine coordinate values listed above; ktab: this lane's key, 88 x 162 limbs
BN_FUNC Fp12 miller_loop_prepared(const Ws& inv, const Ws& ktab_in) {
  Fp12 f = fp12_one();
  Ws p = inv, kt = ktab_in;
  int ti = 0;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    BN_OPAQUE(kt); BN_OPAQUE(p);
    f = ell_pair_expanded(f, ws_at(kt, 162 * (size_t)ti), p);
    ++ti;
    if (ate_naf_digit(j) != 0) {
      BN_OPAQUE(kt); BN_OPAQUE(p);
      f = ell_pair_expanded(f, ws_at(kt, 162 * (size_t)ti), p);
      ++ti;
    }
  }
  for (int e = 0; e < 2; ++e) {
    BN_OPAQUE(kt); BN_OPAQUE(p);
    f = ell_pair_expanded(f, ws_at(kt, 162 * (size_t)ti), p);
    ++ti;
  }
  return f;
}
// The same loop when EVERY lane of the wave reads the same key's table (key-sorted order: the usual case): the entries are read
// through a constant-address-space pointer with no per-lane part, i.e. by SCALAR loads into scalar registers -- no vector
// registers held for them, the fetch runs ahead of the arithmetic -- and the double products take them as scalar operands.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) int32_t* bn_const_i32p;
#else
typedef const int32_t* bn_const_i32p;
#endif
BN_INL Fp2 fp2_load_limbs_const(bn_const_i32p e) {
  Fp2 c;
  BN_UNROLL for (int k = 0; k < NL; ++k) { c.c0.l[k] = e[k]; c.c1.l[k] = e[9 + k]; }
  BN_TRK(set_trk(c.c0, -4e-6, 1.0 + 4e-6, -0.02, 0.02, 1.2); set_trk(c.c1, -4e-6, 1.0 + 4e-6, -0.02, 0.02, 1.2);
         check_actual(c.c0, "pair table entry"); check_actual(c.c1, "pair table entry");)
  return c;
}
BN_FUNC Fp12 ell_pair_expanded_uniform(const Fp12& f, bn_const_i32p e, const Ws& cw) {
  Fp X = fp_load_mem(cw), Y = fp_load_mem(ws_at(cw, 9)), Z = fp_load_mem(ws_at(cw, 18));
  Fp xsX = fp_load_mem(ws_at(cw, 27)), ysY = fp_load_mem(ws_at(cw, 36)), xsZ = fp_load_mem(ws_at(cw, 45));
  Fp ysZ = fp_load_mem(ws_at(cw, 54)), ysX = fp_load_mem(ws_at(cw, 63)), xsY = fp_load_mem(ws_at(cw, 72));
  Fp6 l0 = {fp2_dot_fp(fp2_load_limbs_const(e), ysY, fp2_load_limbs_const(e + 18), Z),
            fp2_mul_fp(fp2_load_limbs_const(e + 36), xsX),
            fp2_dot_fp(fp2_load_limbs_const(e + 54), xsZ, fp2_load_limbs_const(e + 72), X)};
  Fp2 l10 = fp2_dot_fp(fp2_load_limbs_const(e + 90), ysX, fp2_load_limbs_const(e + 108), xsY);
  Fp2 l11 = fp2_dot_fp(fp2_load_limbs_const(e + 126), ysZ, fp2_load_limbs_const(e + 144), Y);
  Fp6 v0 = fp6_mul(f.c0, l0);
  Fp6 v1 = fp6_mul_by_01(f.c1, l10, l11);
  Fp6 dl = {fp2_norm(fp2_sub(l10, l0.c0)), fp2_norm(fp2_sub(l11, l0.c1)), fp2_norm(fp2_neg(l0.c2))};       // l1 - l0
  Fp6 w = fp6_mul(fp6_norm(fp6_sub(f.c0, f.c1)), dl);
  return {fp6_add_mul_v(v0, v1), fp6_norm(fp6_add(fp6_add(w, v0), v1))};
}
BN_FUNC Fp12 miller_loop_prepared_uniform(const Ws& inv, const int32_t* ktab_uniform) {
  Fp12 f = fp12_one();
  Ws p = inv;
  bn_const_i32p kt = (bn_const_i32p)ktab_uniform;
  int ti = 0;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    BN_OPAQUE(p);
    f = ell_pair_expanded_uniform(f, kt + 162 * ti, p);
    ++ti;
    if (ate_naf_digit(j) != 0) {
      BN_OPAQUE(p);
      f = ell_pair_expanded_uniform(f, kt + 162 * ti, p);
      ++ti;
    }
  }
  for (int e = 0; e < 2; ++e) {
    BN_OPAQUE(p);
    f = ell_pair_expanded_uniform(f, kt + 162 * ti, p);
    ++ti;
  }
  return f;
}

// One pair per lane with a prepared key: f = ML(H, Q), the line triples of Q read from its raw table (88 x 54 limbs).  More work
// per pair than the two-pairs-per-lane loop below (f^2 is not shared) but the shortest chain per lane: used when a launch has so
// few pairs that it is bound by the latency of one wave (aggregate verify over a handful of distinct keys).  hh (LDS): H.x, H.y.
BN_FUNC Fp12 miller_loop_1prepared(const Ws& hh_in, const Ws& ta_in) {
  Fp12 f = fp12_one();
  Ws hh = hh_in, ta = ta_in;
  int ti = 0;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    BN_OPAQUE(ta); BN_OPAQUE(hh);
    f = ell(f, line_load_limbs(ws_at(ta, 54 * (size_t)ti)), fp_load_mem(hh), fp_load_mem(ws_at(hh, 9)));
    ++ti;
    if (ate_naf_digit(j) != 0) {
      BN_OPAQUE(ta); BN_OPAQUE(hh);
      f = ell(f, line_load_limbs(ws_at(ta, 54 * (size_t)ti)), fp_load_mem(hh), fp_load_mem(ws_at(hh, 9)));
      ++ti;
    }
  }
  for (int e = 0; e < 2; ++e) {
    BN_OPAQUE(ta); BN_OPAQUE(hh);
    f = ell(f, line_load_limbs(ws_at(ta, 54 * (size_t)ti)), fp_load_mem(hh), fp_load_mem(ws_at(hh, 9)));
    ++ti;
  }
  return f;
}

// Two pairs per lane, BOTH with prepared keys (aggregate verify over a batch that repeats few public keys): f = ML(Ha, Qa) *
// ML(Hb, Qb) with the line triples of Qa and Qb read from their keys' raw tables (88 x 54 limbs each, k_g2_prepare) -- one
// squaring of f and one two-line product per loop digit, no point arithmetic.  hh (LDS): Ha.x, Ha.y, Hb.x, Hb.y.
// A dead half (padding of an odd count) contributes the constant line 1 evaluated at y = 1.
BN_FUNC Fp12 ell2_from_tables(const Fp12& f, const Ws& ta_in, const Ws& tb_in, int ti, const Ws& hh_in, bool live_a, bool live_b) {
  Ws hh = hh_in, ta = ta_in, tb = tb_in;
  BN_OPAQUE(ta); BN_OPAQUE(tb); BN_OPAQUE(hh);
  Line la = line_mask(live_a, line_load_limbs(ws_at(ta, 54 * (size_t)ti)));
  Line lb = line_mask(live_b, line_load_limbs(ws_at(tb, 54 * (size_t)ti)));
  Fp hax = fp_load_mem(hh), hay = fp_load_mem(ws_at(hh, 9)), hbx = fp_load_mem(ws_at(hh, 18)), hby = fp_load_mem(ws_at(hh, 27));
  return ell2(f, la, hax, hay, lb, hbx, hby);
}
BN_FUNC Fp12 miller_loop_2prepared(const Ws& hh, const Ws& ta, const Ws& tb, bool live_a, bool live_b) {
  Fp12 f = fp12_one();
  int ti = 0;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    f = ell2_from_tables(f, ta, tb, ti++, hh, live_a, live_b);
    if (ate_naf_digit(j) != 0) f = ell2_from_tables(f, ta, tb, ti++, hh, live_a, live_b);
  }
  f = ell2_from_tables(f, ta, tb, ti++, hh, live_a, live_b);
  f = ell2_from_tables(f, ta, tb, ti++, hh, live_a, live_b);
  return f;
}

}  // namespace bn
