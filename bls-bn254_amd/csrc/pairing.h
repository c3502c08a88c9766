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
BN_INL Line line_from_table(const int32_t* t) {                  // 54 strict limbs from the generated table
  return {fp2_from_limbs(t), fp2_from_limbs(t + 18), fp2_from_limbs(t + 36)};
}

// One-pair Miller loop f_{6x+2,Q}(P) * l_{T,pi(Q)}(P) * l_{T+pi(Q),-pi^2(Q)}(P)
BN_FUNC Fp12 miller_loop_1(const G1A& p, const G2A& q, const int8_t* naf, int naf_len) {
  BN_CTX;
  Fp12 f = fp12_one();
  G2J T = {q.x, q.y, fp2_one()};
  Fp2 nqy = fp2_norm(fp2_neg(q.y));
  for (int j = naf_len - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    f = ell(f, doubling_step(T), p.x, p.y);
    int d = naf[j];
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
BN_FUNC Fp12 miller_loop_verify(const G1A& sig, const G1A& h, const G2A& pk, const int8_t* naf, int naf_len,
                                     const int32_t (*table)[54]) {
  BN_CTX;
  Fp12 f = fp12_one();
  G2J T = {pk.x, pk.y, fp2_one()};
  Fp2 nqy = fp2_norm(fp2_neg(pk.y));
  int ti = 0;
  for (int j = naf_len - 2; j >= 0; --j) {
    f = fp12_sqr(f);
    f = ell(f, line_from_table(table[ti++]), sig.x, sig.y);
    f = ell(f, doubling_step(T), h.x, h.y);
    int d = naf[j];
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
BN_FUNC Fp12 final_exponentiation(const Fp12& f) {
  BN_CTX;
  // easy part: f^((p^6-1)(p^2+1))
  Fp12 t = fp12_mul(fp12_conj(f), fp12_inv(f));
  t = fp12_mul(fp12_frob<2>(t), t);
  // hard part: t^(l0 + l1 p + l2 p^2 + l3 p^3), Fuentes-Castaneda et al. arrangement
  Fp12 a = fp12_conj(cyclotomic_exp_x(t));                 // t^-x
  a = fp12_cyclotomic_sqr(a);                              // t^-2x
  Fp12 b = fp12_cyclotomic_sqr(a);                         // t^-4x
  b = fp12_mul(a, b);                                      // t^-6x
  Fp12 c = fp12_conj(cyclotomic_exp_x(b));                 // t^(6x^2)
  Fp12 d = fp12_conj(b);                                   // t^(6x)
  b = fp12_mul(c, d);                                      // t^(6x^2+6x)
  d = fp12_cyclotomic_sqr(c);                              // t^(12x^2)
  Fp12 e = cyclotomic_exp_x(d);                            // t^(12x^3)
  e = fp12_mul(b, e);                                      // l2
  d = fp12_mul(a, e);                                      // l1
  a = fp12_mul(c, e);                                      // t^(12x^3+12x^2+6x)
  c = fp12_mul(t, a);                                      // l0
  a = fp12_mul(c, fp12_frob<1>(d));
  a = fp12_mul(a, fp12_frob<2>(e));
  c = fp12_mul(fp12_conj(t), d);                           // l3
  a = fp12_mul(a, fp12_frob<3>(c));
  return a;
}

// Gt byte layout (Gt::to_repr / from_repr, pairings.rs:499-579): c0.c0.c0, c0.c0.c1, c0.c1.c0, ...
BN_FUNC void fp12_to_be(uint8_t* out, const Fp12& a) {
  BN_CTX;
  const Fp2* s[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  for (int i = 0; i < 6; ++i) { fp_to_be(out + 64 * i, s[i]->c0); fp_to_be(out + 64 * i + 32, s[i]->c1); }
}
BN_FUNC Fp12 fp12_from_be(const uint8_t* in, bool& ok) {
  BN_CTX;
  Fp12 a;
  Fp2* s[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  ok = true;
  for (int i = 0; i < 6; ++i) {
    bool o0, o1;
    s[i]->c0 = fp_from_be(in + 64 * i, o0); s[i]->c1 = fp_from_be(in + 64 * i + 32, o1);
    ok &= o0 & o1;
  }
  return a;
}
// Device-side workspace form of an Fp12: 108 strict limbs (canonical Montgomery), c0.c0.c0 first.
BN_FUNC void fp12_store_limbs(int32_t* out, size_t stride, const Fp12& a) {
  BN_CTX;
  const Fp2* s[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  for (int i = 0; i < 6; ++i) {
    Fp c0 = fp_canon(s[i]->c0), c1 = fp_canon(s[i]->c1);
    for (int k = 0; k < NL; ++k) { out[(size_t)(18 * i + k) * stride] = c0.l[k]; out[(size_t)(18 * i + 9 + k) * stride] = c1.l[k]; }
  }
}
BN_FUNC Fp12 fp12_load_limbs(const int32_t* in, size_t stride) {
  BN_CTX;
  Fp12 a;
  Fp2* s[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  for (int i = 0; i < 6; ++i) {
    for (int k = 0; k < NL; ++k) { s[i]->c0.l[k] = in[(size_t)(18 * i + k) * stride]; s[i]->c1.l[k] = in[(size_t)(18 * i + 9 + k) * stride]; }
    BN_TRK(set_trk(s[i]->c0, 0, 1, 0, 0.006, 1); set_trk(s[i]->c1, 0, 1, 0, 0.006, 1);)
  }
  return a;
}

}  // namespace bn
