// quad.h -- G2 point arithmetic with FOUR LANES PER POINT, for the per-key preparation (G2Prepared::from, pairings.rs:614-660):
// the psi subgroup test (63-bit scalar multiplication by the addition chain) and the 88 line steps are serial chains of Fp2
// products, and a launch over a few thousand keys costs the latency of one lane's chain (1.55 ms) -- the largest item of every
// small or mid-size verify_batch.  A line step or a point addition is 7 ... 15 Fp2 products in 3 ... 5 dependency levels: the
// four lanes of a quad form the (up to four) independent products of a level side by side -- each lane picks ITS operand pair,
// all run the same fp2_mul, and the four products reach every lane by DPP quad permutes (tri.h) -- then every lane does the
// cheap linear glue for itself, so all four hold the whole running point.  Chains get ~2 x shorter.
// Same field values as the serial functions of pairing.h / curve.h (doubling_step, addition_step, proj_add, proj_dbl, jac_dbl,
// proj_mul_bn_x, g2_torsion_free), hence the same canonical table limbs and the same validity bit.
#pragma once
#include "tri.h"

namespace bn {

BN_INL Fp2 q_pick4(uint32_t role, const Fp2& a0, const Fp2& a1, const Fp2& a2, const Fp2& a3) {
  return fp2_pick(role == 0u, a0, fp2_pick(role == 1u, a1, fp2_pick(role == 2u, a2, a3)));
}
struct Quad4 { Fp2 r0, r1, r2, r3; };
// lane r forms a * b from ITS operands; every lane receives all four products
BN_INL Quad4 quad_products(const Fp2& a, const Fp2& b) {
  const Fp2 p = fp2_mul(a, b);
  return {tri_fetch2<0, 0, 0, 0>(p), tri_fetch2<1, 1, 1, 1>(p), tri_fetch2<2, 2, 2, 2>(p), tri_fetch2<3, 3, 3, 3>(p)};
}

// ---- line steps (pairing.h doubling_step / addition_step, pairings.rs:901-962): T normalised in and out, on every lane
BN_FUNC Line quad_doubling_step(G2J& r, uint32_t role) {
  BN_CTX;
  const Fp2 zy = fp2_norm(fp2_add(r.z, r.y));
  const Fp2 a1 = q_pick4(role, r.x, r.y, r.z, zy);
  const Quad4 p1 = quad_products(a1, a1);                       // x^2, y^2, z^2, (z + y)^2
  const Fp2 tmp0 = p1.r0, tmp1 = p1.r1, zsq = p1.r2;
  const Fp2 tmp4 = f_lc2<3, 0>(tmp0, tmp0);                      // 3 x^2
  const Fp2 tmp6 = fp2_norm(fp2_add(r.x, tmp4));
  const Fp2 a2 = q_pick4(role, tmp1, fp2_norm(fp2_add(tmp1, r.x)), tmp4, tmp6);
  const Quad4 p2 = quad_products(a2, a2);                       // tmp1^2, (tmp1 + x)^2, tmp4^2, tmp6^2
  const Fp2 tmp2 = p2.r0, tmp5 = p2.r2;
  const Fp2 tmp3 = f_lc3<2, -2, -2>(p2.r1, tmp0, tmp2);
  const Fp2 nx = f_lc2<1, -2>(tmp5, tmp3);
  const Fp2 nz = fp2_norm(fp2_sub(fp2_sub(p1.r3, tmp1), zsq));
  const Fp2 d = fp2_norm(fp2_sub(tmp3, nx));
  const Quad4 p3 = quad_products(q_pick4(role, d, tmp4, nz, nz), q_pick4(role, tmp4, zsq, zsq, zsq));   // (tmp3 - nx) tmp4, tmp4 zsq, nz zsq
  Line l;
  l.c1 = f_lc2<-2, 0>(p3.r1, p3.r1);
  l.c2 = f_lc2<1, -4>(fp2_sub(fp2_sub(p2.r3, tmp0), tmp5), tmp1);
  l.c0 = f_lc2<2, 0>(p3.r2, p3.r2);
  r.x = nx; r.y = f_lc2<1, -8>(p3.r0, tmp2); r.z = nz;
  return l;
}
BN_FUNC Line quad_addition_step(G2J& r, const Fp2& qx, const Fp2& qy, uint32_t role) {
  BN_CTX;
  const Fp2 a1 = q_pick4(role, r.z, qy, fp2_norm(fp2_add(qy, r.z)), r.z);
  const Quad4 p1 = quad_products(a1, a1);                       // z^2, qy^2, (qy + z)^2
  const Fp2 zsq = p1.r0, ysq = p1.r1;
  const Fp2 e = fp2_norm(fp2_sub(fp2_sub(p1.r2, ysq), zsq));
  const Quad4 p2 = quad_products(q_pick4(role, zsq, e, zsq, zsq), q_pick4(role, qx, zsq, qx, qx));      // t0 = zsq qx, t1 = e zsq
  const Fp2 t2 = fp2_norm(fp2_sub(p2.r0, r.x));
  const Fp2 t6 = f_lc2<1, -2>(p2.r1, r.y);
  const Fp2 zt = fp2_norm(fp2_add(r.z, t2));
  const Quad4 p3 = quad_products(q_pick4(role, t2, zt, t6, t6), q_pick4(role, t2, zt, t6, qx));          // t3 = t2^2, (z + t2)^2, t6^2, t9 = t6 qx
  const Fp2 t3 = p3.r0;
  const Fp2 t4 = f_lc2<4, 0>(t3, t3);
  const Fp2 nz = fp2_norm(fp2_sub(fp2_sub(p3.r1, zsq), t3));
  const Fp2 qn = fp2_norm(fp2_add(qy, nz));
  const Quad4 p4 = quad_products(q_pick4(role, t4, t4, qn, nz), q_pick4(role, t2, r.x, qn, nz));         // t5 = t4 t2, t7 = t4 x, (qy + nz)^2, nz^2
  const Fp2 t5 = p4.r0, t7 = p4.r1;
  const Fp2 nx = f_lc3<1, -1, -2>(p3.r2, t5, t7);
  const Fp2 d = fp2_norm(fp2_sub(t7, nx));
  const Quad4 p5 = quad_products(q_pick4(role, d, r.y, d, d), q_pick4(role, t6, t5, t6, t6));            // t8 = (t7 - nx) t6, y t5
  const Fp2 t10 = fp2_sub(fp2_sub(p4.r2, ysq), p4.r3);
  Line l;
  l.c2 = f_lc2<2, -1>(p3.r3, t10);
  l.c0 = f_lc2<2, 0>(nz, nz);
  l.c1 = f_lc2<-2, 0>(t6, t6);
  r.x = nx; r.y = f_lc2<1, -2>(p5.r0, p5.r1); r.z = nz;
  return l;
}
// G2Prepared::from on a quad (g2_prepare_lines, pairing.h): the 88 line triples in evaluation order as canonical limbs; lane c < 3
// canonicalises and stores coefficient c of every line.
BN_INL void quad_line_store(const Ws& w, const Line& l, uint32_t role) {
  const Fp2 c = fp2_pick(role == 0u, l.c0, fp2_pick(role == 1u, l.c1, l.c2));
  const uint32_t slot = role < 2u ? role : 2u;                   // lane 3 repeats lane 2's store (same address, same value)
  fp2_store_limbs(ws_at_lane(w, 18u * slot), c);
}
BN_FUNC void quad_prepare_lines(const G2A& q, const Ws& out, uint32_t role) {
  BN_CTX;
  G2J T = {q.x, q.y, fp2_one()};
  const Fp2 nqy = fp2_norm(fp2_neg(q.y));
  int ti = 0;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    quad_line_store(ws_at(out, 54 * (size_t)ti++), quad_doubling_step(T, role), role);
    const int d = ate_naf_digit(j);
    if (d != 0) quad_line_store(ws_at(out, 54 * (size_t)ti++), quad_addition_step(T, q.x, d > 0 ? q.y : nqy, role), role);
  }
  const Fp2 g2 = fp2_const(bnc::GAMMA1[1]), g3 = fp2_const(bnc::GAMMA1[2]);
  const Fp2 cx = fp2_norm(fp2_conj(q.x)), cy = fp2_norm(fp2_conj(q.y));
  const Quad4 f1 = quad_products(q_pick4(role, cx, cy, cx, cy), q_pick4(role, g2, g3, g2, g3));          // pi(Q)
  const Fp2 q1x = f1.r0, q1y = f1.r1;
  const Fp2 c1x = fp2_norm(fp2_conj(q1x)), c1y = fp2_norm(fp2_conj(q1y));
  const Quad4 f2 = quad_products(q_pick4(role, c1x, c1y, c1x, c1y), q_pick4(role, g2, g3, g2, g3));
  const Fp2 q2x = f2.r0, q2y = fp2_norm(fp2_neg(f2.r1));                                                  // -pi^2(Q)
  quad_line_store(ws_at(out, 54 * (size_t)ti++), quad_addition_step(T, q1x, q1y, role), role);
  quad_line_store(ws_at(out, 54 * (size_t)ti++), quad_addition_step(T, q2x, q2y, role), role);
}

// ---- the complete group law and the Jacobian doubling on a quad (curve.h proj_add / proj_dbl / jac_dbl / conversions)
BN_FUNC G2P quad_proj_add(const G2P& a, const G2P& b, uint32_t role) {
  BN_CTX;
  const Fp2 b3 = fp2_const(bnc::B2_3);
  const Quad4 p1 = quad_products(q_pick4(role, a.x, a.y, a.z, fp2_norm(fp2_sub(a.x, a.y))), q_pick4(role, b.x, b.y, b.z, fp2_norm(fp2_sub(b.x, b.y))));
  const Fp2 t0 = p1.r0, t1 = p1.r1, t2 = p1.r2, m3 = p1.r3;
  const Quad4 p2 = quad_products(q_pick4(role, fp2_norm(fp2_sub(a.y, a.z)), fp2_norm(fp2_sub(a.x, a.z)), t2, t2),
                                 q_pick4(role, fp2_norm(fp2_sub(b.y, b.z)), fp2_norm(fp2_sub(b.x, b.z)), b3, b3));     // m4, m5, 3b t2
  const Fp2 t3 = f_lc3<1, 1, -1>(t0, t1, m3), t4 = f_lc3<1, 1, -1>(t1, t2, p2.r0), y3 = f_lc3<1, 1, -1>(t0, t2, p2.r1);
  const Fp2 t0_3 = f_lc2<3, 0>(t0, t0), bt2 = p2.r2;
  const Fp2 z3 = fp2_norm(fp2_add(t1, bt2)), t1m = fp2_norm(fp2_sub(t1, bt2));
  const Quad4 p3 = quad_products(q_pick4(role, y3, t3, t0_3, z3), q_pick4(role, b3, t1m, t3, t4));       // 3b y3, t3 t1m, 3 t0 t3, z3 t4
  const Fp2 by3 = p3.r0;
  const Quad4 p4 = quad_products(q_pick4(role, t4, t1m, by3, by3), q_pick4(role, by3, z3, t0_3, t0_3)); // t4 by3, t1m z3, by3 3 t0
  return {fp2_norm(fp2_sub(p3.r1, p4.r0)), fp2_norm(fp2_add(p4.r1, p4.r2)), fp2_norm(fp2_add(p3.r3, p3.r2))};
}
BN_FUNC G2P quad_proj_dbl(const G2P& a, uint32_t role) {
  BN_CTX;
  const Fp2 b3 = fp2_const(bnc::B2_3);
  const Quad4 p1 = quad_products(q_pick4(role, a.y, a.y, a.z, a.x), q_pick4(role, a.y, a.z, a.z, a.y)); // y^2, y z, z^2, x y
  const Fp2 t0 = p1.r0, t1 = p1.r1;
  const Fp2 z8 = f_lc2<8, 0>(t0, t0);
  const Quad4 p2 = quad_products(q_pick4(role, p1.r2, p1.r2, p1.r2, p1.r2), q_pick4(role, b3, b3, b3, b3));   // 3b z^2
  const Fp2 t2 = p2.r0;
  const Fp2 y3 = fp2_norm(fp2_add(t0, t2));
  const Fp2 t0m = f_lc2<1, -3>(t0, t2);
  const Quad4 p3 = quad_products(q_pick4(role, t2, t1, t0m, t0m), q_pick4(role, z8, z8, y3, p1.r3));       // x3, z3, t0m y3, t0m xy
  return {f_lc2<2, 0>(p3.r3, t0), fp2_norm(fp2_add(p3.r0, p3.r2)), p3.r1};
}
BN_FUNC G2P quad_jac_dbl(const G2P& p, uint32_t role) {
  BN_CTX;
  const Quad4 p1 = quad_products(q_pick4(role, p.x, p.y, p.y, p.y), q_pick4(role, p.x, p.y, p.z, p.z));       // A = x^2, B = y^2, y z
  const Fp2 A = p1.r0, B = p1.r1;
  const Fp2 E = f_lc2<3, 0>(A, A);
  const Fp2 xb = fp2_norm(fp2_add(p.x, B));
  const Fp2 a2 = q_pick4(role, B, xb, E, E);
  const Quad4 p2 = quad_products(a2, a2);                                                                       // C = B^2, (x + B)^2, E^2
  const Fp2 C = p2.r0;
  const Fp2 D = f_lc3<2, -2, -2>(p2.r1, A, C);
  const Fp2 x3 = f_lc2<1, -2>(p2.r2, D);
  const Fp2 dx = fp2_norm(fp2_sub(D, x3));
  const Quad4 p3 = quad_products(q_pick4(role, E, E, E, E), q_pick4(role, dx, dx, dx, dx));
  return {x3, f_lc2<1, -8>(p3.r0, C), f_lc2<2, 0>(p1.r2, p1.r2)};
}
BN_FUNC G2P quad_proj_to_jac(const G2P& p, uint32_t role) {
  BN_CTX;
  const Quad4 p1 = quad_products(q_pick4(role, p.x, p.z, p.x, p.z), q_pick4(role, p.z, p.z, p.z, p.z));       // x z, z^2
  const Quad4 p2 = quad_products(q_pick4(role, p.y, p.y, p.y, p.y), q_pick4(role, p1.r1, p1.r1, p1.r1, p1.r1));   // y z^2
  return {p1.r0, p2.r0, p.z};
}
BN_FUNC G2P quad_proj_from_jac(const G2P& j, uint32_t role) {
  BN_CTX;
  const bool inf = f_is_zero(j.z);
  const G2P id = proj_identity<Fp2>();
  const Quad4 p1 = quad_products(q_pick4(role, j.x, j.z, j.x, j.z), q_pick4(role, j.z, j.z, j.z, j.z));       // x z, z^2
  const Quad4 p2 = quad_products(q_pick4(role, p1.r1, p1.r1, p1.r1, p1.r1), q_pick4(role, j.z, j.z, j.z, j.z));   // z^3
  return {fp2_select(inf, id.x, p1.r0), fp2_select(inf, id.y, j.y), fp2_select(inf, id.z, p2.r0)};
}
// [x]P for the BN parameter by the addition chain of proj_mul_bn_x (curve.h), runs of doublings in Jacobian coordinates
BN_FUNC G2P quad_mul_bn_x(const G2P& p, uint32_t role) {
  BN_CTX;
  const ChainOp prog[BN_X_CHAIN_LEN] = BN_X_CHAIN;
  G2P slot[BN_X_CHAIN_SLOTS];
  slot[0] = p;
  G2P r = p;
  for (int k = 0; k < BN_X_CHAIN_LEN; ++k) {
    const ChainOp op = prog[k];
    if (op.load >= 0) r = slot[op.load];
    if (op.sq >= 2) {
      G2P j = quad_proj_to_jac(r, role);
      for (int q = 0; q < op.sq; ++q) j = quad_jac_dbl(j, role);
      r = quad_proj_from_jac(j, role);
    } else if (op.sq == 1) r = quad_proj_dbl(r, role);
    if (op.mul >= 0) r = quad_proj_add(r, slot[op.mul], role);
    if (op.store >= 0) slot[op.store] = r;
    if (op.cstore >= 0) slot[op.cstore] = proj_neg(r);
  }
  return r;
}
// psi (g2.rs:938-954): lanes 0, 1 form the two products, z is conjugated by everyone
BN_INL G2P quad_psi(const G2P& a, uint32_t role) {
  const Fp2 cx = fp2_norm(fp2_conj(a.x)), cy = fp2_norm(fp2_conj(a.y));
  const Fp2 g2 = fp2_const(bnc::GAMMA1[1]), g3 = fp2_const(bnc::GAMMA1[2]);
  const Quad4 f = quad_products(q_pick4(role, cx, cy, cx, cy), q_pick4(role, g2, g3, g2, g3));
  return {f.r0, f.r1, fp2_norm(fp2_conj(a.z))};
}
// [x+1]P + psi([x]P) + psi^2([x]P) == psi^3([2x]P)  (g2_torsion_free, curve.h; same boolean as [r]P == O, g2.rs:733-736)
BN_FUNC bool quad_torsion_free(const G2A& a, uint32_t role) {
  BN_CTX;
  const G2P p = proj_from_affine(a);
  const G2P xp = quad_mul_bn_x(p, role);
  const G2P p1 = quad_psi(xp, role);
  const G2P lhs = quad_proj_add(quad_proj_add(xp, p, role), quad_proj_add(p1, quad_psi(p1, role), role), role);
  const G2P rhs = quad_psi(quad_psi(quad_psi(quad_proj_dbl(xp, role), role), role), role);
  return a.inf | proj_eq(lhs, rhs);
}

// ---- the one-pair Miller loop with a VARIABLE G2 point on a quad (miller_loop_1, pairing.h; pairings.rs:760-857): the line steps
// run four lanes per point (above), the Fp12 accumulator three lanes per value (tri.h).  A line l = c0 py + c1 px w + c2 w^3 is the
// sparse element (o0, 0, 0) + (o3, o4, 0) w: every lane multiplies ITS Karatsuba operand by a two-coefficient Fp6 (fp6_mul_by_01,
// five Fp2 products) -- lane 0: f0 (o0, 0), lane 1: f1 (o3, o4), lane 2: (f0 - f1)(o3 - o0, o4) -- then the usual recombination.
// Same values as ell / fp12_mul_by_034, hence the same canonical Miller value as miller_loop_1 and the oracle.
BN_FUNC Fp6 tri_ell(const Fp6& f, const Line& l, const Fp& px, const Fp& py, uint32_t role) {
  BN_CTX;
  const Fp2 o0 = fp2_mul_fp(l.c0, py), o3 = fp2_mul_fp(l.c1, px);
  const Fp2 b0 = fp2_pick(role == 0u, o0, fp2_pick(role == 1u, o3, fp2_norm(fp2_sub(o3, o0))));
  const Fp2 b1 = fp2_pick(role == 0u, fp2_zero(), l.c2);
  const Fp6 p = fp6_mul_by_01(tri_third<false, false>(f, role), b0, b1);
  const Fp6 x1 = tri_fetch6<1, 0, 0, 0>(p);
  const Fp6 x2 = tri_fetch6<2, 2, 2, 2>(p);
  return tri_recombine(p, x1, fp6_norm(fp6_add(fp6_add(p, x1), x2)), role);
}
// p: affine G1 point (normalised coordinates), q: affine G2 point, both validated by the caller and held by every lane
BN_FUNC Fp6 tri_miller_1(const Fp& px, const Fp& py, const G2A& q, uint32_t role) {
  BN_CTX;
  Fp6 f = fp6_norm(fp6_pick(role == 0u, fp6_one(), fp6_zero()));
  G2J T = {q.x, q.y, fp2_one()};
  const Fp2 nqy = fp2_norm(fp2_neg(q.y));
  for (int j = bnc::ATE_NAF_LEN - 2; j >= 0; --j) {
    f = tri_sqr(f, role);
    f = tri_ell(f, quad_doubling_step(T, role), px, py, role);
    const int d = ate_naf_digit(j);
    if (d != 0) f = tri_ell(f, quad_addition_step(T, q.x, d > 0 ? q.y : nqy, role), px, py, role);
  }
  const Fp2 g2 = fp2_const(bnc::GAMMA1[1]), g3 = fp2_const(bnc::GAMMA1[2]);
  const Fp2 cx = fp2_norm(fp2_conj(q.x)), cy = fp2_norm(fp2_conj(q.y));
  const Quad4 f1 = quad_products(q_pick4(role, cx, cy, cx, cy), q_pick4(role, g2, g3, g2, g3));          // pi(Q)
  const Fp2 c1x = fp2_norm(fp2_conj(f1.r0)), c1y = fp2_norm(fp2_conj(f1.r1));
  const Quad4 f2 = quad_products(q_pick4(role, c1x, c1y, c1x, c1y), q_pick4(role, g2, g3, g2, g3));
  f = tri_ell(f, quad_addition_step(T, f1.r0, f1.r1, role), px, py, role);
  f = tri_ell(f, quad_addition_step(T, f2.r0, fp2_norm(fp2_neg(f2.r1)), role), px, py, role);           // -pi^2(Q)
  return f;
}

// One pair with a PREPARED key on a quad (miller_loop_1prepared, pairing.h): f = ML(H, Q), the 88 line triples of Q read from its
// raw table (54 canonical limbs per line; every lane of the quad loads the triple), H affine (px, py) on every lane.
BN_FUNC Fp6 tri_miller_1prepared(const Fp& px, const Fp& py, const Ws& ta_in, uint32_t role) {
  BN_CTX;
  Fp6 f = fp6_norm(fp6_pick(role == 0u, fp6_one(), fp6_zero()));
  Ws ta = ta_in;
  int ti = 0;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= -2; --j) {               // j = -1, -2: the two final lines (no squaring)
    if (j >= 0) f = tri_sqr(f, role);
    const int lines = j >= 0 ? (ate_naf_digit(j) != 0 ? 2 : 1) : 1;
    for (int q = 0; q < lines; ++q) {
      BN_OPAQUE(ta);
      f = tri_ell(f, line_load_limbs(ws_at(ta, 54 * (size_t)ti)), px, py, role);
      ++ti;
    }
  }
  return f;
}

}  // namespace bn
