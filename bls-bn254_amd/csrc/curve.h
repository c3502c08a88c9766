// curve.h -- G1 (y^2 = x^3 + 3 over Fp) and G2 (y^2 = x^3 + 3/(9+u) over Fp2): codecs, on-curve and
// subgroup checks, complete projective group law, SVDW map and hash-to-curve glue.
//
// Reference operators replaced:
//   G1Affine::{from,to}_uncompressed g1.rs:297-302,:339-360 ; is_on_curve :383-391
//   G1Projective::{addition,double,multiply} g1.rs:704-841 ; hash/encode :910-928
//   G2Affine::{from,to}_uncompressed g2.rs:292-300,:350-388 ; is_on_curve :409-414
//   G2Projective::{addition,double,multiply,psi,clear_cofactor,is_torsion_free} g2.rs:685-954
//   MapToCurve for Fp / Fp2 (SVDW) fp.rs:284-371, fp2.rs:221-287
// The group law is the same complete (Renes-Costello-Batina 2015/1060 Alg 7/9) formula set the
// reference uses, so every lane runs an identical instruction stream with no special cases.  The
// cross terms are formed subtractively (X1Y2 + X2Y1 = X1X2 + Y1Y2 - (X1-Y1)(X2-Y2)) to stay inside the
// lazy-limb intervals of fp29.h.
#pragma once
#include "tower.h"

namespace bn {

// ------------------------------------------------------------------ field-generic helpers
BN_INL Fp f_add(const Fp& a, const Fp& b) { return fp_add(a, b); }
BN_INL Fp f_sub(const Fp& a, const Fp& b) { return fp_sub(a, b); }
BN_INL Fp f_mul(const Fp& a, const Fp& b) { return fp_mul(a, b); }
BN_INL Fp f_sqr(const Fp& a) { return fp_sqr(a); }
BN_INL Fp f_norm(const Fp& a) { return fp_norm(a); }
BN_INL Fp f_neg(const Fp& a) { return fp_neg(a); }
BN_INL bool f_is_zero(const Fp& a) { return fp_is_zero(a); }
BN_INL Fp f_select(bool c, const Fp& a, const Fp& b) { return fp_select(c, a, b); }
BN_INL Fp f_mul_b3(const Fp& a) { return fp_lc2<9, 0>(a, a); }                     // 3b = 9 (fp.rs:414)
template <int K1, int K2> BN_INL Fp f_lc2(const Fp& a, const Fp& b) { return fp_lc2<K1, K2>(a, b); }
template <int K1, int K2, int K3> BN_INL Fp f_lc3(const Fp& a, const Fp& b, const Fp& c) { return fp_lc3<K1, K2, K3>(a, b, c); }
BN_INL void f_set_zero(Fp& a) { a = fp_zero(); }
BN_INL void f_set_one(Fp& a) { a = fp_one(); }

BN_INL Fp2 f_add(const Fp2& a, const Fp2& b) { return fp2_add(a, b); }
BN_INL Fp2 f_sub(const Fp2& a, const Fp2& b) { return fp2_sub(a, b); }
BN_INL Fp2 f_mul(const Fp2& a, const Fp2& b) { return fp2_mul(a, b); }
BN_INL Fp2 f_sqr(const Fp2& a) { return fp2_sqr(a); }
BN_INL Fp2 f_norm(const Fp2& a) { return fp2_norm(a); }
BN_INL Fp2 f_neg(const Fp2& a) { return fp2_neg(a); }
BN_INL bool f_is_zero(const Fp2& a) { return fp2_is_zero(a); }
BN_INL Fp2 f_select(bool c, const Fp2& a, const Fp2& b) { return fp2_select(c, a, b); }
BN_INL Fp2 f_mul_b3(const Fp2& a) { return fp2_mul(a, fp2_const(bnc::B2_3)); }      // fp2.rs:411-413
template <int K1, int K2> BN_INL Fp2 f_lc2(const Fp2& a, const Fp2& b) { return {fp_lc2<K1, K2>(a.c0, b.c0), fp_lc2<K1, K2>(a.c1, b.c1)}; }
template <int K1, int K2, int K3> BN_INL Fp2 f_lc3(const Fp2& a, const Fp2& b, const Fp2& c) {
  return {fp_lc3<K1, K2, K3>(a.c0, b.c0, c.c0), fp_lc3<K1, K2, K3>(a.c1, b.c1, c.c1)};
}
BN_INL void f_set_zero(Fp2& a) { a = fp2_zero(); }
BN_INL void f_set_one(Fp2& a) { a = fp2_one(); }

// ------------------------------------------------------------------ points
template <class F> struct Aff { F x, y; bool inf; };
template <class F> struct Proj { F x, y, z; };        // homogeneous; identity = (0, 1, 0)
typedef Aff<Fp> G1A;
typedef Aff<Fp2> G2A;
typedef Proj<Fp> G1P;
typedef Proj<Fp2> G2P;

template <class F> BN_INL Proj<F> proj_identity() { Proj<F> r; f_set_zero(r.x); f_set_one(r.y); f_set_zero(r.z); return r; }
template <class F> BN_INL Proj<F> proj_from_affine(const Aff<F>& a) {
  Proj<F> id = proj_identity<F>(), r;
  F one; f_set_one(one);
  r.x = f_select(a.inf, id.x, a.x); r.y = f_select(a.inf, id.y, a.y); r.z = f_select(a.inf, id.z, one);
  return r;
}
// complete addition, RCB Alg 7 with a = 0 (g1.rs:744-786, g2.rs:789-831); inputs/outputs normalised
template <class F> BN_FUNC Proj<F> proj_add(const Proj<F>& a, const Proj<F>& b) {
  BN_CTX;
  F t0 = f_mul(a.x, b.x), t1 = f_mul(a.y, b.y), t2 = f_mul(a.z, b.z);
  F m3 = f_mul(f_sub(a.x, a.y), f_sub(b.x, b.y));
  F m4 = f_mul(f_sub(a.y, a.z), f_sub(b.y, b.z));
  F m5 = f_mul(f_sub(a.x, a.z), f_sub(b.x, b.z));
  F t3 = f_lc3<1, 1, -1>(t0, t1, m3);              // X1Y2 + X2Y1
  F t4 = f_lc3<1, 1, -1>(t1, t2, m4);              // Y1Z2 + Y2Z1
  F y3 = f_lc3<1, 1, -1>(t0, t2, m5);              // X1Z2 + X2Z1
  F t0_3 = f_lc2<3, 0>(t0, t0);                    // 3 X1X2
  F bt2 = f_mul_b3(t2);
  F z3 = f_norm(f_add(t1, bt2));
  F t1m = f_norm(f_sub(t1, bt2));
  F by3 = f_mul_b3(y3);
  F x3 = f_norm(f_sub(f_mul(t3, t1m), f_mul(t4, by3)));
  F yy = f_norm(f_add(f_mul(t1m, z3), f_mul(by3, t0_3)));
  F zz = f_norm(f_add(f_mul(z3, t4), f_mul(t0_3, t3)));
  return {x3, yy, zz};
}
// doubling, RCB Alg 9 (g1.rs:788-818, g2.rs:834-863)
template <class F> BN_FUNC Proj<F> proj_dbl(const Proj<F>& a) {
  BN_CTX;
  F t0 = f_sqr(a.y);
  F z8 = f_lc2<8, 0>(t0, t0);
  F t1 = f_mul(a.y, a.z);
  F t2 = f_mul_b3(f_sqr(a.z));
  F x3 = f_mul(t2, z8);
  F y3 = f_norm(f_add(t0, t2));
  F z3 = f_mul(t1, z8);
  F t0m = f_lc2<1, -3>(t0, t2);                    // t0 - 3 t2
  F yy = f_norm(f_add(x3, f_mul(t0m, y3)));
  F xy = f_mul(a.x, a.y);
  F xx = f_lc2<2, 0>(f_mul(t0m, xy), t0);
  return {xx, yy, z3};
}
// Jacobian doubling for a = 0 (dbl-2009-l: 2M + 5S), used for runs of doublings between the complete additions.
// (X, Y, Z) Jacobian means x = X / Z^2, y = Y / Z^3.  Valid for every input including Z = 0 and Y = 0 (result Z = 0).
template <class F> BN_FUNC Proj<F> jac_dbl(const Proj<F>& p) {
  BN_CTX;
  F A = f_sqr(p.x), B = f_sqr(p.y);
  F C = f_sqr(B);
  F t = f_sqr(f_norm(f_add(p.x, B)));
  F D = f_lc3<2, -2, -2>(t, A, C);
  F E = f_lc2<3, 0>(A, A);
  F x3 = f_lc2<1, -2>(f_sqr(E), D);
  F y3 = f_lc2<1, -8>(f_mul(E, f_norm(f_sub(D, x3))), C);
  F yz = f_mul(p.y, p.z);
  return {x3, y3, f_lc2<2, 0>(yz, yz)};
}
// homogeneous (X : Y : Z) <-> Jacobian with the same Z: (XZ, YZ^2, Z) and back (X'Z', Y', Z'^3); 2M + 1S each way.
// The identity (0 : 1 : 0) becomes (0, 0, 0), stays (0, 0, 0) under jac_dbl, and is restored on the way back.
template <class F> BN_FUNC Proj<F> proj_to_jac(const Proj<F>& p) {
  BN_CTX;
  return {f_mul(p.x, p.z), f_mul(p.y, f_sqr(p.z)), p.z};
}
template <class F> BN_FUNC Proj<F> proj_from_jac(const Proj<F>& j) {
  BN_CTX;
  bool inf = f_is_zero(j.z);
  Proj<F> id = proj_identity<F>();
  Proj<F> r = {f_mul(j.x, j.z), j.y, f_mul(f_sqr(j.z), j.z)};
  return {f_select(inf, id.x, r.x), f_select(inf, id.y, r.y), f_select(inf, id.z, r.z)};
}
template <class F> BN_INL Proj<F> proj_neg(const Proj<F>& a) { return {a.x, f_norm(f_neg(a.y)), a.z}; }
template <class F> BN_INL Proj<F> proj_select(bool c, const Proj<F>& a, const Proj<F>& b) {
  return {f_select(c, a.x, b.x), f_select(c, a.y, b.y), f_select(c, a.z, b.z)};
}
// k * P for a 64-bit public scalar (uniform bits): left-to-right double and add
template <class F> BN_FUNC Proj<F> proj_mul_u64(const Proj<F>& p, uint64_t k) {
  BN_CTX;
  Proj<F> acc = proj_identity<F>();
  for (int i = 63; i >= 0; --i) {
    acc = proj_dbl(acc);
    if ((k >> i) & 1) acc = proj_add(acc, p);
  }
  return acc;
}
// The BN parameter x = 0x44e992b44a6909f1 as a signed-digit chain over {1, 17, 35} -- negation is free in both places the chain is
// used: a point's inverse is (x, -y), a cyclotomic element's inverse is its conjugate --
//   _17 = 2^4 * 1 + 1,  _35 = 2 * _17 + 1,
//   x = (((((((((((_35 << 6 - _35) << 7 + _17) << 2 + _35) << 4 - _35) << 4 - _35) << 9 + _35) << 4 + _35) << 6 + _17) << 5 + _35) << 5 - _17) << 5 + _17
// = 62 doublings + 13 additions (binary: 62 + 27; the unsigned chain used until round 3: 62 + 17), found by exhaustive search
// over dictionaries of up to five odd values below 64 with an optimal signed recoding for each (tools/x_chain_search.py).
// One interpreter step: r <- slot[load] if load >= 0; r <- 2^sq r; r <- r + slot[mul] if mul >= 0; slot[store] <- r and
// slot[cstore] <- -r if >= 0.  Slot 0 holds the input.  Shared by [x]P (below, quad.h) and t^x (pairing.h, tri.h, wide.h).
struct ChainOp { int8_t load, sq, mul, store, cstore; };
constexpr int BN_X_CHAIN_LEN = 13, BN_X_CHAIN_SLOTS = 5;
#define BN_X_CHAIN {{-1, 4, 0, 1, 2}, {-1, 1, 0, 3, 4}, {-1, 6, 4, -1, -1}, {-1, 7, 1, -1, -1}, {-1, 2, 3, -1, -1}, {-1, 4, 4, -1, -1}, {-1, 4, 4, -1, -1}, \
                    {-1, 9, 3, -1, -1}, {-1, 4, 3, -1, -1}, {-1, 6, 1, -1, -1}, {-1, 5, 3, -1, -1}, {-1, 5, 2, -1, -1}, {-1, 5, 1, -1, -1}}
// [x]P by that chain; runs of doublings are done in Jacobian coordinates.
template <class F> BN_FUNC Proj<F> proj_mul_bn_x(const Proj<F>& p) {
  BN_CTX;
  const ChainOp prog[BN_X_CHAIN_LEN] = BN_X_CHAIN;
  Proj<F> slot[BN_X_CHAIN_SLOTS];
  slot[0] = p;
  Proj<F> r = p;
  for (int k = 0; k < BN_X_CHAIN_LEN; ++k) {
    const ChainOp op = prog[k];
    if (op.load >= 0) r = slot[op.load];
    if (op.sq >= 2) {                                // a run of doublings in Jacobian coordinates (2M + 5S each instead of 7M + 2S)
      Proj<F> j = proj_to_jac(r);
      for (int q = 0; q < op.sq; ++q) j = jac_dbl(j);
      r = proj_from_jac(j);
    } else if (op.sq == 1) r = proj_dbl(r);
    if (op.mul >= 0) r = proj_add(r, slot[op.mul]);
    if (op.store >= 0) slot[op.store] = r;
    if (op.cstore >= 0) slot[op.cstore] = proj_neg(r);
  }
  return r;
}
// k * P for a per-lane 256-bit scalar (4 x u64, little endian): branch-free select per bit
template <class F> BN_FUNC Proj<F> proj_mul_256(const Proj<F>& p, const uint64_t* k) {
  BN_CTX;
  Proj<F> acc = proj_identity<F>();
  for (int i = 255; i >= 0; --i) {
    acc = proj_dbl(acc);
    Proj<F> s = proj_add(acc, p);
    acc = proj_select((k[i >> 6] >> (i & 63)) & 1, s, acc);
  }
  return acc;
}
// k * P for a per-lane 256-bit scalar with 4-bit fixed windows: 252 doublings + 63 additions + 14 table additions instead of
// 256 + 256 (Mul<Scalar>, g1.rs:518-534 / g2.rs:866-886: same point as the reference's double-and-add).  The table index is
// per lane (the table lives in scratch); the control flow is uniform, and adding the table's identity entry is just another
// complete addition.
template <class F> BN_FUNC Proj<F> proj_mul_win4(const Proj<F>& p, const uint64_t* k) {
  BN_CTX;
  Proj<F> tab[16];
  tab[0] = proj_identity<F>();
  tab[1] = p;
  for (int i = 2; i < 16; ++i) tab[i] = proj_add(tab[i - 1], p);
  Proj<F> acc = tab[(int)(k[3] >> 60)];
  for (int w = 62; w >= 0; --w) {
    const int d = (int)((k[w >> 4] >> ((w & 15) * 4)) & 15);
    const Proj<F> t = tab[d];
    acc = proj_dbl(proj_dbl(proj_dbl(proj_dbl(acc))));
    acc = proj_add(acc, t);
  }
  return acc;
}
template <class F> BN_INL bool proj_eq(const Proj<F>& a, const Proj<F>& b) {
  bool ai = f_is_zero(a.z), bi = f_is_zero(b.z);
  bool e = f_is_zero(f_sub(f_mul(a.x, b.z), f_mul(b.x, a.z))) & f_is_zero(f_sub(f_mul(a.y, b.z), f_mul(b.y, a.z)));
  return (ai | bi) ? (ai & bi) : e;
}

// ------------------------------------------------------------------ G1 codec / checks
// strict decode: canonical coordinates, x == 0 -> identity (g1.rs:352-353); no on-curve check (E10)
BN_INL G1A g1_decode(const uint8_t* b, bool& ok) {
  bool okx, oky;
  G1A r;
  r.x = fp_from_be(b, okx);
  r.y = fp_from_be(b + 32, oky);
  r.inf = okx & fp_is_zero(r.x);
  ok = okx & (r.inf | oky);
  return r;
}
BN_INL void g1_encode(uint8_t* b, const G1A& a) {               // identity = (0, 1) (g1.rs:265-271)
  fp_to_be(b, fp_select(a.inf, fp_zero(), a.x));
  fp_to_be(b + 32, fp_select(a.inf, fp_one(), a.y));
}
BN_INL bool g1_on_curve(const G1A& a) {                          // g1.rs:383-391
  Fp rhs = fp_add(fp_mul(fp_sqr(a.x), a.x), fp_const(bnc::THREE));
  return a.inf | fp_is_zero(fp_sub(fp_sqr(a.y), rhs));
}
BN_FUNC G1A g1_to_affine(const G1P& p) {
  BN_CTX;
  G1A r;
  r.inf = fp_is_zero(p.z);
  Fp zi = fp_inv(p.z);
  r.x = fp_mul(p.x, zi); r.y = fp_mul(p.y, zi);
  return r;
}

// ------------------------------------------------------------------ G2 codec / checks
BN_INL G2A g2_decode(const uint8_t* b, bool& ok) {              // x.c1 | x.c0 | y.c1 | y.c0 (g2.rs:350-388)
  bool o0, o1, o2, o3;
  G2A r;
  r.x.c1 = fp_from_be(b, o0); r.x.c0 = fp_from_be(b + 32, o1);
  r.y.c1 = fp_from_be(b + 64, o2); r.y.c0 = fp_from_be(b + 96, o3);
  r.inf = o0 & o1 & fp2_is_zero(r.x);
  ok = o0 & o1 & (r.inf | (o2 & o3));
  return r;
}
BN_INL void g2_encode(uint8_t* b, const G2A& a) {               // g2.rs:292-300
  Fp2 x = fp2_select(a.inf, fp2_zero(), a.x), y = fp2_select(a.inf, fp2_one(), a.y);
  fp_to_be(b, x.c1); fp_to_be(b + 32, x.c0); fp_to_be(b + 64, y.c1); fp_to_be(b + 96, y.c0);
}
BN_INL bool g2_on_curve(const G2A& a) {                          // g2.rs:409-414
  Fp2 rhs = fp2_add(fp2_mul(fp2_sqr(a.x), a.x), fp2_const(bnc::B2));
  return a.inf | fp2_is_zero(fp2_sub(fp2_sqr(a.y), rhs));
}
BN_FUNC G2A g2_to_affine(const G2P& p) {
  BN_CTX;
  G2A r;
  r.inf = fp2_is_zero(p.z);
  Fp2 zi = fp2_inv(p.z);
  r.x = fp2_mul(p.x, zi); r.y = fp2_mul(p.y, zi);
  return r;
}
BN_INL G2P g2_psi(const G2P& a) {                                // g2.rs:938-954
  return {fp2_mul(fp2_norm(fp2_conj(a.x)), fp2_const(bnc::GAMMA1[1])),
          fp2_mul(fp2_norm(fp2_conj(a.y)), fp2_const(bnc::GAMMA1[2])),
          fp2_norm(fp2_conj(a.z))};
}
// Subgroup membership.  The reference computes [r]P (254 doublings + additions, g2.rs:733-736); the
// same boolean is obtained from one 63-bit multiplication and the untwist-Frobenius-twist map:
//   [x+1]P + psi([x]P) + psi^2([x]P) == psi^3([2x]P)
// (equivalence incl. small-order points is tested against [r]P in tests/test_oracle_golden.py).
BN_FUNC bool g2_torsion_free(const G2A& a) {
  BN_CTX;
  G2P p = proj_from_affine(a);
  G2P xp = proj_mul_bn_x(p);
  G2P p1 = g2_psi(xp);
  G2P lhs = proj_add(proj_add(xp, p), proj_add(p1, g2_psi(p1)));
  G2P rhs = g2_psi(g2_psi(g2_psi(proj_dbl(xp))));
  return a.inf | proj_eq(lhs, rhs);
}
BN_FUNC G2P g2_clear_cofactor(const G2P& p) {               // g2.rs:685-693
  G2P p0 = proj_mul_bn_x(p);
  G2P p1 = g2_psi(proj_add(proj_dbl(p0), p0));
  G2P p2 = g2_psi(g2_psi(p0));
  G2P p3 = g2_psi(g2_psi(g2_psi(p)));
  return proj_add(proj_add(p0, p1), proj_add(p2, p3));
}

// ------------------------------------------------------------------ compressed codecs
// G1: 32 B = x big-endian, bit 255 = parity of y (G1Affine::to_compressed, g1.rs:283-288).  Decoding picks the
// root whose parity equals the flag: the corrected rule (the reference's from_compressed selects on
// y.is_high() ^ flag, g1.rs:320, E8).  G2: 64 B = x.c1 || x.c0, bit 255 of the first byte = sgn0(y)
// (g2.rs:274-283, :309-340).  x == 0 <-> identity.
BN_FUNC void g1_compress(uint8_t* out, const G1A& a) {
  Fp x = fp_select(a.inf, fp_zero(), a.x), y = fp_select(a.inf, fp_one(), a.y);
  fp_to_be(out, x);
  out[0] |= (uint8_t)(fp_sgn0(y) << 7);
}
BN_FUNC G1A g1_decompress(const uint8_t* in, bool& ok) {
  uint8_t xb[32];
  for (int i = 0; i < 32; ++i) xb[i] = in[i];
  int flag = xb[0] >> 7; xb[0] &= 0x7f;
  bool okx, sq;
  G1A r;
  r.x = fp_from_be(xb, okx);
  r.inf = okx & fp_is_zero(r.x);
  Fp rhs = fp_norm(fp_add(fp_mul(fp_sqr(r.x), r.x), fp_const(bnc::THREE)));
  Fp y = fp_sqrt_cand(rhs, sq);
  bool flip = fp_sgn0(y) != flag;
  r.y = fp_select(flip, fp_norm(fp_neg(y)), y);
  ok = okx & (r.inf | sq);
  return r;
}
BN_FUNC void g2_compress(uint8_t* out, const G2A& a) {
  Fp2 x = fp2_select(a.inf, fp2_zero(), a.x), y = fp2_select(a.inf, fp2_one(), a.y);
  fp_to_be(out, x.c1); fp_to_be(out + 32, x.c0);
  out[0] |= (uint8_t)(fp2_sgn0(y) << 7);
}

// ------------------------------------------------------------------ SVDW maps
// Straight-line Shallue-van de Woestijne (RFC 9380 F.1), Z = 1, following fp.rs:292-370.  Of the reference's
// four exponentiations per map (inv0, two Euler tests, one square root) two remain: the Euler tests are Jacobi
// symbols (fp_is_square) and the inversion is shared between the two maps of hash_to_curve.
// Split in two so that the two maps of hash_to_curve can share one inversion: svdw_g1_den(u) is the value the
// map inverts, svdw_g1_finish(u, inv0(den)) the rest.
BN_FUNC Fp svdw_g1_den(const Fp& u_in) {
  BN_CTX;
  Fp u2 = fp_sqr(fp_norm(u_in));
  Fp tv1 = fp_lc2<4, 0>(u2, u2);                                 // c1 = g(Z) = 4
  return fp_mul(fp_norm(fp_sub(fp_one(), tv1)), fp_norm(fp_add(fp_one(), tv1)));
}
BN_FUNC G1A svdw_g1_finish(const Fp& u_in, const Fp& tv3) {       // tv3 = inv0(tv1 * tv2)
  BN_CTX;
  Fp u = fp_norm(u_in);
  Fp c2 = fp_const(bnc::SVDW1_C2), c3 = fp_const(bnc::SVDW1_C3), c4 = fp_const(bnc::SVDW1_C4), one = fp_one(), b = fp_const(bnc::THREE);
  Fp u2 = fp_sqr(u);
  Fp tv1 = fp_lc2<4, 0>(u2, u2);
  Fp tv2 = fp_norm(fp_add(one, tv1));
  tv1 = fp_norm(fp_sub(one, tv1));
  Fp tv4 = fp_mul(fp_mul(fp_mul(u, tv1), tv3), c3);
  Fp x1 = fp_norm(fp_sub(c2, tv4));
  Fp gx1 = fp_norm(fp_add(fp_mul(fp_sqr(x1), x1), b));
  Fp x2 = fp_norm(fp_add(c2, tv4));
  Fp gx2 = fp_norm(fp_add(fp_mul(fp_sqr(x2), x2), b));
  Fp x3 = fp_mul(fp_sqr(tv2), tv3);
  x3 = fp_norm(fp_add(fp_mul(fp_sqr(x3), c4), one));
  Fp gx3 = fp_norm(fp_add(fp_mul(fp_sqr(x3), x3), b));
  // the two is_square tests by Jacobi symbol (no exponentiation), then ONE square root of the selected g(x):
  // g(x3) is a square whenever g(x1) and g(x2) are not (the product of the three is one)
  bool e1 = fp_is_square(gx1), e2 = fp_is_square(gx2), sq;
  G1A r;
  r.x = fp_select(e1, x1, fp_select(e2, x2, x3));
  Fp y = fp_sqrt_cand(fp_select(e1, gx1, fp_select(e2, gx2, gx3)), sq);
  bool flip = fp_sgn0(u) != fp_sgn0(y);
  r.y = fp_select(flip, fp_norm(fp_neg(y)), y);
  r.inf = false;
  return r;
}
BN_FUNC G1A svdw_g1(const Fp& u) { return svdw_g1_finish(u, fp_inv(svdw_g1_den(u))); }
// The same map WITHOUT the inversion, for hash_to_curve (whose sum of two mapped points is projective anyway): every candidate
// x_i is kept as a fraction N_i / D_i --
//   x1, x2 = (c2 tv2 -+ c3 u) / tv2,    x3 = Z + c4 (tv2^2 tv3)^2 = (tv1^2 + c4 tv2^2) / tv1^2      (tv3 = 1 / (tv1 tv2))
// -- g(x_i) = (N_i^3 + 3 D_i^3) / D_i^3 is a square iff (N_i^3 + 3 D_i^3) D_i is (Jacobi symbol), and the affine y of the
// selected candidate comes from ONE power (sqrt_ratio, fp.rs:212-243 with v = D^3): y = U V (U V^3)^((p-3)/4), y^2 = U / V.
// Two powers per map (inv0 + square root) become one; the point is the reference's (same x, same y, same sign rule).
// inv0(0) = 0 in the reference (tv1 tv2 = 0): x1 = x2 = c2, x3 = Z = 1.
struct SvdwFrac { Fp n, d, y; };                         // x = n / d (d != 0), y affine
BN_FUNC SvdwFrac svdw_g1_frac(const Fp& u_in) {
  BN_CTX;
  const Fp u = fp_norm(u_in);
  const Fp c2 = fp_const(bnc::SVDW1_C2), c3 = fp_const(bnc::SVDW1_C3), c4 = fp_const(bnc::SVDW1_C4), one = fp_one();
  const Fp u2 = fp_sqr(u);
  Fp tv1 = fp_lc2<4, 0>(u2, u2);                                   // c1 = g(Z) = 4
  const Fp tv2 = fp_norm(fp_add(one, tv1));
  tv1 = fp_norm(fp_sub(one, tv1));
  const bool exc = fp_is_zero(tv1) | fp_is_zero(tv2);
  const Fp c3u = fp_mul(c3, u), c2t = fp_mul(c2, tv2);
  const Fp n1 = fp_select(exc, c2, fp_norm(fp_sub(c2t, c3u))), n2 = fp_select(exc, c2, fp_norm(fp_add(c2t, c3u)));
  const Fp d12 = fp_select(exc, one, tv2);
  const Fp t1s = fp_sqr(tv1);
  const Fp d3 = fp_select(exc, one, fp_norm(t1s));
  const Fp n3 = fp_select(exc, one, fp_norm(fp_add(t1s, fp_mul(c4, fp_sqr(tv2)))));
  const Fp v12 = fp_mul(fp_sqr(d12), d12), v3 = fp_mul(fp_sqr(d3), d3);        // D^3
  const Fp g1n = fp_norm(fp_add(fp_mul(fp_sqr(n1), n1), fp_lc2<3, 0>(v12, v12)));   // N^3 + 3 D^3
  const Fp g2n = fp_norm(fp_add(fp_mul(fp_sqr(n2), n2), fp_lc2<3, 0>(v12, v12)));
  const Fp g3n = fp_norm(fp_add(fp_mul(fp_sqr(n3), n3), fp_lc2<3, 0>(v3, v3)));
  // g(x3) is a square whenever g(x1) and g(x2) are not (the product of the three is one)
  const bool e1 = fp_is_square(fp_mul(g1n, d12)), e2 = fp_is_square(fp_mul(g2n, d12));
  SvdwFrac r;
  r.n = fp_select(e1, n1, fp_select(e2, n2, n3));
  r.d = fp_select(e1 | e2, d12, d3);
  const Fp U = fp_select(e1, g1n, fp_select(e2, g2n, g3n)), V = fp_select(e1 | e2, v12, v3);
  const Fp V3 = fp_mul(fp_sqr(V), V);
  const Fp y = fp_mul(fp_mul(U, V), fp_pow(fp_mul(U, V3), BN_EXP(EXP_PM3_4)));
  const bool flip = fp_sgn0(u) != fp_sgn0(y);
  r.y = fp_select(flip, fp_norm(fp_neg(y)), y);
  return r;
}
// hash_to_curve for G1 (g1.rs:910-919): map two field elements, add, no cofactor.  No inversion at all: the mapped points
// enter the complete addition as (N : y D : D); the sum stays in homogeneous coordinates.
BN_FUNC G1P hash_to_g1_from_fields_proj(const Fp& u0, const Fp& u1) {
  BN_CTX;
#if defined(__HIP_DEVICE_COMPILE__) && defined(BN_HASH_MAP_LOOP)
  // ONE copy of the map in the instruction stream (a two-trip loop the compiler may not unroll): the inlined map is ~150 KB of
  // code, and with two waves per SIMD drifting apart two copies thrash the instruction cache (k_hash_to_g1: 23 % of its wave
  // cycles waited for instructions, profiles/r03_pmc_summary.txt)
  Fp u[2] = {u0, u1};
  G1P q[2];
  _Pragma("unroll 1") for (int k = 0; k < 2; ++k) {
    const SvdwFrac a = svdw_g1_frac(u[k]);
    q[k] = G1P{a.n, fp_mul(a.y, a.d), a.d};
  }
  return proj_add(q[0], q[1]);
#else
  const SvdwFrac a = svdw_g1_frac(u0), b = svdw_g1_frac(u1);
  const G1P q0 = {a.n, fp_mul(a.y, a.d), a.d}, q1 = {b.n, fp_mul(b.y, b.d), b.d};
  return proj_add(q0, q1);
#endif
}
BN_FUNC G1A hash_to_g1_from_fields(const Fp& u0, const Fp& u1) { return g1_to_affine(hash_to_g1_from_fields_proj(u0, u1)); }

// Fp2 helpers for the G2 map
BN_FUNC Fp2 fp2_pow(const Fp2& a, Exp256 e) {
  BN_CTX;
  Fp2 base = fp2_norm(a), r = fp2_one();
  for (int i = 255; i >= 0; --i) {
    r = fp2_sqr(r);
    Fp2 m = fp2_mul(r, base);
    r = fp2_select((e.w[i >> 6] >> (i & 63)) & 1, m, r);
  }
  return r;
}
BN_FUNC bool fp2_is_square(const Fp2& a) {                  // fp2.rs:441-452: norm is a square in Fp
  Fp2 n = fp2_norm(a);
  return fp_is_square(fp_dot2(n.c0, n.c0, n.c1, n.c1));       // Jacobi symbol, no exponentiation
}
// Algorithm 9 of eprint 2012/685 (fp2.rs:172-218); returns a root when a is a square
BN_FUNC Fp2 fp2_sqrt(const Fp2& a_in) {
  BN_CTX;
  Fp2 a = fp2_norm(a_in);
  Fp2 a1 = fp2_pow(a, BN_EXP(EXP_PM3_4));
  Fp2 alpha = fp2_mul(fp2_sqr(a1), a);
  Fp2 x0 = fp2_mul(a1, a);
  bool neg_one = fp2_is_zero(fp2_add(alpha, fp2_one()));
  Fp2 alt = {fp_norm(fp_neg(x0.c1)), x0.c0};
  Fp2 b = fp2_pow(fp2_norm(fp2_add(alpha, fp2_one())), BN_EXP(EXP_PM1_2));
  Fp2 r = fp2_mul(b, x0);
  return fp2_select(neg_one, alt, r);
}
BN_FUNC G2A g2_decompress(const uint8_t* in, bool& ok) {
  uint8_t xb[32];
  for (int i = 0; i < 32; ++i) xb[i] = in[i];
  int flag = xb[0] >> 7; xb[0] &= 0x7f;
  bool o0, o1;
  G2A r;
  r.x.c1 = fp_from_be(xb, o0); r.x.c0 = fp_from_be(in + 32, o1);
  r.inf = o0 & o1 & fp2_is_zero(r.x);
  Fp2 rhs = fp2_norm(fp2_add(fp2_mul(fp2_sqr(r.x), r.x), fp2_const(bnc::B2)));
  Fp2 y = fp2_sqrt(rhs);
  bool sq = fp2_is_zero(fp2_sub(fp2_sqr(y), rhs));
  bool flip = fp2_sgn0(y) != flag;
  r.y = fp2_select(flip, fp2_norm(fp2_neg(y)), y);
  ok = o0 & o1 & (r.inf | sq);
  return r;
}
BN_FUNC G2A svdw_g2(const Fp2& u_in) {                      // fp2.rs:224-286
  Fp2 u = fp2_norm(u_in);
  Fp2 c1 = fp2_const(bnc::SVDW2_C1), c3 = fp2_const(bnc::SVDW2_C3), c4 = fp2_const(bnc::SVDW2_C4), one = fp2_one(), b = fp2_const(bnc::B2);
  Fp2 c2 = {fp_const(bnc::SVDW1_C2), fp_zero()};
  Fp2 tv1 = fp2_mul(fp2_sqr(u), c1);
  Fp2 tv2 = fp2_norm(fp2_add(one, tv1));
  tv1 = fp2_norm(fp2_sub(one, tv1));
  Fp2 tv3 = fp2_inv(fp2_mul(tv1, tv2));
  Fp2 tv4 = fp2_mul(fp2_mul(fp2_mul(u, tv1), tv3), c3);
  Fp2 x1 = fp2_norm(fp2_sub(c2, tv4));
  Fp2 gx1 = fp2_norm(fp2_add(fp2_mul(fp2_sqr(x1), x1), b));
  Fp2 x2 = fp2_norm(fp2_add(c2, tv4));
  Fp2 gx2 = fp2_norm(fp2_add(fp2_mul(fp2_sqr(x2), x2), b));
  Fp2 x3 = fp2_mul(fp2_sqr(tv2), tv3);
  x3 = fp2_norm(fp2_add(fp2_mul(fp2_sqr(x3), c4), one));
  bool e1 = fp2_is_square(gx1), e2 = fp2_is_square(gx2);
  G2A r;
  r.x = fp2_select(e1, x1, fp2_select(e2, x2, x3));
  Fp2 gx = fp2_norm(fp2_add(fp2_mul(fp2_sqr(r.x), r.x), b));
  Fp2 y = fp2_sqrt(gx);
  bool flip = fp2_sgn0(u) != fp2_sgn0(y);
  r.y = fp2_select(flip, fp2_norm(fp2_neg(y)), y);
  r.inf = false;
  return r;
}

}  // namespace bn
