// lane_ops.h -- what ONE lane does for ONE tuple in each kernel of the verification path.
// The __global__ wrappers in kernels.hip only compute the lane's tuple index and call these; the
// same functions are compiled for the host by tests/hostsim (with -DBN_CHECK) to prove the lazy-limb
// interval discipline and to debug against the oracle without a GPU.  They are NOT a CPU fallback:
// nothing in the product's host path calls them.
//
// Workspace layout ("limb-major"): element k of tuple i lives at ws[k * stride + i], so the 64 lanes
// of a wave touch 256 consecutive bytes per limb (fully coalesced global_load/store_dword).
#pragma once
#include "pairing.h"
#include "sha256.h"

namespace bn {

enum : uint8_t { FLAG_SIG_OK = 1, FLAG_PK_OK = 2, FLAG_PK_SUBGROUP = 4, FLAG_IDENTITY = 8 };

BN_FUNC void store_fp(int32_t* ws, size_t stride, const Fp& a) {
  Fp c = fp_canon(a);
  for (int k = 0; k < NL; ++k) ws[(size_t)k * stride] = c.l[k];
}
BN_FUNC Fp load_fp(const int32_t* ws, size_t stride) {
  Fp r;
  for (int k = 0; k < NL; ++k) r.l[k] = ws[(size_t)k * stride];
  BN_TRK(set_trk(r, 0, 1, 0, 0.006, 1);)
  return r;
}

// ---- hash_to_g1: msg -> H(msg) in G1 (g1.rs:910-919).  out: 18 limbs (x, y), Montgomery canonical.
BN_FUNC G1A lane_hash_to_g1(const uint8_t* msg, size_t msg_len, const uint8_t* dst, uint32_t dst_len) {
  uint8_t okm[96];
  expand_message_xmd(okm, 96, msg, msg_len, dst, dst_len);
  return hash_to_g1_from_fields(fp_from_okm(okm), fp_from_okm(okm + 48));
}
// the same point in homogeneous coordinates (X : Y : Z), without the inversion of the affine form: the table-only Miller
// loop evaluates its lines at (X, Y, Z) directly (the factor Z per line pair lies in Fp and dies in the final exponentiation)
BN_FUNC G1P lane_hash_to_g1_proj(const uint8_t* msg, size_t msg_len, const uint8_t* dst, uint32_t dst_len) {
  uint8_t okm[96];
  expand_message_xmd(okm, 96, msg, msg_len, dst, dst_len);
  return hash_to_g1_from_fields_proj(fp_from_okm(okm), fp_from_okm(okm + 48));
}
BN_FUNC G1A lane_encode_to_g1(const uint8_t* msg, size_t msg_len, const uint8_t* dst, uint32_t dst_len) {   // g1.rs:922-928
  uint8_t okm[48];
  expand_message_xmd(okm, 48, msg, msg_len, dst, dst_len);
  return svdw_g1(fp_from_okm(okm));
}
BN_FUNC G2A lane_hash_to_g2(const uint8_t* msg, size_t msg_len, const uint8_t* dst, uint32_t dst_len, bool ro) {  // g2.rs:919-936
  uint8_t okm[192];
  expand_message_xmd(okm, ro ? 192 : 96, msg, msg_len, dst, dst_len);
  Fp2 u0 = {fp_from_okm(okm), fp_from_okm(okm + 48)};
  G2P q = proj_from_affine(svdw_g2(u0));
  if (ro) {
    Fp2 u1 = {fp_from_okm(okm + 96), fp_from_okm(okm + 144)};
    q = proj_add(q, proj_from_affine(svdw_g2(u1)));
  }
  return g2_to_affine(g2_clear_cofactor(q));
}

// ---- point checks
BN_FUNC bool lane_g1_check(const uint8_t* g1) { bool ok; G1A p = g1_decode(g1, ok); return ok & g1_on_curve(p); }
BN_FUNC bool lane_g2_check(const uint8_t* g2) {
  bool ok; G2A q = g2_decode(g2, ok);
  return ok & g2_on_curve(q) & g2_torsion_free(q);
}

// ---- Miller loops.  status bit0: g1 decoded, bit1: g2 decoded, bit2: either point is the identity.
// ws (device kernels, -DBN_MILLER1_WS): this lane's LDS column of 108 limbs -- invariants (54) and the parked running point (54)
BN_FUNC Fp12 lane_miller_1(const uint8_t* g1, const uint8_t* g2, uint8_t& status, const Ws* ws = nullptr) {
  bool ok1, ok2;
  G1A p = g1_decode(g1, ok1);
  G2A q = g2_decode(g2, ok2);
  bool ident = p.inf | q.inf;
  status = (uint8_t)((ok1 ? 1 : 0) | (ok2 ? 2 : 0) | (ident ? 4 : 0));
  // identity or undecodable operands: run the loop on the generators (uniform control flow), result replaced by 1
  bool bad = ident | !ok1 | !ok2;
  G1A gp; gp.x = fp_one(); gp.y = fp_add(fp_one(), fp_one()); gp.inf = false;
  gp.y = fp_norm(gp.y);
  G2A gq; gq.x = fp2_const(bnc::G2_GEN_X); gq.y = fp2_const(bnc::G2_GEN_Y); gq.inf = false;
  p.x = fp_select(bad, gp.x, p.x); p.y = fp_select(bad, gp.y, p.y);
  q.x = fp2_select(bad, gq.x, q.x); q.y = fp2_select(bad, gq.y, q.y);
#ifdef BN_MILLER1_WS
  fp_store_mem(*ws, fp_norm(p.x)); fp_store_mem(ws_at(*ws, 9), fp_norm(p.y));
  fp2_store_mem(ws_at(*ws, 18), fp2_norm(q.x)); fp2_store_mem(ws_at(*ws, 36), fp2_norm(q.y));
  BN_MEM_FENCE;
  Fp12 f = miller_loop_1_ws(*ws, ws_at(*ws, 54));
#else
  Fp12 f = miller_loop_1(p, q);
#endif
  Fp12 one = fp12_one();
  f.c0 = {fp2_select(bad, one.c0.c0, f.c0.c0), fp2_select(bad, one.c0.c1, f.c0.c1), fp2_select(bad, one.c0.c2, f.c0.c2)};
  f.c1 = {fp2_select(bad, one.c1.c0, f.c1.c0), fp2_select(bad, one.c1.c1, f.c1.c1), fp2_select(bad, one.c1.c2, f.c1.c2)};
  return f;
}
// verify: f = ML(sig, -G2gen) * ML(H, pk); flags = FLAG_SIG_OK | FLAG_PK_OK when decodable, on curve, non-identity
// Variant that first writes the validated operands to the limb-major workspace `inv` (72 limbs per tuple) and
// runs the loop that re-loads them per use (miller_loop_verify_ws).
BN_FUNC Fp12 lane_miller_verify_ws(const uint8_t* pk_b, const uint8_t* sig_b, const G1A& h,
                                   const int32_t (*table)[54], uint8_t& flags, const Ws& inv) {
  bool oks, okp;
  G1A sig = g1_decode(sig_b, oks);
  G2A pk = g2_decode(pk_b, okp);
  bool sig_ok = oks & !sig.inf & g1_on_curve(sig);
  bool pk_ok = okp & !pk.inf & g2_on_curve(pk);
  flags = (uint8_t)((sig_ok ? FLAG_SIG_OK : 0) | (pk_ok ? FLAG_PK_OK : 0));
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one())); gp.inf = false;
  fp_store_mem(inv, fp_norm(fp_select(sig_ok, sig.x, gp.x))); fp_store_mem(ws_at(inv, 9), fp_norm(fp_select(sig_ok, sig.y, gp.y)));
  fp_store_mem(ws_at(inv, 18), fp_norm(h.x)); fp_store_mem(ws_at(inv, 27), fp_norm(h.y));
  fp2_store_mem(ws_at(inv, 36), fp2_norm(fp2_select(pk_ok, pk.x, fp2_const(bnc::G2_GEN_X))));
  fp2_store_mem(ws_at(inv, 54), fp2_norm(fp2_select(pk_ok, pk.y, fp2_const(bnc::G2_GEN_Y))));
  BN_MEM_FENCE;
#ifdef BN_VERIFY_PARK_T
  return miller_loop_verify_ws2(inv, ws_at(inv, 72), table);
#else
  return miller_loop_verify_ws(inv, table);
#endif
}
BN_FUNC Fp12 lane_miller_verify(const uint8_t* pk_b, const uint8_t* sig_b, const G1A& h,
                                     const int32_t (*table)[54], uint8_t& flags) {
  bool oks, okp;
  G1A sig = g1_decode(sig_b, oks);
  G2A pk = g2_decode(pk_b, okp);
  bool sig_ok = oks & !sig.inf & g1_on_curve(sig);
  bool pk_ok = okp & !pk.inf & g2_on_curve(pk);
  flags = (uint8_t)((sig_ok ? FLAG_SIG_OK : 0) | (pk_ok ? FLAG_PK_OK : 0));
  // invalid operands are replaced by the generators so that every lane runs the same arithmetic on
  // well-formed values; the flag already makes the tuple invalid
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one())); gp.inf = false;
  sig.x = fp_select(sig_ok, sig.x, gp.x); sig.y = fp_select(sig_ok, sig.y, gp.y);
  pk.x = fp2_select(pk_ok, pk.x, fp2_const(bnc::G2_GEN_X)); pk.y = fp2_select(pk_ok, pk.y, fp2_const(bnc::G2_GEN_Y));
  return miller_loop_verify(sig, h, pk, table);
}

}  // namespace bn
