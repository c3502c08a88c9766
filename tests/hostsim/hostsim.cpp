// hostsim.cpp -- compiles the DEVICE arithmetic headers (bls-bn254_amd/csrc/*.h) for the host with
// -DBN_CHECK: every Fp then carries its proven limb interval / value bound and every multiply asserts
// its precondition (fp29.h "interval discipline").  Used by tests/test_hostsim.py to (1) prove the
// lazy-limb bounds of every code path the kernels run and (2) compare that code with the oracle
// without a GPU.  TEST TOOL ONLY: never linked into the product.
#define BN_WANT_LINE_TABLE
#define BN_LINE_TABLE_QUAL static const
#include "../../bls-bn254_amd/csrc/lane_ops.h"
#include "../../bls-bn254_amd/csrc/keygen.h"
#include "../../bls-bn254_amd/csrc/glv.h"
#include "../../bls-bn254_amd/csrc/tri.h"
#include "../../bls-bn254_amd/csrc/quad.h"
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include "../../bls-bn254_amd/csrc/wide.h"
#include <vector>
#include <cstring>

using namespace bn;

// ---- tri.h on the host: the four lanes of a quad run as four threads; a DPP fetch is a rendezvous (publish, barrier, read, barrier)
namespace {
struct TriQuad {
  bn::Fp slot[4];
  std::mutex m; std::condition_variable cv; int arrived = 0; long gen = 0;
  void barrier() {
    std::unique_lock<std::mutex> lk(m);
    const long g = gen;
    if (++arrived == 4) { arrived = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};
thread_local TriQuad* tri_quad = nullptr;
thread_local uint32_t tri_quad_role = 0;
// run fn(role) on the four lanes; results are whatever fn writes
void tri_run(const std::function<void(uint32_t)>& fn) {
  TriQuad q;
  std::thread th[4];
  for (uint32_t r = 0; r < 4; ++r) th[r] = std::thread([&, r] { tri_quad = &q; tri_quad_role = r; fn(r); });
  for (auto& t : th) t.join();
}
}  // namespace
namespace bn {
void tri_host_run4(void (*fn)(void*, uint32_t), void* arg) { tri_run([&](uint32_t role) { fn(arg, role); }); }
uint32_t tri_host_role() { return tri_quad_role; }
Fp tri_host_fetch(const Fp& x, int p0, int p1, int p2, int p3) {
  const int perm[4] = {p0, p1, p2, p3};
  tri_quad->slot[tri_quad_role] = x;
  tri_quad->barrier();
  Fp r = tri_quad->slot[perm[tri_quad_role]];
  tri_quad->barrier();
  return r;
}
}  // namespace bn

extern "C" {

void hs_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  bool o1, o2;
  Fp x = fp_from_be(a, o1), y = fp_from_be(b, o2);
  fp_to_be(out, fp_mul(x, y));
}
void hs_fp_sqr(const uint8_t* a, uint8_t* out) { bool o; Fp x = fp_from_be(a, o); fp_to_be(out, fp_sqr(x)); }
void hs_fp_inv(const uint8_t* a, uint8_t* out) { bool o; Fp x = fp_from_be(a, o); fp_to_be(out, fp_inv(x)); }
void hs_fp_inv_pow(const uint8_t* a, uint8_t* out) { bool o; Fp x = fp_from_be(a, o); fp_to_be(out, fp_inv_pow(x)); }
// fp_inv (divstep recurrence) on a LAZY operand: x = a - b with limbs anywhere in the lazy range, as callers pass it
void hs_fp_inv_lazy(const uint8_t* a, const uint8_t* b, uint8_t* out) { bool o; Fp x = fp_sub(fp_from_be(a, o), fp_from_be(b, o)); fp_to_be(out, fp_inv(x)); }
// ((a+b)*(a-b) + 9a - b) exercising lazy ops
void hs_fp_mix(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  bool o1, o2;
  Fp x = fp_from_be(a, o1), y = fp_from_be(b, o2);
  Fp t = fp_mul(fp_add(x, y), fp_sub(x, y));
  fp_to_be(out, fp_lc3<1, 9, -1>(t, x, y));
}
int hs_fp_is_square(const uint8_t* a) { bool o; return fp_is_square(fp_from_be(a, o)) ? 1 : 0; }
int hs_fp_decode_ok(const uint8_t* a) { bool o; (void)fp_from_be(a, o); return o; }
void hs_fp_from_okm(const uint8_t* okm, uint8_t* out) { fp_to_be(out, fp_from_okm(okm)); }
void hs_fr_from_okm(const uint8_t* okm, uint8_t* out) { fr_to_be(out, fr_from_okm(okm)); }
int hs_keygen(const uint8_t* ikm, size_t ikm_len, const uint8_t* info, size_t info_len, uint8_t* out) {
  bool ok; Fr sk = lane_keygen(ikm, ikm_len, info, info_len, ok); fr_to_be(out, sk); return ok;
}
void hs_hash_to_scalar(const uint8_t* msg, size_t len, const uint8_t* dst, uint32_t dst_len, uint8_t* out) {
  fr_to_be(out, lane_hash_to_scalar(msg, len, dst, dst_len));
}

void hs_miller1(const uint8_t* g1, const uint8_t* g2, uint8_t* out, int* status) {
  uint8_t st;
  Fp12 f = lane_miller_1(g1, g2, st);
  *status = st;
  fp12_to_be(out, f);
}
// the parked loops of the device kernels on host memory (stride 1): one pair (k_miller_1), two variable pairs sharing f^2
// (k_miller_hpk2), prepared keys (k_g2_prepare + k_g2_expand + k_miller_prepared, and two prepared pairs: k_miller_hpk2p)
void hs_miller1_ws(const uint8_t* g1, const uint8_t* g2, uint8_t* out) {
  bool o1, o2;
  G1A p = g1_decode(g1, o1); G2A q = g2_decode(g2, o2);
  static int32_t mem[108];
  const Ws ws = {mem, 1, 0, false};
  fp_store_mem(ws, fp_norm(p.x)); fp_store_mem(ws_at(ws, 9), fp_norm(p.y));
  fp2_store_mem(ws_at(ws, 18), fp2_norm(q.x)); fp2_store_mem(ws_at(ws, 36), fp2_norm(q.y));
  fp12_to_be(out, miller_loop_1_ws(ws, ws_at(ws, 54)));
}
// ML(Ha, Qa) * ML(Hb, Qb): by the two-variable-pair loop (out1), by prepared raw tables (out2); live_b = 0 pads the second pair
void hs_miller2(const uint8_t* ha, const uint8_t* qa, const uint8_t* hb, const uint8_t* qb, int live_b, uint8_t* out1, uint8_t* out2) {
  bool ok;
  G1A Ha = g1_decode(ha, ok), Hb = g1_decode(hb, ok);
  G2A Qa = g2_decode(qa, ok), Qb = g2_decode(qb, ok);
  static int32_t hh[36], qw[72], park[54], lpark[54], ta[88 * 54], tb[88 * 54];
  const Ws whh = {hh, 1, 0, false}, wqw = {qw, 1, 0, false};
  fp_store_mem(whh, fp_norm(Ha.x)); fp_store_mem(ws_at(whh, 9), fp_norm(Ha.y));
  fp_store_mem(ws_at(whh, 18), fp_norm(Hb.x)); fp_store_mem(ws_at(whh, 27), live_b ? fp_norm(Hb.y) : fp_one());
  fp2_store_mem(wqw, fp2_norm(Qa.x)); fp2_store_mem(ws_at(wqw, 18), fp2_norm(Qa.y));
  fp2_store_mem(ws_at(wqw, 36), fp2_norm(Qb.x)); fp2_store_mem(ws_at(wqw, 54), fp2_norm(Qb.y));
  fp12_to_be(out1, miller_loop_2var_ws(whh, wqw, Ws{park, 1, 0, false}, Ws{lpark, 1, 0, false}, true, live_b != 0));
  g2_prepare_lines(Qa, Ws{ta, 1, 0, false}); g2_prepare_lines(Qb, Ws{tb, 1, 0, false});
  fp12_to_be(out2, miller_loop_2prepared(whh, Ws{ta, 1, 0, false}, Ws{tb, 1, 0, false}, true, live_b != 0));
}
// ML(sig, -G2gen) * ML(H, pk) by the pair tables of the prepared-key verify path, H given homogeneously as (x z : y z : z)
// with z = the Montgomery form of z_small; the loop's value is the textbook Miller product times z^88, so out = the value
// after the final exponentiation (which removes the Fp factor), and z_small = 1 gives the textbook Miller value in ml_out
void hs_miller_prepared(const uint8_t* sig, const uint8_t* h, const uint8_t* pk, int z_small, uint8_t* ml_out, uint8_t* gt_out) {
  bool ok;
  G1A S = g1_decode(sig, ok), H = g1_decode(h, ok);
  G2A Q = g2_decode(pk, ok);
  static int32_t raw[88 * 54], exp_[88 * 162], inv[81];
  g2_prepare_lines(Q, Ws{raw, 1, 0, false});
  for (int t = 0; t < 88; ++t)
    line_pair_expand(line_from_table(BN_NEG_G2_LINE_TABLE[t]), line_load_limbs(Ws{raw + 54 * t, 1, 0, false}), Ws{exp_ + 162 * t, 1, 0, false});
  Fp z = fp_one();
  for (int k = 1; k < z_small; ++k) z = fp_norm(fp_add(z, fp_one()));
  z = fp_canon(z);
  const Ws w = {inv, 1, 0, false};
  Fp xs = fp_norm(S.x), ys = fp_norm(S.y), X = fp_mul(fp_norm(H.x), z), Y = fp_mul(fp_norm(H.y), z), Z = z;
  fp_store_mem(w, X); fp_store_mem(ws_at(w, 9), Y); fp_store_mem(ws_at(w, 18), Z);
  fp_store_mem(ws_at(w, 27), fp_mul(xs, X)); fp_store_mem(ws_at(w, 36), fp_mul(ys, Y)); fp_store_mem(ws_at(w, 45), fp_mul(xs, Z));
  fp_store_mem(ws_at(w, 54), fp_mul(ys, Z)); fp_store_mem(ws_at(w, 63), fp_mul(ys, X)); fp_store_mem(ws_at(w, 72), fp_mul(xs, Y));
  Fp12 f = miller_loop_prepared(w, Ws{exp_, 1, 0, false});
  fp12_to_be(ml_out, f);
  fp12_to_be(gt_out, final_exponentiation(f));
}
int hs_final_exp(const uint8_t* in, uint8_t* out) {
  bool ok;
  Fp12 f = fp12_from_be(in, ok);
  if (!ok) return 4;
  fp12_to_be(out, final_exponentiation(f));
  return 0;
}
// the easy part split at its Fp inversion (k_fe_easy_head / k_fe_inv4 / k_fe_easy_tail) == fe_easy; the inversion is taken
// with three other values in the same fp_inv4 call (a zero among them must stay zero and must not disturb the others)
int hs_fe_easy_split_matches(const uint8_t* in, int slot) {
  bool ok;
  Fp12 f = fp12_from_be(in, ok);
  if (!ok) return -1;
  int32_t head[72], nu[9];
  const FeEasyHead h = fe_easy_head(f);
  fp2_store_limbs(head, 1, h.c0); fp2_store_limbs(head + 18, 1, h.c1); fp2_store_limbs(head + 36, 1, h.c2); fp2_store_limbs(head + 54, 1, h.nrm);
  store_fp(nu, 1, h.nu);
  Fp x[4];
  for (int t = 0; t < 4; ++t) x[t] = t == slot ? load_fp(nu, 1) : t == (slot + 1) % 4 ? fp_zero() : t == (slot + 2) % 4 ? fp_one() : fp_mul(load_fp(nu, 1), load_fp(nu, 1));
  fp_inv4(x);
  if (!fp_is_zero(x[(slot + 1) % 4]) || !fp_eq(x[(slot + 2) % 4], fp_one())) return -2;
  if (!fp_eq(fp_mul(fp_mul(x[(slot + 3) % 4], load_fp(nu, 1)), load_fp(nu, 1)), fp_one())) return -3;
  int32_t inv[9];
  store_fp(inv, 1, x[slot]);
  FeEasyHead g;
  g.c0 = fp2_load_limbs(head, 1); g.c1 = fp2_load_limbs(head + 18, 1); g.c2 = fp2_load_limbs(head + 36, 1); g.nrm = fp2_load_limbs(head + 54, 1);
  g.nu = fp_zero();
  uint8_t a[384], b[384];
  fp12_to_be(a, fe_easy_tail(f, g, load_fp(inv, 1)));
  fp12_to_be(b, fe_easy(f));
  return std::memcmp(a, b, 384) == 0 ? 1 : 0;
}
// wide.h (one workgroup of 128 lanes per tuple) under the interval checker: the lanes of a phase run one after the other over a host
// array standing in for the wave's LDS region.  Final exponentiation = serial easy part + wide hard part.
static void wide_put(const Wide& W, uint32_t v, const Fp12& f) {
  const Fp6* h[2] = {&f.c0, &f.c1};
  for (int j = 0; j < 2; ++j) {
    fp2_store_mem(wide_val(W, v, 3 * j + 0), fp2_norm(h[j]->c0)); fp2_store_mem(wide_val(W, v, 3 * j + 1), fp2_norm(h[j]->c1));
    fp2_store_mem(wide_val(W, v, 3 * j + 2), fp2_norm(h[j]->c2));
  }
}
static Fp12 wide_get(const Wide& W, uint32_t v) {
  return {{fp2_load_mem(wide_val(W, v, 0)), fp2_load_mem(wide_val(W, v, 1)), fp2_load_mem(wide_val(W, v, 2))},
          {fp2_load_mem(wide_val(W, v, 3)), fp2_load_mem(wide_val(W, v, 4)), fp2_load_mem(wide_val(W, v, 5))}};
}
int hs_final_exp_wide(const uint8_t* in, uint8_t* out) {
  bool ok;
  Fp12 f = fp12_from_be(in, ok);
  if (!ok) return 4;
  std::vector<int32_t> lds(WIDE_LDS_DWORDS, 0);
  Wide W{lds.data()};
  wide_put(W, WV_T, fe_easy(f));
  wide_fe_hard(W);
  fp12_to_be(out, wide_get(W, WV_R));
  return 0;
}
// single primitives against their serial counterparts: op 0 a*b, 1 cyclotomic square of a (a must be cyclotomic), 2 conj, 4..6 Frobenius^1..3
int hs_wide_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out_wide, uint8_t* out_serial) {
  bool ok1, ok2 = true;
  Fp12 x = fp12_from_be(a, ok1), y = b ? fp12_from_be(b, ok2) : fp12_one();
  if (!ok1 || !ok2) return 4;
  std::vector<int32_t> lds(WIDE_LDS_DWORDS, 0);
  Wide W{lds.data()};
  wide_put(W, WV_A, x); wide_put(W, WV_B, y);
  Fp12 r;
  switch (op) {
    case 0: wide_exec(W, WOP_MUL, WV_R, WV_A, WV_B); r = fp12_mul(x, y); break;
    case 1: wide_exec(W, WOP_COPY, WV_R, WV_A, 0); wide_exec(W, WOP_SQR, WV_R, 1, 0); r = fp12_cyclotomic_sqr(x); break;
    case 2: wide_exec(W, WOP_CONJ, WV_R, WV_A, 0); r = fp12_conj(x); break;
    case 4: wide_exec(W, WOP_FROB1, WV_R, WV_A, 0); r = fp12_frob<1>(x); break;
    case 5: wide_exec(W, WOP_FROB2, WV_R, WV_A, 0); r = fp12_frob<2>(x); break;
    case 6: wide_exec(W, WOP_FROB3, WV_R, WV_A, 0); r = fp12_frob<3>(x); break;
    default: return -1;
  }
  fp12_to_be(out_wide, wide_get(W, WV_R));
  fp12_to_be(out_serial, r);
  return 0;
}
// wide Miller loops over prepared keys against the serial loops of pairing.h: one pair from the raw line table (out[0] wide,
// out[1] serial) and the verify pair (sig, -G2gen) x (H, pk) from the pair table (out[2] wide, out[3] serial); 4 x 384 bytes
int hs_miller_wide(const uint8_t* sig, const uint8_t* h, const uint8_t* pk, uint8_t* out) {
  bool o1, o2, o3;
  G1A s = g1_decode(sig, o1), hp = g1_decode(h, o2);
  G2A q = g2_decode(pk, o3);
  if (!o1 || !o2 || !o3) return 1;
  std::vector<int32_t> raw(BN_NEG_G2_LINES * 54), exp_(BN_NEG_G2_LINES * 162), lds(WIDE_LDS_DWORDS, 0), col(256, 0);
  const Ws rw = {raw.data(), 1, 0, false}, ew = {exp_.data(), 1, 0, false};
  g2_prepare_lines(q, rw);
  for (int t = 0; t < BN_NEG_G2_LINES; ++t)
    line_pair_expand(line_from_table(BN_NEG_G2_LINE_TABLE[t]), line_load_limbs(ws_at(rw, 54 * (size_t)t)), ws_at(ew, 162 * (size_t)t));
  Wide W{lds.data()};
  const Ws pt = {col.data(), 1, 0, false};
  fp_store_mem(pt, fp_norm(hp.x)); fp_store_mem(ws_at(pt, 9), fp_norm(hp.y));
  wide_miller_prepared(W, rw, pt, false);
  fp12_to_be(out, wide_get(W, WV_R));
  fp12_to_be(out + 384, miller_loop_1prepared(pt, rw));
  const Fp xs = fp_norm(s.x), ys = fp_norm(s.y), X = fp_norm(hp.x), Y = fp_norm(hp.y), Z = fp_one();
  const Ws cw = {col.data() + 32, 1, 0, false};
  fp_store_mem(cw, X); fp_store_mem(ws_at(cw, 9), Y); fp_store_mem(ws_at(cw, 18), Z);
  fp_store_mem(ws_at(cw, 27), fp_mul(xs, X)); fp_store_mem(ws_at(cw, 36), fp_mul(ys, Y)); fp_store_mem(ws_at(cw, 45), fp_mul(xs, Z));
  fp_store_mem(ws_at(cw, 54), fp_mul(ys, Z)); fp_store_mem(ws_at(cw, 63), fp_mul(ys, X)); fp_store_mem(ws_at(cw, 72), fp_mul(xs, Y));
  wide_miller_prepared(W, ew, cw, true);
  fp12_to_be(out + 768, wide_get(W, WV_R));
  fp12_to_be(out + 1152, miller_loop_prepared(cw, ew));
  return 0;
}
// wide one-pair loop with a variable G2 point (lane 0 computes the lines) against miller_loop_1: out[0] wide, out[1] serial
int hs_miller_wide_var(const uint8_t* g1, const uint8_t* g2, uint8_t* out) {
  bool o1, o2;
  G1A p = g1_decode(g1, o1);
  G2A q = g2_decode(g2, o2);
  if (!o1 || !o2) return 1;
  std::vector<int32_t> lds(WIDE_LDS_DWORDS, 0), col(128, 0);
  Wide W{lds.data()};
  const Ws pt = {col.data(), 1, 0, false}, lnw = {col.data() + 32, 1, 0, false};
  fp_store_mem(pt, fp_norm(p.x)); fp_store_mem(ws_at(pt, 9), fp_norm(p.y));
  q.x = fp2_norm(q.x); q.y = fp2_norm(q.y);
  wide_miller_1(W, q, pt, lnw);
  fp12_to_be(out, wide_get(W, WV_R));
  fp12_to_be(out + 384, miller_loop_1(p, q));
  return 0;
}
void hs_pairing(const uint8_t* g1, const uint8_t* g2, uint8_t* out) {
  uint8_t st;
  Fp12 f = lane_miller_1(g1, g2, st);
  fp12_to_be(out, final_exponentiation(f));
}
int hs_expx_chain_matches(const uint8_t* in) {      // addition chain == binary ladder on a cyclotomic element
  bool ok;
  Fp12 t = fe_easy(fp12_from_be(in, ok));
  static int32_t slots[108 * 10];
  static int32_t park[108];
  const Ws pk = {park, 1, 0, false};                 // the kernel's path: partial products parked (LDS on the device)
  Fp12 a = cyclotomic_exp_x(t), c = cyclotomic_exp_x_chain(t, Ws{slots, 1, 0, false}, &pk);
  Fp12 c2 = cyclotomic_exp_x_chain(t, Ws{slots, 1, 0, false});
  uint8_t bc2[384]; fp12_to_be(bc2, c2);
  uint8_t ba[384], bc[384];
  fp12_to_be(ba, a); fp12_to_be(bc, c);
  return std::memcmp(ba, bc, 384) == 0 && std::memcmp(ba, bc2, 384) == 0;
}
int hs_fe_h3_loop_matches(const uint8_t* in) {      // interpreter form of h3 (the kernel's) == register form, on a full final exponentiation
  bool ok;
  Fp12 t = fe_easy(fp12_from_be(in, ok)), a, b, c, b2, d2;
  fe_h1(cyclotomic_exp_x(t), a, b);
  fe_h2(cyclotomic_exp_x(b), b, c, b2, d2);
  Fp12 x0 = cyclotomic_exp_x(d2);
  static int32_t ws[108 * 5], tmp[108 * 4], park[108];
  const Fp12* vals[5] = {&t, &a, &c, &b2, &x0};
  Ws w[5];
  for (int k = 0; k < 5; ++k) {                      // canonical limbs, as fp12_store_limbs leaves them in the phase buffers
    w[k] = Ws{ws + 108 * k, 1, 0, false};
    fp12_store_limbs(ws + 108 * k, 1, *vals[k]);
    fp12_store_mem(w[k], fp12_load_limbs(ws + 108 * k, 1));
  }
  const Ws pk = {park, 1, 0, false};
  Fp12 r1 = fe_h3(t, a, c, b2, x0), r2 = fe_h3_loop(w, Ws{tmp, 1, 0, false}, &pk);
  uint8_t b1[384], bb[384];
  fp12_to_be(b1, r1); fp12_to_be(bb, r2);
  return std::memcmp(b1, bb, 384) == 0;
}
int hs_fe_tails_match(const uint8_t* in) {          // fe_h1 / fe_h2 computed as tails of the t^x kernels == the separate steps
  bool ok;
  Fp12 t = fe_easy(fp12_from_be(in, ok)), a, b, c, b2, d2;
  Fp12 x1 = cyclotomic_exp_x(t);
  fe_h1(x1, a, b);
  Fp12 x2 = cyclotomic_exp_x(b);
  fe_h2(x2, b, c, b2, d2);
  static int32_t park[108], e[5][108];
  const Ws pk = {park, 1, 0, false};
  Ws w[5] = {{e[0], 1, 0, false}, {e[1], 1, 0, false}, {e[2], 1, 0, false}, {e[3], 1, 0, false}, {e[4], 1, 0, false}};
  uint8_t want[384], got[384];
  bool same = true;
  fe_h1_tail(x1, w[0], w[1], &pk);                                         // a, b
  fp12_to_be(want, a); fp12_to_be(got, fp12_load_mem(w[0])); same &= std::memcmp(want, got, 384) == 0;
  fp12_to_be(want, b); fp12_to_be(got, fp12_load_mem(w[1])); same &= std::memcmp(want, got, 384) == 0;
  fe_h2_tail(x2, w[1], w[2], w[3], w[4], &pk);                             // c, b2, d2
  fp12_to_be(want, c); fp12_to_be(got, fp12_load_mem(w[2])); same &= std::memcmp(want, got, 384) == 0;
  fp12_to_be(want, b2); fp12_to_be(got, fp12_load_mem(w[3])); same &= std::memcmp(want, got, 384) == 0;
  fp12_to_be(want, d2); fp12_to_be(got, fp12_load_mem(w[4])); same &= std::memcmp(want, got, 384) == 0;
  return same;
}
void hs_fp12_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  bool o1, o2;
  Fp12 x = fp12_from_be(a, o1), y = fp12_from_be(b, o2);
  fp12_to_be(out, fp12_mul(x, y));
}
void hs_hash_to_g1(const uint8_t* msg, size_t len, const uint8_t* dst, uint32_t dst_len, int ro, uint8_t* out) {
  G1A h = ro ? lane_hash_to_g1(msg, len, dst, dst_len) : lane_encode_to_g1(msg, len, dst, dst_len);
  g1_encode(out, h);
}
void hs_g1_from_fields(const uint8_t* u0, const uint8_t* u1, uint8_t* out) {      // the two-map path with its shared inversion
  bool o0, o1;
  g1_encode(out, hash_to_g1_from_fields(fp_from_be(u0, o0), fp_from_be(u1, o1)));
}
void hs_hash_to_g2(const uint8_t* msg, size_t len, const uint8_t* dst, uint32_t dst_len, int ro, uint8_t* out) {
  g2_encode(out, lane_hash_to_g2(msg, len, dst, dst_len, ro != 0));
}
int hs_g1_check(const uint8_t* g1) { return lane_g1_check(g1); }
int hs_g2_check(const uint8_t* g2) { return lane_g2_check(g2); }
void hs_g1_add(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  bool o1, o2;
  G1A p = g1_decode(a, o1), q = g1_decode(b, o2);
  g1_encode(out, g1_to_affine(proj_add(proj_from_affine(p), proj_from_affine(q))));
}
void hs_g1_mul(const uint8_t* a, const uint8_t* k_be, uint8_t* out) {
  bool o1;
  G1A p = g1_decode(a, o1);
  uint64_t k[4];
  for (int i = 0; i < 4; ++i) { uint64_t v = 0; for (int j = 0; j < 8; ++j) v = (v << 8) | k_be[8 * (3 - i) + j]; k[i] = v; }
  g1_encode(out, g1_to_affine(proj_mul_256(proj_from_affine(p), k)));
}
// Mul<Scalar> with 4-bit windows (proj_mul_win4, curve.h): what blsbn254_g1_mul_batch / g2_mul_batch run per lane
static void hs_scalar_words(const uint8_t* k_be, uint64_t k[4]) {
  for (int i = 0; i < 4; ++i) { uint64_t v = 0; for (int j = 0; j < 8; ++j) v = (v << 8) | k_be[8 * (3 - i) + j]; k[i] = v; }
}
void hs_g1_mul_win4(const uint8_t* a, const uint8_t* k_be, uint8_t* out) {
  bool o1;
  G1A p = g1_decode(a, o1);
  uint64_t k[4]; hs_scalar_words(k_be, k);
  g1_encode(out, g1_to_affine(proj_mul_win4(proj_from_affine(p), k)));
}
void hs_g2_mul_win4(const uint8_t* a, const uint8_t* k_be, uint8_t* out) {
  bool o1;
  G2A p = g2_decode(a, o1);
  uint64_t k[4]; hs_scalar_words(k_be, k);
  g2_encode(out, g2_to_affine(proj_mul_win4(proj_from_affine(p), k)));
}
// full single verify as the kernels compose it
int hs_verify(const uint8_t* pk, const uint8_t* msg, size_t len, const uint8_t* sig, const uint8_t* dst, uint32_t dst_len,
              uint8_t* ml_out) {
  G1A h = lane_hash_to_g1(msg, len, dst, dst_len);
  uint8_t flags;
  Fp12 f = lane_miller_verify(pk, sig, h, BN_NEG_G2_LINE_TABLE, flags);
  if (ml_out) fp12_to_be(ml_out, f);
  bool sub = lane_g2_check(pk);
  bool one = fp12_is_one(final_exponentiation(f));
  return (flags == (FLAG_SIG_OK | FLAG_PK_OK)) && sub && one;
}
// op counts of the two dominant phases in isolation (for the DESIGN.md instruction budget)
void hs_miller_verify_only(const uint8_t* pk, const uint8_t* sig, const uint8_t* h64, uint8_t* ml_out) {
  bool ok; G1A h = g1_decode(h64, ok);
  uint8_t flags;
  check_stats() = CheckStats();
  Fp12 f = lane_miller_verify(pk, sig, h, BN_NEG_G2_LINE_TABLE, flags);
  if (ml_out) fp12_to_be(ml_out, f);
}
// the workspace-reload variant of the verify Miller loop (what k_miller_verify runs)
void hs_miller_verify_ws(const uint8_t* pk, const uint8_t* sig, const uint8_t* h64, uint8_t* ml_out, int* flags_out) {
  bool ok; G1A h = g1_decode(h64, ok);
  uint8_t flags;
  static int32_t inv[126];        // 72 limbs of invariants + 54 for the parked running point (-DBN_VERIFY_PARK_T, as the kernel is built)
  Fp12 f = lane_miller_verify_ws(pk, sig, h, BN_NEG_G2_LINE_TABLE, flags, Ws{inv, 1, 0, false});
  fp12_to_be(ml_out, f);
  *flags_out = flags;
}
int hs_g1_codec_roundtrip(const uint8_t* g1, uint8_t* comp32, uint8_t* back64) {
  bool ok, ok2; G1A p = g1_decode(g1, ok);
  g1_compress(comp32, p);
  G1A q = g1_decompress(comp32, ok2);
  g1_encode(back64, q);
  return ok && ok2;
}
int hs_g2_codec_roundtrip(const uint8_t* g2, uint8_t* comp64, uint8_t* back128) {
  bool ok, ok2; G2A p = g2_decode(g2, ok);
  g2_compress(comp64, p);
  G2A q = g2_decompress(comp64, ok2);
  g2_encode(back128, q);
  return ok && ok2;
}
// GLV split of a canonical scalar (32 B big-endian): out = |k1| (16 B BE) || |k2| (16 B BE), returns sign bits (1: k1 < 0, 2: k2 < 0)
int hs_glv_split(const uint8_t* k_be, uint8_t* out32) {
  uint32_t w[8];
  for (int j = 0; j < 8; ++j) w[j] = load_be32(k_be + 4 * (7 - j));
  GlvSplit g = glv_split(w);
  for (int j = 0; j < 4; ++j) { store_be32(out32 + 4 * (3 - j), g.k1[j]); store_be32(out32 + 16 + 4 * (3 - j), g.k2[j]); }
  return (g.neg1 ? 1 : 0) | (g.neg2 ? 2 : 0);
}
// the endomorphism itself on an affine G1 point: (beta x, y)
void hs_glv_phi(const uint8_t* g1, uint8_t* out64) {
  bool ok; G1A p = g1_decode(g1, ok);
  p.x = fp_mul(p.x, fp_const(bnc::GLV_BETA));
  g1_encode(out64, p);
}
// tri.h primitives against their serial counterparts: op 0 a*b, 1 a^2, 2 cyclotomic square (a cyclotomic), 3 conj, 4..6 Frobenius^1..3
int hs_tri_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out_tri, uint8_t* out_serial) {
  bool ok1, ok2 = true;
  Fp12 x = fp12_from_be(a, ok1), y = b ? fp12_from_be(b, ok2) : fp12_one();
  if (!ok1 || !ok2) return 4;
  Fp6 res[4];
  tri_run([&](uint32_t role) {
    const Fp6 xa = fp6_norm(role == 0 ? x.c0 : x.c1), yb = fp6_norm(role == 0 ? y.c0 : y.c1);      // lanes 2, 3 hold lane 1's half (don't care)
    Fp6 r;
    switch (op) {
      case 0: r = tri_mul(xa, yb, role); break;
      case 1: r = tri_sqr(xa, role); break;
      case 2: r = tri_cyc_sqr(xa, role); break;
      case 3: r = tri_conj(xa, role); break;
      case 4: r = tri_frob<1>(xa, role); break;
      case 5: r = tri_frob<2>(xa, role); break;
      default: r = tri_frob<3>(xa, role); break;
    }
    res[role] = r;
  });
  fp12_to_be(out_tri, Fp12{res[0], res[1]});
  Fp12 want;
  switch (op) {
    case 0: want = fp12_mul(x, y); break;
    case 1: want = fp12_sqr(x); break;
    case 2: want = fp12_cyclotomic_sqr(x); break;
    case 3: want = fp12_conj(x); break;
    case 4: want = fp12_frob<1>(x); break;
    case 5: want = fp12_frob<2>(x); break;
    default: want = fp12_frob<3>(x); break;
  }
  fp12_to_be(out_serial, want);
  return 0;
}
// the verify Miller loop over the pair table on a quad == miller_loop_prepared (bytes of the Miller value); H homogeneous with Z = z_small
int hs_tri_miller(const uint8_t* sig, const uint8_t* h, const uint8_t* pk, int z_small, uint8_t* out_tri, uint8_t* out_serial) {
  bool ok;
  G1A S = g1_decode(sig, ok), H = g1_decode(h, ok);
  G2A Q = g2_decode(pk, ok);
  static int32_t raw[88 * 54], exp_[88 * 162], inv[81];
  g2_prepare_lines(Q, Ws{raw, 1, 0, false});
  for (int t = 0; t < 88; ++t)
    line_pair_expand(line_from_table(BN_NEG_G2_LINE_TABLE[t]), line_load_limbs(Ws{raw + 54 * t, 1, 0, false}), Ws{exp_ + 162 * t, 1, 0, false});
  Fp z = fp_one();
  for (int k = 1; k < z_small; ++k) z = fp_norm(fp_add(z, fp_one()));
  z = fp_canon(z);
  const Ws w = {inv, 1, 0, false};
  Fp xs = fp_norm(S.x), ys = fp_norm(S.y), X = fp_mul(fp_norm(H.x), z), Y = fp_mul(fp_norm(H.y), z), Z = z;
  fp_store_mem(w, X); fp_store_mem(ws_at(w, 9), Y); fp_store_mem(ws_at(w, 18), Z);
  fp_store_mem(ws_at(w, 27), fp_mul(xs, X)); fp_store_mem(ws_at(w, 36), fp_mul(ys, Y)); fp_store_mem(ws_at(w, 45), fp_mul(xs, Z));
  fp_store_mem(ws_at(w, 54), fp_mul(ys, Z)); fp_store_mem(ws_at(w, 63), fp_mul(ys, X)); fp_store_mem(ws_at(w, 72), fp_mul(xs, Y));
  fp12_to_be(out_serial, miller_loop_prepared(w, Ws{exp_, 1, 0, false}));
  Fp6 res[4];
  tri_run([&](uint32_t role) { res[role] = tri_miller_prepared(w, Ws{exp_, 1, 0, false}, role); });
  fp12_to_be(out_tri, Fp12{res[0], res[1]});
  return 0;
}
// the hard part on a quad after the serial easy part == final_exponentiation; also the per-lane comparison with one
int hs_tri_final_exp(const uint8_t* in, uint8_t* out, int* is_one) {
  bool ok;
  Fp12 f = fp12_from_be(in, ok);
  if (!ok) return 4;
  const Fp12 t = fe_easy(f);
  static int32_t tw[108];
  fp12_store_limbs(tw, 1, t);                       // canonical limbs, as k_fe_easy leaves them
  std::vector<int32_t> vals(TRI_VALUES * 108, 0);
  Fp6 res[4]; bool half[4];
  tri_run([&](uint32_t role) {
    const Fp6 th = tri_load_canon(Ws{tw, 1, 0, false}, role);
    res[role] = tri_fe_hard(th, Ws{vals.data(), 1, 0, false}, role);
    half[role] = tri_half_is_one(res[role], role);
  });
  fp12_to_be(out, Fp12{res[0], res[1]});
  *is_one = (half[0] && half[1]) ? 1 : 0;
  return 0;
}
// quad.h (four lanes per G2 point): the key preparation on a quad == the serial one: the 88 x 54 canonical limbs of the line
// table (returns 1 when equal) and the subgroup-test boolean (bit 1 of the result: quad, bit 2: serial)
int hs_quad_prepare(const uint8_t* pk, int* torsion_bits) {
  bool ok;
  G2A Q = g2_decode(pk, ok);
  if (!ok) return -1;
  static int32_t raw_s[88 * 54], raw_q[88 * 54];
  g2_prepare_lines(Q, Ws{raw_s, 1, 0, false});
  const bool tf_s = g2_torsion_free(Q);
  bool tf_q[4];
  tri_run([&](uint32_t role) { quad_prepare_lines(Q, Ws{raw_q, 1, 0, false}, role); tf_q[role] = quad_torsion_free(Q, role); });
  *torsion_bits = (tf_q[0] ? 1 : 0) | (tf_s ? 2 : 0) | ((tf_q[0] == tf_q[1] && tf_q[1] == tf_q[2] && tf_q[2] == tf_q[3]) ? 4 : 0);
  return std::memcmp(raw_s, raw_q, sizeof raw_s) == 0 ? 1 : 0;
}
// ML(P, Q) with a variable G2 point on a quad (tri_miller_1, quad.h) == miller_loop_1: bytes of both
int hs_tri_miller_1(const uint8_t* g1, const uint8_t* g2, uint8_t* out_tri, uint8_t* out_serial) {
  bool ok1, ok2;
  G1A p = g1_decode(g1, ok1);
  G2A q = g2_decode(g2, ok2);
  if (!ok1 || !ok2 || p.inf || q.inf) return -1;
  q.x = fp2_norm(q.x); q.y = fp2_norm(q.y);
  p.x = fp_norm(p.x); p.y = fp_norm(p.y);
  fp12_to_be(out_serial, miller_loop_1(p, q));
  Fp6 res[4];
  tri_run([&](uint32_t role) { res[role] = tri_miller_1(p.x, p.y, q, role); });
  fp12_to_be(out_tri, Fp12{res[0], res[1]});
  return 0;
}
// one pair with a prepared key on a quad (tri_miller_1prepared) == miller_loop_1prepared == ML(H, Q)
int hs_tri_miller_1prepared(const uint8_t* g1, const uint8_t* g2, uint8_t* out_tri, uint8_t* out_serial) {
  bool ok1, ok2;
  G1A p = g1_decode(g1, ok1);
  G2A q = g2_decode(g2, ok2);
  if (!ok1 || !ok2 || p.inf || q.inf) return -1;
  static int32_t raw[88 * 54], hh[18];
  g2_prepare_lines(q, Ws{raw, 1, 0, false});
  const Ws hw = {hh, 1, 0, false};
  fp_store_mem(hw, fp_norm(p.x)); fp_store_mem(ws_at(hw, 9), fp_norm(p.y));
  fp12_to_be(out_serial, miller_loop_1prepared(hw, Ws{raw, 1, 0, false}));
  Fp6 res[4];
  tri_run([&](uint32_t role) { res[role] = tri_miller_1prepared(fp_load_mem(hw), fp_load_mem(ws_at(hw, 9)), Ws{raw, 1, 0, false}, role); });
  fp12_to_be(out_tri, Fp12{res[0], res[1]});
  return 0;
}
void hs_stats(double* out) {
  CheckStats& s = check_stats();
  out[0] = s.worst_mul; out[1] = s.worst_dot; out[2] = s.worst_vb;
  out[3] = (double)s.muls; out[4] = (double)s.sqrs; out[5] = (double)s.dots; out[6] = (double)s.norms; out[7] = (double)s.lcs;
}
void hs_stats_reset() { check_stats() = CheckStats(); }
// Executed operation counts of the two dominant loops as the KERNELS run them (data independent: one run counts for all inputs):
// out[6 k + 0..5] = fp_mul, fp_sqr, fp_dot2, fp_norm, fp_lc passes, fp_lc terms of phase k:
//   k = 0 miller_loop_prepared (k_miller_prepared)      k = 1 miller_loop_verify_ws2 (k_miller_verify, exact path)
//   k = 2 cyclotomic_exp_x_chain (one t^x launch)       k = 3 fe_easy        k = 4 fe_h1 + fe_h2 + fe_h3 (the glue steps)
// Executed MADs follow as 162 mul + 126 sqr + 243 dot2 + 9 lc-terms (scripts/executed_mads.py).
void hs_executed_ops(const uint8_t* sig, const uint8_t* h, const uint8_t* pk, double* out) {
  bool ok;
  G1A S = g1_decode(sig, ok), H = g1_decode(h, ok);
  G2A Q = g2_decode(pk, ok);
  static int32_t raw[88 * 54], exp_[88 * 162], inv[81], park[54], slots[10 * 108], parkf[108];
  g2_prepare_lines(Q, Ws{raw, 1, 0, false});
  for (int t = 0; t < 88; ++t)
    line_pair_expand(line_from_table(BN_NEG_G2_LINE_TABLE[t]), line_load_limbs(Ws{raw + 54 * t, 1, 0, false}), Ws{exp_ + 162 * t, 1, 0, false});
  const Ws w = {inv, 1, 0, false};
  Fp xs = fp_norm(S.x), ys = fp_norm(S.y), X = fp_norm(H.x), Y = fp_norm(H.y), Z = fp_one();
  fp_store_mem(w, X); fp_store_mem(ws_at(w, 9), Y); fp_store_mem(ws_at(w, 18), Z);
  fp_store_mem(ws_at(w, 27), fp_mul(xs, X)); fp_store_mem(ws_at(w, 36), fp_mul(ys, Y)); fp_store_mem(ws_at(w, 45), fp_mul(xs, Z));
  fp_store_mem(ws_at(w, 54), fp_mul(ys, Z)); fp_store_mem(ws_at(w, 63), fp_mul(ys, X)); fp_store_mem(ws_at(w, 72), fp_mul(xs, Y));
  auto snap = [&](int k) {
    CheckStats& s = check_stats();
    out[6 * k + 0] = (double)s.muls; out[6 * k + 1] = (double)s.sqrs; out[6 * k + 2] = (double)s.dots;
    out[6 * k + 3] = (double)s.norms; out[6 * k + 4] = (double)s.lcs; out[6 * k + 5] = (double)s.lc_terms;
    check_stats() = CheckStats();
  };
  check_stats() = CheckStats();
  Fp12 f = miller_loop_prepared(w, Ws{exp_, 1, 0, false});
  snap(0);
  // the exact path's loop: invariants sig.x, sig.y, h.x, h.y, pk.x, pk.y in `inv` (72 limbs), T parked
  fp_store_mem(w, xs); fp_store_mem(ws_at(w, 9), ys); fp_store_mem(ws_at(w, 18), X); fp_store_mem(ws_at(w, 27), Y);
  fp2_store_mem(ws_at(w, 36), fp2_norm(Q.x)); fp2_store_mem(ws_at(w, 54), fp2_norm(Q.y));
  check_stats() = CheckStats();
  (void)miller_loop_verify_ws2(w, Ws{park, 1, 0, false}, BN_NEG_G2_LINE_TABLE);
  snap(1);
  Fp12 t = fe_easy(f);
  snap(3);
  const Ws pk_ws = {parkf, 1, 0, false};
  Fp12 x0 = cyclotomic_exp_x_chain(t, Ws{slots, 1, 0, false}, &pk_ws);
  snap(2);
  Fp12 a, b, c, b2, d2;
  fe_h1(x0, a, b);
  fe_h2(cyclotomic_exp_x(b), b, c, b2, d2);
  check_stats() = CheckStats();
  fe_h1(x0, a, b); fe_h2(x0, b, c, b2, d2); (void)fe_h3(t, a, c, b2, x0);       // counts only (operands need not be the real chain values)
  snap(4);
}
}
