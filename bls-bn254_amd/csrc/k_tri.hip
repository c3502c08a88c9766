// k_tri.hip -- three lanes per tuple (tri.h): the table-only verify Miller loop and the hard part of the final exponentiation
// for launches between the wave-per-tuple limit (~4 000 tuples) and a quarter of a round of lanes (16 384 tuples = 65 536
// lanes): the sizes where the lane-per-tuple kernels cost one lane's whole chain (4.4 + 5.4 ms) however few tuples there are.
// Same values as k_miller_prepared / k_fe_expx* + k_fe_h3, same workspace formats (f_ws: canonical limbs, c0 then c1), so
// either form can follow the other.  Tuple = global lane / 4, role = lane & 3 (0: c0, 1: c1, 2: the Karatsuba cross product,
// 3: idle).  Compile policy: the Miller units' (-DBN_FORCE_INLINE -DBN_LC_MAD), one wave per SIMD.
#include "quad.h"
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

// sorted position s -> tuple perm[s] with key id kid[perm[s]] (k_miller_prepared's arguments and outputs)
BN_KERNEL k_miller_tri_prepared(const uint32_t* perm, const uint32_t* kid, const uint8_t* sigs, const int32_t* h_ws, size_t h_stride,
                                const int32_t* table, const uint8_t* key_ok, size_t n, int32_t* f_ws, uint8_t* flags) {
  __shared__ int32_t cw_lds[81 * 64];            // 64 tuples per workgroup, [limb][tuple]; a quad's lanes write and read identical values
  const size_t s = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (s >= n) return;                            // whole quads leave together
  const uint32_t role = tri_role(), tq = threadIdx.x >> 2;
  const uint32_t i = perm[s], k = kid[i];
  bool oks;
  G1A sig = g1_decode(sigs + 64 * (size_t)i, oks);
  const bool sig_ok = oks & !sig.inf & g1_on_curve(sig);
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one()));
  const Ws inv = {cw_lds, 64, tq * 4u, false};
  const Ws hw = {const_cast<int32_t*>(h_ws), h_stride, i * 4u, true};
  const Fp xs = fp_norm(fp_select(sig_ok, sig.x, gp.x)), ys = fp_norm(fp_select(sig_ok, sig.y, gp.y));
  const Fp X = fp_load_mem(hw), Y = fp_load_mem(ws_at(hw, 9)), Z = fp_load_mem(ws_at(hw, 18));       // H(msg) = (X : Y : Z), k_hash_to_g1 mode 3
  fp_store_mem(inv, X); fp_store_mem(ws_at(inv, 9), Y); fp_store_mem(ws_at(inv, 18), Z);
  fp_store_mem(ws_at(inv, 27), fp_mul(xs, X)); fp_store_mem(ws_at(inv, 36), fp_mul(ys, Y)); fp_store_mem(ws_at(inv, 45), fp_mul(xs, Z));
  fp_store_mem(ws_at(inv, 54), fp_mul(ys, Z)); fp_store_mem(ws_at(inv, 63), fp_mul(ys, X)); fp_store_mem(ws_at(inv, 72), fp_mul(xs, Y));
  BN_MEM_FENCE;
  const Ws kt = {const_cast<int32_t*>(table), 1, k * (uint32_t)(BN_NEG_G2_LINES * 162 * 4), true};
  tri_store_canon(Ws{f_ws, n, (uint32_t)s * 4u, true}, tri_miller_prepared(inv, kt, role), role);
  if (role == 0u) flags[s] = (sig_ok && key_ok[k] != 0) ? 1 : 0;
}

// t = f^((p^6-1)(p^2+1)) at t_ws (canonical limbs, as the easy part leaves it) -> the final exponentiation's result:
//   mode 0: one[i] = (result == 1) && flags ok && subgroup ok (k_pack_bitmap makes the bitmap)   mode 1/2: Gt bytes
//   mode 3: *is_one (n == 1)                                                                     mode 4: gt_bytes[i] = (result == 1)
// vals: TRI_VALUES x 108 limbs per tuple (limb-major, stride n): the named values of the hard part.
BN_KERNEL k_fe_tri_hard(const int32_t* t_ws, size_t n, size_t stride, int32_t* vals, const uint8_t* flags, const uint8_t* sub_ok,
                        uint8_t* one, uint8_t* gt_bytes, int* is_one, int mode) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (i >= n) return;
  const uint32_t role = tri_role();
  const Fp6 t = tri_load_canon(Ws{const_cast<int32_t*>(t_ws), stride, (uint32_t)i * 4u, true}, role);
  const Fp6 r = tri_fe_hard(t, Ws{vals, n, (uint32_t)i * 4u, true}, role);
  const int half = tri_half_is_one(r, role) ? 1 : 0;
  const int h0 = __builtin_amdgcn_update_dpp(0, half, 0x00, 0xf, 0xf, false), h1 = __builtin_amdgcn_update_dpp(0, half, 0x55, 0xf, 0xf, false);
  const bool isone = (h0 & h1) != 0;
  if (mode == 1 || mode == 2) {
    const Fp6 hi = tri_fetch6<1, 1, 1, 1>(r);                       // lane 0 assembles the value
    if (role == 0u) fp12_to_be(gt_bytes + 384 * i, Fp12{r, hi});
  } else if (role == 0u) {
    if (mode == 0) one[i] = (isone && flags[i] == (FLAG_SIG_OK | FLAG_PK_OK) && sub_ok[i] != 0) ? 1 : 0;
    else if (mode == 3) *is_one = isone ? 1 : 0;
    else gt_bytes[i] = isone ? 1 : 0;
  }
}

// ML(P_i, Q_i) from byte inputs with a quad per pair (k_miller_1's arguments and outputs): the line steps four lanes per point
// (quad.h), the accumulator three lanes per value.  status as lane_miller_1: bit 0 g1 decoded, bit 1 g2 decoded, bit 2 either is
// the identity; such pairs run on the generators (uniform control flow) and yield 1.
BN_KERNEL k_miller_tri_1(const uint8_t* g1, const uint8_t* g2, size_t n, int32_t* f_ws, size_t f_stride, uint8_t* status) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (i >= n) return;
  const uint32_t role = tri_role();
  bool ok1, ok2;
  G1A p = g1_decode(g1 + 64 * i, ok1);
  G2A q = g2_decode(g2 + 128 * i, ok2);
  const bool ident = p.inf | q.inf;
  const bool bad = ident | !ok1 | !ok2;
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one())); gp.inf = false;
  p.x = fp_select(bad, gp.x, p.x); p.y = fp_select(bad, gp.y, p.y);
  q.x = fp2_select(bad, fp2_const(bnc::G2_GEN_X), q.x); q.y = fp2_select(bad, fp2_const(bnc::G2_GEN_Y), q.y); q.inf = false;
  q.x = fp2_norm(q.x); q.y = fp2_norm(q.y);
  Fp6 f = tri_miller_1(fp_norm(p.x), fp_norm(p.y), q, role);
  f = fp6_select(bad, fp6_pick(role == 0u, fp6_one(), fp6_zero()), f);
  tri_store_canon(Ws{f_ws, f_stride, (uint32_t)i * 4u, true}, f, role);
  if (role == 0u) status[i] = (uint8_t)((ok1 ? 1 : 0) | (ok2 ? 2 : 0) | (ident ? 4 : 0));
}

// ONE pair (H, pk) per quad with the key's lines read from its prepared raw table: k_miller_hpk1p's arguments and outputs
// (aggregate verify by per-key sums over a few thousand distinct keys: u + 1 pairs).
BN_KERNEL k_miller_tri_1p(const int32_t* h_ws, size_t h_stride, const uint32_t* kid, const int32_t* table, const uint8_t* key_ok, size_t n,
                          int32_t* f_ws, size_t f_stride, uint8_t* flags, const uint8_t* skip) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (i >= n) return;
  const uint32_t role = tri_role();
  const bool live = !(skip && (skip[i] & 2));
  const uint32_t key = kid[i];
  const Ws hw = {const_cast<int32_t*>(h_ws), h_stride, (uint32_t)i * 4u, true};
  const Fp px = fp_load_mem(hw), py = fp_load_mem(ws_at(hw, 9));
  const Ws ta = {const_cast<int32_t*>(table), 1, key * (uint32_t)(BN_NEG_G2_LINES * 54 * 4), true};
  Fp6 f = tri_miller_1prepared(px, py, ta, role);
  f = fp6_select(live, f, fp6_pick(role == 0u, fp6_one(), fp6_zero()));
  tri_store_canon(Ws{f_ws, f_stride, (uint32_t)i * 4u, true}, f, role);
  if (role == 0u) flags[i] = key_ok[key];
}
