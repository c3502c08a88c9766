// k_miller_prep.hip -- the verify Miller loop for PREPARED public keys: f = ML(sig, -G2gen) * ML(H(msg), pk) with both
// pairs' line coefficients taken from tables, i.e. multi_miller_loop(&[(&G1Affine, &G2Prepared)]) of the reference
// (pairings.rs:808-857) for two terms -- and since the first term's table (-G2gen) is the same for every tuple, the per-key
// table holds the nine coefficient products of each step's line PAIR (k_g2_expand; pairing.h line_pair_expand).  No point
// arithmetic, no running point: per loop digit one squaring of f, five coefficient evaluations and one sparse product.
// Lanes run in key-sorted order (perm), so a wave reads one key's table at (mostly) one address.
// Same compile policy as the other Miller units (-DBN_FORCE_INLINE -DBN_LC_MAD).
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

// sorted position s -> tuple perm[s] with key id kid[perm[s]].  f_ws / flags are in SORTED order.  h_ws: H(msg) homogeneous, 27 x h_stride limbs.
// flags[s] = signature decodes, is not the identity, is on the curve, AND the key passed its checks (key_ok).
BN_KERNEL k_miller_prepared(const uint32_t* perm, const uint32_t* kid, const uint8_t* sigs, const int32_t* h_ws, size_t h_stride,
                            const int32_t* table, const uint8_t* key_ok, size_t n, int32_t* f_ws, uint8_t* flags) {
  __shared__ int32_t inv_lds[81 * 256];          // each lane touches only its own column: no barrier needed
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t i = perm[s], k = kid[i];
  bool oks;
  G1A sig = g1_decode(sigs + 64 * (size_t)i, oks);
  const bool sig_ok = oks & !sig.inf & g1_on_curve(sig);
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one()));
  const Ws inv = {inv_lds, 256, threadIdx.x * 4u, false};
  const Ws hw = {const_cast<int32_t*>(h_ws), h_stride, i * 4u, true};
  const Fp xs = fp_norm(fp_select(sig_ok, sig.x, gp.x)), ys = fp_norm(fp_select(sig_ok, sig.y, gp.y));
  const Fp X = fp_load_mem(hw), Y = fp_load_mem(ws_at(hw, 9)), Z = fp_load_mem(ws_at(hw, 18));       // H(msg) = (X : Y : Z), from k_hash_to_g1 mode 3
  fp_store_mem(inv, X); fp_store_mem(ws_at(inv, 9), Y); fp_store_mem(ws_at(inv, 18), Z);
  fp_store_mem(ws_at(inv, 27), fp_mul(xs, X)); fp_store_mem(ws_at(inv, 36), fp_mul(ys, Y)); fp_store_mem(ws_at(inv, 45), fp_mul(xs, Z));
  fp_store_mem(ws_at(inv, 54), fp_mul(ys, Z)); fp_store_mem(ws_at(inv, 63), fp_mul(ys, X)); fp_store_mem(ws_at(inv, 72), fp_mul(xs, Y));
  BN_MEM_FENCE;
  const Ws kt = {const_cast<int32_t*>(table), 1, k * (uint32_t)(BN_NEG_G2_LINES * 162 * 4), true};
#ifdef BN_MILLER_UNIFORM_KEY
  const uint32_t k0 = __builtin_amdgcn_readfirstlane(k);
  Fp12 f;
  if (__builtin_amdgcn_ballot_w64(k != k0) == 0)          // one key for the whole wave: its table through scalar loads
    f = miller_loop_prepared_uniform(inv, table + (size_t)k0 * (BN_NEG_G2_LINES * 162));
  else
    f = miller_loop_prepared(inv, kt);
  fp12_store_limbs(Ws{f_ws, n, s * 4u, true}, f);
#else
  fp12_store_limbs(Ws{f_ws, n, s * 4u, true}, miller_loop_prepared(inv, kt));
#endif
  flags[s] = (sig_ok && key_ok[k] != 0) ? 1 : 0;
}
