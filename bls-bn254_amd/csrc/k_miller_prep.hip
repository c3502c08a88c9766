// k_miller_prep.hip -- the verify Miller loop for PREPARED public keys: f = ML(sig, -G2gen) * ML(H(msg), pk) with both
// pairs' line coefficients read from tables (the generated -G2gen table and the per-key table k_g2_prepare wrote), i.e.
// multi_miller_loop(&[(&G1Affine, &G2Prepared)]) of the reference (pairings.rs:808-857) for two terms.  No point
// arithmetic, no running point: per loop digit one squaring of f and one two-line product.  Lanes run in key-sorted
// order (perm), so a wave reads one key's lines at (mostly) one address.
// Same compile policy as the other Miller units (-DBN_FORCE_INLINE -DBN_LC_MAD).
#define BN_WANT_LINE_TABLE
#define BN_LINE_TABLE_QUAL static __device__ const
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

// sorted position s -> tuple perm[s] with key id kid[perm[s]].  f_ws / flags are in SORTED order.
// flags[s] = signature decodes, is not the identity, is on the curve, AND the key passed its checks (key_ok).
BN_KERNEL k_miller_prepared(const uint32_t* perm, const uint32_t* kid, const uint8_t* sigs, const int32_t* h_ws, size_t h_stride,
                            const int32_t* table, const uint8_t* key_ok, size_t n, int32_t* f_ws, uint8_t* flags) {
  __shared__ int32_t inv_lds[36 * 256];          // each lane touches only its own column: no barrier needed
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t i = perm[s], k = kid[i];
  bool oks;
  G1A sig = g1_decode(sigs + 64 * (size_t)i, oks);
  const bool sig_ok = oks & !sig.inf & g1_on_curve(sig);
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one()));
  const Ws inv = {inv_lds, 256, threadIdx.x * 4u, false};
  const Ws hw = {const_cast<int32_t*>(h_ws), h_stride, i * 4u, true};
  fp_store_mem(inv, fp_norm(fp_select(sig_ok, sig.x, gp.x))); fp_store_mem(ws_at(inv, 9), fp_norm(fp_select(sig_ok, sig.y, gp.y)));
  fp_store_mem(ws_at(inv, 18), fp_load_mem(hw)); fp_store_mem(ws_at(inv, 27), fp_load_mem(ws_at(hw, 9)));
  BN_MEM_FENCE;
  const Ws kt = {const_cast<int32_t*>(table), 1, k * (uint32_t)(BN_NEG_G2_LINES * 54 * 4), true};
  fp12_store_limbs(Ws{f_ws, n, s * 4u, true}, miller_loop_prepared(inv, BN_NEG_G2_LINE_TABLE, kt));
  flags[s] = (sig_ok && key_ok[k] != 0) ? 1 : 0;
}
