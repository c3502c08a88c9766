// k_miller_hpk1p.hip -- ONE pair (H, pk) per lane with the key's lines read from its prepared table (G2Prepared, pairings.rs:609-660):
// f = ML(H, pk) per pair (multi_miller_loop terms, pairings.rs:808-857; the product is formed by the tree afterwards).  The two-pairs-
// per-lane kernel k_miller_hpk2p shares f^2 between two pairs and is the one for large batches; with only a few pairs a launch
// is the latency of one wave, and one pair per lane is the shorter chain (aggregate verify by per-key sums: u + 1 pairs).
// Same compile policy as the other Miller units.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

// n pairs, n lanes; kid[pair] = key id; table: raw line triples, 88 x 54 limbs per key; flags[pair] = the pair's key is valid.
// skip (optional): bit 1 of skip[pair] set = the pair contributes 1 (its G1 member is the identity).
BN_KERNEL k_miller_hpk1p(const int32_t* h_ws, size_t h_stride, const uint32_t* kid, const int32_t* table, const uint8_t* key_ok, size_t n,
                         int32_t* f_ws, size_t f_stride, uint8_t* flags, const uint8_t* skip) {
  __shared__ int32_t lds[18 * 256];              // each lane touches only its own column: no barrier needed
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Ws hh = {lds, 256, threadIdx.x * 4u, false};
  const bool live = !(skip && (skip[i] & 2));
  const uint32_t key = kid[i];
  const Ws hw = {const_cast<int32_t*>(h_ws), h_stride, i * 4u, true};
  fp_store_mem(hh, fp_load_mem(hw)); fp_store_mem(ws_at(hh, 9), fp_load_mem(ws_at(hw, 9)));
  flags[i] = key_ok[key];
  BN_MEM_FENCE;
  const Ws ta = {const_cast<int32_t*>(table), 1, key * (uint32_t)(BN_NEG_G2_LINES * 54 * 4), true};
  Fp12 f = miller_loop_1prepared(hh, ta);
  const Fp12 one = fp12_one();
  f.c0 = {fp2_select(live, f.c0.c0, one.c0.c0), fp2_select(live, f.c0.c1, one.c0.c1), fp2_select(live, f.c0.c2, one.c0.c2)};
  f.c1 = {fp2_select(live, f.c1.c0, one.c1.c0), fp2_select(live, f.c1.c1, one.c1.c1), fp2_select(live, f.c1.c2, one.c1.c2)};
  fp12_store_limbs(Ws{f_ws, f_stride, i * 4u, true}, f);
}
