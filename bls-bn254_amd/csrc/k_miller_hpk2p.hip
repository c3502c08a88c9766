// k_miller_hpk2p.hip -- aggregate verify over a batch that repeats few public keys: two pairs (H(msg), pk) per lane sharing
// one f^2 per loop digit (multi_miller_loop, pairings.rs:808-857), the line coefficients of BOTH keys read from their
// prepared tables (G2Prepared, pairings.rs:609-660; written once per distinct key by k_g2_prepare).  Per digit: one squaring
// of f and one two-line product -- no point arithmetic at all.  Same compile policy as the other Miller units.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

// n pairs, ceil(n / 2) lanes; kid[pair] = key id; table: raw line triples, 88 x 54 limbs per key; key_ok per key.
// flags[pair] = the pair's key is valid (decodes, not the identity, on the curve, in the r-torsion).
// skip (optional): bit 1 of skip[pair] set = the pair contributes 1 (its G1 member is the identity; multi_miller_loop skips such terms).
BN_KERNEL k_miller_hpk2p(const int32_t* h_ws, size_t h_stride, const uint32_t* kid, const int32_t* table, const uint8_t* key_ok, size_t n,
                         int32_t* f_ws, size_t f_stride, uint8_t* flags, const uint8_t* skip) {
  __shared__ int32_t lds[36 * 256];              // each lane touches only its own column: no barrier needed
  const size_t n_lanes = (n + 1) >> 1;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  const Ws hh = {lds, 256, threadIdx.x * 4u, false};
  // pair a = 2i (always present), pair b = 2i + 1 (padding when n is odd: re-reads pair 0, masked out, evaluated at y = 1)
  const size_t pa = 2 * (size_t)i, pbq = pa + 1;
  const bool has_b = pbq < n;
  const size_t pb = has_b ? pbq : 0;
  const bool live_a = !(skip && (skip[pa] & 2)), live_b = has_b && !(skip && (skip[pb] & 2));
  const uint32_t key_a = kid[pa], key_b = kid[pb];
  const Ws hwa = {const_cast<int32_t*>(h_ws), h_stride, (uint32_t)pa * 4u, true}, hwb = {const_cast<int32_t*>(h_ws), h_stride, (uint32_t)pb * 4u, true};
  fp_store_mem(hh, fp_load_mem(hwa)); fp_store_mem(ws_at(hh, 9), fp_select(live_a, fp_load_mem(ws_at(hwa, 9)), fp_one()));
  fp_store_mem(ws_at(hh, 18), fp_load_mem(hwb)); fp_store_mem(ws_at(hh, 27), fp_select(live_b, fp_load_mem(ws_at(hwb, 9)), fp_one()));
  flags[pa] = key_ok[key_a];
  if (has_b) flags[pbq] = key_ok[key_b];
  BN_MEM_FENCE;
  const Ws ta = {const_cast<int32_t*>(table), 1, key_a * (uint32_t)(BN_NEG_G2_LINES * 54 * 4), true};
  const Ws tb = {const_cast<int32_t*>(table), 1, key_b * (uint32_t)(BN_NEG_G2_LINES * 54 * 4), true};
  fp12_store_limbs(Ws{f_ws, f_stride, i * 4u, true}, miller_loop_2prepared(hh, ta, tb, live_a, live_b));
}
