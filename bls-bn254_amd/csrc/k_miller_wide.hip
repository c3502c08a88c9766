// k_miller_wide.hip -- Miller loops over prepared keys with ONE WAVE PER TUPLE (wide.h), for launches of few tuples where the
// lane-per-tuple kernels are the latency of one lane's chain (k_miller_prepared 4.4 ms, k_miller_hpk1p 3.5 ms, whatever n):
//   k_miller_wide_prepared  = k_miller_prepared (verify: both pairs from the key's pair table), same arguments
//   k_miller_wide_1p        = k_miller_hpk1p (one pair (H, pk) from the key's raw line table), same arguments
// Same values as those kernels (multi_miller_loop over prepared terms, pairings.rs:808-857), hence the same bytes.
// Workgroup = 128 lanes (two waves) = tuple blockIdx.x.
#include "wide.h"
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

namespace {
__device__ inline void wide_store_result(const Wide& W, int32_t* f_ws, size_t f_stride, size_t col, bool live) {
  const uint32_t lane = threadIdx.x;
  if (lane < 6u) {
    const Fp2 one = lane == 0u ? fp2_one() : fp2_zero();
    const Fp2 v = fp2_select(live, fp2_load_mem(wide_val(W, WV_R, lane)), one);
    fp2_store_limbs(ws_at_lane(ws_uniform(Ws{f_ws, f_stride, (uint32_t)col * 4u, true}), 18u * lane), v);      // lane-dependent offset, uniform buffer base
  }
}
}  // namespace

__global__ void __launch_bounds__(128) k_miller_wide_prepared(const uint32_t* perm, const uint32_t* kid, const uint8_t* sigs, const int32_t* h_ws, size_t h_stride,
                                                             const int32_t* table, const uint8_t* key_ok, size_t n, int32_t* f_ws, uint8_t* flags) {
  __shared__ int32_t lds[WIDE_PA_LIMBS + 108 * 3 + 81];            // product area, R, L, L2, the nine coordinate values
  const size_t s = blockIdx.x;
  if (s >= n) return;
  const Wide W(lds);
  const uint32_t i = perm[s], k = kid[i];
  const Ws cw = {lds, 1, (WIDE_PA_LIMBS + 108u * 3u) * 4u, false};
  if (threadIdx.x == 0) {                                          // prologue of k_miller_prepared, on one lane
    bool oks;
    G1A sig = g1_decode(sigs + 64 * (size_t)i, oks);
    const bool sig_ok = oks & !sig.inf & g1_on_curve(sig);
    G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one()));
    const Ws hw = {const_cast<int32_t*>(h_ws), h_stride, i * 4u, true};
    const Fp xs = fp_norm(fp_select(sig_ok, sig.x, gp.x)), ys = fp_norm(fp_select(sig_ok, sig.y, gp.y));
    const Fp X = fp_load_mem(hw), Y = fp_load_mem(ws_at(hw, 9)), Z = fp_load_mem(ws_at(hw, 18));
    fp_store_mem(cw, X); fp_store_mem(ws_at(cw, 9), Y); fp_store_mem(ws_at(cw, 18), Z);
    fp_store_mem(ws_at(cw, 27), fp_mul(xs, X)); fp_store_mem(ws_at(cw, 36), fp_mul(ys, Y)); fp_store_mem(ws_at(cw, 45), fp_mul(xs, Z));
    fp_store_mem(ws_at(cw, 54), fp_mul(ys, Z)); fp_store_mem(ws_at(cw, 63), fp_mul(ys, X)); fp_store_mem(ws_at(cw, 72), fp_mul(xs, Y));
    flags[s] = (sig_ok && key_ok[k] != 0) ? 1 : 0;
  }
  __syncthreads();
  wide_miller_prepared(W, Ws{const_cast<int32_t*>(table), 1, k * (uint32_t)(BN_NEG_G2_LINES * 162 * 4), true}, cw, true);
  wide_store_result(W, f_ws, n, s, true);
}

__global__ void __launch_bounds__(128) k_miller_wide_1p(const int32_t* h_ws, size_t h_stride, const uint32_t* kid, const int32_t* table, const uint8_t* key_ok, size_t n,
                                                       int32_t* f_ws, size_t f_stride, uint8_t* flags, const uint8_t* skip) {
  __shared__ int32_t lds[WIDE_PA_LIMBS + 108 * 3 + 18];            // product area, R, L, L2, (px, py)
  const size_t i = blockIdx.x;
  if (i >= n) return;
  const Wide W(lds);
  const uint32_t key = kid[i];
  const bool live = !(skip && (skip[i] & 2));
  const Ws pt = {lds, 1, (WIDE_PA_LIMBS + 108u * 3u) * 4u, false};
  if (threadIdx.x == 0) {
    const Ws hw = {const_cast<int32_t*>(h_ws), h_stride, (uint32_t)i * 4u, true};
    fp_store_mem(pt, fp_load_mem(hw)); fp_store_mem(ws_at(pt, 9), fp_load_mem(ws_at(hw, 9)));
    flags[i] = key_ok[key];
  }
  __syncthreads();
  wide_miller_prepared(W, Ws{const_cast<int32_t*>(table), 1, key * (uint32_t)(BN_NEG_G2_LINES * 54 * 4), true}, pt, false);
  wide_store_result(W, f_ws, f_stride, i, live);
}

// k_miller_1 (pairing / miller_loop of arbitrary (G1, G2) pairs) with one wave per pair: lane 0 runs the point arithmetic of the
// variable G2 point, the wave the Fp12 arithmetic.  status as lane_miller_1: bit 0 g1 decodes, bit 1 g2 decodes, bit 2 identity.
__global__ void __launch_bounds__(128) k_miller_wide_1(const uint8_t* g1, const uint8_t* g2, size_t n, int32_t* f_ws, size_t f_stride, uint8_t* status) {
  __shared__ int32_t lds[WIDE_PA_LIMBS + 108 * 3 + 18 + 54];       // product area, R, L, L2, (px, py), the parked line triple
  const size_t i = blockIdx.x;
  if (i >= n) return;
  const Wide W(lds);
  bool ok1, ok2;
  G1A p = g1_decode(g1 + 64 * i, ok1);
  G2A q = g2_decode(g2 + 128 * i, ok2);
  const bool ident = p.inf | q.inf, bad = ident | !ok1 | !ok2;      // identity or undecodable operands: generators, result replaced by 1
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one()));
  p.x = fp_select(bad, gp.x, p.x); p.y = fp_select(bad, gp.y, p.y);
  q.x = fp2_norm(fp2_select(bad, fp2_const(bnc::G2_GEN_X), q.x)); q.y = fp2_norm(fp2_select(bad, fp2_const(bnc::G2_GEN_Y), q.y)); q.inf = false;
  const Ws pt = {lds, 1, (WIDE_PA_LIMBS + 108u * 3u) * 4u, false}, lnw = {lds, 1, (WIDE_PA_LIMBS + 108u * 3u + 18u) * 4u, false};
  if (threadIdx.x == 0) {
    fp_store_mem(pt, fp_norm(p.x)); fp_store_mem(ws_at(pt, 9), fp_norm(p.y));
    status[i] = (uint8_t)((ok1 ? 1 : 0) | (ok2 ? 2 : 0) | (ident ? 4 : 0));
  }
  __syncthreads();
  wide_miller_1(W, q, pt, lnw);
  wide_store_result(W, f_ws, f_stride, i, !bad);
}
