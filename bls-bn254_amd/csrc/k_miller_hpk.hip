// k_miller_hpk.hip -- aggregate verify: per-pair Miller loop ML(H(msg_i), pk_i).
// Compiled with the tower functions force-inlined (-DBN_FORCE_INLINE) so that the register allocator sees
// the whole loop body and f / T stay in the 512 VGPR+AGPR of a 1-wave-per-SIMD kernel instead of
// round-tripping through scratch (r01 profile: 270 KB of scratch traffic per tuple with outlined calls,
// 41.9 -> 32.5 ms once inlined).  One kernel per translation unit: they compile in parallel.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_miller_hpk(const int32_t* h_ws, const uint8_t* pks, size_t n, int32_t* f_ws, size_t f_stride, uint8_t* flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;      // 32-bit lane index, buffer-addressed workspaces (see k_miller_verify.hip)
  if (i >= n) return;
  bool okp;
  G2A pk = g2_decode(pks + 128 * (size_t)i, okp);
  bool pk_ok = okp & !pk.inf & g2_on_curve(pk);
  pk.x = fp2_select(pk_ok, pk.x, fp2_const(bnc::G2_GEN_X)); pk.y = fp2_select(pk_ok, pk.y, fp2_const(bnc::G2_GEN_Y));
  pk.inf = false;
  __shared__ int32_t lds[108 * 256];             // invariants (54 limbs) + the parked running point (54): own column per lane, no barrier
  const Ws ws = {lds, 256, threadIdx.x * 4u, false};
  const Ws hw = {const_cast<int32_t*>(h_ws), n, i * 4u, true};
  fp_store_mem(ws, fp_load_mem(hw)); fp_store_mem(ws_at(ws, 9), fp_load_mem(ws_at(hw, 9)));
  fp2_store_mem(ws_at(ws, 18), fp2_norm(pk.x)); fp2_store_mem(ws_at(ws, 36), fp2_norm(pk.y));
  BN_MEM_FENCE;
  fp12_store_limbs(Ws{f_ws, f_stride, i * 4u, true}, miller_loop_1_ws(ws, ws_at(ws, 54)));
  flags[i] = pk_ok ? 1 : 0;
}
