// k_miller_1.hip -- primitive per-pair Miller loop ML(P_i, Q_i) from byte inputs (pairing / multi_miller_loop).
// Compiled with the tower functions force-inlined (-DBN_FORCE_INLINE) so that the register allocator sees
// the whole loop body and f / T stay in the 512 VGPR+AGPR of a 1-wave-per-SIMD kernel instead of
// round-tripping through scratch (r01 profile: 270 KB of scratch traffic per tuple with outlined calls,
// 41.9 -> 32.5 ms once inlined).  One kernel per translation unit: they compile in parallel.
#define BN_MILLER1_WS
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_miller_1(const uint8_t* g1, const uint8_t* g2, size_t n, int32_t* f_ws, size_t f_stride, uint8_t* status) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;      // 32-bit lane index, buffer-addressed workspace (see k_miller_verify.hip)
  if (i >= n) return;
  __shared__ int32_t lds[108 * 256];             // each lane touches only its own column: no barrier needed
  const Ws ws = {lds, 256, threadIdx.x * 4u, false};
  uint8_t st;
  Fp12 f = lane_miller_1(g1 + 64 * (size_t)i, g2 + 128 * (size_t)i, st, &ws);
  fp12_store_limbs(Ws{f_ws, f_stride, i * 4u, true}, f);
  status[i] = st;
}
