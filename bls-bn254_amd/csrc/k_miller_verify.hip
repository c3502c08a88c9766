// k_miller_verify.hip -- the dominant kernel: f = ML(sig, -G2gen) * ML(H(msg), pk), first pair from the fixed line table.
// Compiled with the tower functions force-inlined (-DBN_FORCE_INLINE) so that the register allocator sees
// the whole loop body and f / T stay in the 512 VGPR+AGPR of a 1-wave-per-SIMD kernel instead of
// round-tripping through scratch (r01 profile: 270 KB of scratch traffic per tuple with outlined calls,
// 41.9 -> 32.5 ms once inlined).  The loop invariants (sig, H, pk: 72 limbs per lane) are written once to LDS,
// limb-major (72 KB per 256-lane workgroup), and re-loaded per use instead of occupying 90 registers.
// One kernel per translation unit.
#define BN_WANT_LINE_TABLE
#define BN_LINE_TABLE_QUAL static __device__ const
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_miller_verify(const uint8_t* pks, const uint8_t* sigs, const int32_t* h_ws, size_t n, int32_t* f_ws, uint8_t* flags) {
#ifdef BN_VERIFY_PARK_T
  __shared__ int32_t inv_lds[126 * 256];         // + 54 limbs: the parked running point
#else
  __shared__ int32_t inv_lds[72 * 256];
#endif          // each lane touches only its own column: no barrier needed
  // 32-bit lane index and buffer-addressed workspaces: no 64-bit per-lane value lives across the loop (a
  // zero-extended thread id did, and its zero half was re-loaded from scratch wherever a zero was needed)
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Ws hw = {const_cast<int32_t*>(h_ws), n, i * 4u, true};
  G1A h; h.x = fp_load_mem(hw); h.y = fp_load_mem(ws_at(hw, 9)); h.inf = false;
  uint8_t fl;
  Fp12 f = lane_miller_verify_ws(pks + 128 * (size_t)i, sigs + 64 * (size_t)i, h, BN_NEG_G2_LINE_TABLE, fl,
                                 Ws{inv_lds, 256, threadIdx.x * 4u, false});
  fp12_store_limbs(Ws{f_ws, n, i * 4u, true}, f);
  flags[i] = fl;
}
