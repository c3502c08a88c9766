// k_fe_expx.hip -- t -> t^x in the cyclotomic subgroup (x = 0x44e992b44a6909f1) by the addition chain of
// pairing.h (62 Granger-Scott squarings + 13 multiplications), fully inlined.  The running value r lives in
// registers; the five named powers of the chain (t, t^17, t^35 and the conjugates of the last two) are parked in a limb-major
// HBM workspace (432 B each per tuple, re-read at most five times: ~8 KB per tuple against ~0.4 M VALU ops) so that r plus the
// temporaries of one product fit the 512 registers of a 1-wave-per-SIMD kernel.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_fe_expx(const int32_t* in, int32_t* out, int32_t* slots, size_t n, size_t stride) {
  __shared__ int32_t park_lds[108 * 256];        // each lane touches only its own column: no barrier needed
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Ws park = {park_lds, 256, threadIdx.x * 4u, false};
  fp12_store_limbs(out + i, stride, cyclotomic_exp_x_chain(fp12_load_limbs(in + i, stride), Ws{slots, stride, (uint32_t)i * 4u, true}, &park));
}
