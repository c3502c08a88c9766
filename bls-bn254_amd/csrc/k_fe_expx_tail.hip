// k_fe_expx_tail.hip -- the first two t -> t^x phases of the final exponentiation with the glue step that follows
// them (fe_h1, fe_h2) computed from the value still in registers: two launches and two Fp12 round trips fewer than
// k_fe_expx + k_fe_h1 / k_fe_h2.  Separate kernels (not modes of k_fe_expx) so that the shared chain loop keeps its
// code generation; see k_fe_expx.hip for the chain itself.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_fe_expx_h1(const int32_t* in, int32_t* slots, size_t n, size_t stride, int32_t* a_out, int32_t* b_out) {
  __shared__ int32_t park_lds[108 * 256];        // each lane touches only its own column: no barrier needed
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t l4 = (uint32_t)i * 4u;
  const Ws park = {park_lds, 256, threadIdx.x * 4u, false};
  Fp12 r = cyclotomic_exp_x_chain(fp12_load_limbs(in + i, stride), Ws{slots, stride, l4, true}, &park);
  fe_h1_tail(r, Ws{a_out, stride, l4, true}, Ws{b_out, stride, l4, true}, &park);
}
BN_KERNEL k_fe_expx_h2(const int32_t* in, int32_t* slots, size_t n, size_t stride, int32_t* b_in, int32_t* c_out, int32_t* b2_out, int32_t* d2_out) {
  __shared__ int32_t park_lds[108 * 256];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t l4 = (uint32_t)i * 4u;
  const Ws park = {park_lds, 256, threadIdx.x * 4u, false};
  Fp12 r = cyclotomic_exp_x_chain(fp12_load_limbs(in + i, stride), Ws{slots, stride, l4, true}, &park);
  fe_h2_tail(r, Ws{b_in, stride, l4, true}, Ws{c_out, stride, l4, true}, Ws{b2_out, stride, l4, true}, Ws{d2_out, stride, l4, true}, &park);
}
