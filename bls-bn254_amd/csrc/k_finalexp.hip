// k_finalexp.hip -- hard-part glue kernels of the final exponentiation between the t -> t^x phases (h1, h2);
// small enough to be compiled fully inlined (0.46 -> 0.37 / 0.39 ms).  The last step lives in k_fe_h3.hip.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

// hard-part glue between the three t -> t^x kernels (see pairing.h fe_h1 / fe_h2 / fe_h3)
BN_KERNEL k_fe_h1(const int32_t* x0, int32_t* a_out, int32_t* b_out, size_t n, size_t stride) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp12 a, b;
  fe_h1(fp12_load_limbs(x0 + i, stride), a, b);
  fp12_store_limbs(a_out + i, stride, a); fp12_store_limbs(b_out + i, stride, b);
}
BN_KERNEL k_fe_h2(const int32_t* x0, const int32_t* b_in, int32_t* c_out, int32_t* b2_out, int32_t* d2_out, size_t n, size_t stride) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp12 c, b2, d2;
  fe_h2(fp12_load_limbs(x0 + i, stride), fp12_load_limbs(b_in + i, stride), c, b2, d2);
  fp12_store_limbs(c_out + i, stride, c); fp12_store_limbs(b2_out + i, stride, b2); fp12_store_limbs(d2_out + i, stride, d2);
}
