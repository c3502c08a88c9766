// k_fe_easy.hip -- easy part of the final exponentiation: f^((p^6-1)(p^2+1)) (one Fp12 inversion).
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_fe_easy(const int32_t* in, int32_t* out, size_t n, size_t stride) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fp12_store_limbs(out + i, stride, fe_easy(fp12_load_limbs(in + i, stride)));
}
