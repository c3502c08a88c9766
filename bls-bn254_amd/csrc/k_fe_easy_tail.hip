// k_fe_easy_tail.hip -- third launch of the split easy part (k_fe_easy.hip): f^-1 from the parked pieces and the shared inversion,
// then t = conj(f) f^-1 and t^(p^2) t.  Its own unit: two Fp6 and two Fp12 products want the 512 registers of one wave per SIMD
// (at 256 it spills 70 registers and runs 0.71 instead of ~0.4 ms).
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_fe_easy_tail(const int32_t* in, int32_t* out, size_t n, size_t stride, const int32_t* head, const int32_t* nu) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  FeEasyHead h;
  h.c0 = fp2_load_limbs(head + i, stride); h.c1 = fp2_load_limbs(head + 18 * stride + i, stride);
  h.c2 = fp2_load_limbs(head + 36 * stride + i, stride); h.nrm = fp2_load_limbs(head + 54 * stride + i, stride);
  h.nu = fp_zero();
  fp12_store_limbs(out + i, stride, fe_easy_tail(fp12_load_limbs(in + i, stride), h, load_fp(nu + i, stride)));
}
