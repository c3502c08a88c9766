// k_fe_easy.hip -- easy part of the final exponentiation: f^((p^6-1)(p^2+1)) (one Fp12 inversion).
// k_fe_easy: one launch, every lane inverts its own norm (launches below a round of waves: one chain either way).
// k_fe_easy_head / k_fe_inv4 / k_fe_easy_tail: the same values with the Fp inversion shared by four tuples per lane
// (pairing.h fe_easy_head / fp_inv4 / fe_easy_tail): the inversion is ~60 % of the one-launch kernel's instructions.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_fe_easy(const int32_t* in, int32_t* out, size_t n, size_t stride) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fp12_store_limbs(out + i, stride, fe_easy(fp12_load_limbs(in + i, stride)));
}
// head[0..71] = c0, c1, c2, N (canonical limbs, limb-major, stride), nu[0..8] likewise
BN_KERNEL k_fe_easy_head(const int32_t* in, size_t n, size_t stride, int32_t* head, int32_t* nu) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const FeEasyHead h = fe_easy_head(fp12_load_limbs(in + i, stride));
  fp2_store_limbs(head + i, stride, h.c0); fp2_store_limbs(head + 18 * stride + i, stride, h.c1);
  fp2_store_limbs(head + 36 * stride + i, stride, h.c2); fp2_store_limbs(head + 54 * stride + i, stride, h.nrm);
  store_fp(nu + i, stride, h.nu);
}
// lane j inverts tuples j, j + q, j + 2q, j + 3q (q = ceil(n / 4)) in place: coalesced, and a quarter of the lanes
BN_KERNEL k_fe_inv4(int32_t* nu, size_t n, size_t stride) {
  const size_t q = (n + 3) / 4;
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= q) return;
  Fp x[4];
  for (int t = 0; t < 4; ++t) { const size_t i = j + (size_t)t * q; x[t] = i < n ? load_fp(nu + i, stride) : fp_one(); }
  fp_inv4(x);
  for (int t = 0; t < 4; ++t) { const size_t i = j + (size_t)t * q; if (i < n) store_fp(nu + i, stride, x[t]); }
}
