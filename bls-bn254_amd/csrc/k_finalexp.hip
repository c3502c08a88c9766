// k_finalexp.hip -- hard-part glue kernels of the final exponentiation (h1, h2, h3).
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

__device__ inline void write_ballot(uint8_t* bitmap, size_t n, size_t i, bool bit) {
  unsigned long long m = __ballot(bit);
  unsigned lane = threadIdx.x & 63;
  size_t base = (i - lane) >> 3;                       // first byte of this wave's 64 tuples
  size_t nbytes = (n + 7) >> 3;
  if (lane < 8 && base + lane < nbytes) bitmap[base + lane] = (uint8_t)(m >> (8 * lane));
}

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
// mode 0: verify -> bitmap bit = flags ok && subgroup ok && result == 1 ; mode 1/2: Gt bytes ; mode 3: *is_one (n == 1)
// mode 4: gt_bytes[i] = (result == 1) as one byte per element (RLC group check)
BN_KERNEL k_fe_h3(const int32_t* t, const int32_t* a, const int32_t* c, const int32_t* b2, const int32_t* x0, size_t n, size_t stride,
                  const uint8_t* flags, const uint8_t* sub_ok, uint8_t* bitmap, uint8_t* gt_bytes, int* is_one, int mode) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool bit = false;
  if (i < n) {
    Fp12 r = fe_h3(fp12_load_limbs(t + i, stride), fp12_load_limbs(a + i, stride), fp12_load_limbs(c + i, stride),
                   fp12_load_limbs(b2 + i, stride), fp12_load_limbs(x0 + i, stride));
    if (mode == 0) bit = fp12_is_one(r) & (flags[i] == (FLAG_SIG_OK | FLAG_PK_OK)) & (sub_ok[i] != 0);
    else if (mode == 3) *is_one = fp12_is_one(r) ? 1 : 0;
    else if (mode == 4) gt_bytes[i] = fp12_is_one(r) ? 1 : 0;
    else fp12_to_be(gt_bytes + 384 * i, r);
  }
  if (mode == 0) write_ballot(bitmap, n, i, bit);
}
