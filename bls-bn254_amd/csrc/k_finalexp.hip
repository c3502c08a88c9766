// k_finalexp.hip -- final exponentiation kernels (easy part + Fuentes-Castaneda hard part).
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

BN_KERNEL k_final_exp(const int32_t* f_ws, size_t n, size_t f_stride, const uint8_t* flags, const uint8_t* sub_ok,
                      uint8_t* bitmap, uint8_t* gt_bytes, int mode) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool bit = false;
  if (i < n) {
    Fp12 f = final_exponentiation(fp12_load_limbs(f_ws + i, f_stride));
    if (mode == 0) bit = fp12_is_one(f) & (flags[i] == (FLAG_SIG_OK | FLAG_PK_OK)) & (sub_ok[i] != 0);
    else fp12_to_be(gt_bytes + 384 * i, f);
  }
  if (mode == 0) write_ballot(bitmap, n, i, bit);
}
BN_KERNEL k_final_exp_is_one(const int32_t* f_ws, size_t stride, int* out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  *out = fp12_is_one(final_exponentiation(fp12_load_limbs(f_ws, stride))) ? 1 : 0;
}


