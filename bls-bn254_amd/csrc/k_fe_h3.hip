// k_fe_h3.hip -- last step of the final exponentiation: eight Fp12 products and three Frobenius maps over the five
// phase results, comparison with 1 and the bitmap write.  Compiled with outlined tower functions: fully inlined it
// is slower (1.9 -> 2.1 ms, and 4.1 ms with the operand loads staged), its five live Fp12 operands spill more than
// the calls cost.
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
