// k_fe_h3.hip -- last step of the final exponentiation: eight Fp12 products and three Frobenius maps over the five
// phase results, comparison with 1 and the bitmap write.  Run as a small interpreter (fe_h3_loop, pairing.h): one
// inlined multiply-by-memory-operand in a loop; the straight-line form with eight inlined products was slower than
// outlined calls (2.1 vs 1.9 ms).
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;


// mode 0: verify -> bitmap bit = flags ok && subgroup ok && result == 1 ; mode 1/2: Gt bytes ; mode 3: *is_one (n == 1)
// mode 4: gt_bytes[i] = (result == 1) as one byte per element (RLC group check)
BN_KERNEL k_fe_h3(const int32_t* t, const int32_t* a, const int32_t* c, const int32_t* b2, const int32_t* x0, int32_t* tmp, size_t n, size_t stride,
                  const uint8_t* flags, const uint8_t* sub_ok, uint8_t* bitmap, uint8_t* gt_bytes, int* is_one, int mode) {
  __shared__ int32_t park_lds[108 * 256];        // each lane touches only its own column: no barrier needed
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool bit = false;
  if (i < n) {
    const uint32_t l4 = i * 4u;
    const Ws src[5] = {{const_cast<int32_t*>(t), stride, l4, true}, {const_cast<int32_t*>(a), stride, l4, true}, {const_cast<int32_t*>(c), stride, l4, true},
                       {const_cast<int32_t*>(b2), stride, l4, true}, {const_cast<int32_t*>(x0), stride, l4, true}};
    const Ws park = {park_lds, 256, threadIdx.x * 4u, false};
    Fp12 r = fe_h3_loop(src, Ws{tmp, stride, l4, true}, &park);
    if (mode == 0) bit = fp12_is_one(r) & (flags[i] == (FLAG_SIG_OK | FLAG_PK_OK)) & (sub_ok[i] != 0);
    else if (mode == 3) *is_one = fp12_is_one(r) ? 1 : 0;
    else if (mode == 4) gt_bytes[i] = fp12_is_one(r) ? 1 : 0;
    else fp12_to_be(gt_bytes + 384 * (size_t)i, r);
  }
  if (mode == 0) write_ballot(bitmap, n, i, bit);
}
