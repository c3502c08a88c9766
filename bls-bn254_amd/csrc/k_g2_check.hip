// k_g2_check.hip -- public-key validation: decode, on curve, psi-based subgroup test (g2.rs:350-414, :733-736).
// Own translation unit, tower / curve functions force-inlined (-DBN_FORCE_INLINE): no Fp2-sized values
// passed through the stack between outlined functions.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_g2_check(const uint8_t* g2, size_t n, uint8_t* ok_bytes, uint8_t* bitmap) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = i < n ? lane_g2_check(g2 + 128 * i) : false;
  if (ok_bytes && i < n) ok_bytes[i] = ok;
  if (bitmap) write_ballot(bitmap, n, i, ok);
}
