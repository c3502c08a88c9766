// k_hash_g1.hip -- hash_to_curve for G1: XMD-SHA-256, two SVDW maps, one addition (g1.rs:910-928).
// Own translation unit, tower / curve functions force-inlined (-DBN_FORCE_INLINE): no Fp2-sized values
// passed through the stack between outlined functions.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_hash_to_g1(const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t* dst, uint32_t dst_len,
                       int32_t* h_ws, size_t h_stride, uint8_t* out_bytes, int mode) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* m = msgs + off[i];
  size_t len = (size_t)(off[i + 1] - off[i]);
  if (mode == 3) {            // homogeneous (X : Y : Z), 27 limbs, for the table-only Miller loop
    G1P hp = lane_hash_to_g1_proj(m, len, dst, dst_len);
    store_fp(h_ws + i, h_stride, hp.x); store_fp(h_ws + 9 * h_stride + i, h_stride, hp.y); store_fp(h_ws + 18 * h_stride + i, h_stride, hp.z);
    return;
  }
  G1A h = mode == 2 ? lane_encode_to_g1(m, len, dst, dst_len) : lane_hash_to_g1(m, len, dst, dst_len);
  if (mode == 0) { store_fp(h_ws + i, h_stride, h.x); store_fp(h_ws + 9 * h_stride + i, h_stride, h.y); }
  else g1_encode(out_bytes + 64 * i, h);
}
