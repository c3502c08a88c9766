// k_sign.hip -- signing-side kernels: sig = [sk] H(msg) in G1, pk = [sk] G2gen.
// Branch-free double-and-add with the complete group law (every lane runs the same 256 steps).
// Replaces Mul<Scalar> for G1Projective / G2Projective (g1.rs:518-534,:821-841; g2.rs:866-886) composed with
// G1Projective::hash (g1.rs:910-919).  Scalars: 32 bytes big-endian, must be < r (scalar.rs:229-239).
#include "lane_ops.h"
#include "fr29.h"
#include "kernels.h"
using namespace bn;

__device__ inline bool load_scalar(const uint8_t* sk, uint64_t k[4]) {
  bool ok;
  (void)fr_from_be(sk, ok);                      // range check against r
  for (int w = 0; w < 4; ++w) {
    uint64_t v = 0;
    for (int j = 0; j < 8; ++j) v = (v << 8) | sk[8 * (3 - w) + j];
    k[w] = v;
  }
  return ok;
}
BN_KERNEL k_sign(const uint8_t* sks, const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t* dst, uint32_t dst_len,
                 uint8_t* sigs, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t k[4];
  bool ok = load_scalar(sks + 32 * i, k);
  G1A h = lane_hash_to_g1(msgs + off[i], (size_t)(off[i + 1] - off[i]), dst, dst_len);
  g1_encode(sigs + 64 * i, g1_to_affine(proj_mul_256(proj_from_affine(h), k)));
  status[i] = ok ? 1 : 0;
}
BN_KERNEL k_sk_to_pk(const uint8_t* sks, size_t n, uint8_t* pks, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t k[4];
  bool ok = load_scalar(sks + 32 * i, k);
  G2A g; g.x = fp2_const(bnc::G2_GEN_X); g.y = fp2_const(bnc::G2_GEN_Y); g.inf = false;
  g2_encode(pks + 128 * i, g2_to_affine(proj_mul_256(proj_from_affine(g), k)));
  status[i] = ok ? 1 : 0;
}
