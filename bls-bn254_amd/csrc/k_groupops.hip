// k_groupops.hip -- group operations on caller-supplied points: Mul<Scalar> for G1 / G2 (g1.rs:518-534, :821-841;
// g2.rs:866-886), impl Sum for G2Projective (g2.rs:579-583) as a segmented sum, and the affine encodings that hand the
// per-group sums of public keys to the verify pipeline (FastAggregateVerify: one message signed by many keys).
// One lane per point / per chunk; the complete RCB group law of curve.h, so no lane ever branches on its data.
#include "lane_ops.h"
#include "fr29.h"
#include "kernels.h"
using namespace bn;

__device__ inline void store_g2p(int32_t* ws, size_t stride, const G2P& p) {
  store_fp(ws, stride, p.x.c0); store_fp(ws + 9 * stride, stride, p.x.c1);
  store_fp(ws + 18 * stride, stride, p.y.c0); store_fp(ws + 27 * stride, stride, p.y.c1);
  store_fp(ws + 36 * stride, stride, p.z.c0); store_fp(ws + 45 * stride, stride, p.z.c1);
}
__device__ inline G2P load_g2p(const int32_t* ws, size_t stride) {
  return {{load_fp(ws, stride), load_fp(ws + 9 * stride, stride)},
          {load_fp(ws + 18 * stride, stride), load_fp(ws + 27 * stride, stride)},
          {load_fp(ws + 36 * stride, stride), load_fp(ws + 45 * stride, stride)}};
}
// 32 bytes big-endian -> four little-endian 64-bit words; ok = value < r (scalar.rs:229-239)
__device__ inline bool load_scalar_words(const uint8_t* sk, uint64_t k[4]) {
  bool ok;
  (void)fr_from_be(sk, ok);
  for (int w = 0; w < 4; ++w) {
    uint64_t v = 0;
    for (int j = 0; j < 8; ++j) v = (v << 8) | sk[8 * (3 - w) + j];
    k[w] = v;
  }
  return ok;
}
// out_i = [k_i] P_i.  status bit 0: the point decodes and is on the curve; bit 1: the scalar is canonical (< r).
BN_KERNEL k_g1_mul(const uint8_t* g1, const uint8_t* scalars, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t k[4];
  const bool oks = load_scalar_words(scalars + 32 * i, k);
  bool okd;
  G1A p = g1_decode(g1 + 64 * i, okd);
  const bool okp = okd & g1_on_curve(p);
  G1A g; g.x = fp_one(); g.y = fp_norm(fp_add(fp_one(), fp_one())); g.inf = false;     // stand-in for an invalid point: keeps the lane on the curve
  p.x = fp_select(okp, p.x, g.x); p.y = fp_select(okp, p.y, g.y); p.inf = okp & p.inf;
  g1_encode(out + 64 * i, g1_to_affine(proj_mul_win4(proj_from_affine(p), k)));
  status[i] = (uint8_t)((okp ? 1 : 0) | (oks ? 2 : 0));
}
BN_KERNEL k_g2_mul(const uint8_t* g2, const uint8_t* scalars, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t k[4];
  const bool oks = load_scalar_words(scalars + 32 * i, k);
  bool okd;
  G2A p = g2_decode(g2 + 128 * i, okd);
  const bool okp = okd & g2_on_curve(p);
  G2A g; g.x = fp2_const(bnc::G2_GEN_X); g.y = fp2_const(bnc::G2_GEN_Y); g.inf = false;
  p.x = fp2_select(okp, p.x, g.x); p.y = fp2_select(okp, p.y, g.y); p.inf = okp & p.inf;
  g2_encode(out + 128 * i, g2_to_affine(proj_mul_win4(proj_from_affine(p), k)));
  status[i] = (uint8_t)((okp ? 1 : 0) | (oks ? 2 : 0));
}
// n uncompressed G2 points -> homogeneous limb-major workspace (54 x n limbs); ok[i] = decodes and is on the curve
// (an invalid point is stored as the identity so that the sums stay defined; its flag poisons its group).
BN_KERNEL k_g2_load(const uint8_t* g2, size_t n, int32_t* ws, uint8_t* ok) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool okd;
  G2A p = g2_decode(g2 + 128 * i, okd);
  const bool good = okd & g2_on_curve(p);
  p.inf = p.inf | !good;
  store_g2p(ws + i, n, proj_from_affine(p));
  ok[i] = good ? 1 : 0;
}
// One lane per chunk: out[c] = sum of items chunk_start[c] .. + chunk_len[c] of in_ws, ok_out[c] = AND of their flags
// (an empty chunk -- an empty group -- gives the identity with ok = 0).
BN_KERNEL k_g2_seg_sum(const int32_t* in_ws, size_t in_stride, const uint8_t* ok_in, const uint32_t* chunk_start, const uint32_t* chunk_len, size_t m,
                       int32_t* out_ws, size_t out_stride, uint8_t* ok_out) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m) return;
  const size_t s0 = chunk_start[c];
  const uint32_t len = chunk_len[c];
  G2P acc = proj_identity<Fp2>();
  uint8_t ok = len ? 1 : 0;
#pragma unroll 1
  for (uint32_t j = 0; j < len; ++j) {
    acc = proj_add(acc, load_g2p(in_ws + s0 + j, in_stride));
    ok &= ok_in[s0 + j];
  }
  store_g2p(out_ws + c, out_stride, acc);
  ok_out[c] = ok;
}
// m sums -> uncompressed encodings.  poison = true (FastAggregateVerify): a group whose flag is 0 gets an encoding that
// cannot decode (x.c1 = 0xff..ff >= p), so the verify pipeline clears its bit; the identity keeps its own encoding, which
// KeyValidate rejects there.
BN_KERNEL k_g2p_to_bytes(const int32_t* ws, size_t stride, const uint8_t* ok, size_t m, uint8_t* out, int poison) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  g2_encode(out + 128 * i, g2_to_affine(load_g2p(ws + i, stride)));
  if (poison && !ok[i]) for (int b = 0; b < 32; ++b) out[128 * i + b] = 0xff;
}
