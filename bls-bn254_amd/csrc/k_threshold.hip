// k_threshold.hip -- threshold-signature combine (BASELINE configs[4]: 1000-of-2000):
//     sigma = sum_i lambda_i sigma_i,   lambda_i = prod_{j != i} x_j / (x_j - x_i)
// (Scalar arithmetic scalar.rs:523-548, :216-219; Mul<Scalar> g1.rs:518-534, :821-841; Sum g1.rs:561-565).
// t is small (10^3), so the work is laid out to fill the chip and to keep the sequential chain short:
//   k_fr_decode          ids -> Montgomery limbs (limb-major), validity
//   k_lagrange_partial   lane (i, s): the s-th slice of the two length-t products of lambda_i          (t x S lanes)
//   k_lagrange_finish    lane i: combine the S partials, ONE inversion per lane (all lanes invert concurrently: the SIMD
//                        lanes are there anyway, so Montgomery's batch-inversion trick would lengthen the critical path
//                        by a scan instead of shortening it), lambda_i -> canonical words -> GLV halves k1, k2 (glv.h)
//   k_msm_window         lane (i, h, w): digit_w(k_h) * P_h  with P_0 = sigma_i, P_1 = phi(sigma_i) = (beta x, y), signs
//                        folded into the point; the 256 lanes of a workgroup are summed through LDS       (2t x 32 lanes)
//   k_msm_finish         lane w: sum of its window's partials, 4w doublings, tree over the 32 windows, affine bytes
// The value is the same group element the reference's t x 255-step ladders + Sum produce, so the bytes are identical.
#include "lane_ops.h"
#include "fr29.h"
#include "glv.h"
#include "kernels.h"
using namespace bn;

namespace {
constexpr int MSM_WINDOWS = 32;        // 4-bit windows over the 128-bit GLV halves

__device__ inline void store_fr(int32_t* ws, size_t stride, const Fr& a) { for (int k = 0; k < NL; ++k) ws[(size_t)k * stride] = a.l[k]; }
__device__ inline Fr load_fr(const int32_t* ws, size_t stride) { Fr r; for (int k = 0; k < NL; ++k) r.l[k] = ws[(size_t)k * stride]; return r; }

// a^(r-2) with 4-bit windows: 252 squarings + 63 multiplications + 14 for the table (fr_inv: 256 + 256)
__device__ Fr fr_inv_w4(const Fr& a) {
  Fr tab[16];
  tab[0] = fr_const(bnc::FR_ONE); tab[1] = a;
  for (int i = 2; i < 16; ++i) tab[i] = fr_mul(tab[i - 1], a);
  Fr r = tab[(int)(bnc::EXP_RM2[3] >> 60)];
  for (int w = 62; w >= 0; --w) {
    uint64_t word = w >= 48 ? bnc::EXP_RM2[3] : w >= 32 ? bnc::EXP_RM2[2] : w >= 16 ? bnc::EXP_RM2[1] : bnc::EXP_RM2[0];
    int d = (int)((word >> ((w & 15) * 4)) & 15);
    r = fr_mul(r, r); r = fr_mul(r, r); r = fr_mul(r, r); r = fr_mul(r, r);
    r = fr_mul(r, tab[d]);
  }
  return r;
}
__device__ inline void store_g1p_lds(int32_t* lds, unsigned tid, const G1P& p) {
  for (int k = 0; k < NL; ++k) { lds[k * 256 + tid] = p.x.l[k]; lds[(9 + k) * 256 + tid] = p.y.l[k]; lds[(18 + k) * 256 + tid] = p.z.l[k]; }
}
__device__ inline G1P load_g1p_lds(const int32_t* lds, unsigned tid) {
  G1P p;
  for (int k = 0; k < NL; ++k) { p.x.l[k] = lds[k * 256 + tid]; p.y.l[k] = lds[(9 + k) * 256 + tid]; p.z.l[k] = lds[(18 + k) * 256 + tid]; }
  BN_TRK(set_trk(p.x, 0, 1, 0, 0.012, 2); set_trk(p.y, 0, 1, 0, 0.012, 2); set_trk(p.z, 0, 1, 0, 0.012, 2);)
  return p;
}
__device__ inline G1P canon_g1p(const G1P& p) { return {fp_canon(p.x), fp_canon(p.y), fp_canon(p.z)}; }
}  // namespace

// status[i] bit 0: id decodes (< r) and is non-zero
BN_KERNEL k_fr_decode(const uint8_t* ids, size_t t, int32_t* x_ws, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t) return;
  bool ok;
  Fr x = fr_from_be(ids + 32 * i, ok);
  store_fr(x_ws + i, t, x);
  status[i] = (ok && !fr_is_zero(x)) ? 1 : 0;
}
// grid (ceil(t / 256), S): slice s = blockIdx.y covers j in [s J, (s + 1) J).  x_j is read at a wave-uniform address.
BN_KERNEL k_lagrange_partial(const int32_t* x_ws, size_t t, size_t J, int32_t* pnum, int32_t* pden, uint8_t* dup) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t) return;
  const size_t s = blockIdx.y, lo = s * J, hi = (lo + J < t) ? lo + J : t;
  Fr xi = load_fr(x_ws + i, t);
  Fr num = fr_const(bnc::FR_ONE), den = num, one = num;
  bool d = false;
  for (size_t j = lo; j < hi; ++j) {
    Fr xj = load_fr(x_ws + j, t);
    Fr df = fr_sub(xj, xi);
    bool self = j == i;
    d |= !self & fr_is_zero(df);
    num = fr_mul(num, fr_select(self, one, xj));
    den = fr_mul(den, fr_select(self, one, df));
  }
  const size_t st = t * gridDim.y;
  store_fr(pnum + s * t + i, st, num);
  store_fr(pden + s * t + i, st, den);
  if (d) dup[i] = 1;                 // some other id equals id_i
}
// lambda_i = prod_s num / prod_s den; outputs: canonical big-endian bytes (scalars, for the debug / test surface) and the
// GLV halves, limb-major words: glv_ws[(w) * t + i] for w = 0..3 (k1), 4..7 (k2), 8 (bit 0: k1 < 0, bit 1: k2 < 0)
BN_KERNEL k_lagrange_finish(const int32_t* pnum, const int32_t* pden, size_t t, size_t S, uint8_t* scalars, uint32_t* glv_ws) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t) return;
  const size_t st = t * S;
  Fr num = load_fr(pnum + i, st), den = load_fr(pden + i, st);
  for (size_t s = 1; s < S; ++s) { num = fr_mul(num, load_fr(pnum + s * t + i, st)); den = fr_mul(den, load_fr(pden + s * t + i, st)); }
  Fr lam = fr_mul(num, fr_inv_w4(den));
  Fr one; for (int k = 0; k < NL; ++k) one.l[k] = k == 0;
  Fr c = fr_mul(lam, one);                                  // out of Montgomery form, canonical
  uint32_t w[8];
  limbs_to_words(w, c.l);
  if (scalars) for (int j = 0; j < 8; ++j) store_be32(scalars + 32 * i + 4 * (7 - j), w[j]);
  GlvSplit g = glv_split(w);
  for (int j = 0; j < 4; ++j) { glv_ws[(size_t)j * t + i] = g.k1[j]; glv_ws[(size_t)(4 + j) * t + i] = g.k2[j]; }
  glv_ws[(size_t)8 * t + i] = (g.neg1 ? 1u : 0u) | (g.neg2 ? 2u : 0u);
}
// grid (ceil(2t / 256), 32): window w = blockIdx.y; lane q < t: (sigma_q, k1), lane q >= t: (phi(sigma_{q-t}), k2).
// part: limb-major projective points, 27 limbs, element index w * n_chunks + blockIdx.x, stride 32 * n_chunks.
BN_KERNEL k_msm_window(const uint8_t* g1, const uint32_t* glv_ws, size_t t, int32_t* part, uint8_t* status) {
  __shared__ int32_t lds[27 * 256];
  const unsigned tid = threadIdx.x;
  const size_t q = (size_t)blockIdx.x * 256 + tid;
  const unsigned w = blockIdx.y;
  G1P acc = proj_identity<Fp>();
  if (q < 2 * t) {
    const bool second = q >= t;
    const size_t i = second ? q - t : q;
    bool ok;
    G1A a = g1_decode(g1 + 64 * i, ok);
    if (w == 0 && !second) status[i] = ok ? 1 : 0;
    const uint32_t flags = glv_ws[(size_t)8 * t + i];
    const bool neg = second ? (flags & 2u) != 0 : (flags & 1u) != 0;
    const uint32_t word = glv_ws[(size_t)((second ? 4 : 0) + (w >> 3)) * t + i];
    const uint32_t d = (word >> ((w & 7) * 4)) & 15u;
    if (second) a.x = fp_mul(a.x, fp_const(bnc::GLV_BETA));
    a.y = fp_select(neg, fp_norm(fp_neg(a.y)), a.y);
    G1P p = proj_from_affine(a);
    for (int b = 3; b >= 0; --b) {
      acc = proj_dbl(acc);
      G1P s = proj_add(acc, p);
      acc = proj_select((d >> b) & 1, s, acc);
    }
  }
  // sum of the workgroup's 256 points through LDS (complete additions: identities are fine)
  for (unsigned s = 128; s > 0; s >>= 1) {
    if (tid >= s && tid < 2 * s) store_g1p_lds(lds, tid, canon_g1p(acc));
    __syncthreads();
    if (tid < s) acc = proj_add(acc, load_g1p_lds(lds, tid + s));
    __syncthreads();
  }
  if (tid == 0) {
    const size_t n_chunks = gridDim.x, e = (size_t)w * n_chunks + blockIdx.x, st = (size_t)MSM_WINDOWS * n_chunks;
    G1P c = canon_g1p(acc);
    for (int k = 0; k < NL; ++k) { part[(size_t)k * st + e] = c.x.l[k]; part[(size_t)(9 + k) * st + e] = c.y.l[k]; part[(size_t)(18 + k) * st + e] = c.z.l[k]; }
  }
}
// one workgroup of 64 lanes; lanes 0..31 own a window each
__global__ void __launch_bounds__(64) k_msm_finish(const int32_t* part, size_t n_chunks, uint8_t* out) {
  __shared__ int32_t lds[27 * 256];
  const unsigned tid = threadIdx.x;
  G1P acc = proj_identity<Fp>();
  if (tid < MSM_WINDOWS) {
    const size_t st = (size_t)MSM_WINDOWS * n_chunks;
    for (size_t c = 0; c < n_chunks; ++c) {
      const size_t e = (size_t)tid * n_chunks + c;
      G1P p;
      for (int k = 0; k < NL; ++k) { p.x.l[k] = part[(size_t)k * st + e]; p.y.l[k] = part[(size_t)(9 + k) * st + e]; p.z.l[k] = part[(size_t)(18 + k) * st + e]; }
      BN_TRK(set_trk(p.x, 0, 1, 0, 0.006, 1); set_trk(p.y, 0, 1, 0, 0.006, 1); set_trk(p.z, 0, 1, 0, 0.006, 1);)
      acc = proj_add(acc, p);
    }
    // times 2^(4 tid): a run of doublings in Jacobian coordinates (2M + 5S each); the loop bound differs per lane, the
    // wave runs max = 124 iterations with the finished lanes masked off
    G1P j = proj_to_jac(acc);
    for (unsigned b = 0; b < 4 * tid; ++b) j = jac_dbl(j);
    acc = proj_from_jac(j);
  }
  for (unsigned s = 16; s > 0; s >>= 1) {
    if (tid >= s && tid < 2 * s) store_g1p_lds(lds, tid, canon_g1p(acc));
    __syncthreads();
    if (tid < s) acc = proj_add(acc, load_g1p_lds(lds, tid + s));
    __syncthreads();
  }
  if (tid == 0) g1_encode(out, g1_to_affine(acc));
}
// pairwise sums along the chunk axis of the window partials, for batches with many chunks: out[w][c] = in[w][2c] + in[w][2c+1]
BN_KERNEL k_msm_fold(const int32_t* in, size_t n_in, int32_t* out) {
  const size_t n_out = (n_in + 1) >> 1;
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_out * MSM_WINDOWS) return;
  const size_t w = e / n_out, c = e % n_out, si = (size_t)MSM_WINDOWS * n_in, so = (size_t)MSM_WINDOWS * n_out;
  auto load = [&](size_t idx) {
    G1P p;
    for (int k = 0; k < NL; ++k) { p.x.l[k] = in[(size_t)k * si + idx]; p.y.l[k] = in[(size_t)(9 + k) * si + idx]; p.z.l[k] = in[(size_t)(18 + k) * si + idx]; }
    return p;
  };
  G1P a = load(w * n_in + 2 * c);
  if (2 * c + 1 < n_in) a = proj_add(a, load(w * n_in + 2 * c + 1));
  G1P r = canon_g1p(a);
  for (int k = 0; k < NL; ++k) { out[(size_t)k * so + e] = r.x.l[k]; out[(size_t)(9 + k) * so + e] = r.y.l[k]; out[(size_t)(18 + k) * so + e] = r.z.l[k]; }
}
