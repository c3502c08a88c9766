// k_rlc.hip -- random-linear-combination batch verification (builds on
// Gt::mul_by_scalar-style bilinearity, pairings.rs:585-600).
//
// For a group of G tuples with independent 64-bit scalars r_i:
//     prod_i [ e(sig_i, -G2gen) e(H_i, pk_i) ]^(r_i)  =  e(sum_i r_i sig_i, -G2gen) * prod_i e(r_i H_i, pk_i)
// so ONE final exponentiation (and one fixed-Q Miller loop) serves the whole group; if the product is 1 every
// eligible tuple of the group is valid except with probability 2^-64, otherwise the host re-verifies the group
// tuple by tuple with the exact path.  Tuples that fail decoding / on-curve / subgroup checks are not eligible:
// they contribute nothing to the product and are reported invalid directly.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

__device__ inline void store_g1p_ws(int32_t* ws, size_t stride, const G1P& p) {
  store_fp(ws, stride, p.x); store_fp(ws + 9 * stride, stride, p.y); store_fp(ws + 18 * stride, stride, p.z);
}
__device__ inline G1P load_g1p_ws(const int32_t* ws, size_t stride) {
  return {load_fp(ws, stride), load_fp(ws + 9 * stride, stride), load_fp(ws + 18 * stride, stride)};
}
// 64-bit scalar * P, branch-free
__device__ inline G1P g1_mul_u64(const G1P& p, uint64_t k) {
  G1P acc = proj_identity<Fp>();
  for (int i = 63; i >= 0; --i) {
    acc = proj_dbl(acc);
    G1P s = proj_add(acc, p);
    acc = proj_select((k >> i) & 1, s, acc);
  }
  return acc;
}
// Per tuple: eligibility, r_i = SHA-256(seed || i || pk_i || sig_i || H(msg_i))[0..8) (non-zero), A_i = r_i sig_i (projective,
// a_ws, stride n_pad), B_i = r_i H_i (affine limbs, b_ws, stride n).  Lanes n <= i < n_pad are padding: identity / not eligible.
BN_KERNEL k_rlc_prep(const uint8_t* pks, const uint8_t* sigs, const int32_t* h_ws, const uint8_t* sub_ok, const uint8_t* seed,
                     size_t n, size_t n_pad, int32_t* a_ws, int32_t* b_ws, uint8_t* elig) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  G1P A = proj_identity<Fp>();
  G1A B; B.x = fp_one(); B.y = fp_norm(fp_add(fp_one(), fp_one())); B.inf = false;      // generator placeholder
  bool ok = false;
  if (i < n) {
    bool oks, okp;
    G1A sig = g1_decode(sigs + 64 * i, oks);
    G2A pk = g2_decode(pks + 128 * i, okp);
    ok = oks & !sig.inf & g1_on_curve(sig) & okp & !pk.inf & (sub_ok[i] != 0);         // sub_ok already covers pk on curve + torsion free
    Sha256 s; sha256_init(s);
    sha256_update(s, seed, 32);
    for (int k = 0; k < 8; ++k) sha256_byte(s, (uint8_t)((uint64_t)i >> (8 * k)));
    sha256_update(s, pks + 128 * i, 128); sha256_update(s, sigs + 64 * i, 64);
    // bind the message too, through its hash point H(msg_i) (canonical Montgomery limbs as k_hash_to_g1 stored them)
    for (int k = 0; k < 18; ++k) {
      uint32_t v = (uint32_t)h_ws[(size_t)k * n + i];
      sha256_byte(s, (uint8_t)v); sha256_byte(s, (uint8_t)(v >> 8)); sha256_byte(s, (uint8_t)(v >> 16)); sha256_byte(s, (uint8_t)(v >> 24));
    }
    uint8_t dg[32]; sha256_final(s, dg);
    uint64_t r = 0;
    for (int k = 0; k < 8; ++k) r = (r << 8) | dg[k];
    r = r ? r : 1;                                                                        // never zero; all 64 bits of the digest are kept
    G1A h; h.x = load_fp(h_ws + i, n); h.y = load_fp(h_ws + 9 * n + i, n); h.inf = false;
    G1A gp = B;
    sig.x = fp_select(ok, sig.x, gp.x); sig.y = fp_select(ok, sig.y, gp.y); sig.inf = false;
    G1P As = g1_mul_u64(proj_from_affine(sig), r);
    A = proj_select(ok, As, A);
    G1A Bh = g1_to_affine(g1_mul_u64(proj_from_affine(h), r));
    B.x = fp_select(ok, Bh.x, B.x); B.y = fp_select(ok, Bh.y, B.y);
  }
  store_g1p_ws(a_ws + i, n_pad, A);
  if (i < n) { store_fp(b_ws + i, n, B.x); store_fp(b_ws + 9 * n + i, n, B.y); }      // stride n: the layout k_miller_hpk reads
  elig[i] = ok ? 1 : 0;
}
// f_ws[i] = ONE where the tuple is not eligible (or padding)
BN_KERNEL k_fp12_mask_one(int32_t* f_ws, size_t stride, const uint8_t* elig, size_t n_pad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad || elig[i]) return;
  fp12_store_limbs(f_ws + i, stride, fp12_one());
}
// a[i] *= b[i]
BN_KERNEL k_fp12_mul_elem(int32_t* a, size_t sa, const int32_t* b, size_t sb, size_t m) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  fp12_store_limbs(a + i, sa, fp12_mul(fp12_load_limbs(a + i, sa), fp12_load_limbs(b + i, sb)));
}
// m projective G1 points (limb-major, stride) -> 64-byte encodings
BN_KERNEL k_g1p_to_bytes(const int32_t* ws, size_t stride, size_t m, uint8_t* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  g1_encode(out + 64 * i, g1_to_affine(load_g1p_ws(ws + i, stride)));
}
// compact the tuples listed in idx (fallback of failed groups): fixed-size records only (H is already computed)
BN_KERNEL k_rlc_gather(const uint32_t* idx, size_t m, const uint8_t* pks, const uint8_t* sigs, const int32_t* h_ws, size_t n,
                       const uint8_t* sub_ok, uint8_t* c_pks, uint8_t* c_sigs, int32_t* c_h, uint8_t* c_sub) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  size_t i = idx[j];
  for (int k = 0; k < 128; ++k) c_pks[128 * j + k] = pks[128 * i + k];
  for (int k = 0; k < 64; ++k) c_sigs[64 * j + k] = sigs[64 * i + k];
  for (int k = 0; k < 18; ++k) c_h[(size_t)k * m + j] = h_ws[(size_t)k * n + i];
  c_sub[j] = sub_ok[i];
}
