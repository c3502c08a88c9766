// k_misc.hip -- hash-to-curve, point checks, product / sum trees, Lagrange coefficients, reductions,
// and the VALU roofline probe.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;


__device__ inline void store_g1p(int32_t* ws, size_t stride, const G1P& p) {
  store_fp(ws, stride, p.x); store_fp(ws + 9 * stride, stride, p.y); store_fp(ws + 18 * stride, stride, p.z);
}
__device__ inline G1P load_g1p(const int32_t* ws, size_t stride) {
  return {load_fp(ws, stride), load_fp(ws + 9 * stride, stride), load_fp(ws + 18 * stride, stride)};
}
BN_KERNEL k_hash_to_g2(const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t* dst, uint32_t dst_len,
                       uint8_t* out_bytes, int ro) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g2_encode(out_bytes + 128 * i, lane_hash_to_g2(msgs + off[i], (size_t)(off[i + 1] - off[i]), dst, dst_len, ro != 0));
}
BN_KERNEL k_g1_check(const uint8_t* g1, size_t n, uint8_t* bitmap) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = i < n ? lane_g1_check(g1 + 64 * i) : false;
  write_ballot(bitmap, n, i, ok);
}
BN_KERNEL k_fp12_from_bytes(const uint8_t* in, size_t n, int32_t* f_ws, size_t f_stride, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok;
  Fp12 f = fp12_from_be(in + 384 * i, ok);
  fp12_store_limbs(f_ws + i, f_stride, f);
  status[i] = ok ? 1 : 0;
}
BN_KERNEL k_fp12_to_bytes(const int32_t* f_ws, size_t n, size_t stride, uint8_t* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fp12_to_be(out + 384 * i, fp12_load_limbs(f_ws + i, stride));
}
BN_KERNEL k_fp12_mul_pairs(const int32_t* in, size_t n_in, size_t in_stride, int32_t* out, size_t out_stride) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t n_out = (n_in + 1) >> 1;
  if (i >= n_out) return;
  Fp12 a = fp12_load_limbs(in + 2 * i, in_stride);
  if (2 * i + 1 < n_in) a = fp12_mul(a, fp12_load_limbs(in + 2 * i + 1, in_stride));
  fp12_store_limbs(out + i, out_stride, a);
}
BN_KERNEL k_g1_load(const uint8_t* g1, size_t n, int32_t* ws, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok;
  G1A a = g1_decode(g1 + 64 * i, ok);
  store_g1p(ws + i, n, proj_from_affine(a));
  status[i] = ok ? 1 : 0;
}
// one G1 point (64 bytes) -> affine Montgomery limbs in slot `slot` of a limb-major H workspace; *ok = decodes, not the
// identity, on the curve (the aggregate signature joins the batch of (H(msg), pk) pairs as the pair (sig, -G2gen))
BN_KERNEL k_g1_to_ws(const uint8_t* g1, int32_t* h_ws, size_t slot, size_t stride, uint8_t* ok) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  bool okd;
  G1A p = g1_decode(g1, okd);
  const bool good = okd & !p.inf & g1_on_curve(p);
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one()));
  store_fp(h_ws + slot, stride, fp_select(good, p.x, gp.x)); store_fp(h_ws + 9 * stride + slot, stride, fp_select(good, p.y, gp.y));
  *ok = good ? 1 : 0;
}
// n G1 points (64 bytes each) -> affine Montgomery limbs in a limb-major H workspace (stride n); status bit 0: decodes,
// bit 1: is the identity (such a pair contributes 1 to a Miller product and is replaced by the generator here)
BN_KERNEL k_g1_to_ws_batch(const uint8_t* g1, size_t n, int32_t* h_ws, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool okd;
  G1A p = g1_decode(g1 + 64 * i, okd);
  const bool use = okd & !p.inf;
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one()));
  store_fp(h_ws + i, n, fp_select(use, p.x, gp.x)); store_fp(h_ws + 9 * n + i, n, fp_select(use, p.y, gp.y));
  status[i] = (uint8_t)((okd ? 1 : 0) | (p.inf ? 2 : 0));
}
BN_KERNEL k_g1_add_pairs(const int32_t* in, size_t n_in, size_t in_stride, int32_t* out, size_t out_stride) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t n_out = (n_in + 1) >> 1;
  if (i >= n_out) return;
  G1P a = load_g1p(in + 2 * i, in_stride);
  if (2 * i + 1 < n_in) a = proj_add(a, load_g1p(in + 2 * i + 1, in_stride));
  store_g1p(out + i, out_stride, a);
}
BN_KERNEL k_g1_to_bytes(const int32_t* ws, size_t stride, uint8_t* out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  g1_encode(out, g1_to_affine(load_g1p(ws, stride)));
}
// compressed codecs: mode 0 compress (uncompressed in -> compressed out), 1 decompress; status 1 = ok
BN_KERNEL k_g1_codec(const uint8_t* in, size_t n, uint8_t* out, uint8_t* status, int mode) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok;
  if (mode == 0) { G1A p = g1_decode(in + 64 * i, ok); g1_compress(out + 32 * i, p); }
  else { G1A p = g1_decompress(in + 32 * i, ok); g1_encode(out + 64 * i, p); }
  status[i] = ok ? 1 : 0;
}
BN_KERNEL k_g2_codec(const uint8_t* in, size_t n, uint8_t* out, uint8_t* status, int mode) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok;
  if (mode == 0) { G2A q = g2_decode(in + 128 * i, ok); g2_compress(out + 64 * i, q); }
  else { G2A q = g2_decompress(in + 64 * i, ok); g2_encode(out + 128 * i, q); }
  status[i] = ok ? 1 : 0;
}
// VALU roofline probe: 8 independent 64-bit multiply-accumulate chains per lane (compiler-selected
// v_mad_u64_u32, each with its own carry-out SGPR pair, no shared VCC), every CU busy at 4 waves per SIMD.  The
// denominator of bench.py's roofline.frac is measured in the same run (BASELINE.md section 3 "same-run rule").
// kind 1 runs the same shape with plain 32-bit VOP2 work (v_add_u32 / v_xor_b32) instead: the full-rate issue ceiling.
// Lane 0 of every wave stamps the shader-cycle counter and the 100 MHz constant clock around the loop into
// `stamps` (4 x u64 per wave; a buffer nothing else reads): their ratio is the clock the chip holds under this load.
__global__ void __launch_bounds__(256) k_valu_peak(uint32_t* out, uint32_t seed, int iters, int kind, uint64_t* stamps) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a = seed * 2654435761u + tid, b = (seed ^ 0x9e3779b9u) + tid * 7u;
  uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  uint32_t res;
  if (kind == 0) {
    uint64_t acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = tid + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        acc[c] = (uint64_t)a * (uint32_t)(b + c) + acc[c];
        asm("" : "+v"(acc[c]));
      }
    }
    uint64_t s = 0;
    for (int c = 0; c < 8; ++c) s ^= acc[c];
    res = (uint32_t)s ^ (uint32_t)(s >> 32);
  } else {
    uint32_t acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = tid + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        uint32_t v = acc[c];
        v = (c & 1) ? (v ^ b) : v + a;                  // exactly one VOP2 instruction per chain step
        asm("" : "+v"(v));
        acc[c] = v;
      }
    }
    res = 0;
    for (int c = 0; c < 8; ++c) res ^= acc[c];
  }
  uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[tid] = res;
  if (stamps && (threadIdx.x & 63) == 0) {
    uint64_t* w = stamps + 2 * (size_t)(tid >> 6);
    w[0] = c1 - c0; w[1] = r1 - r0;
  }
}
// status reductions
__global__ void k_status_reduce(const uint8_t* status, size_t n, uint8_t want_mask, uint8_t want_val, int* first_bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (status[i] & want_mask) != want_val) atomicMin(first_bad, (int)(i > 0x7ffffffe ? 0x7ffffffe : i));
}
__global__ void k_and_reduce(const uint8_t* flags, const uint8_t* sub_ok, size_t n, int* all_ok) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (flags[i] != 1 || sub_ok[i] == 0)) atomicAnd(all_ok, 0);
}
