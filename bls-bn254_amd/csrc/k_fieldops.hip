// k_fieldops.hip -- batched field / tower primitives behind the debug ABI blsbn254_field_op_batch, and the Gt group
// operations (Gt multiply, Gt::mul_by_scalar pairings.rs:585-600).  One lane = one element.  The reference holds no
// vectors for Fp6 / Fp12, so this entry point fuzzed against the CPU oracle is the isolated pin of
// fp.rs:388-416, fp2.rs:377-437, fp6.rs:225-287, fp12.rs:120-219 on the device arithmetic (fp29.h / tower.h).
// Operands and results are canonical big-endian bytes: Fp 32 B; Fp2 64 B (c0 || c1); Fp6 192 B (c0.c0, c0.c1, c1.c0 ..);
// Fp12 384 B in the Gt::to_repr order (pairings.rs:499-514).
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

namespace {
__device__ inline Fp6 fp6_from_be(const uint8_t* in, bool& ok) {
  Fp6 a;
  a.c0 = fp2_from_be(in, ok); a.c1 = fp2_from_be(in + 64, ok); a.c2 = fp2_from_be(in + 128, ok);
  return a;
}
__device__ inline void fp6_to_be(uint8_t* out, const Fp6& a) { fp2_to_be(out, a.c0); fp2_to_be(out + 64, a.c1); fp2_to_be(out + 128, a.c2); }
}  // namespace

// op codes: include/blsbn254.h (BLSBN254_OP_*).  status[i] = 1 when every coefficient decoded (< p).
BN_KERNEL k_field_op(int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok = true;
  if (op < 16) {                                           // ---- Fp
    bool o1 = true, o2 = true;
    Fp x = fp_from_be(a + 32 * i, o1), y = b ? fp_from_be(b + 32 * i, o2) : fp_one(), r;
    ok = o1 & o2;
    switch (op) {
      case 0: r = fp_mul(x, y); break;
      case 1: r = fp_sqr(x); break;
      case 2: r = fp_inv(x); break;
      case 3: r = fp_add(x, y); break;
      case 4: r = fp_sub(x, y); break;
      case 5: r = fp_neg(x); break;
      case 6: { bool sq; Fp s = fp_sqrt_cand(x, sq); r = fp_select(sq, s, fp_zero()); } break;     // a root, or 0 when x is not a square
      case 7: r = fp_select(fp_is_square(x), fp_one(), fp_zero()); break;                          // Jacobi-symbol is_square as 1 / 0
      default: r = fp_mul(fp_lc2<9, 0>(x, x), fp_one()); break;                                     // 8: mul_by_3b (x9), fp.rs:414
    }
    fp_to_be(out + 32 * i, r);
  } else if (op < 32) {                                    // ---- Fp2
    bool o1 = true, o2 = true;
    Fp2 x = fp2_from_be(a + 64 * i, o1), y = b ? fp2_from_be(b + 64 * i, o2) : fp2_one(), r;
    ok = o1 & o2;
    switch (op) {
      case 16: r = fp2_mul(x, y); break;
      case 17: r = fp2_sqr(x); break;
      case 18: r = fp2_inv(x); break;
      case 19: r = fp2_mul_xi(x); break;
      case 20: r = fp2_norm(fp2_conj(x)); break;
      default: { Fp2 s = fp2_sqrt(x); r = fp2_select(fp2_is_zero(fp2_sub(fp2_sqr(s), x)), s, fp2_zero()); } break;   // 21: a root or 0
    }
    fp2_to_be(out + 64 * i, r);
  } else if (op < 48) {                                    // ---- Fp6
    bool o1 = true, o2 = true;
    Fp6 x = fp6_from_be(a + 192 * i, o1), y = b ? fp6_from_be(b + 192 * i, o2) : fp6_one(), r;
    ok = o1 & o2;
    switch (op) {
      case 32: r = fp6_mul(x, y); break;
      case 33: r = fp6_sqr(x); break;
      case 34: r = fp6_inv(x); break;
      default: r = fp6_mul_v(x); break;                    // 35: mul_by_non_residue, fp6.rs:146-152
    }
    fp6_to_be(out + 192 * i, r);
  } else {                                                 // ---- Fp12
    bool o1 = true, o2 = true;
    Fp12 x = fp12_from_be(a + 384 * i, o1), y = fp12_one(), r;
    if (b) y = fp12_from_be(b + 384 * i, o2);
    ok = o1 & o2;
    switch (op) {
      case 48: r = fp12_mul(x, y); break;
      case 49: r = fp12_sqr(x); break;
      case 50: r = fp12_inv(x); break;
      case 51: r = fp12_conj(x); break;
      case 52: r = fp12_frob<1>(x); break;
      case 53: r = fp12_frob<2>(x); break;
      case 54: r = fp12_frob<3>(x); break;
      case 55: r = fp12_cyclotomic_sqr(x); break;          // equals x^2 only for x in the cyclotomic subgroup; compared formula against formula
      default: r = fp12_mul_by_034(x, y.c0.c0, y.c1.c0, y.c1.c1); break;   // 56: sparse product with (b.c0.c0, b.c1.c0, b.c1.c1), the line slots (E7)
    }
    fp12_to_be(out + 384 * i, r);
  }
  status[i] = ok ? 1 : 0;
}

// Gt::mul_by_scalar (pairings.rs:585-600): gt^k for a 256-bit big-endian scalar, square-and-multiply with a branch-free
// select (per-lane scalars).  Plain Fp12 squarings: valid for every Fp12 input, like the reference's loop.
BN_KERNEL k_gt_pow(const uint8_t* gt, const uint8_t* scalars, size_t n, uint8_t* out, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok;
  Fp12 x = fp12_from_be(gt + 384 * i, ok), acc = fp12_one();
  for (int bit = 0; bit < 256; ++bit) {
    const int byte = bit >> 3, sh = 7 - (bit & 7);
    const bool on = (scalars[32 * i + byte] >> sh) & 1;
    acc = fp12_sqr(acc);
    Fp12 m = fp12_mul(acc, x);
    acc.c0 = {fp2_select(on, m.c0.c0, acc.c0.c0), fp2_select(on, m.c0.c1, acc.c0.c1), fp2_select(on, m.c0.c2, acc.c0.c2)};
    acc.c1 = {fp2_select(on, m.c1.c0, acc.c1.c0), fp2_select(on, m.c1.c1, acc.c1.c1), fp2_select(on, m.c1.c2, acc.c1.c2)};
  }
  fp12_to_be(out + 384 * i, acc);
  status[i] = ok ? 1 : 0;
}
