// k_miller_hpk2.hip -- aggregate verify: TWO pairs (H(msg), pk) per lane sharing one f^2 per loop digit
// (multi_miller_loop's shape, pairings.rs:808-857), lane i takes pairs 2i and 2i + 1.  Per pair the shared squaring is
// halved and both lines of a step are folded into f as one product (23 Fp2 products for two lines instead of 2 x 13 + a
// second squaring): ~22 % fewer multiplications per pair than k_miller_hpk.  The product over lanes (k_fp12_mul_pairs tree)
// is then bit-identical to the product of the one-pair loops: field arithmetic is exact.
// Same compile policy as the other Miller units (-DBN_FORCE_INLINE -DBN_LC_MAD).  LDS per lane: the two G1 points (36 limbs),
// the parked running point (54) and the parked first line (54) = 144 of the 160 limbs a lane has at 256 lanes per CU.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

// n pairs, ceil(n / 2) lanes.  h_ws: 18 x n limbs (stride n), q_ws: 72 x n_lanes limbs (stride n_lanes, written here),
// f_ws: 108 x f_stride, flags[pair] = 1 when the public key decodes, is not the identity and is on the curve.
BN_KERNEL k_miller_hpk2(const int32_t* h_ws, const uint8_t* pks, size_t n, int32_t* q_ws, int32_t* f_ws, size_t f_stride, uint8_t* flags) {
  __shared__ int32_t lds[144 * 256];             // each lane touches only its own column: no barrier needed
  const size_t n_lanes = (n + 1) >> 1;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  const Ws hh = {lds, 256, threadIdx.x * 4u, false};
  const Ws park = ws_at(hh, 36), lpark = ws_at(hh, 90);
  const Ws qw = {q_ws, n_lanes, i * 4u, true};
  bool live[2];
  for (int k = 0; k < 2; ++k) {
    const size_t pr = 2 * (size_t)i + k;
    const bool present = pr < n;
    const size_t src = present ? pr : 0;                         // padding lane half: re-read pair 0, masked out below
    bool okp;
    G2A pk = g2_decode(pks + 128 * src, okp);
    const bool pk_ok = okp & !pk.inf & g2_on_curve(pk);
    pk.x = fp2_select(pk_ok, pk.x, fp2_const(bnc::G2_GEN_X)); pk.y = fp2_select(pk_ok, pk.y, fp2_const(bnc::G2_GEN_Y));
    fp2_store_mem(ws_at(qw, 36 * k), fp2_norm(pk.x)); fp2_store_mem(ws_at(qw, 36 * k + 18), fp2_norm(pk.y));
    const Ws hw = {const_cast<int32_t*>(h_ws), n, (uint32_t)src * 4u, true};
    // a padding half evaluates its (masked, constant 1) line at y = 1, so that it multiplies f by exactly 1
    fp_store_mem(ws_at(hh, 18 * k), fp_load_mem(hw)); fp_store_mem(ws_at(hh, 18 * k + 9), fp_select(present, fp_load_mem(ws_at(hw, 9)), fp_one()));
    if (present) flags[pr] = pk_ok ? 1 : 0;
    live[k] = present;            // an invalid key still runs on the generator (uniform arithmetic); its flag fails the whole aggregate
  }
  BN_MEM_FENCE;
  fp12_store_limbs(Ws{f_ws, f_stride, i * 4u, true}, miller_loop_2var_ws(hh, qw, park, lpark, live[0], live[1]));
}
