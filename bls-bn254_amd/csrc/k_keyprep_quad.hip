// k_keyprep_quad.hip -- k_g2_prepare with FOUR LANES PER KEY (quad.h): for the usual case of a few thousand distinct keys the
// per-key preparation is the latency of one lane's chain (psi subgroup test || 88 line steps, 1.55 ms each), and every small
// or mid-size verify_batch waits for it.  A quad of lanes forms the independent Fp2 products of each point operation side by
// side.  Same outputs as k_g2_prepare (k_keyprep.hip): the raw line table (canonical limbs) and the validity bytes.
// Launch with 2 * ceil(4 u / 256) workgroups of 256: the first half computes lines, the second half validity.
#include "quad.h"
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_g2_prepare_quad(const uint8_t* pks, const uint32_t* keys, uint32_t u, int32_t* table, uint8_t* key_ok, const uint32_t* d_u) {
  const uint32_t nb = (4u * u + 255u) / 256u;
  const bool check_role = blockIdx.x >= nb;
  const uint32_t lane = (check_role ? blockIdx.x - nb : blockIdx.x) * blockDim.x + threadIdx.x;
  const uint32_t k = lane >> 2, role = threadIdx.x & 3u;
  if (k >= u || (d_u && k >= *d_u)) return;        // whole quads leave together
  const uint8_t* b = pks + 128 * (size_t)(keys ? keys[k] : k);
  bool okd;
  G2A q = g2_decode(b, okd);
  const bool curve_ok = okd & !q.inf & g2_on_curve(q);
  if (check_role) {
    const bool tf = quad_torsion_free(q, role);
    if (role == 0u) key_ok[k] = (curve_ok & tf) ? 1 : 0;
  } else {
    // as in k_g2_prepare: a point that fails decoding / the curve equation is replaced by the generator (its table is never used)
    q.x = fp2_select(curve_ok, q.x, fp2_const(bnc::G2_GEN_X)); q.y = fp2_select(curve_ok, q.y, fp2_const(bnc::G2_GEN_Y)); q.inf = false;
    quad_prepare_lines(q, Ws{table, 1, k * (uint32_t)(BN_NEG_G2_LINES * 54 * 4), true}, role);
  }
}
