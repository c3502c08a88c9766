// k_fe_wide.hip -- hard part of the final exponentiation with ONE WAVE PER TUPLE (wide.h): for launches of few tuples (the one
// final exponentiation of an aggregate verify, single pairings, small batches), where the lane-per-tuple kernels
// k_fe_expx* / k_fe_h3 are the latency of one lane's serial chain (4.5 ms whatever n <= 65536).  Same values, same bytes.
// Input: t = f^((p^6-1)(p^2+1)) as k_fe_easy leaves it (limb-major, stride).  Workgroup = 128 lanes (two waves) = tuple blockIdx.x.
//   mode 0: one[i] = (result == 1) && flags ok && subgroup ok  (k_pack_bitmap turns the bytes into the bitmap)
//   mode 1/2: Gt bytes      mode 3: *is_one (n == 1)      mode 4: gt_bytes[i] = (result == 1)
#include "wide.h"
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

__global__ void __launch_bounds__(128) k_fe_hard_wide(const int32_t* t_ws, size_t n, size_t stride, const uint8_t* flags, const uint8_t* sub_ok,
                                                     uint8_t* one, uint8_t* gt_bytes, int* is_one, int mode) {
  __shared__ int32_t lds[WIDE_LDS_DWORDS];
  const size_t i = blockIdx.x;                       // uniform over the workgroup
  if (i >= n) return;
  const Wide W(lds);
  const uint32_t lane = threadIdx.x;
  if (lane < 6u) fp2_store_mem(wide_val(W, WV_T, lane), fp2_load_limbs(ws_at_lane(Ws{const_cast<int32_t*>(t_ws), stride, (uint32_t)i * 4u, true}, 18u * lane)));
  __syncthreads();
  wide_fe_hard(W);
  if (lane == 0u) {
    const Fp12 r = {{fp2_load_mem(wide_val(W, WV_R, 0)), fp2_load_mem(wide_val(W, WV_R, 1)), fp2_load_mem(wide_val(W, WV_R, 2))},
                    {fp2_load_mem(wide_val(W, WV_R, 3)), fp2_load_mem(wide_val(W, WV_R, 4)), fp2_load_mem(wide_val(W, WV_R, 5))}};
    if (mode == 0) one[i] = (fp12_is_one(r) && flags[i] == (FLAG_SIG_OK | FLAG_PK_OK) && sub_ok[i] != 0) ? 1 : 0;
    else if (mode == 3) *is_one = fp12_is_one(r) ? 1 : 0;
    else if (mode == 4) gt_bytes[i] = fp12_is_one(r) ? 1 : 0;
    else fp12_to_be(gt_bytes + 384 * i, r);
  }
}
