// wide_ops.hip -- cost of the wave-per-tuple primitives (csrc/wide.h) on ONE wave: nanoseconds per wide_mul, wide_cyc_sqr,
// line evaluation and Frobenius, measured with the 100 MHz wall clock inside the kernel (the chains of the small-call kernels
// k_miller_wide_* / k_fe_hard_wide are sequences of these).  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBN_FORCE_INLINE
// -DBN_LC_MAD -mllvm -amdgpu-use-amdgpu-trackers=1 -I bls-bn254_amd/csrc bench_micro/wide_ops.hip -o bench_micro/wide_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "wide.h"
using namespace bn;

__global__ void __launch_bounds__(128) k_wide_ops(const int32_t* table, uint64_t* out, int reps) {
  __shared__ int32_t lds[WIDE_LDS_DWORDS + 81];
  const Wide W(lds);
  const uint32_t lane = threadIdx.x;
  Ws cw = {lds, 1, (uint32_t)WIDE_LDS_DWORDS * 4u + ((uint32_t)reps & 0u), false};     // not a compile-time constant: it is passed by reference to a real function
  if (lane < 6u) {
    Fp2 v = {fp_const(bnc::R2), fp_const(bnc::R3)};
    fp2_store_mem(wide_val(W, WV_R, lane), fp2_mul(v, v));
    fp2_store_mem(wide_val(W, WV_T, lane), fp2_mul(v, fp2_mul(v, v)));
  }
  if (lane < 9u) fp_store_mem(ws_at(cw, 9u * lane), fp_mul(fp_const(bnc::R2), fp_const(bnc::R3)));
  __syncthreads();
  const Ws tb = {const_cast<int32_t*>(table), 1, 0, true};
  uint64_t t0 = wall_clock64();
  for (int i = 0; i < reps; ++i) wide_exec(W, WOP_MUL, WV_R, WV_R, WV_T);
  uint64_t t1 = wall_clock64();
  for (int i = 0; i < reps; ++i) wide_exec(W, WOP_SQR, WV_R, 1, 0);
  uint64_t t2 = wall_clock64();
  for (int i = 0; i < reps; ++i) { wide_mul_products_lines_call(W, true, WV_R, WV_R, 2u, tb, cw, 0); wide_exec(W, WOP_SUMS, WV_R, 0, 0); }
  uint64_t t3 = wall_clock64();
  for (int i = 0; i < reps; ++i) wide_exec(W, WOP_FROB1, WV_A, WV_R, 0);
  uint64_t t4 = wall_clock64();
  for (int i = 0; i < reps; ++i) wide_exec(W, WOP_CONJ, WV_A, WV_R, 0);
  uint64_t t5 = wall_clock64();
  wide_miller_prepared(W, tb, cw, true);                            // the whole two-pair loop over a (meaningless) pair table
  uint64_t t6 = wall_clock64();
  if (lane == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; out[5] = t6 - t5; }
  if (lane < 6u && blockIdx.x == 0) out[8 + lane] = (uint64_t)fp2_load_mem(wide_val(W, WV_R, lane)).c0.l[0] + (uint64_t)fp2_load_mem(wide_val(W, WV_L, lane)).c0.l[0];
}

int main() {
  int32_t* table; uint64_t* out;
  hipMalloc(&table, 162 * 4 * 90); hipMemset(table, 1, 162 * 4 * 90);   // a pair table (88 entries)
  hipMalloc(&out, 16 * 8);
  const int reps = 200;
  for (int blocks : {1, 256, 1024}) {
    hipLaunchKernelGGL(k_wide_ops, dim3(blocks), dim3(128), 0, 0, table, out, reps);
    hipDeviceSynchronize();
    uint64_t h[16]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    const char* names[5] = {"wide_mul", "wide_cyc_sqr", "wide_sqr_with_two_lines", "wide_frob1", "wide_conj"};
    printf("{\"blocks\": %d", blocks);
    for (int k = 0; k < 5; ++k) printf(", \"%s_ns\": %.0f", names[k], (double)h[k] * 10.0 / reps);
    printf(", \"wide_miller_prepared_us\": %.1f}\n", (double)h[5] * 0.01);
  }
  return 0;
}
