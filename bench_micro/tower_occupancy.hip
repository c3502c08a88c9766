// Does a second wave per SIMD pay for the tower arithmetic of the final exponentiation?  62 Granger-Scott squarings (the
// squaring run of one t^x chain; 210 registers, no spills at either occupancy) and 17 register-resident Fp12 products, at one
// wave per SIMD (forced by a 96 KB LDS pad) and at two.  Build twice:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBN_FORCE_INLINE -DBN_LC_MAD -DWPS=1 -o tower_occ_w1 tower_occupancy.hip   (and -DWPS=2)
// Result on MI355X (262144 lanes, profiles/r02_tower_occupancy.json): squarings 2.77 ms -> 2.58 ms (-7 %), products 3.95 -> 4.53 ms
// (the generic product spills at 256 registers): occupancy is not the lever, the instruction stream is.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../bls-bn254_amd/csrc/pairing.h"
using namespace bn;
#ifndef WPS
#define WPS 1
#endif
__global__ void __launch_bounds__(256, WPS) k_sq(const int32_t* in, int32_t* out, size_t n, int reps) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
#if WPS == 1
  __shared__ int32_t pad[24 * 1024];          // 96 KB: one workgroup per CU = one wave per SIMD
  pad[threadIdx.x] = (int32_t)i;
  __syncthreads();
  if (pad[(threadIdx.x + 1) & 255] == -7) return;
#endif
  if (i >= n) return;
  Fp12 r = fp12_load_limbs(Ws{const_cast<int32_t*>(in), n, i * 4u, true});
  for (int k = 0; k < reps; ++k) r = fp12_cyclotomic_sqr(r);
  fp12_store_limbs(Ws{out, n, i * 4u, true}, r);
}
__global__ void __launch_bounds__(256, WPS) k_mul(const int32_t* in, int32_t* out, size_t n, int reps) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp12 r = fp12_load_limbs(Ws{const_cast<int32_t*>(in), n, i * 4u, true});
  Fp12 b = r;
  for (int k = 0; k < reps; ++k) r = fp12_mul(r, b);
  fp12_store_limbs(Ws{out, n, i * 4u, true}, r);
}
int main() {
  size_t n = 262144;
  int32_t *in, *out;
  (void)hipMalloc(&in, n * 108 * 4); (void)hipMalloc(&out, n * 108 * 4);
  (void)hipMemset(in, 1, n * 108 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    int reps = which == 0 ? 62 : 17;
    for (int it = 0; it < 3; ++it) {
      (void)hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_sq, dim3(n / 256), dim3(256), 0, 0, in, out, n, reps);
      else hipLaunchKernelGGL(k_mul, dim3(n / 256), dim3(256), 0, 0, in, out, n, reps);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (it == 2) printf("WPS=%d %s reps=%d: %.3f ms\n", WPS, which == 0 ? "cyc_sqr" : "fp12_mul", reps, ms);
    }
  }
  return 0;
}
