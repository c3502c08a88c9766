// lazy_tower.hip -- VERDICT r02 item 2(i), measured: is a LAZILY REDUCED tower (unreduced double-width Fp2 products kept as
// 64-bit column sums across the Fp6 Karatsuba recombination, ONE Montgomery reduction per output coefficient) cheaper on
// gfx950 than the shipped form (9 x 29-bit limbs, every Fp2 product reduced: tower.h fp6_mul)?
//
// Column headroom decides the radix.  With 29-bit limbs a column of 9 products is 2^61.2: at most 3 products fit a signed
// 64-bit accumulator, and the Fp6 recombination needs v0 + xi (v1 + v2 - w0) = 1 + 10 x 6 product-magnitudes in one column.
// So the lazy form needs 10 x 26-bit limbs (R = 2^260): a column of 10 products is 2^55.3, 200 product-magnitudes of room;
// the price is 100 MADs per limb product and per reduction instead of 81.
//
// What is compared (the lazy form is checked on the host against Python big integers: tests/test_lazy_tower_model.py):
//   fp6_mul                tower.h, shipped: 6 Fp2 products = 24 limb products + 12 reductions (2916 MADs) + recombination passes
//   lz::fp6_mul_lazy       10 x 26: 24 limb products (2400 MADs) into raw columns, recombination on 64-bit columns, 6 reductions (600)
// Build for the GPU (timing, registers, spills):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBN_FORCE_INLINE -DBN_LC_MAD -o lazy_tower lazy_tower.hip
// Result on MI355X: profiles/r03_lazy_tower.json (and DESIGN.md section 5).
#include <stdio.h>
#include <hip/hip_runtime.h>
#include "lazy_tower.h"

using namespace bn;
#ifndef WPS
#define WPS 1
#endif
// r <- r * b, `reps` times, one Fp6 per lane (both forms keep r and b in registers, like the inner products of the kernels)
__global__ void __launch_bounds__(256, WPS) k_fp6_shipped(const int32_t* in, int32_t* out, size_t n, int reps) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp6 r, b;
  Fp* rp[6] = {&r.c0.c0, &r.c0.c1, &r.c1.c0, &r.c1.c1, &r.c2.c0, &r.c2.c1};
  Fp* bp[6] = {&b.c0.c0, &b.c0.c1, &b.c1.c0, &b.c1.c1, &b.c2.c0, &b.c2.c1};
  for (int c = 0; c < 6; ++c) for (int k = 0; k < 9; ++k) { rp[c]->l[k] = in[(9 * c + k) * n + i] & 0x1fffffff; bp[c]->l[k] = in[(54 + 9 * c + k) * n + i] & 0x1fffffff; }
#pragma unroll 1
  for (int k = 0; k < reps; ++k) r = fp6_mul(r, b);
  for (int c = 0; c < 6; ++c) for (int k = 0; k < 9; ++k) out[(9 * c + k) * n + i] = rp[c]->l[k];
}
__global__ void __launch_bounds__(256, WPS) k_fp6_lazy(const int32_t* in, int32_t* out, size_t n, int reps) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  lz::F6 r, b;
  lz::F* rp[6] = {&r.c0.c0, &r.c0.c1, &r.c1.c0, &r.c1.c1, &r.c2.c0, &r.c2.c1};
  lz::F* bp[6] = {&b.c0.c0, &b.c0.c1, &b.c1.c0, &b.c1.c1, &b.c2.c0, &b.c2.c1};
  for (int c = 0; c < 6; ++c) for (int k = 0; k < 10; ++k) { rp[c]->l[k] = in[(10 * c + k) * n + i] & 0x3ffffff; bp[c]->l[k] = in[(60 + 10 * c + k) * n + i] & 0x3ffffff; }
#pragma unroll 1
  for (int k = 0; k < reps; ++k) r = lz::f6_norm(lz::fp6_mul_lazy(r, b));
  for (int c = 0; c < 6; ++c) for (int k = 0; k < 10; ++k) out[(10 * c + k) * n + i] = rp[c]->l[k];
}
int main() {
  size_t n = 262144;
  int32_t *in, *out;
  (void)hipMalloc(&in, n * 120 * 4); (void)hipMalloc(&out, n * 60 * 4);
  (void)hipMemset(in, 0x15, n * 120 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int reps = 60;
  for (int which = 0; which < 2; ++which) {
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
      (void)hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_fp6_shipped, dim3(n / 256), dim3(256), 0, 0, in, out, n, reps);
      else hipLaunchKernelGGL(k_fp6_lazy, dim3(n / 256), dim3(256), 0, 0, in, out, n, reps);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (it > 0 && ms < best) best = ms;
    }
    printf("{\"form\": \"%s\", \"waves_per_simd\": %d, \"lanes\": %zu, \"fp6_products\": %d, \"ms\": %.3f, \"ns_per_fp6_product_per_wave_round\": %.1f}\n",
           which == 0 ? "shipped 9x29, 12 reductions" : "lazy 10x26, 6 reductions", WPS, n, reps, best, best * 1e6 / reps / 4.0);
  }
  return 0;
}
