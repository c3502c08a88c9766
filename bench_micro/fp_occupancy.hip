// How much does a second (third, fourth) wave per SIMD buy for the real Fp instruction mix?
// Same fp_dot2 / fp_mul / lazy-add loop on register-resident operands, occupancy limited by a dummy
// LDS allocation (160 KB per CU: 1 block of 256 threads = 1 wave/SIMD needs > 80 KB, etc.).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../bls-bn254_amd/csrc/tower.h"
using namespace bn;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k_mix(const int32_t* in, int32_t* out, int iters, int n) {
  extern __shared__ int32_t pad[];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  Fp2 a, b;
  for (int k = 0; k < 9; ++k) { a.c0.l[k] = in[k * n + i]; a.c1.l[k] = in[(9 + k) * n + i]; b.c0.l[k] = in[(18 + k) * n + i]; b.c1.l[k] = in[(27 + k) * n + i]; }
  if (threadIdx.x == 0) pad[0] = iters;     // keep the allocation alive
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) { Fp2 c = fp2_mul(a, b); a = b; b = c; }                       // 2 x fp_dot2
    if (KIND == 1) { Fp2 c = fp2_sqr(a); a = fp2_norm(fp2_add(b, c)); b = c; }    // 2 x fp_mul + lazy ops
    if (KIND == 2) { Fp2 c = fp2_mul(a, b); Fp2 d = fp2_add_mul_xi(a, c); a = fp2_norm(fp2_sub(b, d)); b = c; }  // mul + lc + norm
  }
  for (int k = 0; k < 9; ++k) { out[k * n + i] = b.c0.l[k] + a.c0.l[k]; out[(9 + k) * n + i] = b.c1.l[k] + a.c1.l[k]; }
}
typedef void (*kfn)(const int32_t*, int32_t*, int, int);
int main() {
  hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  const int iters = 4000;
  kfn fns[3] = {k_mix<0>, k_mix<1>, k_mix<2>};
  const char* names[3] = {"fp2_mul (2 dot2)", "fp2_sqr+add+norm", "fp2_mul+lc+norm"};
  for (int k = 0; k < 3; ++k) CHK(hipFuncSetAttribute((const void*)fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int maxn = cus * 8 * 256;
  int32_t *in, *out; CHK(hipMalloc(&in, (size_t)maxn * 36 * 4)); CHK(hipMalloc(&out, (size_t)maxn * 18 * 4));
  CHK(hipMemset(in, 1, (size_t)maxn * 36 * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  printf("{\"fp_occupancy\": [\n");
  bool first = true;
  for (int k = 0; k < 3; ++k) for (int wps : {1, 2, 3, 4, 8}) {
    size_t lds = wps == 8 ? 0 : (size_t)(160 * 1024 / wps) - 1024;      // allows exactly wps blocks per CU
    int blocks = cus * wps, n = blocks * 256;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(fns[k], dim3(blocks), dim3(256), lds, 0, in, out, iters, n);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    double per_s = (double)n * iters / (best * 1e-3);
    printf("%s {\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"G_ops_per_s\": %.2f}", first ? " " : ",\n ", names[k], wps, best, per_s / 1e9);
    first = false;
  }
  printf("\n]}\n");
  return 0;
}
