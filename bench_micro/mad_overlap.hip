// Do plain VALU ops overlap with v_mad_u64_u32?  Loop of 8 independent MAD chains with K extra
// independent ops (v_add_u32 / v_lshl_add_u64) per MAD.  If time is flat in K the MAD pipe is the
// bound and the fillers are free; if it grows by ~4 cycles per filler they share one issue port.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 20000;
template <int K, int KIND>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a = seed * 2654435761u + tid, b = (seed ^ 0x9e3779b9u) + tid * 7u;
  uint64_t acc[8]; uint32_t f[8]; uint64_t g[8];
  for (int c = 0; c < 8; ++c) { acc[c] = tid + c; f[c] = tid * 3 + c; g[c] = tid * 5 + c; }
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      acc[c] = (uint64_t)a * (uint32_t)(b + c) + acc[c];          // v_mad_u64_u32 (compiler-selected carry sink)
      asm("" : "+v"(acc[c]));
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (KIND == 0) { f[(c + j) & 7] += a; asm("" : "+v"(f[(c + j) & 7])); }
        else { g[(c + j) & 7] += ((uint64_t)b << 32 | a); asm("" : "+v"(g[(c + j) & 7])); }
      }
    }
  }
  uint64_t s = 0; for (int c = 0; c < 8; ++c) s ^= acc[c] ^ f[c] ^ g[c];
  out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
typedef void (*kfn)(uint32_t*, uint32_t);
int main() {
  hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  kfn fns[2][4] = {{k<0,0>, k<1,0>, k<2,0>, k<3,0>}, {k<0,1>, k<1,1>, k<2,1>, k<3,1>}};
  const char* kn[2] = {"v_add_u32", "v_lshl_add_u64"};
  uint32_t* out; CHK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  printf("{\"mad_overlap\": [\n"); bool first = true;
  for (int kind = 0; kind < 2; ++kind) for (int K = 0; K < 4; ++K) for (int wps : {1, 2, 4}) {
    int blocks = cus * wps; float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CHK(hipEventRecord(e0)); hipLaunchKernelGGL(fns[kind][K], dim3(blocks), dim3(256), 0, 0, out, 1u + rep);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    double cyc_per_mad = best * 1e-3 * 2.4e9 / ((double)wps * ITERS * 8);
    printf("%s {\"filler\": \"%s\", \"fillers_per_mad\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"cycles_per_mad_slot\": %.2f, \"T_mad_per_s\": %.2f}",
           first ? " " : ",\n ", kn[kind], K, wps, best, cyc_per_mad, (double)blocks * 256 * ITERS * 8 / (best * 1e-3) / 1e12);
    first = false;
  }
  printf("\n]}\n"); return 0;
}
