// VALU integer/FP64 issue-rate microbenchmark for gfx950.
// Measures the per-SIMD issue cost of the instructions a 256-bit Montgomery
// multiplier can be built from, at 1/2/4/8 waves per SIMD, so that the
// "VALU roofline" used by bench.py is a measured number, not a datasheet guess.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 16384;   // ~1-5 ms per launch: long enough that launch ramp and DVFS settling do not dominate (r01 used 2048: 0.05-0.3 ms)
constexpr int CHAINS = 8;   // independent dependency chains per lane

enum Kind { MAD64 = 0, MULLO, MULHI, FMA64, ADDCO, MAD24, MULHI24, LSHLADD64, ADD3, MADU32, MAD64_DEP, MAD64_SGPR, ADDU32, ANDB32, LSHRREV, ASHR64, NKINDS };
static const char* kind_name[NKINDS] = {
  "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_fma_f64", "v_add_co/addc pair",
  "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_lshl_add_u64", "v_add3_u32", "v_mad_u32(lo)+", "v_mad_u64_u32 dep-chain(1)",
  "v_mad_u64_u32 own-sgpr-carry", "v_add_u32 (VOP2)", "v_and_b32 (VOP2)", "v_lshrrev_b32 (VOP2)", "v_ashrrev_i64"};
// instructions of the measured kind per chain step
static const int kind_ops[NKINDS] = {1,1,1,1,2,1,1,1,1,1,1,1,1,1,1,1};

template <int KIND>
__global__ void __launch_bounds__(256) k_peak(uint32_t* out, uint32_t seed) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a = seed * 2654435761u + tid, b = (seed ^ 0x9e3779b9u) + tid * 7u;
  if constexpr (KIND == MAD64) {
    uint64_t acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = tid + c;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        uint64_t r;
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc[c]) : "vcc");
        acc[c] = r;
      }
    }
    uint64_t s = 0; for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32);
  } else if constexpr (KIND == MAD64_SGPR) {
    // the same independent chains, but every MAD writes its carry to its OWN SGPR pair instead of VCC (what the compiler
    // emits in the library's kernels and in blsbn254_valu_probe): no write-after-write serialisation on VCC
    uint64_t acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = tid + c;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        uint64_t r, carry;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b), "v"(acc[c]));
        acc[c] = r;
      }
    }
    uint64_t s = 0; for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32);
  } else if constexpr (KIND == ASHR64) {
    uint64_t acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = ((uint64_t)a << 32) | (tid + c);
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        uint64_t r;
        asm volatile("v_ashrrev_i64 %0, 1, %1" : "=v"(r) : "v"(acc[c]));
        acc[c] = r;
      }
    }
    uint64_t s = 0; for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32);
  } else if constexpr (KIND == MAD64_DEP) {
    uint64_t acc = tid;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        uint64_t r;
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc) : "vcc");
        acc = r;
      }
    }
    out[tid] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
  } else if constexpr (KIND == FMA64) {
    double acc[CHAINS];
    double x = 1.0 + 1e-9 * tid, y = 1e-12 * (double)seed;
    for (int c = 0; c < CHAINS; ++c) acc[c] = c;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        double r;
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(acc[c]), "v"(y));
        acc[c] = r;
      }
    }
    double s = 0; for (int c = 0; c < CHAINS; ++c) s += acc[c];
    out[tid] = (uint32_t)(int64_t)s;
  } else if constexpr (KIND == LSHLADD64) {
    uint64_t acc[CHAINS];
    uint64_t inc = ((uint64_t)a << 32) | b;
    for (int c = 0; c < CHAINS; ++c) acc[c] = tid + c;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        uint64_t r;
        asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(acc[c]), "v"(inc));
        acc[c] = r;
      }
    }
    uint64_t s = 0; for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32);
  } else {
    uint32_t acc[CHAINS], hi[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { acc[c] = tid + c; hi[c] = c; }
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        uint32_t r;
        if constexpr (KIND == MULLO)   { asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(acc[c]), "v"(b)); acc[c] = r; }
        if constexpr (KIND == MULHI)   { asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"(acc[c]), "v"(b)); acc[c] = r; }
        if constexpr (KIND == MAD24)   { asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(acc[c]), "v"(b), "v"(a)); acc[c] = r; }
        if constexpr (KIND == MULHI24) { asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(acc[c]), "v"(b)); acc[c] = r; }
        if constexpr (KIND == ADDU32)  { asm volatile("v_add_u32_e32 %0, %1, %2" : "=v"(r) : "v"(b), "v"(acc[c])); acc[c] = r; }
        if constexpr (KIND == ANDB32)  { asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(r) : "v"(b), "v"(acc[c])); acc[c] = r; }
        if constexpr (KIND == LSHRREV) { asm volatile("v_lshrrev_b32_e32 %0, 1, %1" : "=v"(r) : "v"(acc[c])); acc[c] = r; }
        if constexpr (KIND == ADD3)    { asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(acc[c]), "v"(b), "v"(a)); acc[c] = r; }
        if constexpr (KIND == MADU32)  { asm volatile("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(acc[c]), "v"(b), "v"(a)); acc[c] = r; }
        if constexpr (KIND == ADDCO) {
          uint32_t r2;
          asm volatile("v_add_co_u32 %0, vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, %4, %5, vcc"
                       : "=&v"(r), "=v"(r2) : "v"(acc[c]), "v"(b), "v"(hi[c]), "v"(a) : "vcc");
          acc[c] = r; hi[c] = r2;
        }
      }
    }
    uint32_t s = 0; for (int c = 0; c < CHAINS; ++c) s ^= acc[c] ^ hi[c];
    out[tid] = s;
  }
}

typedef void (*kfn)(uint32_t*, uint32_t);

int main() {
  hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double clk_ghz = prop.clockRate / 1e6;
  printf("device %s CUs=%d clock=%.3f GHz\n", prop.name, cus, clk_ghz);
  kfn fns[NKINDS] = { k_peak<MAD64>, k_peak<MULLO>, k_peak<MULHI>, k_peak<FMA64>, k_peak<ADDCO>, k_peak<MAD24>,
                      k_peak<MULHI24>, k_peak<LSHLADD64>, k_peak<ADD3>, k_peak<MADU32>, k_peak<MAD64_DEP>, k_peak<MAD64_SGPR>,
                      k_peak<ADDU32>, k_peak<ANDB32>, k_peak<LSHRREV>, k_peak<ASHR64> };
  uint32_t* out; CHK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  printf("{\"valu_peak\": [\n");
  bool first = true;
  for (int k = 0; k < NKINDS; ++k) {
    for (int wps : {1, 2, 4, 8}) {           // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
      int blocks = cus * wps;
      hipLaunchKernelGGL(fns[k], dim3(blocks), dim3(256), 0, 0, out, 1u);
      CHK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(fns[k], dim3(blocks), dim3(256), 0, 0, out, 2u + rep);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double lane_ops = (double)blocks * 256 * ITERS * CHAINS * kind_ops[k];
      double tops = lane_ops / (best * 1e-3) / 1e12;
      // cycles per wave-instruction per SIMD at nominal clock
      double wave_instr_per_simd = (double)wps * ITERS * CHAINS * kind_ops[k];
      double cyc = best * 1e-3 * clk_ghz * 1e9 / wave_instr_per_simd;
      printf("%s {\"instr\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"T_lane_ops_per_s\": %.3f, \"cyc_per_wave_instr_per_simd\": %.2f}",
             first ? " " : ",\n ", kind_name[k], wps, best, tops, cyc);
      first = false;
    }
  }
  printf("\n]}\n");
  return 0;
}
