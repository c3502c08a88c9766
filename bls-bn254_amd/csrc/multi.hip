// multi.hip -- the N-GPU side of the C ABI (include/blsbn254.h, "multi-device" section).
//
// One blsbn254_multi = a list of devices, one blsbn254_ctx (stream + workspace) per entry, one host thread per
// entry for the duration of a call.  Verify tuples are independent (the per-term independence of
// multi_miller_loop, pairings.rs:819-824), so device g takes the contiguous range [lo_g, lo_{g+1}) and there is
// no data-path collective:
//   * host-pointer entry points: every device copies ITS slice of the validity bitmap straight into the caller's
//     buffer (slice boundaries are multiples of 8 tuples, so no byte is shared) -- a host gather of disjoint slices;
//   * device-resident entry point: every device holds its own shard in HBM, writes its bits into a zeroed
//     full-length word array and the arrays are summed with ONE ncclAllReduce(ncclSum, ncclUint32) over xGMI
//     (RCCL has no bitwise OR; the bit sets are disjoint so SUM == OR), leaving the full bitmap on every device.
//     librccl is opened with dlopen at first use: the library has no link-time dependency on it.
//   * aggregate verify: per-device Fp12 partial products (blsbn254_aggregate_partial), gathered on the host
//     (8 x 384 bytes), one blsbn254_aggregate_finish on the first device.
// Only the public C ABI of host.hip is used here.  There is no CPU fallback.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/blsbn254.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load(std::string& err) {
    if (handle) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (handle) break;
    }
    if (!handle) { err = std::string("librccl not found: ") + dlerror(); return false; }
    CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
    AllReduce = (decltype(AllReduce))dlsym(handle, "ncclAllReduce");
    GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) { err = "librccl lacks a required symbol"; return false; }
    return true;
  }
};

__global__ void k_place_bitmap(const uint8_t* local_bm, size_t lo_bytes, size_t nbytes_local, uint8_t* full, size_t nbytes_full) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nbytes_full) return;
  full[i] = (i >= lo_bytes && i < lo_bytes + nbytes_local) ? local_bm[i - lo_bytes] : (uint8_t)0;
}

}  // namespace

struct blsbn254_multi {
  std::vector<int> devices;
  std::vector<blsbn254_ctx*> ctx;
  std::vector<ncclComm_t> comms;      // created lazily by the device-resident entry point
  std::vector<void*> local_bm;        // per-device scratch bitmap of the device-resident entry point
  std::vector<size_t> local_cap;
  Rccl rccl;
  std::string last_error;
};

// contiguous shard boundaries, multiples of 8 tuples so that bitmap bytes never straddle two devices
static size_t shard_lo(size_t n, size_t g, size_t G) { return g >= G ? n : ((n * g / G) & ~(size_t)7); }

extern "C" {
// library-internal (host_verify.hip; hidden visibility): blsbn254_verify_batch_dev on the counting path, nothing left pending
__attribute__((visibility("hidden"))) int blsbn254_internal_verify_batch_dev_sync(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                                                                                  const uint8_t* d_sigs, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* d_bitmap);


int blsbn254_multi_create(const int* devices, int ndev, blsbn254_multi** out) {
  if (!out || !devices || ndev < 1 || ndev > 64) return BLSBN254_E_ARG;
  *out = nullptr;
  blsbn254_multi* m = new blsbn254_multi();
  for (int i = 0; i < ndev; ++i) {
    blsbn254_ctx* c = nullptr;
    int rc = blsbn254_ctx_create(devices[i], &c);
    if (rc) { for (blsbn254_ctx* x : m->ctx) blsbn254_ctx_destroy(x); delete m; return rc; }
    m->devices.push_back(devices[i]); m->ctx.push_back(c);
  }
  m->local_bm.assign(ndev, nullptr); m->local_cap.assign(ndev, 0);
  *out = m;
  return 0;
}
void blsbn254_multi_destroy(blsbn254_multi* m) {
  if (!m) return;
  for (size_t g = 0; g < m->ctx.size(); ++g) {
    (void)hipSetDevice(m->devices[g]);
    if (g < m->comms.size() && m->comms[g]) (void)m->rccl.CommDestroy(m->comms[g]);
    if (m->local_bm[g]) (void)hipFree(m->local_bm[g]);
    blsbn254_ctx_destroy(m->ctx[g]);
  }
  delete m;
}
int blsbn254_multi_device_count(blsbn254_multi* m) { return m ? (int)m->ctx.size() : 0; }
blsbn254_ctx* blsbn254_multi_ctx(blsbn254_multi* m, int i) { return (m && i >= 0 && (size_t)i < m->ctx.size()) ? m->ctx[i] : nullptr; }
const char* blsbn254_multi_last_error(blsbn254_multi* m) { return m ? m->last_error.c_str() : ""; }

// first non-zero return code of the per-device workers (the reference's error codes are positive, device errors negative)
static int first_rc(blsbn254_multi* m, const std::vector<int>& rcs) {
  for (size_t g = 0; g < rcs.size(); ++g)
    if (rcs[g]) { m->last_error = "device " + std::to_string(m->devices[g]) + ": " + blsbn254_last_error(m->ctx[g]); return rcs[g]; }
  return 0;
}

int blsbn254_verify_batch_multi(blsbn254_multi* m, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs,
                                size_t n, const uint8_t* dst, size_t dst_len, uint8_t* bm) {
  if (!m || !off || (n && (!pks || !sigs || !bm)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  const size_t G = m->ctx.size();
  std::vector<int> rcs(G, 0);
  std::vector<std::thread> th;
  for (size_t g = 0; g < G; ++g) {
    const size_t lo = shard_lo(n, g, G), hi = shard_lo(n, g + 1, G);
    if (hi == lo) continue;
    th.emplace_back([=, &rcs]() {
      rcs[g] = blsbn254_verify_batch(m->ctx[g], pks + 128 * lo, msgs, off + lo, sigs + 64 * lo, hi - lo, dst, dst_len, bm + lo / 8);
    });
  }
  for (std::thread& t : th) t.join();
  return first_rc(m, rcs);
}

// blsbn254_verify_batch_rlc over all devices: every device draws its own seed (seed == NULL) or uses the caller's (tests)
int blsbn254_verify_batch_rlc_multi(blsbn254_multi* m, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs,
                                    size_t n, const uint8_t* dst, size_t dst_len, const uint8_t seed[32], uint8_t* bm) {
  if (!m || !off || (n && (!pks || !sigs || !bm)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  const size_t G = m->ctx.size();
  std::vector<int> rcs(G, 0);
  std::vector<std::thread> th;
  for (size_t g = 0; g < G; ++g) {
    const size_t lo = shard_lo(n, g, G), hi = shard_lo(n, g + 1, G);
    if (hi == lo) continue;
    th.emplace_back([=, &rcs]() {
      rcs[g] = blsbn254_verify_batch_rlc(m->ctx[g], pks + 128 * lo, msgs, off + lo, sigs + 64 * lo, hi - lo, dst, dst_len, seed, bm + lo / 8);
    });
  }
  for (std::thread& t : th) t.join();
  return first_rc(m, rcs);
}

int blsbn254_aggregate_verify_multi(blsbn254_multi* m, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, size_t n,
                                    const uint8_t agg_sig[64], const uint8_t* dst, size_t dst_len, int* valid) {
  if (!m || !valid || !agg_sig || !off || (n && !pks) || (dst_len && !dst)) return BLSBN254_E_ARG;
  *valid = 0;
  if (n == 0) return 0;
  const size_t G = m->ctx.size();
  std::vector<int> rcs(G, 0), oks(G, 1);
  std::vector<uint8_t> partials(384 * G);
  std::vector<std::thread> th;
  int sig_ok = 0;
  for (size_t g = 0; g < G; ++g) {
    const size_t lo = n * g / G, hi = n * (g + 1) / G;
    th.emplace_back([=, &rcs, &oks, &partials, &sig_ok]() {
      if (g == 0)      // the first device also carries the pair (agg_sig, -G2gen)
        rcs[g] = blsbn254_aggregate_partial_with_sig(m->ctx[g], pks + 128 * lo, msgs, off + lo, hi - lo, dst, dst_len, agg_sig, partials.data(), &oks[g], &sig_ok);
      else
        rcs[g] = blsbn254_aggregate_partial(m->ctx[g], pks + 128 * lo, msgs, off + lo, hi - lo, dst, dst_len, partials.data() + 384 * g, &oks[g]);
    });
  }
  for (std::thread& t : th) t.join();
  int rc = first_rc(m, rcs);
  if (rc) return rc;
  int v = 0;
  rc = blsbn254_aggregate_finish(m->ctx[0], partials.data(), G, nullptr, &v);
  if (rc) { m->last_error = blsbn254_last_error(m->ctx[0]); return rc; }
  bool all_ok = sig_ok == 1;
  for (int o : oks) all_ok &= o == 1;
  *valid = (all_ok && v == 1) ? 1 : 0;
  return 0;
}

#define MHIP(m, x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { (m)->last_error = std::string(#x) + ": " + hipGetErrorString(e_); return BLSBN254_E_HIP; } } while (0)
#define MNCCL(m, x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { (m)->last_error = std::string(#x) + ": " + (m)->rccl.GetErrorString(r_); return BLSBN254_E_RCCL; } } while (0)

// Device-resident N-GPU verify.  Device g holds ITS shard: d_pks[g] (n_g x 128), d_msgs[g] / d_off[g] (n_g + 1 offsets
// relative to d_msgs[g]), d_sigs[g] (n_g x 64), with n_g = counts[g] a multiple of 32 for every g but the last.  On return
// (after the streams are synchronized by this call) d_full_bitmap[g] -- 4 * ceil(N / 32) bytes on device g, N = sum n_g --
// holds the validity bitmap of the WHOLE batch in global tuple order on every device.
int blsbn254_verify_batch_multi_dev(blsbn254_multi* m, const uint8_t* const* d_pks, const uint8_t* const* d_msgs, const uint64_t* const* d_off,
                                    const uint8_t* const* d_sigs, const size_t* counts, const uint8_t* dst, size_t dst_len,
                                    uint8_t* const* d_full_bitmap) {
  if (!m || !d_pks || !d_msgs || !d_off || !d_sigs || !counts || !d_full_bitmap || (dst_len && !dst)) return BLSBN254_E_ARG;
  const size_t G = m->ctx.size();
  size_t N = 0;
  for (size_t g = 0; g < G; ++g) {
    if (g + 1 < G && (counts[g] & 31)) { m->last_error = "every shard but the last must hold a multiple of 32 tuples"; return BLSBN254_E_ARG; }
    if (counts[g] && (!d_pks[g] || !d_off[g] || !d_sigs[g])) return BLSBN254_E_ARG;
    if (!d_full_bitmap[g]) return BLSBN254_E_ARG;
    N += counts[g];
  }
  if (N == 0) return 0;
  const size_t nwords = (N + 31) / 32, nbytes_full = 4 * nwords;
  if (m->comms.empty()) {
    if (!m->rccl.load(m->last_error)) return BLSBN254_E_RCCL;
    std::vector<ncclComm_t> comms(G, nullptr);
    MNCCL(m, m->rccl.CommInitAll(comms.data(), (int)G, m->devices.data()));
    m->comms = comms;
  }
  // 1. every device verifies its shard into a local bitmap (own thread: the host side of a launch sequence is serial)
  std::vector<int> rcs(G, 0);
  std::vector<std::thread> th;
  size_t lo = 0;
  std::vector<size_t> los(G);
  for (size_t g = 0; g < G; ++g) {
    los[g] = lo; lo += counts[g];
    const size_t need = (counts[g] + 7) / 8 + 8;
    MHIP(m, hipSetDevice(m->devices[g]));
    if (m->local_cap[g] < need) {
      if (m->local_bm[g]) (void)hipFree(m->local_bm[g]);
      m->local_bm[g] = nullptr; m->local_cap[g] = 0;
      MHIP(m, hipMalloc(&m->local_bm[g], need + need / 8));
      m->local_cap[g] = need + need / 8;
    }
  }
  for (size_t g = 0; g < G; ++g) {
    if (!counts[g]) continue;
    th.emplace_back([=, &rcs]() {
      // the bitmap slice is consumed on the stream right behind (k_place_bitmap, the all-reduce): the variant that leaves nothing pending
      rcs[g] = blsbn254_internal_verify_batch_dev_sync(m->ctx[g], d_pks[g], d_msgs[g], d_off[g], d_sigs[g], counts[g], dst, dst_len, (uint8_t*)m->local_bm[g]);
    });
  }
  for (std::thread& t : th) t.join();
  int rc = first_rc(m, rcs);
  if (rc) return rc;
  // 2. place the slice into the zeroed full-length array (same stream, so ordered after the verify kernels) and all-reduce
  for (size_t g = 0; g < G; ++g) {
    MHIP(m, hipSetDevice(m->devices[g]));
    hipStream_t s = (hipStream_t)blsbn254_ctx_stream(m->ctx[g]);
    hipLaunchKernelGGL(k_place_bitmap, dim3((unsigned)((nbytes_full + 255) / 256)), dim3(256), 0, s, (const uint8_t*)m->local_bm[g], los[g] / 8,
                       (counts[g] + 7) / 8, d_full_bitmap[g], nbytes_full);
    MHIP(m, hipGetLastError());
  }
  MNCCL(m, m->rccl.GroupStart());
  for (size_t g = 0; g < G; ++g) {
    ncclResult_t r = m->rccl.AllReduce(d_full_bitmap[g], d_full_bitmap[g], nwords, ncclUint32, ncclSum, m->comms[g], (hipStream_t)blsbn254_ctx_stream(m->ctx[g]));
    if (r != ncclSuccess) { (void)m->rccl.GroupEnd(); m->last_error = std::string("ncclAllReduce: ") + m->rccl.GetErrorString(r); return BLSBN254_E_RCCL; }
  }
  MNCCL(m, m->rccl.GroupEnd());
  for (size_t g = 0; g < G; ++g) {
    int s = blsbn254_ctx_synchronize(m->ctx[g]);
    if (s) { m->last_error = blsbn254_last_error(m->ctx[g]); return s; }
  }
  return 0;
}

}  // extern "C"
