// host.hip -- C-ABI host side (include/blsbn254.h): contexts, workspace, staging, kernel launches.
// There is no CPU fallback in this file: every entry point launches kernels or fails.
#include <hip/hip_runtime.h>
#include <sys/random.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "sha256.h"      // host-side use: pre-hashing an oversize DST only (RFC 9380 5.3.3)
#include "lane_ops.h"    // flag constants
#include "kernels.h"
#include "../../include/blsbn254.h"

using namespace bn;

// ------------------------------------------------------------------ host side
struct DevBuf {
  void* p = nullptr; size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct ProfEntry { uint64_t launches = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; double ms = 0; };

struct blsbn254_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  DevBuf in_a, in_b, in_c, in_off, dst, h_ws, f_ws, f_ws2, flags, sub_ok, status, bitmap, out, scalars, misc;
  DevBuf fe[6];          // final-exponentiation phase buffers (x, a, b, c, b2, d), 108 x n limbs each
  DevBuf rlc_a2, rlc_a, rlc_b, rlc_elig, rlc_f2, rlc_bytes, rlc_neg, rlc_ok, rlc_idx, rlc_cpk, rlc_csig, rlc_ch, rlc_csub, rlc_cbm;   // RLC batch verification
  // RLC over repeated keys (k_rlc2.hip): weighted points, chunk descriptions, virtual tuples, fallback list
  DevBuf r2_seed, r2_a, r2_b, r2_sigok, r2_tchunk, r2_ccnt, r2_cbase, r2_ckid, r2_cstart, r2_clen, r2_csig, r2_ch, r2_cstate, r2_iota, r2_cisone,
         r2_need, r2_bcnt, r2_bbase, r2_list, r2_valid;
  DevBuf ks_cnt[2], ks_base[2], ks_kid[2], ks_start, ks_len, ks_tchunk, ks_iota, ks_out[2], ks_out2[2];   // key_sums scratch (levels of chunk sums)
  DevBuf r2_sa, r2_sb, r2_celig, r2_kelig, r2_ksig, r2_kh, r2_kstate, r2_kisone, r2_kpass, r2_cpass, r2_clist, r2_cneed, r2_cbcnt, r2_cbbase;   // chunk sums, key round of the RLC path
  bool rlc_key_round = true;         // RLC: first check every key's whole run as ONE virtual tuple (BLSBN254_RLC_KEY_ROUND=0 disables)
  unsigned rlc_key_skip = 0, rlc_key_streak = 0;   // ... backing off while batches keep failing it (skip the next 2, 4, 8, 16 chunks of work)
  size_t rlc_group = 16;             // tuples per chunk (BLSBN254_RLC_GROUP / blsbn254_set_rlc_group)
  bool rlc_group_auto = true;        // no explicit setting: 16, raised (to at most 32) when that saves a whole round of waves
  size_t lanes_per_round = 65536;    // CUs x 256: the lanes resident at one wave per SIMD (the big kernels' occupancy)
  uint64_t stat_rlc_key_rounds = 0, stat_rlc_key_rounds_passed = 0;
  uint64_t stat_rlc[4] = {0, 0, 0, 0};   // tuples on the chunked path, chunks checked, tuples sent to the exact fallback, tuples on the exact path (distinct keys)
  DevBuf status_all;     // per-element decode status of a chunked call, all chunks
  // prepared-key verify path (k_keyprep.hip, k_miller_prep.hip)
  DevBuf kd_slots, kd_rep, kd_kid, kd_keys, kd_hist, kd_cursor, kd_perm, kd_cnt, prep_table, prep_raw, prep_ok, prep_isone, prep_valid;
  hipStream_t stream2 = nullptr;     // the per-key preparation runs beside hash-to-G1
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  uint32_t kd_seed = 0;              // per-context random seed of the key hash table
  bool auto_prepare = true;          // verify_batch: de-duplicate the public keys and prepare each distinct key once (BLSBN254_AUTO_PREPARE=0 disables)
  uint64_t stat_prepared_chunks = 0, stat_exact_chunks = 0, stat_grouped_aggregates = 0, stat_pairwise_aggregates = 0;
  DevBuf q_ws;           // decoded public keys of the two-pairs-per-lane Miller kernel, 72 x lanes limbs
  DevBuf th_x, th_num, th_den, th_glv, th_part, th_part2;   // threshold combine: ids, partial products, GLV halves, window sums
  DevBuf fe_wide_one;    // validity bytes of the wave-per-tuple final exponentiation (mode 0)
  bool wide_fe = true;               // BLSBN254_WIDE_FE=0 disables the wave-per-tuple hard part
  size_t wide_fe_max = 4096;         // ... used for launches of at most this many tuples (BLSBN254_WIDE_FE_MAX)
  DevBuf fe_slots;       // the ten named powers of the t -> t^x addition chain, 10 x 108 x n limbs
  uint8_t dst_host[256];  // the (pre-hashed if oversize) DST currently resident in `dst`, and its length; -1 = none
  int dst_host_len = -1;
  size_t chunk = (size_t)1 << 22;   // tuples per launch of the chunked entry points (BLSBN254_CHUNK_LANES overrides: tests)
  bool profiling = false;
  std::map<std::string, ProfEntry> prof;
  std::string last_error;
};

#define HIPCHK(ctx, x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { (ctx)->last_error = std::string(#x) + ": " + hipGetErrorString(e_); \
    return e_ == hipErrorOutOfMemory ? BLSBN254_E_NOMEM : BLSBN254_E_HIP; } } while (0)

static inline unsigned nblocks(size_t n) { return (unsigned)((n + 255) / 256); }

struct ProfScope {
  blsbn254_ctx* c; const char* name; hipEvent_t e0 = nullptr, e1 = nullptr;
  ProfScope(blsbn254_ctx* c_, const char* n) : c(c_), name(n) {
    if (c->profiling) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, c->stream); }
  }
  ~ProfScope() {
    if (c->profiling) { (void)hipEventRecord(e1, c->stream); ProfEntry& p = c->prof[name]; ++p.launches; p.pending.emplace_back(e0, e1); }
  }
};
#define LAUNCH(ctx, name, kernel, n, ...) do { ProfScope ps_(ctx, name); \
    hipLaunchKernelGGL(kernel, dim3(nblocks(n)), dim3(256), 0, (ctx)->stream, __VA_ARGS__); } while (0); HIPCHK(ctx, hipGetLastError())

// one workgroup of 64 lanes (one wave) per element: the wave-per-tuple kernels (wide.h)
#define LAUNCH_WIDE(ctx, name, kernel, n, ...) do { ProfScope ps_(ctx, name); \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(n)), dim3(64), 0, (ctx)->stream, __VA_ARGS__); } while (0); HIPCHK(ctx, hipGetLastError())

// the same on the context's second stream (events recorded there)
struct ProfScope2 {
  blsbn254_ctx* c; const char* name; hipEvent_t e0 = nullptr, e1 = nullptr;
  ProfScope2(blsbn254_ctx* c_, const char* n) : c(c_), name(n) {
    if (c->profiling) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, c->stream2); }
  }
  ~ProfScope2() {
    if (c->profiling) { (void)hipEventRecord(e1, c->stream2); ProfEntry& p = c->prof[name]; ++p.launches; p.pending.emplace_back(e0, e1); }
  }
};
#define LAUNCH2(ctx, name, kernel, n, ...) do { ProfScope2 ps_(ctx, name); \
    hipLaunchKernelGGL(kernel, dim3(nblocks(n)), dim3(256), 0, (ctx)->stream2, __VA_ARGS__); } while (0); HIPCHK(ctx, hipGetLastError())

extern "C" {

// -G2gen = (x, p - y) of the generator fp2.rs:305-333, as bytes
static const uint8_t NEG_G2_BYTES[128] = {
  0x19,0x8e,0x93,0x93,0x92,0x0d,0x48,0x3a,0x72,0x60,0xbf,0xb7,0x31,0xfb,0x5d,0x25,0xf1,0xaa,0x49,0x33,0x35,0xa9,0xe7,0x12,0x97,0xe4,0x85,0xb7,0xae,0xf3,0x12,0xc2,
  0x18,0x00,0xde,0xef,0x12,0x1f,0x1e,0x76,0x42,0x6a,0x00,0x66,0x5e,0x5c,0x44,0x79,0x67,0x43,0x22,0xd4,0xf7,0x5e,0xda,0xdd,0x46,0xde,0xbd,0x5c,0xd9,0x92,0xf6,0xed,
  0x27,0x5d,0xc4,0xa2,0x88,0xd1,0xaf,0xb3,0xcb,0xb1,0xac,0x09,0x18,0x75,0x24,0xc7,0xdb,0x36,0x39,0x5d,0xf7,0xbe,0x3b,0x99,0xe6,0x73,0xb1,0x3a,0x07,0x5a,0x65,0xec,
  0x1d,0x9b,0xef,0xcd,0x05,0xa5,0x32,0x3e,0x6d,0xa4,0xd4,0x35,0xf3,0xb6,0x17,0xcd,0xb3,0xaf,0x83,0x28,0x5c,0x2d,0xf7,0x11,0xef,0x39,0xc0,0x15,0x71,0x82,0x7f,0x9d};


const char* blsbn254_strerror(int code) {
  switch (code) {
    case 0: return "ok";
    case 1: return "invalid scalar bytes";
    case 2: return "invalid G1 bytes";
    case 3: return "invalid G2 bytes";
    case 4: return "invalid Gt bytes";
    case BLSBN254_E_ARG: return "invalid argument";
    case BLSBN254_E_HIP: return "HIP runtime error";
    case BLSBN254_E_NOMEM: return "out of device memory";
    case BLSBN254_E_NO_DEVICE: return "no gfx950 device available (there is no CPU fallback)";
    case BLSBN254_E_RCCL: return "RCCL error";
    default: return "unknown error";
  }
}
const char* blsbn254_last_error(blsbn254_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int blsbn254_ctx_create(int device, blsbn254_ctx** out) {
  if (!out) return BLSBN254_E_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return BLSBN254_E_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return BLSBN254_E_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BLSBN254_E_NO_DEVICE;
  blsbn254_ctx* c = new blsbn254_ctx();
  c->device = device;
  if (const char* e = std::getenv("BLSBN254_CHUNK_LANES")) {
    size_t v = (size_t)std::strtoull(e, nullptr, 10) & ~(size_t)7;        // multiples of 8: bitmap bytes must not straddle chunks
    if (v >= 8 && v <= ((size_t)1 << 23)) c->chunk = v;
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return BLSBN254_E_HIP; }
  if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return BLSBN254_E_HIP; }
  if (getrandom(&c->kd_seed, sizeof c->kd_seed, 0) != (ssize_t)sizeof c->kd_seed) c->kd_seed = 0x5bd1e995u;
  if (const char* e = std::getenv("BLSBN254_AUTO_PREPARE")) c->auto_prepare = std::atoi(e) != 0;
  if (const char* e = std::getenv("BLSBN254_RLC_GROUP")) { long v = std::atol(e); if (v >= 2 && v <= 4096) { c->rlc_group = (size_t)v; c->rlc_group_auto = false; } }
  c->lanes_per_round = (size_t)prop.multiProcessorCount * 256;
  if (const char* e = std::getenv("BLSBN254_RLC_KEY_ROUND")) c->rlc_key_round = std::atoi(e) != 0;
  if (const char* e = std::getenv("BLSBN254_WIDE_FE")) c->wide_fe = std::atoi(e) != 0;
  if (const char* e = std::getenv("BLSBN254_WIDE_FE_MAX")) { long v = std::atol(e); if (v >= 0 && v <= (1 << 20)) c->wide_fe_max = (size_t)v; }
  *out = c;
  return 0;
}
void blsbn254_ctx_destroy(blsbn254_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& kv : c->prof) for (auto& pr : kv.second.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  DevBuf* bufs[] = {&c->in_a, &c->in_b, &c->in_c, &c->in_off, &c->dst, &c->h_ws, &c->f_ws, &c->f_ws2, &c->flags, &c->sub_ok, &c->status, &c->status_all, &c->bitmap, &c->out, &c->scalars, &c->misc};
  for (DevBuf* b : bufs) b->release();
  for (DevBuf& b : c->fe) b.release();
  c->fe_slots.release(); c->fe_wide_one.release();
  { DevBuf* tb[] = {&c->th_x, &c->th_num, &c->th_den, &c->th_glv, &c->th_part, &c->th_part2, &c->q_ws}; for (DevBuf* b : tb) b->release(); }
  { DevBuf* rb[] = {&c->rlc_a2, &c->rlc_a, &c->rlc_b, &c->rlc_elig, &c->rlc_f2, &c->rlc_bytes, &c->rlc_neg, &c->rlc_ok, &c->rlc_idx, &c->rlc_cpk, &c->rlc_csig, &c->rlc_ch, &c->rlc_csub, &c->rlc_cbm};
    for (DevBuf* b : rb) b->release(); }
  { DevBuf* kb[] = {&c->kd_slots, &c->kd_rep, &c->kd_kid, &c->kd_keys, &c->kd_hist, &c->kd_cursor, &c->kd_perm, &c->kd_cnt, &c->prep_table, &c->prep_raw, &c->prep_ok,
                    &c->prep_isone, &c->prep_valid};
    for (DevBuf* b : kb) b->release(); }
  { DevBuf* rb[] = {&c->r2_seed, &c->r2_a, &c->r2_b, &c->r2_sigok, &c->r2_tchunk, &c->r2_ccnt, &c->r2_cbase, &c->r2_ckid, &c->r2_cstart, &c->r2_clen, &c->r2_csig,
                    &c->r2_ch, &c->r2_cstate, &c->r2_iota, &c->r2_cisone, &c->r2_need, &c->r2_bcnt, &c->r2_bbase, &c->r2_list, &c->r2_valid,
                    &c->ks_cnt[0], &c->ks_cnt[1], &c->ks_base[0], &c->ks_base[1], &c->ks_kid[0], &c->ks_kid[1], &c->ks_start, &c->ks_len, &c->ks_tchunk, &c->ks_iota,
                    &c->ks_out[0], &c->ks_out[1], &c->ks_out2[0], &c->ks_out2[1], &c->r2_sa, &c->r2_sb, &c->r2_celig, &c->r2_kelig, &c->r2_ksig, &c->r2_kh, &c->r2_kstate, &c->r2_kisone,
                    &c->r2_kpass, &c->r2_cpass, &c->r2_clist, &c->r2_cneed, &c->r2_cbcnt, &c->r2_cbbase};
    for (DevBuf* b : rb) b->release(); }
  (void)hipStreamSynchronize(c->stream2);
  (void)hipEventDestroy(c->ev_fork); (void)hipEventDestroy(c->ev_join);
  (void)hipStreamDestroy(c->stream2);
  (void)hipStreamDestroy(c->stream);
  delete c;
}
int blsbn254_ctx_synchronize(blsbn254_ctx* c) { if (!c) return BLSBN254_E_ARG; HIPCHK(c, hipStreamSynchronize(c->stream)); return 0; }
void* blsbn254_ctx_stream(blsbn254_ctx* c) { return c ? (void*)c->stream : nullptr; }

int blsbn254_profile_enable(blsbn254_ctx* c, int on) { if (!c) return BLSBN254_E_ARG; c->profiling = on != 0; return 0; }
int blsbn254_profile_reset(blsbn254_ctx* c) {
  if (!c) return BLSBN254_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto& kv : c->prof) for (auto& pr : kv.second.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  c->prof.clear();
  return 0;
}
int blsbn254_profile_read(blsbn254_ctx* c, char* names, uint64_t* launches, double* total_ms, int max_entries) {
  if (!c) return BLSBN254_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int k = 0;
  for (auto& kv : c->prof) {
    ProfEntry& p = kv.second;
    for (auto& pr : p.pending) { float ms = 0; (void)hipEventElapsedTime(&ms, pr.first, pr.second); p.ms += ms; (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    p.pending.clear();
    if (k < max_entries) { std::snprintf(names + 32 * k, 32, "%s", kv.first.c_str()); launches[k] = p.launches; total_ms[k] = p.ms; ++k; }
  }
  return k;
}

// VALU roofline probe (k_valu_peak, k_misc.hip), 4 waves per SIMD on every CU:
//   out[0] v_mad_u64_u32 lane-MADs per second          out[1] plain VOP2 (v_add_u32 / v_xor_b32) lane-ops per second
//   out[2] shader clock held under the MAD probe (Hz)  out[3] shader clock held under the VOP2 probe (Hz)
//   out[4] compute units                               out[5] 4-cycle issue ceiling = CUs x 4 SIMDs x 16 lane-ops/cycle x out[2]
//   (what ONE wave per SIMD can issue: a wave64 VALU instruction every 4 cycles; with several waves per SIMD plain VOP2
//   work issues faster than that -- out[1] -- while v_mad_u64_u32 does not -- out[0])
// The clock is delta(s_memtime) / delta(s_memrealtime) x 100 MHz, median over all waves (MI355X_MICROARCH.md, DVFS item 6).
struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};
int blsbn254_valu_probe(blsbn254_ctx* c, double out[6]) {
  if (!c || !out) return BLSBN254_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  hipDeviceProp_t prop;
  HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
  const int blocks = prop.multiProcessorCount * 4, iters = 1 << 16;
  const size_t nwaves = (size_t)blocks * 4;
  HIPCHK(c, c->misc.reserve((size_t)blocks * 256 * 4 + 64 + nwaves * 16));
  uint64_t* d_stamps = (uint64_t*)((char*)c->misc.p + (((size_t)blocks * 256 * 4 + 63) & ~(size_t)63));
  EventPair ev;                                   // destroyed on every return path
  HIPCHK(c, hipEventCreate(&ev.e0)); HIPCHK(c, hipEventCreate(&ev.e1));
  std::vector<uint64_t> st(nwaves * 2);
  std::vector<double> clk(nwaves);
  for (int kind = 0; kind < 2; ++kind) {
    double best = 0, best_clk = 0;
    for (int rep = 0; rep < 4; ++rep) {
      HIPCHK(c, hipEventRecord(ev.e0, c->stream));
      hipLaunchKernelGGL(k_valu_peak, dim3(blocks), dim3(256), 0, c->stream, (uint32_t*)c->misc.p, 1u + rep, iters, kind, d_stamps);
      HIPCHK(c, hipEventRecord(ev.e1, c->stream));
      HIPCHK(c, hipEventSynchronize(ev.e1));
      float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
      double rate = (double)blocks * 256 * iters * 8 / (ms * 1e-3);
      if (rep > 0 && rate > best) {                    // rep 0 warms the clocks
        best = rate;
        HIPCHK(c, hipMemcpy(st.data(), d_stamps, nwaves * 16, hipMemcpyDeviceToHost));
        for (size_t w = 0; w < nwaves; ++w) clk[w] = st[2 * w + 1] ? (double)st[2 * w] / (double)st[2 * w + 1] * 100e6 : 0;
        std::nth_element(clk.begin(), clk.begin() + nwaves / 2, clk.end());
        best_clk = clk[nwaves / 2];
      }
    }
    out[kind] = best; out[2 + kind] = best_clk;
  }
  out[4] = prop.multiProcessorCount;
  out[5] = (double)prop.multiProcessorCount * 4.0 * 16.0 * out[2];   // one wave64 instruction per 4 cycles per SIMD = 16 lane-ops per cycle per SIMD
  return 0;
}
// Measured v_mad_u64_u32 issue rate of the whole chip (lane-MADs per second): out[0] of the probe above.
int blsbn254_valu_peak(blsbn254_ctx* c, double* mads_per_s) {
  if (!c || !mads_per_s) return BLSBN254_E_ARG;
  double o[6];
  int rc = blsbn254_valu_probe(c, o);
  if (rc) return rc;
  *mads_per_s = o[0];
  return 0;
}

// DST handling: RFC 9380 5.3.3 (oversize DSTs are pre-hashed); staged into device memory once per call
static int stage_dst(blsbn254_ctx* c, const uint8_t* dst, size_t dst_len, uint32_t* out_len) {
  uint8_t tmp[256];
  if (dst_len > 255) {
    Sha256 s; sha256_init(s);
    sha256_update(s, (const uint8_t*)"H2C-OVERSIZE-DST-", 17); sha256_update(s, dst, dst_len); sha256_final(s, tmp);
    dst_len = 32;
  } else if (dst_len) std::memcpy(tmp, dst, dst_len);
  *out_len = (uint32_t)dst_len;
  // the same tag as the previous call (every step of a steady-state caller): already resident, nothing to copy or wait for
  if (c->dst_host_len == (int)dst_len && (dst_len == 0 || std::memcmp(c->dst_host, tmp, dst_len) == 0)) return 0;
  HIPCHK(c, c->dst.reserve(256));
  c->dst_host_len = -1;
  if (dst_len) {
    HIPCHK(c, hipStreamSynchronize(c->stream));           // kernels of an earlier call may still be reading the old tag
    std::memcpy(c->dst_host, tmp, dst_len);                // ctx-owned source: outlives the asynchronous copy
    HIPCHK(c, hipMemcpyAsync(c->dst.p, c->dst_host, dst_len, hipMemcpyHostToDevice, c->stream));
  }
  c->dst_host_len = (int)dst_len;
  return 0;
}
static int check_offsets(const uint64_t* off, size_t n) {
  for (size_t i = 0; i < n; ++i) if (off[i + 1] < off[i]) return BLSBN254_E_ARG;
  return 0;
}
// first index whose status differs from the wanted value, or -1
static int first_bad(blsbn254_ctx* c, const uint8_t* d_status, size_t n, uint8_t mask, uint8_t val, int* out) {
  HIPCHK(c, c->misc.reserve(64));
  int init = 0x7fffffff;
  HIPCHK(c, hipMemcpyAsync(c->misc.p, &init, 4, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "status_reduce", k_status_reduce, n, d_status, n, mask, val, (int*)c->misc.p);
  HIPCHK(c, hipMemcpyAsync(out, c->misc.p, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (*out == 0x7fffffff) *out = -1;
  return 0;
}
// status byte of the tuple at index idx (host read)
static int read_status(blsbn254_ctx* c, const uint8_t* d_status, int idx, uint8_t* st) {
  HIPCHK(c, hipMemcpyAsync(st, d_status + idx, 1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// The kernels address limb-major workspaces through a buffer descriptor with a 32-bit scalar byte offset
// (limb index x stride x 4, tower.h `Ws`): a launch may span at most MAX_LANES tuples (107 x 8 Mi x 4 B < 4 GiB).
// Independent-element entry points are processed in chunks of ctx->chunk (4 Mi); the product-type ones reject more.
static const size_t MAX_LANES = (size_t)1 << 23;
#define CHECK_LANES(c, n) do { if ((n) > MAX_LANES) { (c)->last_error = "more than 2^23 elements in one product-type call"; return BLSBN254_E_ARG; } } while (0)

// Final exponentiation of the n Fp12 values at f (limb-major, `stride`), in place for the easy part.
//   mode 0: verify bitmap (flags / sub_ok / d_bitmap)   mode 1: Gt bytes   mode 3: single is_one flag (n == 1)
static int run_final_exp(blsbn254_ctx* c, int32_t* f, size_t n, size_t stride, int mode, const uint8_t* flags, const uint8_t* sub_ok,
                         uint8_t* d_bitmap, uint8_t* d_gt, int* d_is_one) {
  CHECK_LANES(c, stride);
  for (DevBuf& b : c->fe) HIPCHK(c, b.reserve(stride * 108 * 4));
  HIPCHK(c, c->fe_slots.reserve(stride * 108 * 4 * 10));
  int32_t* S = (int32_t*)c->fe_slots.p;
  int32_t *X = (int32_t*)c->fe[0].p, *A = (int32_t*)c->fe[1].p, *B = (int32_t*)c->fe[2].p, *C = (int32_t*)c->fe[3].p,
          *B2 = (int32_t*)c->fe[4].p, *D = (int32_t*)c->fe[5].p;
  LAUNCH(c, "fe_easy", k_fe_easy, n, (const int32_t*)f, f, n, stride);                       // t (in place)
  // Few tuples: the lane-per-tuple kernels below would be the latency of one lane's chain (4.5 ms for any n <= 65536); the hard
  // part runs with one WAVE per tuple instead (k_fe_wide.hip): ~0.6 ms per round of CUs x 16 tuples.  Same values.
  if (c->wide_fe && n <= c->wide_fe_max) {
    uint8_t* one = nullptr;
    if (mode == 0) { HIPCHK(c, c->fe_wide_one.reserve(n)); one = (uint8_t*)c->fe_wide_one.p; }
    LAUNCH_WIDE(c, "fe_hard_wide", k_fe_hard_wide, n, (const int32_t*)f, n, stride, flags, sub_ok, one, d_gt, d_is_one, mode);
    if (mode == 0) { LAUNCH(c, "pack_bitmap", k_pack_bitmap, n, (const uint8_t*)one, n, d_bitmap); }
    return 0;
  }
  // t^x three times; the glue steps fe_h1 / fe_h2 are computed by the first two launches themselves (k_fe_expx_tail.hip)
  LAUNCH(c, "fe_expx_h1", k_fe_expx_h1, n, (const int32_t*)f, S, n, stride, A, B);
  LAUNCH(c, "fe_expx_h2", k_fe_expx_h2, n, (const int32_t*)B, S, n, stride, B, C, B2, D);
  LAUNCH(c, "fe_expx", k_fe_expx, n, (const int32_t*)D, X, S, n, stride);
  LAUNCH(c, "fe_h3", k_fe_h3, n, (const int32_t*)f, (const int32_t*)A, (const int32_t*)C, (const int32_t*)B2, (const int32_t*)X, S, n, stride,
         flags, sub_ok, d_bitmap, d_gt, d_is_one, mode);
  return 0;
}

// ---------------- pairing / Miller loop / final exponentiation
static int miller_to_ws(blsbn254_ctx* c, const uint8_t* d_g1, const uint8_t* d_g2, size_t n) {
  CHECK_LANES(c, n);
  HIPCHK(c, c->f_ws.reserve(n * 108 * 4));
  HIPCHK(c, c->status.reserve(n));
  if (c->wide_fe && n <= c->wide_fe_max / 2) {        // few pairs: one wave per pair (lane 0 runs the G2 point arithmetic); the serial part makes the chain ~2 x a prepared one
    LAUNCH_WIDE(c, "miller_wide_1", k_miller_wide_1, n, d_g1, d_g2, n, (int32_t*)c->f_ws.p, n, (uint8_t*)c->status.p);
  } else {
    LAUNCH(c, "miller_1", k_miller_1, n, d_g1, d_g2, n, (int32_t*)c->f_ws.p, n, (uint8_t*)c->status.p);
  }
  return 0;
}
static int decode_status_rc(blsbn254_ctx* c, const uint8_t* d_status, size_t n) {
  int bad;
  int rc = first_bad(c, d_status, n, 3, 3, &bad);
  if (rc) return rc;
  if (bad < 0) return 0;
  uint8_t st; rc = read_status(c, d_status, bad, &st);
  if (rc) return rc;
  return (st & 1) ? BLSBN254_ERR_G2 : BLSBN254_ERR_G1;
}
int blsbn254_pairing_batch_dev(blsbn254_ctx* c, const uint8_t* d_g1, const uint8_t* d_g2, size_t n, uint8_t* d_gt, uint8_t* d_status) {
  if (!c || (n && (!d_g1 || !d_g2 || !d_gt))) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->status_all.reserve(n));
  for (size_t lo = 0; lo < n; lo += c->chunk) {
    size_t m = n - lo < c->chunk ? n - lo : c->chunk;
    int rc = miller_to_ws(c, d_g1 + 64 * lo, d_g2 + 128 * lo, m);
    if (rc) return rc;
    rc = run_final_exp(c, (int32_t*)c->f_ws.p, m, m, 1, nullptr, nullptr, nullptr, d_gt + 384 * lo, nullptr);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync((uint8_t*)c->status_all.p + lo, c->status.p, m, hipMemcpyDeviceToDevice, c->stream));
  }
  if (d_status) HIPCHK(c, hipMemcpyAsync(d_status, c->status_all.p, n, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}
int blsbn254_pairing_batch(blsbn254_ctx* c, const uint8_t* g1, const uint8_t* g2, size_t n, uint8_t* gt) {
  if (!c || (n && (!g1 || !g2 || !gt))) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->in_a.reserve(64 * n)); HIPCHK(c, c->in_b.reserve(128 * n)); HIPCHK(c, c->out.reserve(384 * n));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, g1, 64 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, g2, 128 * n, hipMemcpyHostToDevice, c->stream));
  int rc = blsbn254_pairing_batch_dev(c, (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_b.p, n, (uint8_t*)c->out.p, nullptr);
  if (rc) return rc;
  rc = decode_status_rc(c, (const uint8_t*)c->status_all.p, n);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(gt, c->out.p, 384 * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_miller_loop_batch(blsbn254_ctx* c, const uint8_t* g1, const uint8_t* g2, size_t n, uint8_t* ml_out) {
  if (!c || (n && (!g1 || !g2 || !ml_out))) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->in_a.reserve(64 * n)); HIPCHK(c, c->in_b.reserve(128 * n)); HIPCHK(c, c->out.reserve(384 * n));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, g1, 64 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, g2, 128 * n, hipMemcpyHostToDevice, c->stream));
  for (size_t lo = 0; lo < n; lo += c->chunk) {
    size_t m = n - lo < c->chunk ? n - lo : c->chunk;
    int rc = miller_to_ws(c, (const uint8_t*)c->in_a.p + 64 * lo, (const uint8_t*)c->in_b.p + 128 * lo, m);
    if (rc) return rc;
    rc = decode_status_rc(c, (const uint8_t*)c->status.p, m);
    if (rc) return rc;
    LAUNCH(c, "fp12_to_bytes", k_fp12_to_bytes, m, (const int32_t*)c->f_ws.p, m, m, (uint8_t*)c->out.p + 384 * lo);
  }
  HIPCHK(c, hipMemcpyAsync(ml_out, c->out.p, 384 * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
// product of the n Fp12 values in f_ws (stride n) -> left in *result (pointer into f_ws or f_ws2), stride *rs
static int product_tree(blsbn254_ctx* c, size_t n, const int32_t** result, size_t* rs) {
  HIPCHK(c, c->f_ws2.reserve(((n + 1) / 2) * 108 * 4));
  int32_t* a = (int32_t*)c->f_ws.p; int32_t* b = (int32_t*)c->f_ws2.p;
  size_t sa = n, m = n;
  while (m > 1) {
    size_t mo = (m + 1) / 2;
    LAUNCH(c, "fp12_mul_pairs", k_fp12_mul_pairs, mo, (const int32_t*)a, m, sa, b, mo);
    std::swap(a, b); sa = mo; m = mo;
  }
  *result = a; *rs = sa;
  return 0;
}
int blsbn254_multi_miller_loop(blsbn254_ctx* c, const uint8_t* g1, const uint8_t* g2, size_t n, uint8_t ml_out[384]) {
  if (!c || !ml_out || (n && (!g1 || !g2))) return BLSBN254_E_ARG;
  if (n == 0) { std::memset(ml_out, 0, 384); ml_out[31] = 1; return 0; }       // empty product = Fp12::ONE
  CHECK_LANES(c, n);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->in_a.reserve(64 * n)); HIPCHK(c, c->in_b.reserve(128 * n)); HIPCHK(c, c->out.reserve(384));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, g1, 64 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, g2, 128 * n, hipMemcpyHostToDevice, c->stream));
  int rc = miller_to_ws(c, (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_b.p, n);
  if (rc) return rc;
  rc = decode_status_rc(c, (const uint8_t*)c->status.p, n);
  if (rc) return rc;
  const int32_t* res; size_t rs;
  rc = product_tree(c, n, &res, &rs);
  if (rc) return rc;
  LAUNCH(c, "fp12_to_bytes", k_fp12_to_bytes, 1, res, (size_t)1, rs, (uint8_t*)c->out.p);
  HIPCHK(c, hipMemcpyAsync(ml_out, c->out.p, 384, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_final_exponentiation(blsbn254_ctx* c, const uint8_t* ml, size_t n, uint8_t* gt) {
  if (!c || (n && (!ml || !gt))) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->in_a.reserve(384 * n)); HIPCHK(c, c->out.reserve(384 * n));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, ml, 384 * n, hipMemcpyHostToDevice, c->stream));
  for (size_t lo = 0; lo < n; lo += c->chunk) {
    size_t m = n - lo < c->chunk ? n - lo : c->chunk;
    HIPCHK(c, c->f_ws.reserve(m * 108 * 4)); HIPCHK(c, c->status.reserve(m));
    LAUNCH(c, "fp12_from_bytes", k_fp12_from_bytes, m, (const uint8_t*)c->in_a.p + 384 * lo, m, (int32_t*)c->f_ws.p, m, (uint8_t*)c->status.p);
    int bad; int rc = first_bad(c, (const uint8_t*)c->status.p, m, 1, 1, &bad);
    if (rc) return rc;
    if (bad >= 0) return BLSBN254_ERR_GT;
    rc = run_final_exp(c, (int32_t*)c->f_ws.p, m, m, 1, nullptr, nullptr, nullptr, (uint8_t*)c->out.p + 384 * lo, nullptr);
    if (rc) return rc;
  }
  HIPCHK(c, hipMemcpyAsync(gt, c->out.p, 384 * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ---------------- hash to curve
static int stage_msgs(blsbn254_ctx* c, const uint8_t* msgs, const uint64_t* off, size_t n) {
  if (check_offsets(off, n)) return BLSBN254_E_ARG;
  size_t total = (size_t)(off[n] - off[0]);
  HIPCHK(c, c->in_c.reserve(total + 1)); HIPCHK(c, c->in_off.reserve(8 * (n + 1)));
  if (total) HIPCHK(c, hipMemcpyAsync(c->in_c.p, msgs + off[0], total, hipMemcpyHostToDevice, c->stream));
  if (off[0] == 0) HIPCHK(c, hipMemcpyAsync(c->in_off.p, off, 8 * (n + 1), hipMemcpyHostToDevice, c->stream));
  else {
    std::vector<uint64_t> rel(n + 1);
    for (size_t i = 0; i <= n; ++i) rel[i] = off[i] - off[0];
    HIPCHK(c, hipMemcpyAsync(c->in_off.p, rel.data(), 8 * (n + 1), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return 0;
}
static int h2c_common(blsbn254_ctx* c, const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* out, int g2, int ro) {
  if (!c || (n && (!msgs && off && off[n] != off[0])) || !off || (n && !out) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t dl; int rc = stage_dst(c, dst, dst_len, &dl);
  if (rc) return rc;
  rc = stage_msgs(c, msgs, off, n);
  if (rc) return rc;
  size_t sz = g2 ? 128 : 64;
  HIPCHK(c, c->out.reserve(sz * n));
  if (g2) { LAUNCH(c, "hash_to_g2", k_hash_to_g2, n, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, n, (const uint8_t*)c->dst.p, dl, (uint8_t*)c->out.p, ro); }
  else { LAUNCH(c, "hash_to_g1", k_hash_to_g1, n, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, n, (const uint8_t*)c->dst.p, dl, (int32_t*)nullptr, n, (uint8_t*)c->out.p, ro ? 1 : 2); }
  HIPCHK(c, hipMemcpyAsync(out, c->out.p, sz * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_hash_to_g1_batch(blsbn254_ctx* c, const uint8_t* m, const uint64_t* o, size_t n, const uint8_t* d, size_t dl, uint8_t* out) { return h2c_common(c, m, o, n, d, dl, out, 0, 1); }
int blsbn254_encode_to_g1_batch(blsbn254_ctx* c, const uint8_t* m, const uint64_t* o, size_t n, const uint8_t* d, size_t dl, uint8_t* out) { return h2c_common(c, m, o, n, d, dl, out, 0, 0); }
int blsbn254_hash_to_g2_batch(blsbn254_ctx* c, const uint8_t* m, const uint64_t* o, size_t n, const uint8_t* d, size_t dl, uint8_t* out) { return h2c_common(c, m, o, n, d, dl, out, 1, 1); }
int blsbn254_encode_to_g2_batch(blsbn254_ctx* c, const uint8_t* m, const uint64_t* o, size_t n, const uint8_t* d, size_t dl, uint8_t* out) { return h2c_common(c, m, o, n, d, dl, out, 1, 0); }

// ---------------- point checks
static int check_common(blsbn254_ctx* c, const uint8_t* pts, size_t n, uint8_t* bm, int g2) {
  if (!c || (n && (!pts || !bm))) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  size_t sz = g2 ? 128 : 64, nb = (n + 7) / 8;
  HIPCHK(c, c->in_a.reserve(sz * n)); HIPCHK(c, c->bitmap.reserve(nb + 8));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, pts, sz * n, hipMemcpyHostToDevice, c->stream));
  if (g2) { LAUNCH(c, "g2_check", k_g2_check, n, (const uint8_t*)c->in_a.p, n, (uint8_t*)nullptr, (uint8_t*)c->bitmap.p); }
  else { LAUNCH(c, "g1_check", k_g1_check, n, (const uint8_t*)c->in_a.p, n, (uint8_t*)c->bitmap.p); }
  HIPCHK(c, hipMemcpyAsync(bm, c->bitmap.p, nb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_g1_check_batch(blsbn254_ctx* c, const uint8_t* g1, size_t n, uint8_t* bm) { return check_common(c, g1, n, bm, 0); }
int blsbn254_g2_check_batch(blsbn254_ctx* c, const uint8_t* g2, size_t n, uint8_t* bm) { return check_common(c, g2, n, bm, 1); }

// ---------------- verify
// Workspace is ~7.4 KB per tuple (H, f, six final-exponentiation phase buffers, ten chain slots); batches
// larger than ctx->chunk (4 Mi) tuples are processed chunk by chunk so that any n fits the 288 GB of HBM.
// ---- prepared-key path.  Limits: key ids and table offsets are 32-bit (88 x 54 x 4 B per key): at most PREP_MAX_KEYS keys.
static const size_t PREP_MAX_KEYS = (size_t)1 << 16;
static const size_t PREP_RAW_LIMBS = (size_t)BN_NEG_G2_LINES * 54;       // a key's 88 line triples
static const size_t PREP_KEY_LIMBS = (size_t)BN_NEG_G2_LINES * 162;      // a key's 88 expanded line pairs (key line x -G2gen line)

// G2Prepared::from for u keys on the second stream (after ev_fork), ev_join recorded behind it.
// keys == nullptr: key k = pks[128 k]; else key k = the public key of tuple keys[k].
static int prepare_keys_async(blsbn254_ctx* c, const uint8_t* d_pks, const uint32_t* d_keys, size_t u, int32_t* table, uint8_t* key_ok) {
  HIPCHK(c, c->prep_raw.reserve(u * PREP_RAW_LIMBS * 4));
  HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
  LAUNCH2(c, "g2_prepare", k_g2_prepare, 2 * 256 * (size_t)nblocks(u), d_pks, d_keys, (uint32_t)u, (int32_t*)c->prep_raw.p, key_ok);
  LAUNCH2(c, "g2_expand", k_g2_expand, u * (size_t)BN_NEG_G2_LINES, (const int32_t*)c->prep_raw.p, (uint32_t)u, table);
  HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
  return 0;
}
// Verify n tuples whose keys are given by index into a prepared table (d_kid[i] < u), everything device-resident.
// The caller has put the preparation of the table on stream2 (ev_join) or the table is final (join = false).
static int verify_prepared_dev(blsbn254_ctx* c, const int32_t* table, const uint8_t* key_ok, size_t u, const uint32_t* d_kid, bool hist_done,
                               const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sigs, size_t n, uint32_t dl, uint8_t* d_bitmap, bool join) {
  HIPCHK(c, c->h_ws.reserve(n * 27 * 4)); HIPCHK(c, c->f_ws.reserve(n * 108 * 4)); HIPCHK(c, c->flags.reserve(n));
  HIPCHK(c, c->kd_hist.reserve(4 * (u + 1))); HIPCHK(c, c->kd_cursor.reserve(4 * (u + 1))); HIPCHK(c, c->kd_perm.reserve(4 * n));
  HIPCHK(c, c->prep_isone.reserve(n)); HIPCHK(c, c->prep_valid.reserve(n)); HIPCHK(c, c->misc.reserve(64));
  uint32_t* hist = (uint32_t*)c->kd_hist.p; uint32_t* cursor = (uint32_t*)c->kd_cursor.p; uint32_t* perm = (uint32_t*)c->kd_perm.p;
  LAUNCH(c, "hash_to_g1", k_hash_to_g1, n, d_msgs, d_off, n, (const uint8_t*)c->dst.p, dl, (int32_t*)c->h_ws.p, n, (uint8_t*)nullptr, 3);   // homogeneous H: no inversion
  if (!hist_done) {
    int* d_bad = (int*)c->misc.p;
    static const int init = 0x7fffffff;
    HIPCHK(c, hipMemcpyAsync(d_bad, &init, 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(hist, 0, 4 * u, c->stream));
    LAUNCH(c, "kd_hist", k_kd_hist, n, d_kid, (uint32_t)n, (uint32_t)u, hist, d_bad);
    int bad;
    HIPCHK(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bad != 0x7fffffff) { c->last_error = "key index out of range at tuple " + std::to_string(bad); return BLSBN254_E_ARG; }
  }
  { ProfScope ps_(c, "kd_scan"); hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)hist, (uint32_t)u, cursor); }
  HIPCHK(c, hipGetLastError());
  LAUNCH(c, "kd_scatter", k_kd_scatter, n, d_kid, (uint32_t)n, (uint32_t)u, cursor, perm);
  if (join) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
  if (c->wide_fe && n <= c->wide_fe_max) {            // few tuples: one wave per tuple (k_miller_wide.hip), same values
    LAUNCH_WIDE(c, "miller_wide_prepared", k_miller_wide_prepared, n, (const uint32_t*)perm, d_kid, d_sigs, (const int32_t*)c->h_ws.p, n, table, key_ok, n,
                (int32_t*)c->f_ws.p, (uint8_t*)c->flags.p);
  } else {
    LAUNCH(c, "miller_prepared", k_miller_prepared, n, (const uint32_t*)perm, d_kid, d_sigs, (const int32_t*)c->h_ws.p, n, table, key_ok, n,
           (int32_t*)c->f_ws.p, (uint8_t*)c->flags.p);
  }
  int rc = run_final_exp(c, (int32_t*)c->f_ws.p, n, n, 4, nullptr, nullptr, nullptr, (uint8_t*)c->prep_isone.p, nullptr);
  if (rc) return rc;
  LAUNCH(c, "prep_unsort", k_prep_unsort, n, (const uint8_t*)c->prep_isone.p, (const uint8_t*)c->flags.p, (const uint32_t*)perm, (uint32_t)n, (uint8_t*)c->prep_valid.p);
  LAUNCH(c, "pack_bitmap", k_pack_bitmap, n, (const uint8_t*)c->prep_valid.p, n, d_bitmap);
  return 0;
}
// De-duplicate the public keys of a chunk.  *u_out = number of distinct keys; kd_kid / kd_keys / kd_hist are filled.
static int dedup_keys(blsbn254_ctx* c, const uint8_t* d_pks, size_t n, size_t* u_out) {
  size_t m = 1;
  while (m < 2 * n) m <<= 1;
  HIPCHK(c, c->kd_slots.reserve(4 * m)); HIPCHK(c, c->kd_rep.reserve(4 * n)); HIPCHK(c, c->kd_kid.reserve(4 * n)); HIPCHK(c, c->kd_keys.reserve(4 * n));
  HIPCHK(c, c->kd_hist.reserve(4 * (n + 1))); HIPCHK(c, c->kd_cnt.reserve(64));
  HIPCHK(c, hipMemsetAsync(c->kd_slots.p, 0xff, 4 * m, c->stream));
  HIPCHK(c, hipMemsetAsync(c->kd_cnt.p, 0, 4, c->stream));
  LAUNCH(c, "kd_insert", k_kd_insert, n, d_pks, (uint32_t)n, (uint32_t*)c->kd_slots.p, (uint32_t)(m - 1), c->kd_seed, (uint32_t*)c->kd_rep.p);
  LAUNCH(c, "kd_assign", k_kd_assign, n, (const uint32_t*)c->kd_rep.p, (uint32_t)n, (uint32_t*)c->kd_kid.p, (uint32_t*)c->kd_cnt.p, (uint32_t*)c->kd_keys.p);
  uint32_t u = 0;
  HIPCHK(c, hipMemcpyAsync(&u, c->kd_cnt.p, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *u_out = u;
  return 0;
}
// the exact per-tuple path: every tuple validates its own key and runs the two-pair Miller loop with a variable-Q pair
static int verify_exact_dev(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                            const uint8_t* d_sigs, size_t n, uint32_t dl, uint8_t* d_bitmap) {
  ++c->stat_exact_chunks;
  HIPCHK(c, c->h_ws.reserve(n * 18 * 4)); HIPCHK(c, c->f_ws.reserve(n * 108 * 4));
  HIPCHK(c, c->flags.reserve(n)); HIPCHK(c, c->sub_ok.reserve(n));
  LAUNCH(c, "hash_to_g1", k_hash_to_g1, n, d_msgs, d_off, n, (const uint8_t*)c->dst.p, dl, (int32_t*)c->h_ws.p, n, (uint8_t*)nullptr, 0);
  LAUNCH(c, "g2_check", k_g2_check, n, d_pks, n, (uint8_t*)c->sub_ok.p, (uint8_t*)nullptr);
  LAUNCH(c, "miller_verify", k_miller_verify, n, d_pks, d_sigs, (const int32_t*)c->h_ws.p, n, (int32_t*)c->f_ws.p, (uint8_t*)c->flags.p);
  return run_final_exp(c, (int32_t*)c->f_ws.p, n, n, 0, (const uint8_t*)c->flags.p, (const uint8_t*)c->sub_ok.p, d_bitmap, nullptr, nullptr);
}
// Workspace is ~7.4 KB per tuple (H, f, six final-exponentiation phase buffers, ten chain slots); batches
// larger than ctx->chunk (4 Mi) tuples are processed chunk by chunk so that any n fits the 288 GB of HBM.
static int verify_chunk_dev(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                            const uint8_t* d_sigs, size_t n, uint32_t dl, uint8_t* d_bitmap) {
  // Few distinct keys (a validator set signing many messages): every distinct key is validated and turned into its line
  // table ONCE (G2Prepared), beside hash-to-G1, and the tuples run the table-only Miller loop in key-sorted order.
  // Same bitmap as the exact per-tuple path below, which batches of mostly distinct keys keep taking.
  // Small chunks (at most wide_fe_max tuples) take it whatever their keys: with tables, the Miller loop and the final
  // exponentiation can run one WAVE per tuple (k_miller_wide.hip, k_fe_wide.hip) instead of at the latency of one lane.
  const bool small = c->wide_fe && n <= c->wide_fe_max;
  if (c->auto_prepare && (n >= 1024 || small)) {
    size_t u = 0;
    int rc = dedup_keys(c, d_pks, n, &u);
    if (rc) return rc;
    if ((u * 2 <= n || small) && u <= PREP_MAX_KEYS) {
      HIPCHK(c, c->prep_table.reserve(u * PREP_KEY_LIMBS * 4)); HIPCHK(c, c->prep_ok.reserve(u));
      rc = prepare_keys_async(c, d_pks, (const uint32_t*)c->kd_keys.p, u, (int32_t*)c->prep_table.p, (uint8_t*)c->prep_ok.p);
      if (rc) return rc;
      HIPCHK(c, hipMemsetAsync(c->kd_hist.p, 0, 4 * u, c->stream));
      LAUNCH(c, "kd_propagate", k_kd_propagate, n, (const uint32_t*)c->kd_rep.p, (uint32_t)n, (uint32_t)u, (uint32_t*)c->kd_kid.p, (uint32_t*)c->kd_hist.p);
      ++c->stat_prepared_chunks;
      return verify_prepared_dev(c, (const int32_t*)c->prep_table.p, (const uint8_t*)c->prep_ok.p, u, (const uint32_t*)c->kd_kid.p, true,
                                 d_msgs, d_off, d_sigs, n, dl, d_bitmap, true);
    }
  }
  return verify_exact_dev(c, d_pks, d_msgs, d_off, d_sigs, n, dl, d_bitmap);
}
int blsbn254_verify_batch_dev(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                              const uint8_t* d_sigs, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* d_bitmap) {
  if (!c || (n && (!d_pks || !d_off || !d_sigs || !d_bitmap)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t dl; int rc = stage_dst(c, dst, dst_len, &dl);
  if (rc) return rc;
  for (size_t lo = 0; lo < n; lo += c->chunk) {        // chunk starts are multiples of 8: bitmap bytes do not straddle
    size_t m = n - lo < c->chunk ? n - lo : c->chunk;
    rc = verify_chunk_dev(c, d_pks + 128 * lo, d_msgs, d_off + lo, d_sigs + 64 * lo, m, dl, d_bitmap + lo / 8);
    if (rc) return rc;
  }
  return 0;
}
static int fp12_tree(blsbn254_ctx* c, int32_t* a, size_t cnt, size_t sa, int32_t** res, size_t* rs);

// ---------------- G2Prepared: explicit API
struct blsbn254_g2prepared { blsbn254_ctx* ctx; size_t u; DevBuf table, raw, ok; };   // pair tables (verify), raw line triples (multi_miller_loop), validity
int blsbn254_g2_prepare_batch(blsbn254_ctx* c, const uint8_t* pks, size_t u, blsbn254_g2prepared** out) {
  if (!c || !out || (u && !pks)) return BLSBN254_E_ARG;
  *out = nullptr;
  if (u + 1 > PREP_MAX_KEYS) { c->last_error = "more than 65535 keys in one prepared table"; return BLSBN254_E_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  blsbn254_g2prepared* p = new blsbn254_g2prepared();
  p->ctx = c; p->u = u;
  // entry u (one past the caller's keys) is -G2gen: the second member of the aggregate signature's pair
  const size_t u1 = u + 1;
  hipError_t e1 = p->table.reserve(u1 * PREP_KEY_LIMBS * 4), e2 = p->ok.reserve(u1), e3 = c->in_a.reserve(128 * u1), e4 = p->raw.reserve(u1 * PREP_RAW_LIMBS * 4);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) { p->table.release(); p->raw.release(); p->ok.release(); delete p; return BLSBN254_E_NOMEM; }
  if (u) HIPCHK(c, hipMemcpyAsync(c->in_a.p, pks, 128 * u, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync((uint8_t*)c->in_a.p + 128 * u, NEG_G2_BYTES, 128, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "g2_prepare", k_g2_prepare, 2 * 256 * (size_t)nblocks(u1), (const uint8_t*)c->in_a.p, (const uint32_t*)nullptr, (uint32_t)u1, (int32_t*)p->raw.p, (uint8_t*)p->ok.p);
  LAUNCH(c, "g2_expand", k_g2_expand, u1 * (size_t)BN_NEG_G2_LINES, (const int32_t*)p->raw.p, (uint32_t)u1, (int32_t*)p->table.p);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *out = p;
  return 0;
}
void blsbn254_g2prepared_destroy(blsbn254_g2prepared* p) {
  if (!p) return;
  (void)hipSetDevice(p->ctx->device);
  (void)hipStreamSynchronize(p->ctx->stream);
  p->table.release(); p->raw.release(); p->ok.release();
  delete p;
}
size_t blsbn254_g2prepared_count(const blsbn254_g2prepared* p) { return p ? p->u : 0; }
// key validity (on curve, not the identity, in the r-torsion) of every prepared key, as a bitmap
int blsbn254_g2prepared_valid(blsbn254_ctx* c, const blsbn254_g2prepared* p, uint8_t* ok_bitmap) {
  if (!c || !p || p->ctx != c || (p->u && !ok_bitmap)) return BLSBN254_E_ARG;
  if (!p->u) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->bitmap.reserve((p->u + 7) / 8 + 8));
  LAUNCH(c, "pack_bitmap", k_pack_bitmap, p->u, (const uint8_t*)p->ok.p, p->u, (uint8_t*)c->bitmap.p);
  HIPCHK(c, hipMemcpyAsync(ok_bitmap, c->bitmap.p, (p->u + 7) / 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_verify_batch_prepared(blsbn254_ctx* c, const blsbn254_g2prepared* keys, const uint32_t* key_idx, const uint8_t* msgs, const uint64_t* off,
                                   const uint8_t* sigs, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* bm) {
  if (!c || !keys || keys->ctx != c || !off || (n && (!key_idx || !sigs || !bm)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t dl; int rc = stage_dst(c, dst, dst_len, &dl);
  if (rc) return rc;
  rc = stage_msgs(c, msgs, off, n);
  if (rc) return rc;
  const size_t nb = (n + 7) / 8;
  HIPCHK(c, c->in_b.reserve(64 * n)); HIPCHK(c, c->kd_kid.reserve(4 * n)); HIPCHK(c, c->bitmap.reserve(nb + 8));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, sigs, 64 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->kd_kid.p, key_idx, 4 * n, hipMemcpyHostToDevice, c->stream));
  for (size_t lo = 0; lo < n; lo += c->chunk) {
    size_t m = n - lo < c->chunk ? n - lo : c->chunk;
    rc = verify_prepared_dev(c, (const int32_t*)keys->table.p, (const uint8_t*)keys->ok.p, keys->u, (const uint32_t*)c->kd_kid.p + lo, false,
                             (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p + lo, (const uint8_t*)c->in_b.p + 64 * lo, m, dl, (uint8_t*)c->bitmap.p + lo / 8, false);
    if (rc) return rc;
  }
  HIPCHK(c, hipMemcpyAsync(bm, c->bitmap.p, nb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
// multi_miller_loop(&[(&G1Affine, &G2Prepared)]) (pairings.rs:808-857) over prepared keys named by index: the Fp12 product of
// the n Miller values, two pairs per lane sharing f^2, every line read from the keys' tables.  A pair whose G1 member is
// the identity contributes 1 (the reference skips such terms); every referenced key must be valid (else InvalidG2Bytes).
int blsbn254_multi_miller_loop_prepared(blsbn254_ctx* c, const blsbn254_g2prepared* keys, const uint32_t* key_idx, const uint8_t* g1, size_t n,
                                        uint8_t ml_out[384]) {
  if (!c || !keys || keys->ctx != c || !ml_out || (n && (!key_idx || !g1))) return BLSBN254_E_ARG;
  if (n == 0) { std::memset(ml_out, 0, 384); ml_out[31] = 1; return 0; }
  CHECK_LANES(c, n);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n_lanes = (n + 1) / 2;
  HIPCHK(c, c->in_a.reserve(64 * n)); HIPCHK(c, c->kd_kid.reserve(4 * n)); HIPCHK(c, c->h_ws.reserve(n * 18 * 4)); HIPCHK(c, c->f_ws.reserve(n_lanes * 108 * 4));
  HIPCHK(c, c->status.reserve(n)); HIPCHK(c, c->flags.reserve(n)); HIPCHK(c, c->kd_hist.reserve(4 * (keys->u + 1))); HIPCHK(c, c->misc.reserve(64)); HIPCHK(c, c->out.reserve(384));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, g1, 64 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->kd_kid.p, key_idx, 4 * n, hipMemcpyHostToDevice, c->stream));
  // key indices in range?
  int* d_bad = (int*)c->misc.p;
  static const int init = 0x7fffffff;
  HIPCHK(c, hipMemcpyAsync(d_bad, &init, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->kd_hist.p, 0, 4 * keys->u, c->stream));
  LAUNCH(c, "kd_hist", k_kd_hist, n, (const uint32_t*)c->kd_kid.p, (uint32_t)n, (uint32_t)keys->u, (uint32_t*)c->kd_hist.p, d_bad);
  int bad;
  HIPCHK(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (bad != 0x7fffffff) { c->last_error = "key index out of range at pair " + std::to_string(bad); return BLSBN254_E_ARG; }
  LAUNCH(c, "g1_to_ws", k_g1_to_ws_batch, n, (const uint8_t*)c->in_a.p, n, (int32_t*)c->h_ws.p, (uint8_t*)c->status.p);
  int rc = first_bad(c, (const uint8_t*)c->status.p, n, 1, 1, &bad);
  if (rc) return rc;
  if (bad >= 0) return BLSBN254_ERR_G1;
  const bool wide = c->wide_fe && n <= c->wide_fe_max;       // few pairs: one wave per pair, product of n values instead of n / 2
  const size_t f_cnt = wide ? n : n_lanes;
  if (wide) {
    HIPCHK(c, c->f_ws.reserve(n * 108 * 4));
    LAUNCH_WIDE(c, "miller_wide_1p", k_miller_wide_1p, n, (const int32_t*)c->h_ws.p, n, (const uint32_t*)c->kd_kid.p, (const int32_t*)keys->raw.p,
                (const uint8_t*)keys->ok.p, n, (int32_t*)c->f_ws.p, n, (uint8_t*)c->flags.p, (const uint8_t*)c->status.p);
  } else {
    LAUNCH(c, "miller_hpk2p", k_miller_hpk2p, n_lanes, (const int32_t*)c->h_ws.p, n, (const uint32_t*)c->kd_kid.p, (const int32_t*)keys->raw.p,
           (const uint8_t*)keys->ok.p, n, (int32_t*)c->f_ws.p, n_lanes, (uint8_t*)c->flags.p, (const uint8_t*)c->status.p);
  }
  rc = first_bad(c, (const uint8_t*)c->flags.p, n, 1, 1, &bad);
  if (rc) return rc;
  if (bad >= 0) return BLSBN254_ERR_G2;
  int32_t* res; size_t rs;
  rc = fp12_tree(c, (int32_t*)c->f_ws.p, f_cnt, f_cnt, &res, &rs);
  if (rc) return rc;
  LAUNCH(c, "fp12_to_bytes", k_fp12_to_bytes, 1, (const int32_t*)res, (size_t)1, rs, (uint8_t*)c->out.p);
  HIPCHK(c, hipMemcpyAsync(ml_out, c->out.p, 384, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
// CoreAggregateVerify with the public keys given as prepared keys by index: prod_i e(H(msg_i), pk_[key_idx_i]) * e(agg_sig, -G2gen) == 1.
// The signature's pair uses the table's own -G2gen entry; two pairs per lane, every line from the tables, ONE final exponentiation.
int blsbn254_aggregate_verify_prepared(blsbn254_ctx* c, const blsbn254_g2prepared* keys, const uint32_t* key_idx, const uint8_t* msgs, const uint64_t* off,
                                       size_t n, const uint8_t agg_sig[64], const uint8_t* dst, size_t dst_len, int* valid) {
  if (!c || !keys || keys->ctx != c || !valid || !agg_sig || !off || (n && !key_idx) || (dst_len && !dst)) return BLSBN254_E_ARG;
  *valid = 0;
  if (n == 0) return 0;
  const size_t np = n + 1, n_lanes = (np + 1) / 2;
  CHECK_LANES(c, np);
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t dl; int rc = stage_dst(c, dst, dst_len, &dl);
  if (rc) return rc;
  rc = stage_msgs(c, msgs, off, n);
  if (rc) return rc;
  HIPCHK(c, c->in_b.reserve(64)); HIPCHK(c, c->kd_kid.reserve(4 * np)); HIPCHK(c, c->h_ws.reserve(np * 18 * 4)); HIPCHK(c, c->f_ws.reserve(n_lanes * 108 * 4));
  HIPCHK(c, c->flags.reserve(np)); HIPCHK(c, c->kd_hist.reserve(4 * (keys->u + 2))); HIPCHK(c, c->misc.reserve(64));
  const uint32_t sig_key = (uint32_t)keys->u;
  HIPCHK(c, hipMemcpyAsync(c->kd_kid.p, key_idx, 4 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync((uint32_t*)c->kd_kid.p + n, &sig_key, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, agg_sig, 64, hipMemcpyHostToDevice, c->stream));
  int* d_ok = (int*)c->misc.p;               // [0] first bad key index (kd_hist), [1] all keys valid, [2] (byte) signature valid
  static const int init[3] = {0x7fffffff, 1, 1};
  HIPCHK(c, hipMemcpyAsync(d_ok, init, 12, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->kd_hist.p, 0, 4 * keys->u, c->stream));
  LAUNCH(c, "kd_hist", k_kd_hist, n, (const uint32_t*)c->kd_kid.p, (uint32_t)n, (uint32_t)keys->u, (uint32_t*)c->kd_hist.p, d_ok);
  int bad;
  HIPCHK(c, hipMemcpyAsync(&bad, d_ok, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));          // also: sig_key is on the stack
  if (bad != 0x7fffffff) { c->last_error = "key index out of range at pair " + std::to_string(bad); return BLSBN254_E_ARG; }
  LAUNCH(c, "hash_to_g1", k_hash_to_g1, n, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, n, (const uint8_t*)c->dst.p, dl, (int32_t*)c->h_ws.p, np, (uint8_t*)nullptr, 0);
  LAUNCH(c, "g1_to_ws", k_g1_to_ws, 1, (const uint8_t*)c->in_b.p, (int32_t*)c->h_ws.p, n, np, (uint8_t*)(d_ok + 2));
  LAUNCH(c, "miller_hpk2p", k_miller_hpk2p, n_lanes, (const int32_t*)c->h_ws.p, np, (const uint32_t*)c->kd_kid.p, (const int32_t*)keys->raw.p,
         (const uint8_t*)keys->ok.p, np, (int32_t*)c->f_ws.p, n_lanes, (uint8_t*)c->flags.p, (const uint8_t*)nullptr);
  LAUNCH(c, "and_reduce", k_and_reduce, n, (const uint8_t*)c->flags.p, (const uint8_t*)c->flags.p, n, d_ok + 1);
  int32_t* res; size_t rs;
  rc = fp12_tree(c, (int32_t*)c->f_ws.p, n_lanes, n_lanes, &res, &rs);
  if (rc) return rc;
  int* d_one = d_ok + 4;
  rc = run_final_exp(c, res, 1, rs, 3, nullptr, nullptr, nullptr, nullptr, d_one);
  if (rc) return rc;
  int h[5] = {0, 0, 0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(h, d_ok, 20, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *valid = (h[1] == 1 && (h[2] & 0xff) == 1 && h[4] == 1) ? 1 : 0;
  return 0;
}
// how many verify chunks took the prepared-key path / the exact per-tuple path on this context (tests, bench)
int blsbn254_path_stats(blsbn254_ctx* c, uint64_t out[2]) {
  if (!c || !out) return BLSBN254_E_ARG;
  out[0] = c->stat_prepared_chunks; out[1] = c->stat_exact_chunks;
  return 0;
}
int blsbn254_aggregate_path_stats(blsbn254_ctx* c, uint64_t out[2]) {
  if (!c || !out) return BLSBN254_E_ARG;
  out[0] = c->stat_grouped_aggregates; out[1] = c->stat_pairwise_aggregates;
  return 0;
}
int blsbn254_set_auto_prepare(blsbn254_ctx* c, int on) { if (!c) return BLSBN254_E_ARG; c->auto_prepare = on != 0; return 0; }

// ---------------- sums of G1 points per key
// `items` homogeneous points (limb-major at pts, stride pts_stride; optionally a second array pts2 summed alongside) are grouped
// by key in key order: item j has key kid[mark_perm[j]] and is the point in column pt_perm[j] (or j when pt_perm is NULL); key k
// owns hist[k] consecutive items ending at run_end[k].  Level by level, runs are cut into chunks of at most KEY_SUM_GROUP items,
// every chunk is summed by one lane, and the chunk sums (in key order too) are the items of the next level, until every key has
// ONE sum: *out / *out2 (stride u), indexed by key id.  One host synchronisation per level (the chunk count).
static const size_t KEY_SUM_GROUP = 32;
static int key_sums(blsbn254_ctx* c, const int32_t* pts, const int32_t* pts2, size_t pts_stride, const uint32_t* mark_perm, const uint32_t* pt_perm,
                    const uint32_t* kid, const uint32_t* hist, const uint32_t* run_end, size_t items, size_t u, const int32_t** out, const int32_t** out2) {
  const size_t G = KEY_SUM_GROUP, m_max = items / G + u;
  const uint32_t u32 = (uint32_t)u, G32 = (uint32_t)G;
  for (int t = 0; t < 2; ++t) {
    HIPCHK(c, c->ks_cnt[t].reserve(4 * (u + 2))); HIPCHK(c, c->ks_base[t].reserve(4 * (u + 2))); HIPCHK(c, c->ks_kid[t].reserve(4 * m_max));
    HIPCHK(c, c->ks_out[t].reserve(27 * 4 * m_max));
    if (pts2) HIPCHK(c, c->ks_out2[t].reserve(27 * 4 * m_max));
  }
  HIPCHK(c, c->ks_start.reserve(4 * m_max)); HIPCHK(c, c->ks_len.reserve(4 * m_max)); HIPCHK(c, c->ks_tchunk.reserve(4 * items)); HIPCHK(c, c->ks_iota.reserve(4 * m_max));
  LAUNCH(c, "iota", k_iota_u32, m_max, (uint32_t*)c->ks_iota.p, (uint32_t)m_max);
  int a = 0;
  for (int level = 0; ; ++level) {
    if (level > 8) { c->last_error = "internal: key sums do not converge"; return BLSBN254_E_HIP; }
    uint32_t *cnt = (uint32_t*)c->ks_cnt[a].p, *base = (uint32_t*)c->ks_base[a].p, *ckid = (uint32_t*)c->ks_kid[a].p;
    LAUNCH(c, "rlc2_counts", k_rlc2_chunk_counts, u + 1, hist, u32, G32, cnt);
    { ProfScope ps_(c, "kd_scan"); hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)cnt, u32 + 1, base); }
    HIPCHK(c, hipGetLastError());
    uint32_t m32 = 0;
    HIPCHK(c, hipMemcpyAsync(&m32, base + u, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t m = m32;
    if (m < u || m > m_max || m > items) { c->last_error = "internal: chunk count out of range"; return BLSBN254_E_HIP; }
    LAUNCH(c, "rlc2_mark", k_rlc2_mark, items, mark_perm, kid, hist, run_end, (const uint32_t*)base, (uint32_t)items, G32,
           (uint32_t*)c->ks_tchunk.p, ckid, (uint32_t*)c->ks_start.p, (uint32_t*)c->ks_len.p);
    LAUNCH(c, "g1_seg_sum", k_g1_seg_sum, m, pts, pts_stride, pt_perm, (const uint32_t*)c->ks_start.p, (const uint32_t*)c->ks_len.p, m, (int32_t*)c->ks_out[a].p, m);
    if (pts2) { LAUNCH(c, "g1_seg_sum", k_g1_seg_sum, m, pts2, pts_stride, pt_perm, (const uint32_t*)c->ks_start.p, (const uint32_t*)c->ks_len.p, m, (int32_t*)c->ks_out2[a].p, m); }
    pts = (const int32_t*)c->ks_out[a].p; pts2 = pts2 ? (const int32_t*)c->ks_out2[a].p : nullptr; pts_stride = m; items = m;
    if (m == u) break;                                                          // one chunk per key: chunk index == key id
    // next level: item j has key ckid[j]; key k owns items base[k] .. base[k + 1]
    mark_perm = (const uint32_t*)c->ks_iota.p; pt_perm = nullptr; kid = ckid; hist = cnt; run_end = base + 1;
    a ^= 1;
  }
  *out = pts;
  if (out2) *out2 = pts2;
  return 0;
}

// ---------------- random-linear-combination batch verification over repeated keys (k_rlc2.hip)
static int draw_seed(blsbn254_ctx* c, uint8_t out[32]) {
  size_t got = 0;
  while (got < 32) {
    ssize_t k = getrandom(out + got, 32 - got, 0);
    if (k <= 0) { c->last_error = "getrandom failed"; return BLSBN254_E_HIP; }
    got += (size_t)k;
  }
  return 0;
}
// One table-only Miller loop + final exponentiation over `cnt` (virtual or real) tuples: is_one bytes to d_isone, flags in ctx->flags.
static int prepared_round(blsbn254_ctx* c, const uint32_t* perm, const uint32_t* kid, const uint8_t* sigs, const int32_t* h_ws, size_t h_stride,
                          size_t cnt, uint8_t* d_isone) {
  HIPCHK(c, c->f_ws.reserve(cnt * 108 * 4)); HIPCHK(c, c->flags.reserve(cnt));
  if (c->wide_fe && cnt <= c->wide_fe_max) {
    LAUNCH_WIDE(c, "miller_wide_prepared", k_miller_wide_prepared, cnt, perm, kid, sigs, h_ws, h_stride, (const int32_t*)c->prep_table.p, (const uint8_t*)c->prep_ok.p, cnt,
                (int32_t*)c->f_ws.p, (uint8_t*)c->flags.p);
  } else {
    LAUNCH(c, "miller_prepared", k_miller_prepared, cnt, perm, kid, sigs, h_ws, h_stride, (const int32_t*)c->prep_table.p, (const uint8_t*)c->prep_ok.p, cnt,
           (int32_t*)c->f_ws.p, (uint8_t*)c->flags.p);
  }
  return run_final_exp(c, (int32_t*)c->f_ws.p, cnt, cnt, 4, nullptr, nullptr, nullptr, d_isone, nullptr);
}
// n <= ctx->chunk tuples, everything device-resident; d_seed = 32 bytes in device memory.  *took = 0 when the keys do not repeat
// (nothing was done: the caller takes another path).
static int rlc2_chunk_dev(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sigs, size_t n,
                          uint32_t dl, const uint8_t* d_seed, uint8_t* d_bitmap, bool* took) {
  *took = false;
  if (n < 2) return 0;
  size_t u = 0;
  int rc = dedup_keys(c, d_pks, n, &u);
  if (rc) return rc;
  if (!(u * 2 <= n && u <= PREP_MAX_KEYS)) return 0;
  *took = true;
  // Chunk size.  The chunk round runs one wave per SIMD, i.e. lanes_per_round virtual tuples at a time, and a launch that is a
  // few chunks over a multiple of that pays a whole extra round: unless the caller fixed G, take the next G (at most 2 G)
  // whose chunk-count bound n / G + u needs a round less.
  size_t G = c->rlc_group;
  if (c->rlc_group_auto) {
    const size_t R = c->lanes_per_round;
    const size_t r0 = (n / G + u + R - 1) / R;
    for (size_t g = G + 1; r0 > 1 && g <= 2 * G; ++g)
      if ((n / g + u + R - 1) / R < r0) { G = g; break; }
  }
  const size_t nblk = (n + 255) / 256;
  const uint32_t G32 = (uint32_t)G, n32 = (uint32_t)n, u32 = (uint32_t)u;
  HIPCHK(c, c->prep_table.reserve(u * PREP_KEY_LIMBS * 4)); HIPCHK(c, c->prep_ok.reserve(u));
  rc = prepare_keys_async(c, d_pks, (const uint32_t*)c->kd_keys.p, u, (int32_t*)c->prep_table.p, (uint8_t*)c->prep_ok.p);
  if (rc) return rc;
  HIPCHK(c, c->h_ws.reserve(n * 27 * 4)); HIPCHK(c, c->kd_cursor.reserve(4 * (u + 1))); HIPCHK(c, c->kd_perm.reserve(4 * n));
  HIPCHK(c, c->r2_a.reserve(n * 27 * 4)); HIPCHK(c, c->r2_b.reserve(n * 27 * 4)); HIPCHK(c, c->r2_sigok.reserve(n)); HIPCHK(c, c->r2_tchunk.reserve(4 * n));
  HIPCHK(c, c->r2_ccnt.reserve(4 * (u + 2))); HIPCHK(c, c->r2_cbase.reserve(4 * (u + 2)));
  HIPCHK(c, c->r2_need.reserve(n)); HIPCHK(c, c->r2_bcnt.reserve(4 * (nblk + 2))); HIPCHK(c, c->r2_bbase.reserve(4 * (nblk + 2)));
  HIPCHK(c, c->r2_list.reserve(4 * n)); HIPCHK(c, c->r2_valid.reserve(n));
  uint32_t *hist = (uint32_t*)c->kd_hist.p, *cursor = (uint32_t*)c->kd_cursor.p, *perm = (uint32_t*)c->kd_perm.p, *kid = (uint32_t*)c->kd_kid.p;
  uint32_t *ccnt = (uint32_t*)c->r2_ccnt.p, *cbase = (uint32_t*)c->r2_cbase.p;
  // key ids, key-sorted order, chunk numbering
  HIPCHK(c, hipMemsetAsync(hist, 0, 4 * u, c->stream));
  LAUNCH(c, "kd_propagate", k_kd_propagate, n, (const uint32_t*)c->kd_rep.p, n32, u32, kid, hist);
  { ProfScope ps_(c, "kd_scan"); hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)hist, u32, cursor); }
  HIPCHK(c, hipGetLastError());
  LAUNCH(c, "kd_scatter", k_kd_scatter, n, (const uint32_t*)kid, n32, u32, cursor, perm);          // cursor[k] is now the END of run k
  LAUNCH(c, "rlc2_counts", k_rlc2_chunk_counts, u + 1, (const uint32_t*)hist, u32, G32, ccnt);
  { ProfScope ps_(c, "kd_scan"); hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)ccnt, u32 + 1, cbase); }
  HIPCHK(c, hipGetLastError());
  uint32_t m32 = 0;
  HIPCHK(c, hipMemcpyAsync(&m32, cbase + u, 4, hipMemcpyDeviceToHost, c->stream));
  // the hash points and the weighted points r_i sig_i, r_i H_i (the host learns the chunk count while these run)
  LAUNCH(c, "hash_to_g1", k_hash_to_g1, n, d_msgs, d_off, n, (const uint8_t*)c->dst.p, dl, (int32_t*)c->h_ws.p, n, (uint8_t*)nullptr, 3);
  LAUNCH(c, "rlc2_prep", k_rlc2_prep, n, (const uint32_t*)perm, d_pks, d_sigs, (const int32_t*)c->h_ws.p, n, d_seed, (int32_t*)c->r2_a.p, (int32_t*)c->r2_b.p,
         (uint8_t*)c->r2_sigok.p);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t m = m32;
  if (m == 0 || m > n) { c->last_error = "internal: chunk count out of range"; return BLSBN254_E_HIP; }
  HIPCHK(c, c->r2_ckid.reserve(4 * m)); HIPCHK(c, c->r2_cstart.reserve(4 * m)); HIPCHK(c, c->r2_clen.reserve(4 * m)); HIPCHK(c, c->r2_csig.reserve(64 * m));
  HIPCHK(c, c->r2_ch.reserve(27 * 4 * m)); HIPCHK(c, c->r2_cstate.reserve(m)); HIPCHK(c, c->r2_iota.reserve(4 * m)); HIPCHK(c, c->r2_cisone.reserve(m));
  LAUNCH(c, "rlc2_mark", k_rlc2_mark, n, (const uint32_t*)perm, (const uint32_t*)kid, (const uint32_t*)hist, (const uint32_t*)cursor, (const uint32_t*)cbase, n32, G32,
         (uint32_t*)c->r2_tchunk.p, (uint32_t*)c->r2_ckid.p, (uint32_t*)c->r2_cstart.p, (uint32_t*)c->r2_clen.p);
  HIPCHK(c, c->r2_sa.reserve(27 * 4 * m)); HIPCHK(c, c->r2_sb.reserve(27 * 4 * m)); HIPCHK(c, c->r2_celig.reserve(4 * m));
  LAUNCH(c, "rlc2_sum", k_rlc2_sum, m, (const int32_t*)c->r2_a.p, (const int32_t*)c->r2_b.p, n, (const uint8_t*)c->r2_sigok.p, (const uint32_t*)c->r2_cstart.p,
         (const uint32_t*)c->r2_clen.p, m, (int32_t*)c->r2_sa.p, (int32_t*)c->r2_sb.p, (uint32_t*)c->r2_celig.p);
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));                      // the key tables are ready
  // The key round: ALL tuples of a key as one virtual tuple (the chunk sums of the key, summed) -- u checks, few enough for the
  // wave-per-tuple kernels.  A batch without invalid signatures (the usual case) is decided here, in a fraction of a chunk round;
  // otherwise only the chunks of the keys that failed are looked at below.  Same weights, hence the same 2^-64 bound per check.
  const size_t mblk = (m + 255) / 256;
  HIPCHK(c, c->r2_cpass.reserve(m)); HIPCHK(c, c->r2_clist.reserve(4 * m)); HIPCHK(c, c->r2_cneed.reserve(m)); HIPCHK(c, c->r2_cbcnt.reserve(4 * (mblk + 2)));
  HIPCHK(c, c->r2_cbbase.reserve(4 * (mblk + 2)));
  const uint32_t* clist = nullptr;                     // chunks of the chunk round (NULL: all of them, in order)
  size_t mc = m;
  LAUNCH(c, "iota", k_iota_u32, m, (uint32_t*)c->r2_iota.p, (uint32_t)m);
  // Batches that keep failing it (a stream with invalid signatures spread over all keys) would pay for the key round every time:
  // after a failure the next 2 (then 4, 8, 16) batches skip it; a pass resets the back-off.
  const bool key_round = c->rlc_key_round && c->rlc_key_skip == 0;
  if (c->rlc_key_round && c->rlc_key_skip) --c->rlc_key_skip;
  if (key_round) {
    HIPCHK(c, c->r2_kelig.reserve(4 * u)); HIPCHK(c, c->r2_ksig.reserve(64 * u)); HIPCHK(c, c->r2_kh.reserve(27 * 4 * u)); HIPCHK(c, c->r2_kstate.reserve(u));
    HIPCHK(c, c->r2_kisone.reserve(u)); HIPCHK(c, c->r2_kpass.reserve(u)); HIPCHK(c, c->misc.reserve(64));
    const int32_t *ksa = nullptr, *ksb = nullptr;
    rc = key_sums(c, (const int32_t*)c->r2_sa.p, (const int32_t*)c->r2_sb.p, m, (const uint32_t*)c->r2_iota.p, nullptr, (const uint32_t*)c->r2_ckid.p,
                  (const uint32_t*)ccnt, (const uint32_t*)cbase + 1, m, u, &ksa, &ksb);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->r2_kelig.p, 0, 4 * u, c->stream));
    LAUNCH(c, "rlc2_key_elig", k_rlc2_key_elig, m, (const uint32_t*)c->r2_ckid.p, (const uint32_t*)c->r2_celig.p, (uint32_t)m, (uint32_t*)c->r2_kelig.p);
    LAUNCH(c, "rlc2_virtual", k_rlc2_virtual, u, ksa, ksb, u, (const uint32_t*)c->r2_kelig.p, (const uint32_t*)nullptr, u, (uint8_t*)c->r2_ksig.p, (int32_t*)c->r2_kh.p,
           (uint8_t*)c->r2_kstate.p);
    rc = prepared_round(c, (const uint32_t*)c->r2_iota.p, (const uint32_t*)c->r2_iota.p, (const uint8_t*)c->r2_ksig.p, (const int32_t*)c->r2_kh.p, u, u, (uint8_t*)c->r2_kisone.p);
    if (rc) return rc;
    int* d_all = (int*)c->misc.p;
    static const int one_i = 1;
    HIPCHK(c, hipMemcpyAsync(d_all, &one_i, 4, hipMemcpyHostToDevice, c->stream));
    LAUNCH(c, "rlc2_keys_pass", k_rlc2_keys_pass, u, (const uint8_t*)c->prep_ok.p, (const uint8_t*)c->r2_kstate.p, (const uint8_t*)c->r2_kisone.p, u32, (uint8_t*)c->r2_kpass.p, d_all);
    // the chunks of the keys that failed, as an ordered list (counted while the host waits for the verdict)
    LAUNCH(c, "rlc2_chunk_need", k_rlc2_chunk_need, m, (const uint32_t*)c->r2_ckid.p, (const uint8_t*)c->r2_kpass.p, (uint32_t)m, (uint8_t*)c->r2_cneed.p, (uint32_t*)c->r2_cbcnt.p);
    { ProfScope ps_(c, "kd_scan"); hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)c->r2_cbcnt.p, (uint32_t)mblk + 1, (uint32_t*)c->r2_cbbase.p); }
    HIPCHK(c, hipGetLastError());
    int all_pass = 0; uint32_t mc32 = 0;
    HIPCHK(c, hipMemcpyAsync(&all_pass, d_all, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&mc32, (uint32_t*)c->r2_cbbase.p + mblk, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stat_rlc[0] += n; ++c->stat_rlc_key_rounds;
    if (all_pass == 1) {
      ++c->stat_rlc_key_rounds_passed;
      c->rlc_key_streak = 0;
      LAUNCH(c, "rlc2_valid_fast", k_rlc2_valid_fast, n, (const uint32_t*)perm, (const uint32_t*)kid, (const uint8_t*)c->r2_sigok.p, (const uint8_t*)c->prep_ok.p, n32, (uint8_t*)c->r2_valid.p);
      LAUNCH(c, "pack_bitmap", k_pack_bitmap, n, (const uint8_t*)c->r2_valid.p, n, d_bitmap);
      return 0;
    }
    if (c->rlc_key_streak < 4) ++c->rlc_key_streak;
    c->rlc_key_skip = 1u << c->rlc_key_streak;
    if (mc32 == 0 || mc32 > m) { c->last_error = "internal: chunk list out of range"; return BLSBN254_E_HIP; }
    mc = mc32;
    LAUNCH(c, "rlc2_compact", k_rlc2_compact, m, (const uint8_t*)c->r2_cneed.p, (const uint32_t*)c->r2_iota.p, (uint32_t)m, (const uint32_t*)c->r2_cbbase.p, (uint32_t*)c->r2_clist.p);
    clist = (const uint32_t*)c->r2_clist.p;
    HIPCHK(c, hipMemsetAsync(c->r2_cpass.p, 1, m, c->stream));               // chunks of the keys that passed
  } else {
    c->stat_rlc[0] += n;
  }
  // the chunk round: every (listed) chunk is one virtual tuple on the prepared-key verify path
  LAUNCH(c, "rlc2_virtual", k_rlc2_virtual, mc, (const int32_t*)c->r2_sa.p, (const int32_t*)c->r2_sb.p, m, (const uint32_t*)c->r2_celig.p, clist, mc, (uint8_t*)c->r2_csig.p,
         (int32_t*)c->r2_ch.p, (uint8_t*)c->r2_cstate.p);
  rc = prepared_round(c, clist ? clist : (const uint32_t*)c->r2_iota.p, (const uint32_t*)c->r2_ckid.p, (const uint8_t*)c->r2_csig.p, (const int32_t*)c->r2_ch.p, m, mc, (uint8_t*)c->r2_cisone.p);
  if (rc) return rc;
  LAUNCH(c, "rlc2_chunk_pass", k_rlc2_chunk_pass, mc, clist, (const uint8_t*)c->r2_cstate.p, (const uint8_t*)c->r2_cisone.p, (const uint8_t*)c->flags.p, (uint32_t)mc, (uint8_t*)c->r2_cpass.p);
  LAUNCH(c, "rlc2_resolve", k_rlc2_resolve, n, (const uint32_t*)perm, (const uint32_t*)kid, (const uint32_t*)c->r2_tchunk.p, (const uint8_t*)c->r2_sigok.p,
         (const uint8_t*)c->prep_ok.p, (const uint8_t*)c->r2_cpass.p, n32, (uint8_t*)c->r2_valid.p, (uint8_t*)c->r2_need.p, (uint32_t*)c->r2_bcnt.p);
  { ProfScope ps_(c, "kd_scan"); hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)c->r2_bcnt.p, (uint32_t)nblk + 1, (uint32_t*)c->r2_bbase.p); }
  HIPCHK(c, hipGetLastError());
  uint32_t m2 = 0;
  HIPCHK(c, hipMemcpyAsync(&m2, (uint32_t*)c->r2_bbase.p + nblk, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (m2 > n) { c->last_error = "internal: fallback count out of range"; return BLSBN254_E_HIP; }
  c->stat_rlc[1] += mc; c->stat_rlc[2] += m2;
  if (m2) {                                                                     // eligible tuples of failed chunks: the exact prepared-key path
    HIPCHK(c, c->prep_isone.reserve(m2));
    LAUNCH(c, "rlc2_compact", k_rlc2_compact, n, (const uint8_t*)c->r2_need.p, (const uint32_t*)perm, n32, (const uint32_t*)c->r2_bbase.p, (uint32_t*)c->r2_list.p);
    rc = prepared_round(c, (const uint32_t*)c->r2_list.p, (const uint32_t*)kid, d_sigs, (const int32_t*)c->h_ws.p, n, m2, (uint8_t*)c->prep_isone.p);
    if (rc) return rc;
    LAUNCH(c, "prep_unsort", k_prep_unsort, m2, (const uint8_t*)c->prep_isone.p, (const uint8_t*)c->flags.p, (const uint32_t*)c->r2_list.p, m2, (uint8_t*)c->r2_valid.p);
  }
  LAUNCH(c, "pack_bitmap", k_pack_bitmap, n, (const uint8_t*)c->r2_valid.p, n, d_bitmap);
  return 0;
}
static int stage_seed(blsbn254_ctx* c, const uint8_t* seed) {
  uint8_t own[32];
  if (!seed) {                                        // the normal case: 32 bytes from the OS, drawn now -- after the batch is fixed
    int rc = draw_seed(c, own);
    if (rc) return rc;
    seed = own;
  }
  HIPCHK(c, c->r2_seed.reserve(32));
  HIPCHK(c, hipMemcpyAsync(c->r2_seed.p, seed, 32, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));         // `own` is on the stack
  return 0;
}
int blsbn254_verify_batch_rlc_dev(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sigs, size_t n,
                                  const uint8_t* dst, size_t dst_len, const uint8_t seed[32], uint8_t* d_bitmap) {
  if (!c || (n && (!d_pks || !d_off || !d_sigs || !d_bitmap)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t dl; int rc = stage_dst(c, dst, dst_len, &dl);
  if (rc) return rc;
  rc = stage_seed(c, seed);
  if (rc) return rc;
  for (size_t lo = 0; lo < n; lo += c->chunk) {        // chunk starts are multiples of 8: bitmap bytes do not straddle
    size_t m = n - lo < c->chunk ? n - lo : c->chunk;
    bool took = false;
    rc = rlc2_chunk_dev(c, d_pks + 128 * lo, d_msgs, d_off + lo, d_sigs + 64 * lo, m, dl, (const uint8_t*)c->r2_seed.p, d_bitmap + lo / 8, &took);
    if (rc) return rc;
    if (!took) {                                       // keys do not repeat: nothing to share per key, the exact per-tuple path
      c->stat_rlc[3] += m;
      rc = verify_exact_dev(c, d_pks + 128 * lo, d_msgs, d_off + lo, d_sigs + 64 * lo, m, dl, d_bitmap + lo / 8);
      if (rc) return rc;
    }
  }
  return 0;
}
int blsbn254_set_rlc_group(blsbn254_ctx* c, size_t group) {
  if (!c || group == 1 || group > 4096) return BLSBN254_E_ARG;
  c->rlc_group = group ? group : 16;
  c->rlc_group_auto = group == 0;
  return 0;
}
int blsbn254_rlc_stats(blsbn254_ctx* c, uint64_t out[6]) {
  if (!c || !out) return BLSBN254_E_ARG;
  for (int k = 0; k < 4; ++k) out[k] = c->stat_rlc[k];
  out[4] = c->stat_rlc_key_rounds; out[5] = c->stat_rlc_key_rounds_passed;
  return 0;
}
int blsbn254_set_rlc_key_round(blsbn254_ctx* c, int on) { if (!c) return BLSBN254_E_ARG; c->rlc_key_round = on != 0; c->rlc_key_skip = c->rlc_key_streak = 0; return 0; }

// ---------------- random-linear-combination batch verification
static const size_t RLC_GROUP = 16;      // distinct-key variant: tuples per shared final exponentiation (power of two)
int blsbn254_verify_batch_rlc(blsbn254_ctx* c, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs,
                              size_t n, const uint8_t* dst, size_t dst_len, const uint8_t seed[32], uint8_t* bm) {
  if (!c || !off || (n && (!pks || !sigs || !bm)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t dl; int rc = stage_dst(c, dst, dst_len, &dl);
  if (rc) return rc;
  rc = stage_msgs(c, msgs, off, n);
  if (rc) return rc;
  const size_t nb = (n + 7) / 8;
  HIPCHK(c, c->in_a.reserve(128 * n)); HIPCHK(c, c->in_b.reserve(64 * n)); HIPCHK(c, c->bitmap.reserve(nb + 8));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, pks, 128 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, sigs, 64 * n, hipMemcpyHostToDevice, c->stream));
  rc = stage_seed(c, seed);
  if (rc) return rc;
  // Repeated keys: per-key chunks as virtual tuples on the prepared-key path (k_rlc2.hip).  Batches beyond one launch chunk
  // go chunk by chunk through the device entry point (which takes the exact path for a chunk of distinct keys).
  bool took = false;
  if (n <= c->chunk) {
    rc = rlc2_chunk_dev(c, (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, (const uint8_t*)c->in_b.p, n, dl,
                        (const uint8_t*)c->r2_seed.p, (uint8_t*)c->bitmap.p, &took);
    if (rc) return rc;
  } else {
    rc = blsbn254_verify_batch_rlc_dev(c, (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, (const uint8_t*)c->in_b.p, n,
                                       dst, dst_len, seed, (uint8_t*)c->bitmap.p);
    if (rc) return rc;
    took = true;
  }
  if (took) {
    HIPCHK(c, hipMemcpyAsync(bm, c->bitmap.p, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
  }
  // Distinct keys: groups of RLC_GROUP tuples in the caller's order share the signature-side Miller loop and the final exponentiation.
  c->stat_rlc[3] += n;
  const size_t G = RLC_GROUP, n_pad = (n + G - 1) / G * G, ng = n_pad / G;
  HIPCHK(c, c->misc.reserve(64));
  HIPCHK(c, c->h_ws.reserve(n * 18 * 4)); HIPCHK(c, c->sub_ok.reserve(n)); HIPCHK(c, c->flags.reserve(n_pad));
  HIPCHK(c, c->rlc_a.reserve(n_pad * 27 * 4)); HIPCHK(c, c->rlc_b.reserve(n_pad * 18 * 4)); HIPCHK(c, c->rlc_elig.reserve(n_pad));
  HIPCHK(c, c->f_ws.reserve(n_pad * 108 * 4)); HIPCHK(c, c->rlc_f2.reserve(ng * 108 * 4)); HIPCHK(c, c->rlc_bytes.reserve(ng * 64));
  HIPCHK(c, c->rlc_neg.reserve(ng * 128)); HIPCHK(c, c->rlc_ok.reserve(ng)); HIPCHK(c, c->status.reserve(ng + 8));
  HIPCHK(c, hipMemcpyAsync(c->misc.p, c->r2_seed.p, 32, hipMemcpyDeviceToDevice, c->stream));
  { std::vector<uint8_t> neg(ng * 128);
    for (size_t g = 0; g < ng; ++g) std::memcpy(neg.data() + 128 * g, NEG_G2_BYTES, 128);
    HIPCHK(c, hipMemcpyAsync(c->rlc_neg.p, neg.data(), ng * 128, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); }
  const uint8_t* d_pks = (const uint8_t*)c->in_a.p; const uint8_t* d_sigs = (const uint8_t*)c->in_b.p;
  int32_t* f = (int32_t*)c->f_ws.p;
  LAUNCH(c, "hash_to_g1", k_hash_to_g1, n, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, n, (const uint8_t*)c->dst.p, dl, (int32_t*)c->h_ws.p, n, (uint8_t*)nullptr, 0);
  LAUNCH(c, "g2_check", k_g2_check, n, d_pks, n, (uint8_t*)c->sub_ok.p, (uint8_t*)nullptr);
  LAUNCH(c, "rlc_prep", k_rlc_prep, n_pad, d_pks, d_sigs, (const int32_t*)c->h_ws.p, (const uint8_t*)c->sub_ok.p, (const uint8_t*)c->misc.p,
         n, n_pad, (int32_t*)c->rlc_a.p, (int32_t*)c->rlc_b.p, (uint8_t*)c->rlc_elig.p);
  // prod_i ML(r_i H_i, pk_i) per group: per-pair loops, ONE where not eligible, log2(G) levels of pairwise products
  LAUNCH(c, "miller_hpk", k_miller_hpk, n, (const int32_t*)c->rlc_b.p, d_pks, n, f, n_pad, (uint8_t*)c->flags.p);
  LAUNCH(c, "fp12_mask_one", k_fp12_mask_one, n_pad, f, n_pad, (const uint8_t*)c->rlc_elig.p, n_pad);
  HIPCHK(c, c->f_ws2.reserve((n_pad / 2) * 108 * 4)); HIPCHK(c, c->rlc_a2.reserve((n_pad / 2) * 27 * 4));
  int32_t *pa = f, *pb = (int32_t*)c->f_ws2.p, *ga = (int32_t*)c->rlc_a.p, *gb = (int32_t*)c->rlc_a2.p;
  size_t cnt = n_pad, st = n_pad;
  for (size_t lvl = 1; lvl < G; lvl <<= 1) {                          // adjacent pairs never straddle a group
    size_t mo = cnt / 2;
    LAUNCH(c, "fp12_mul_pairs", k_fp12_mul_pairs, mo, (const int32_t*)pa, cnt, st, pb, mo);
    LAUNCH(c, "g1_add_pairs", k_g1_add_pairs, mo, (const int32_t*)ga, cnt, st, gb, mo);
    std::swap(pa, pb); std::swap(ga, gb); st = mo; cnt = mo;
  }
  // e(sum_i r_i sig_i, -G2gen) per group, multiplied in; one final exponentiation per group
  LAUNCH(c, "g1p_to_bytes", k_g1p_to_bytes, ng, (const int32_t*)ga, st, ng, (uint8_t*)c->rlc_bytes.p);
  LAUNCH(c, "miller_1", k_miller_1, ng, (const uint8_t*)c->rlc_bytes.p, (const uint8_t*)c->rlc_neg.p, ng, (int32_t*)c->rlc_f2.p, ng, (uint8_t*)c->status.p);
  LAUNCH(c, "fp12_mul_elem", k_fp12_mul_elem, ng, pa, st, (const int32_t*)c->rlc_f2.p, ng, ng);
  rc = run_final_exp(c, pa, ng, st, 4, nullptr, nullptr, nullptr, (uint8_t*)c->rlc_ok.p, nullptr);
  if (rc) return rc;
  std::vector<uint8_t> h_ok(ng), h_elig(n_pad);
  HIPCHK(c, hipMemcpyAsync(h_ok.data(), c->rlc_ok.p, ng, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_elig.data(), c->rlc_elig.p, n_pad, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memset(bm, 0, nb);
  std::vector<uint32_t> idx;
  for (size_t g = 0; g < ng; ++g) {
    for (size_t i = g * G; i < (g + 1) * G && i < n; ++i) {
      if (!h_elig[i]) continue;                                        // failed a precheck: invalid, not part of any product
      if (h_ok[g]) bm[i >> 3] |= (uint8_t)(1u << (i & 7));
      else idx.push_back((uint32_t)i);
    }
  }
  if (!idx.empty()) {                                                  // exact per-tuple path for the groups that failed
    size_t m = idx.size(), mb = (m + 7) / 8;
    HIPCHK(c, c->rlc_idx.reserve(4 * m)); HIPCHK(c, c->rlc_cpk.reserve(128 * m)); HIPCHK(c, c->rlc_csig.reserve(64 * m));
    HIPCHK(c, c->rlc_ch.reserve(18 * 4 * m)); HIPCHK(c, c->rlc_csub.reserve(m)); HIPCHK(c, c->rlc_cbm.reserve(mb + 8));
    HIPCHK(c, c->f_ws.reserve(m * 108 * 4)); HIPCHK(c, c->flags.reserve(m));
    HIPCHK(c, hipMemcpyAsync(c->rlc_idx.p, idx.data(), 4 * m, hipMemcpyHostToDevice, c->stream));
    LAUNCH(c, "rlc_gather", k_rlc_gather, m, (const uint32_t*)c->rlc_idx.p, m, d_pks, d_sigs, (const int32_t*)c->h_ws.p, n, (const uint8_t*)c->sub_ok.p,
           (uint8_t*)c->rlc_cpk.p, (uint8_t*)c->rlc_csig.p, (int32_t*)c->rlc_ch.p, (uint8_t*)c->rlc_csub.p);
    LAUNCH(c, "miller_verify", k_miller_verify, m, (const uint8_t*)c->rlc_cpk.p, (const uint8_t*)c->rlc_csig.p, (const int32_t*)c->rlc_ch.p, m,
           (int32_t*)c->f_ws.p, (uint8_t*)c->flags.p);
    rc = run_final_exp(c, (int32_t*)c->f_ws.p, m, m, 0, (const uint8_t*)c->flags.p, (const uint8_t*)c->rlc_csub.p, (uint8_t*)c->rlc_cbm.p, nullptr, nullptr);
    if (rc) return rc;
    std::vector<uint8_t> cb(mb);
    HIPCHK(c, hipMemcpyAsync(cb.data(), c->rlc_cbm.p, mb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t j = 0; j < m; ++j) if (cb[j >> 3] & (1u << (j & 7))) { size_t i = idx[j]; bm[i >> 3] |= (uint8_t)(1u << (i & 7)); }
  }
  return 0;
}

int blsbn254_verify_batch(blsbn254_ctx* c, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs,
                          size_t n, const uint8_t* dst, size_t dst_len, uint8_t* bm) {
  if (!c || !off || (n && (!pks || !sigs || !bm)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = stage_msgs(c, msgs, off, n);
  if (rc) return rc;
  size_t nb = (n + 7) / 8;
  HIPCHK(c, c->in_a.reserve(128 * n)); HIPCHK(c, c->in_b.reserve(64 * n)); HIPCHK(c, c->bitmap.reserve(nb + 8));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, pks, 128 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, sigs, 64 * n, hipMemcpyHostToDevice, c->stream));
  rc = blsbn254_verify_batch_dev(c, (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p,
                                 (const uint8_t*)c->in_b.p, n, dst, dst_len, (uint8_t*)c->bitmap.p);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(bm, c->bitmap.p, nb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
// in-place product tree over `cnt` Fp12 values at a (stride sa); result pointer / stride returned
static int fp12_tree(blsbn254_ctx* c, int32_t* a, size_t cnt, size_t sa, int32_t** res, size_t* rs) {
  HIPCHK(c, c->f_ws2.reserve(((cnt + 1) / 2) * 108 * 4 + 432));
  int32_t* b = (int32_t*)c->f_ws2.p;
  while (cnt > 1) {
    size_t mo = (cnt + 1) / 2;
    LAUNCH(c, "fp12_mul_pairs", k_fp12_mul_pairs, mo, (const int32_t*)a, cnt, sa, b, mo);
    std::swap(a, b); sa = mo; cnt = mo;
  }
  *res = a; *rs = sa;
  return 0;
}
// prod_i ML(H(msg_i), pk_i) over the caller's n pairs, optionally times ML(extra_sig, -G2gen): the aggregate signature
// then simply joins the batch as pair n (one more lane half among the million) instead of a latency-bound one-lane launch.
// Two pairs per lane sharing one f^2 (k_miller_hpk2), then the pairwise product tree.
// staged = true: the caller (aggregate_verify_grouped, which found the keys distinct) has already put dst, messages, keys and
// the signature where this function stages them
static int aggregate_partial_impl(blsbn254_ctx* c, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, size_t n,
                                  const uint8_t* dst, size_t dst_len, const uint8_t* extra_sig, uint8_t ml_out[384], int* all_pks_ok, int* sig_ok,
                                  bool staged = false) {
  *all_pks_ok = 1;
  if (sig_ok) *sig_ok = 1;
  const size_t np = n + (extra_sig ? 1 : 0);                 // pairs in the loop
  if (np == 0) { std::memset(ml_out, 0, 384); ml_out[31] = 1; return 0; }
  CHECK_LANES(c, np);
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t dl = 0; int rc;
  if (n) {
    rc = stage_dst(c, dst, dst_len, &dl);                    // (a no-op when the tag is already resident)
    if (rc) return rc;
    if (!staged) {
      rc = stage_msgs(c, msgs, off, n);
      if (rc) return rc;
    }
  }
  const size_t n_lanes = (np + 1) / 2;
  HIPCHK(c, c->in_a.reserve(128 * np)); HIPCHK(c, c->in_b.reserve(64)); HIPCHK(c, c->h_ws.reserve(np * 18 * 4)); HIPCHK(c, c->f_ws.reserve(n_lanes * 108 * 4));
  HIPCHK(c, c->q_ws.reserve(n_lanes * 72 * 4));
  HIPCHK(c, c->flags.reserve(np)); HIPCHK(c, c->sub_ok.reserve(np)); HIPCHK(c, c->misc.reserve(64)); HIPCHK(c, c->out.reserve(384));
  if (n && !staged) HIPCHK(c, hipMemcpyAsync(c->in_a.p, pks, 128 * n, hipMemcpyHostToDevice, c->stream));
  int32_t* f = (int32_t*)c->f_ws.p;
  int* d_ok = (int*)c->misc.p;                               // [0] all keys valid, [1] (byte) signature valid
  static const int ones[2] = {1, 1};
  HIPCHK(c, hipMemcpyAsync(d_ok, ones, 8, hipMemcpyHostToDevice, c->stream));
  if (extra_sig && !staged) {
    HIPCHK(c, hipMemcpyAsync(c->in_b.p, extra_sig, 64, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((uint8_t*)c->in_a.p + 128 * n, NEG_G2_BYTES, 128, hipMemcpyHostToDevice, c->stream));
  }
  // few distinct keys among the np pairs (the pair (sig, -G2gen) included)?  then prepare each once, beside the hashing
  bool prepared = false;
  if (c->auto_prepare && np >= 1024) {
    size_t u = 0;
    rc = dedup_keys(c, (const uint8_t*)c->in_a.p, np, &u);
    if (rc) return rc;
    if (u * 2 <= np && u <= PREP_MAX_KEYS) {
      HIPCHK(c, c->prep_table.reserve(64)); HIPCHK(c, c->prep_ok.reserve(u)); HIPCHK(c, c->prep_raw.reserve(u * PREP_RAW_LIMBS * 4));
      HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
      HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      LAUNCH2(c, "g2_prepare", k_g2_prepare, 2 * 256 * (size_t)nblocks(u), (const uint8_t*)c->in_a.p, (const uint32_t*)c->kd_keys.p, (uint32_t)u, (int32_t*)c->prep_raw.p, (uint8_t*)c->prep_ok.p);
      HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
      LAUNCH(c, "kd_propagate", k_kd_propagate, np, (const uint32_t*)c->kd_rep.p, (uint32_t)np, (uint32_t)u, (uint32_t*)c->kd_kid.p, (uint32_t*)nullptr);   // no sorting here: no histogram
      prepared = true;
    }
  }
  // hash (and, on the exact path, key checks) over the caller's n pairs; their H points land in slots 0..n-1 of a stride-np workspace
  if (n) {
    LAUNCH(c, "hash_to_g1", k_hash_to_g1, n, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, n, (const uint8_t*)c->dst.p, dl, (int32_t*)c->h_ws.p, np, (uint8_t*)nullptr, 0);
    if (!prepared) { LAUNCH(c, "g2_check", k_g2_check, n, (const uint8_t*)c->in_a.p, n, (uint8_t*)c->sub_ok.p, (uint8_t*)nullptr); }
  }
  if (extra_sig) { LAUNCH(c, "g1_to_ws", k_g1_to_ws, 1, (const uint8_t*)c->in_b.p, (int32_t*)c->h_ws.p, n, np, (uint8_t*)(d_ok + 1)); }
  if (prepared) {
    // few distinct keys: every key (and -G2gen, when the signature's pair is carried) was validated and turned into its line
    // table once, beside hash-to-G1; the pairs read their lines from those tables
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    LAUNCH(c, "miller_hpk2p", k_miller_hpk2p, n_lanes, (const int32_t*)c->h_ws.p, np, (const uint32_t*)c->kd_kid.p, (const int32_t*)c->prep_raw.p,
           (const uint8_t*)c->prep_ok.p, np, f, n_lanes, (uint8_t*)c->flags.p, (const uint8_t*)nullptr);
    if (n) { LAUNCH(c, "and_reduce", k_and_reduce, n, (const uint8_t*)c->flags.p, (const uint8_t*)c->flags.p, n, d_ok); }
  } else {
    LAUNCH(c, "miller_hpk2", k_miller_hpk2, n_lanes, (const int32_t*)c->h_ws.p, (const uint8_t*)c->in_a.p, np, (int32_t*)c->q_ws.p, f, n_lanes, (uint8_t*)c->flags.p);
    if (n) { LAUNCH(c, "and_reduce", k_and_reduce, n, (const uint8_t*)c->flags.p, (const uint8_t*)c->sub_ok.p, n, d_ok); }
  }
  int32_t* res; size_t rs;
  rc = fp12_tree(c, f, n_lanes, n_lanes, &res, &rs);
  if (rc) return rc;
  LAUNCH(c, "fp12_to_bytes", k_fp12_to_bytes, 1, (const int32_t*)res, (size_t)1, rs, (uint8_t*)c->out.p);
  int h_ok[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(ml_out, c->out.p, 384, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_ok, d_ok, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *all_pks_ok = h_ok[0];
  if (sig_ok) *sig_ok = (h_ok[1] & 0xff) == 1;
  return 0;
}
int blsbn254_aggregate_partial(blsbn254_ctx* c, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, size_t n,
                               const uint8_t* dst, size_t dst_len, uint8_t ml_out[384], int* all_pks_ok) {
  if (!c || !ml_out || !all_pks_ok || !off || (n && !pks) || (dst_len && !dst)) return BLSBN254_E_ARG;
  return aggregate_partial_impl(c, pks, msgs, off, n, dst, dst_len, nullptr, ml_out, all_pks_ok, nullptr);
}
// The first shard of a sharded aggregate verify may carry the aggregate signature's pair as well (then the finishing
// call passes agg_sig = NULL): *sig_ok = the signature decodes, is not the identity and is on the curve.
int blsbn254_aggregate_partial_with_sig(blsbn254_ctx* c, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, size_t n,
                                        const uint8_t* dst, size_t dst_len, const uint8_t agg_sig[64], uint8_t ml_out[384], int* all_pks_ok, int* sig_ok) {
  if (!c || !ml_out || !all_pks_ok || !sig_ok || !agg_sig || !off || (n && !pks) || (dst_len && !dst)) return BLSBN254_E_ARG;
  return aggregate_partial_impl(c, pks, msgs, off, n, dst, dst_len, agg_sig, ml_out, all_pks_ok, sig_ok);
}
int blsbn254_aggregate_finish(blsbn254_ctx* c, const uint8_t* partials, size_t k, const uint8_t agg_sig[64], int* valid) {
  if (!c || !valid || (k && !partials) || (!k && !agg_sig)) return BLSBN254_E_ARG;
  *valid = 0;
  HIPCHK(c, hipSetDevice(c->device));
  const bool with_sig = agg_sig != nullptr;                // NULL: a partial already carries ML(agg_sig, -G2gen)
  size_t m = k + (with_sig ? 1 : 0);                       // slot k holds ML(agg_sig, -G2gen)
  HIPCHK(c, c->in_a.reserve(384 * (k ? k : 1))); HIPCHK(c, c->in_b.reserve(64 + 128)); HIPCHK(c, c->f_ws.reserve(m * 108 * 4));
  HIPCHK(c, c->status.reserve(k + 8)); HIPCHK(c, c->misc.reserve(64)); HIPCHK(c, c->bitmap.reserve(16));
  int32_t* f = (int32_t*)c->f_ws.p;
  if (with_sig) {
    uint8_t last[64 + 128];
    std::memcpy(last, agg_sig, 64); std::memcpy(last + 64, NEG_G2_BYTES, 128);
    HIPCHK(c, hipMemcpyAsync(c->in_b.p, last, sizeof last, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));       // `last` is on the stack
  }
  if (k) {
    HIPCHK(c, hipMemcpyAsync(c->in_a.p, partials, 384 * k, hipMemcpyHostToDevice, c->stream));
    // partials arrive as bytes: decode into slots 0..k-1 of the stride-m array
    LAUNCH(c, "fp12_from_bytes", k_fp12_from_bytes, k, (const uint8_t*)c->in_a.p, k, f, m, (uint8_t*)c->status.p);
    int bad; int rc = first_bad(c, (const uint8_t*)c->status.p, k, 1, 1, &bad);
    if (rc) return rc;
    if (bad >= 0) return BLSBN254_ERR_GT;
  }
  if (with_sig) {
    LAUNCH(c, "miller_1", k_miller_1, 1, (const uint8_t*)c->in_b.p, (const uint8_t*)c->in_b.p + 64, (size_t)1, f + k, m, (uint8_t*)c->status.p);
    LAUNCH(c, "g1_check", k_g1_check, 1, (const uint8_t*)c->in_b.p, (size_t)1, (uint8_t*)c->bitmap.p);
  }
  int32_t* res; size_t rs;
  int rc = fp12_tree(c, f, m, m, &res, &rs);
  if (rc) return rc;
  int* d_one = (int*)c->misc.p;
  rc = run_final_exp(c, res, 1, rs, 3, nullptr, nullptr, nullptr, nullptr, d_one);
  if (rc) return rc;
  int h_one = 0; uint8_t sig_st = 3, sig_on_curve = 1;
  HIPCHK(c, hipMemcpyAsync(&h_one, d_one, 4, hipMemcpyDeviceToHost, c->stream));
  if (with_sig) {
    HIPCHK(c, hipMemcpyAsync(&sig_st, c->status.p, 1, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&sig_on_curve, c->bitmap.p, 1, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  bool sig_ok = (sig_st & 7) == 3 && (sig_on_curve & 1);      // decodes, not the identity, on the curve
  *valid = (sig_ok && h_one == 1) ? 1 : 0;
  return 0;
}
// Aggregate verify over a batch that repeats public keys, by bilinearity in the first argument (exact, no randomness):
//   prod_i e(H_i, pk_(k_i)) = prod_k e( sum_{i: k_i = k} H_i , pk_k )
// so only one Miller loop per DISTINCT key (plus the signature's pair) runs, after n G1 additions: key de-duplication and
// key-sorted order as in verify_batch, the sums by levels of chunks of KEY_SUM_GROUP points (key_sums, k_g1_seg_sum), the u + 1 pairs on the
// prepared two-pairs-per-lane loop, product tree, ONE final exponentiation.  The boolean is aggregate_verify's; the Miller
// value is not the product of the n per-pair values (blsbn254_aggregate_partial keeps that bit-exact form for the sharded API).
// *took = false: keys do not repeat, nothing was done.
static int aggregate_verify_grouped(blsbn254_ctx* c, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t agg_sig[64],
                                    const uint8_t* dst, size_t dst_len, int* valid, bool* took) {
  *took = false;
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t dl = 0;
  int rc = stage_dst(c, dst, dst_len, &dl);
  if (rc) return rc;
  rc = stage_msgs(c, msgs, off, n);
  if (rc) return rc;
  HIPCHK(c, c->in_a.reserve(128 * (n + 1))); HIPCHK(c, c->in_b.reserve(64));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, pks, 128 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync((uint8_t*)c->in_a.p + 128 * n, NEG_G2_BYTES, 128, hipMemcpyHostToDevice, c->stream));   // "tuple n": the key of the signature's pair
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, agg_sig, 64, hipMemcpyHostToDevice, c->stream));
  size_t u = 0;
  rc = dedup_keys(c, (const uint8_t*)c->in_a.p, n, &u);
  if (rc) return rc;
  const bool small = c->wide_fe && n + 1 <= c->wide_fe_max;       // few pairs: from tables, one wave per pair, whatever the keys
  if (!((u * 2 <= n || small) && u + 1 <= PREP_MAX_KEYS)) return 0;
  *took = true;
  const size_t np = u + 1, n_lanes = (np + 1) / 2;
  const uint32_t n32 = (uint32_t)n, u32 = (uint32_t)u;
  // the u keys and -G2gen (key id u) become line tables on the second stream, beside the hashing
  const uint32_t last_key = n32;
  HIPCHK(c, hipMemcpyAsync((uint32_t*)c->kd_keys.p + u, &last_key, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));                                   // last_key and the staged copies are consumed
  HIPCHK(c, c->prep_ok.reserve(np)); HIPCHK(c, c->prep_raw.reserve(np * PREP_RAW_LIMBS * 4));
  HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
  LAUNCH2(c, "g2_prepare", k_g2_prepare, 2 * 256 * (size_t)nblocks(np), (const uint8_t*)c->in_a.p, (const uint32_t*)c->kd_keys.p, (uint32_t)np, (int32_t*)c->prep_raw.p, (uint8_t*)c->prep_ok.p);
  HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
  // key ids, key-sorted order
  HIPCHK(c, c->h_ws.reserve(n * 27 * 4)); HIPCHK(c, c->kd_cursor.reserve(4 * (u + 1))); HIPCHK(c, c->kd_perm.reserve(4 * n));
  HIPCHK(c, c->f_ws.reserve(n_lanes * 108 * 4)); HIPCHK(c, c->flags.reserve(np)); HIPCHK(c, c->status.reserve(np + 8)); HIPCHK(c, c->misc.reserve(64));
  HIPCHK(c, c->rlc_b.reserve(np * 18 * 4)); HIPCHK(c, c->rlc_idx.reserve(4 * np));
  uint32_t *hist = (uint32_t*)c->kd_hist.p, *cursor = (uint32_t*)c->kd_cursor.p, *perm = (uint32_t*)c->kd_perm.p, *kid = (uint32_t*)c->kd_kid.p;
  HIPCHK(c, hipMemsetAsync(hist, 0, 4 * u, c->stream));
  LAUNCH(c, "kd_propagate", k_kd_propagate, n, (const uint32_t*)c->kd_rep.p, n32, u32, kid, hist);
  { ProfScope ps_(c, "kd_scan"); hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)hist, u32, cursor); }
  HIPCHK(c, hipGetLastError());
  LAUNCH(c, "kd_scatter", k_kd_scatter, n, (const uint32_t*)kid, n32, u32, cursor, perm);          // cursor[k] is now the END of run k
  LAUNCH(c, "hash_to_g1", k_hash_to_g1, n, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, n, (const uint8_t*)c->dst.p, dl, (int32_t*)c->h_ws.p, n, (uint8_t*)nullptr, 3);
  // sums per key: the tuples in sorted order (columns perm[s] of h_ws) -> one sum per key (key_sums)
  const int32_t* pts = nullptr; size_t pts_stride = u;
  rc = key_sums(c, (const int32_t*)c->h_ws.p, nullptr, n, perm, perm, kid, hist, cursor, n, u, &pts, nullptr);
  if (rc) return rc;
  // the u + 1 pairs: (sum_k, pk_k) for k < u and (agg_sig, -G2gen) as pair u with key id u
  int32_t* h2 = (int32_t*)c->rlc_b.p; uint8_t* st = (uint8_t*)c->status.p; uint32_t* kid2 = (uint32_t*)c->rlc_idx.p;
  int* d_ok = (int*)c->misc.p;                                                  // [0] all keys valid, [1] (byte) signature valid, [4] is_one
  static const int ones[2] = {1, 1};
  HIPCHK(c, hipMemcpyAsync(d_ok, ones, 8, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "g1p_to_h", k_g1p_to_h_affine, u, pts, pts_stride, u, h2, np, st);
  HIPCHK(c, hipMemsetAsync(st + u, 1, 1, c->stream));
  LAUNCH(c, "g1_to_ws", k_g1_to_ws, 1, (const uint8_t*)c->in_b.p, h2, u, np, (uint8_t*)(d_ok + 1));
  LAUNCH(c, "iota", k_iota_u32, np, kid2, (uint32_t)np);
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
  // few pairs: the launch is the latency of one wave, so one pair per lane (no shared f^2, shorter chain); else two per lane
  const bool one_per_lane = np * 2 <= c->lanes_per_round;
  const size_t f_cnt = one_per_lane ? np : n_lanes;
  HIPCHK(c, c->f_ws.reserve(f_cnt * 108 * 4));
  if (c->wide_fe && np <= c->wide_fe_max) {           // a handful of keys: one WAVE per pair
    LAUNCH_WIDE(c, "miller_wide_1p", k_miller_wide_1p, np, (const int32_t*)h2, np, (const uint32_t*)kid2, (const int32_t*)c->prep_raw.p, (const uint8_t*)c->prep_ok.p, np,
                (int32_t*)c->f_ws.p, np, (uint8_t*)c->flags.p, (const uint8_t*)st);
  } else if (one_per_lane) {
    LAUNCH(c, "miller_hpk1p", k_miller_hpk1p, np, (const int32_t*)h2, np, (const uint32_t*)kid2, (const int32_t*)c->prep_raw.p, (const uint8_t*)c->prep_ok.p, np,
           (int32_t*)c->f_ws.p, np, (uint8_t*)c->flags.p, (const uint8_t*)st);
  } else {
    LAUNCH(c, "miller_hpk2p", k_miller_hpk2p, n_lanes, (const int32_t*)h2, np, (const uint32_t*)kid2, (const int32_t*)c->prep_raw.p, (const uint8_t*)c->prep_ok.p, np,
           (int32_t*)c->f_ws.p, n_lanes, (uint8_t*)c->flags.p, (const uint8_t*)st);
  }
  LAUNCH(c, "and_reduce", k_and_reduce, u, (const uint8_t*)c->flags.p, (const uint8_t*)c->flags.p, u, d_ok);
  int32_t* res; size_t rs;
  rc = fp12_tree(c, (int32_t*)c->f_ws.p, f_cnt, f_cnt, &res, &rs);
  if (rc) return rc;
  rc = run_final_exp(c, res, 1, rs, 3, nullptr, nullptr, nullptr, nullptr, d_ok + 4);
  if (rc) return rc;
  int h[5] = {0, 0, 0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(h, d_ok, 20, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *valid = (h[0] == 1 && (h[1] & 0xff) == 1 && h[4] == 1) ? 1 : 0;
  ++c->stat_grouped_aggregates;
  return 0;
}
int blsbn254_aggregate_verify(blsbn254_ctx* c, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, size_t n,
                              const uint8_t agg_sig[64], const uint8_t* dst, size_t dst_len, int* valid) {
  if (!c || !valid || !agg_sig || !off || (n && !pks) || (dst_len && !dst)) return BLSBN254_E_ARG;
  *valid = 0;
  if (n == 0) return 0;
  if (n + 1 > MAX_LANES) { c->last_error = "more than 2^23 - 1 pairs in one aggregate_verify call"; return BLSBN254_E_ARG; }
  uint8_t ml[384]; int ok = 0, sig_ok = 0, v = 0;
  int rc;
  bool staged = false;
  if (c->auto_prepare && (n >= 1024 || (c->wide_fe && n + 1 <= c->wide_fe_max))) {   // repeated keys (or few pairs): one pair per distinct key
    bool took = false;
    rc = aggregate_verify_grouped(c, pks, msgs, off, n, agg_sig, dst, dst_len, valid, &took);
    if (rc || took) return rc;
    staged = true;
  }
  ++c->stat_pairwise_aggregates;
  rc = aggregate_partial_impl(c, pks, msgs, off, n, dst, dst_len, agg_sig, ml, &ok, &sig_ok, staged);
  if (rc) return rc;
  rc = blsbn254_aggregate_finish(c, ml, 1, nullptr, &v);
  if (rc) return rc;
  *valid = (ok == 1 && sig_ok == 1 && v == 1) ? 1 : 0;
  return 0;
}
// n G1 points (limb-major projective, stride n) in c->h_ws -> their sum as 64 bytes
static int g1_sum_to_bytes(blsbn254_ctx* c, size_t n, uint8_t out[64]) {
  HIPCHK(c, c->f_ws2.reserve(((n + 1) / 2) * 27 * 4)); HIPCHK(c, c->out.reserve(64));
  int32_t* a = (int32_t*)c->h_ws.p; int32_t* b = (int32_t*)c->f_ws2.p;
  size_t sa = n, cnt = n;
  while (cnt > 1) {
    size_t mo = (cnt + 1) / 2;
    LAUNCH(c, "g1_add_pairs", k_g1_add_pairs, mo, (const int32_t*)a, cnt, sa, b, mo);
    std::swap(a, b); sa = mo; cnt = mo;
  }
  LAUNCH(c, "g1_to_bytes", k_g1_to_bytes, 1, (const int32_t*)a, sa, (uint8_t*)c->out.p);
  HIPCHK(c, hipMemcpyAsync(out, c->out.p, 64, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_aggregate_sigs(blsbn254_ctx* c, const uint8_t* sigs, size_t n, uint8_t out[64]) {
  if (!c || !out || (n && !sigs)) return BLSBN254_E_ARG;
  if (n == 0) { std::memset(out, 0, 64); out[63] = 1; return 0; }            // empty sum = identity (0, 1)
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->in_a.reserve(64 * n)); HIPCHK(c, c->h_ws.reserve(n * 27 * 4)); HIPCHK(c, c->status.reserve(n));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, sigs, 64 * n, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "g1_load", k_g1_load, n, (const uint8_t*)c->in_a.p, n, (int32_t*)c->h_ws.p, (uint8_t*)c->status.p);
  int bad; int rc = first_bad(c, (const uint8_t*)c->status.p, n, 1, 1, &bad);
  if (rc) return rc;
  if (bad >= 0) return BLSBN254_ERR_G1;
  return g1_sum_to_bytes(c, n, out);
}
// Threshold combine (k_threshold.hip): Lagrange coefficients over t x sqrt(t) lanes, GLV-split 4-bit-window MSM over
// 2t x 32 lanes with in-workgroup sums, one short finishing kernel.  One host synchronisation at the end.
int blsbn254_threshold_combine(blsbn254_ctx* c, const uint8_t* ids, const uint8_t* partial_sigs, size_t t, uint8_t out_sig[64]) {
  if (!c || !out_sig || (t && (!ids || !partial_sigs))) return BLSBN254_E_ARG;
  if (t == 0) { std::memset(out_sig, 0, 64); out_sig[63] = 1; return 0; }
  CHECK_LANES(c, t);
  HIPCHK(c, hipSetDevice(c->device));
  size_t S = 1;
  while (S < 64 && S * S < t) ++S;                       // ~sqrt(t) slices: t x S lanes, critical path 2 (t / S + S) products
  const size_t J = (t + S - 1) / S;
  size_t n_chunks = (2 * t + 255) / 256;
  HIPCHK(c, c->in_a.reserve(64 * t)); HIPCHK(c, c->in_b.reserve(32 * t)); HIPCHK(c, c->scalars.reserve(32 * t));
  HIPCHK(c, c->status.reserve(2 * t)); HIPCHK(c, c->flags.reserve(t)); HIPCHK(c, c->misc.reserve(64)); HIPCHK(c, c->out.reserve(64));
  HIPCHK(c, c->th_x.reserve(9 * t * 4)); HIPCHK(c, c->th_num.reserve(9 * t * S * 4)); HIPCHK(c, c->th_den.reserve(9 * t * S * 4));
  HIPCHK(c, c->th_glv.reserve(9 * t * 4)); HIPCHK(c, c->th_part.reserve(27 * 32 * n_chunks * 4)); HIPCHK(c, c->th_part2.reserve(27 * 32 * ((n_chunks + 1) / 2) * 4));
  uint8_t* st_ids = (uint8_t*)c->status.p; uint8_t* st_pts = st_ids + t; uint8_t* dup = (uint8_t*)c->flags.p;
  int* d_bad = (int*)c->misc.p;
  static const int init[2] = {0x7fffffff, 0x7fffffff};      // static: outlives the asynchronous copy
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, partial_sigs, 64 * t, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, ids, 32 * t, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_bad, init, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dup, 0, t, c->stream));
  LAUNCH(c, "fr_decode", k_fr_decode, t, (const uint8_t*)c->in_b.p, t, (int32_t*)c->th_x.p, st_ids);
  { ProfScope ps_(c, "lagrange_partial");
    hipLaunchKernelGGL(k_lagrange_partial, dim3(nblocks(t), (unsigned)S), dim3(256), 0, c->stream, (const int32_t*)c->th_x.p, t, J,
                       (int32_t*)c->th_num.p, (int32_t*)c->th_den.p, dup); }
  HIPCHK(c, hipGetLastError());
  LAUNCH(c, "lagrange_finish", k_lagrange_finish, t, (const int32_t*)c->th_num.p, (const int32_t*)c->th_den.p, t, S, (uint8_t*)c->scalars.p, (uint32_t*)c->th_glv.p);
  { ProfScope ps_(c, "msm_window");
    hipLaunchKernelGGL(k_msm_window, dim3((unsigned)n_chunks, 32), dim3(256), 0, c->stream, (const uint8_t*)c->in_a.p, (const uint32_t*)c->th_glv.p, t,
                       (int32_t*)c->th_part.p, st_pts); }
  HIPCHK(c, hipGetLastError());
  int32_t* pa = (int32_t*)c->th_part.p; int32_t* pb = (int32_t*)c->th_part2.p;
  while (n_chunks > 16) {                                  // large t only: fold the chunk axis pairwise
    const size_t no = (n_chunks + 1) / 2;
    LAUNCH(c, "msm_fold", k_msm_fold, no * 32, (const int32_t*)pa, n_chunks, pb);
    std::swap(pa, pb); n_chunks = no;
  }
  { ProfScope ps_(c, "msm_finish");
    hipLaunchKernelGGL(k_msm_finish, dim3(1), dim3(64), 0, c->stream, (const int32_t*)pa, n_chunks, (uint8_t*)c->out.p); }
  HIPCHK(c, hipGetLastError());
  // ids: decoded, non-zero (status 1) and pairwise distinct (dup 0); points: decoded
  LAUNCH(c, "status_reduce", k_status_reduce, t, (const uint8_t*)st_ids, t, (uint8_t)1, (uint8_t)1, d_bad);
  LAUNCH(c, "status_reduce", k_status_reduce, t, (const uint8_t*)dup, t, (uint8_t)1, (uint8_t)0, d_bad);
  LAUNCH(c, "status_reduce", k_status_reduce, t, (const uint8_t*)st_pts, t, (uint8_t)1, (uint8_t)1, d_bad + 1);
  int bad[2];
  HIPCHK(c, hipMemcpyAsync(bad, d_bad, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(out_sig, c->out.p, 64, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (bad[0] != 0x7fffffff) return BLSBN254_ERR_SCALAR;
  if (bad[1] != 0x7fffffff) return BLSBN254_ERR_G1;
  return 0;
}
// The Lagrange coefficients at zero alone (t x 32 bytes big-endian), for callers that combine elsewhere and for tests.
int blsbn254_lagrange_at_zero(blsbn254_ctx* c, const uint8_t* ids, size_t t, uint8_t* out) {
  if (!c || (t && (!ids || !out))) return BLSBN254_E_ARG;
  if (t == 0) return 0;
  CHECK_LANES(c, t);
  HIPCHK(c, hipSetDevice(c->device));
  size_t S = 1;
  while (S < 64 && S * S < t) ++S;
  const size_t J = (t + S - 1) / S;
  HIPCHK(c, c->in_b.reserve(32 * t)); HIPCHK(c, c->scalars.reserve(32 * t)); HIPCHK(c, c->status.reserve(t)); HIPCHK(c, c->flags.reserve(t));
  HIPCHK(c, c->misc.reserve(64));
  HIPCHK(c, c->th_x.reserve(9 * t * 4)); HIPCHK(c, c->th_num.reserve(9 * t * S * 4)); HIPCHK(c, c->th_den.reserve(9 * t * S * 4)); HIPCHK(c, c->th_glv.reserve(9 * t * 4));
  uint8_t* st_ids = (uint8_t*)c->status.p; uint8_t* dup = (uint8_t*)c->flags.p;
  int* d_bad = (int*)c->misc.p;
  const int init = 0x7fffffff;
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, ids, 32 * t, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_bad, &init, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dup, 0, t, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));             // `init` is on the stack
  LAUNCH(c, "fr_decode", k_fr_decode, t, (const uint8_t*)c->in_b.p, t, (int32_t*)c->th_x.p, st_ids);
  { ProfScope ps_(c, "lagrange_partial");
    hipLaunchKernelGGL(k_lagrange_partial, dim3(nblocks(t), (unsigned)S), dim3(256), 0, c->stream, (const int32_t*)c->th_x.p, t, J,
                       (int32_t*)c->th_num.p, (int32_t*)c->th_den.p, dup); }
  HIPCHK(c, hipGetLastError());
  LAUNCH(c, "lagrange_finish", k_lagrange_finish, t, (const int32_t*)c->th_num.p, (const int32_t*)c->th_den.p, t, S, (uint8_t*)c->scalars.p, (uint32_t*)c->th_glv.p);
  LAUNCH(c, "status_reduce", k_status_reduce, t, (const uint8_t*)st_ids, t, (uint8_t)1, (uint8_t)1, d_bad);
  LAUNCH(c, "status_reduce", k_status_reduce, t, (const uint8_t*)dup, t, (uint8_t)1, (uint8_t)0, d_bad);
  int bad;
  HIPCHK(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(out, c->scalars.p, 32 * t, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return bad != 0x7fffffff ? BLSBN254_ERR_SCALAR : 0;
}

// ---------------- field / tower primitives (debug ABI) and Gt group operations
static size_t field_op_width(int op) { return op < 0 ? 0 : op <= 8 ? 32 : (op >= 16 && op <= 21) ? 64 : (op >= 32 && op <= 35) ? 192 : (op >= 48 && op <= 56) ? 384 : 0; }
static bool field_op_binary(int op) { return op == 0 || op == 3 || op == 4 || op == 16 || op == 32 || op == 48 || op == 56; }
int blsbn254_field_op_batch(blsbn254_ctx* c, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
  const size_t w = field_op_width(op);
  if (!c || w == 0 || (n && (!a || !out || (field_op_binary(op) && !b)))) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  const bool bin = field_op_binary(op);
  HIPCHK(c, c->in_a.reserve(w * n)); HIPCHK(c, c->out.reserve(w * n)); HIPCHK(c, c->status.reserve(n));
  if (bin) HIPCHK(c, c->in_b.reserve(w * n));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, a, w * n, hipMemcpyHostToDevice, c->stream));
  if (bin) HIPCHK(c, hipMemcpyAsync(c->in_b.p, b, w * n, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "field_op", k_field_op, n, op, (const uint8_t*)c->in_a.p, bin ? (const uint8_t*)c->in_b.p : (const uint8_t*)nullptr, n,
         (uint8_t*)c->out.p, (uint8_t*)c->status.p);
  int bad; int rc = first_bad(c, (const uint8_t*)c->status.p, n, 1, 1, &bad);
  if (rc) return rc;
  if (bad >= 0) return BLSBN254_ERR_GT;
  HIPCHK(c, hipMemcpyAsync(out, c->out.p, w * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_gt_mul_batch(blsbn254_ctx* c, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
  return blsbn254_field_op_batch(c, BLSBN254_OP_FP12_MUL, a, b, n, out);
}
int blsbn254_gt_pow_batch(blsbn254_ctx* c, const uint8_t* gt, const uint8_t* scalars, size_t n, uint8_t* out) {
  if (!c || (n && (!gt || !scalars || !out))) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->in_a.reserve(384 * n)); HIPCHK(c, c->in_b.reserve(32 * n)); HIPCHK(c, c->out.reserve(384 * n)); HIPCHK(c, c->status.reserve(n));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, gt, 384 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, scalars, 32 * n, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "gt_pow", k_gt_pow, n, (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_b.p, n, (uint8_t*)c->out.p, (uint8_t*)c->status.p);
  int bad; int rc = first_bad(c, (const uint8_t*)c->status.p, n, 1, 1, &bad);
  if (rc) return rc;
  if (bad >= 0) return BLSBN254_ERR_GT;
  HIPCHK(c, hipMemcpyAsync(out, c->out.p, 384 * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ---------------- compressed codecs
static int codec_common(blsbn254_ctx* c, const uint8_t* in, size_t n, uint8_t* out, int g2, int mode) {
  if (!c || (n && (!in || !out))) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  size_t full = g2 ? 128 : 64, comp = full / 2;
  size_t isz = mode == 0 ? full : comp, osz = mode == 0 ? comp : full;
  HIPCHK(c, c->in_a.reserve(isz * n)); HIPCHK(c, c->out.reserve(osz * n)); HIPCHK(c, c->status.reserve(n));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, in, isz * n, hipMemcpyHostToDevice, c->stream));
  if (g2) { LAUNCH(c, "g2_codec", k_g2_codec, n, (const uint8_t*)c->in_a.p, n, (uint8_t*)c->out.p, (uint8_t*)c->status.p, mode); }
  else { LAUNCH(c, "g1_codec", k_g1_codec, n, (const uint8_t*)c->in_a.p, n, (uint8_t*)c->out.p, (uint8_t*)c->status.p, mode); }
  int bad; int rc = first_bad(c, (const uint8_t*)c->status.p, n, 1, 1, &bad);
  if (rc) return rc;
  if (bad >= 0) return g2 ? BLSBN254_ERR_G2 : BLSBN254_ERR_G1;
  HIPCHK(c, hipMemcpyAsync(out, c->out.p, osz * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_g1_compress_batch(blsbn254_ctx* c, const uint8_t* g1, size_t n, uint8_t* out) { return codec_common(c, g1, n, out, 0, 0); }
int blsbn254_g1_decompress_batch(blsbn254_ctx* c, const uint8_t* in, size_t n, uint8_t* g1) { return codec_common(c, in, n, g1, 0, 1); }
int blsbn254_g2_compress_batch(blsbn254_ctx* c, const uint8_t* g2, size_t n, uint8_t* out) { return codec_common(c, g2, n, out, 1, 0); }
int blsbn254_g2_decompress_batch(blsbn254_ctx* c, const uint8_t* in, size_t n, uint8_t* g2) { return codec_common(c, in, n, g2, 1, 1); }

// ---------------- signing side
int blsbn254_sign_batch(blsbn254_ctx* c, const uint8_t* sks, const uint8_t* msgs, const uint64_t* off, size_t n,
                        const uint8_t* dst, size_t dst_len, uint8_t* sigs_out) {
  if (!c || !off || (n && (!sks || !sigs_out)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t dl; int rc = stage_dst(c, dst, dst_len, &dl);
  if (rc) return rc;
  rc = stage_msgs(c, msgs, off, n);
  if (rc) return rc;
  HIPCHK(c, c->in_a.reserve(32 * n)); HIPCHK(c, c->out.reserve(64 * n)); HIPCHK(c, c->status.reserve(n));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, sks, 32 * n, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "sign", k_sign, n, (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, n, (const uint8_t*)c->dst.p, dl,
         (uint8_t*)c->out.p, (uint8_t*)c->status.p);
  int bad; rc = first_bad(c, (const uint8_t*)c->status.p, n, 1, 1, &bad);
  if (rc) return rc;
  if (bad >= 0) return BLSBN254_ERR_SCALAR;
  HIPCHK(c, hipMemcpyAsync(sigs_out, c->out.p, 64 * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemsetAsync(c->in_a.p, 0, 32 * n, c->stream));          // the staged secret keys do not outlive the call
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_sk_to_pk_batch(blsbn254_ctx* c, const uint8_t* sks, size_t n, uint8_t* pks_out) {
  if (!c || (n && (!sks || !pks_out))) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->in_a.reserve(32 * n)); HIPCHK(c, c->out.reserve(128 * n)); HIPCHK(c, c->status.reserve(n));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, sks, 32 * n, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "sk_to_pk", k_sk_to_pk, n, (const uint8_t*)c->in_a.p, n, (uint8_t*)c->out.p, (uint8_t*)c->status.p);
  int bad; int rc = first_bad(c, (const uint8_t*)c->status.p, n, 1, 1, &bad);
  if (rc) return rc;
  if (bad >= 0) return BLSBN254_ERR_SCALAR;
  HIPCHK(c, hipMemcpyAsync(pks_out, c->out.p, 128 * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemsetAsync(c->in_a.p, 0, 32 * n, c->stream));          // the staged secret keys do not outlive the call
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ---------------- key derivation, hash-to-scalar, proof of possession
int blsbn254_keygen_batch(blsbn254_ctx* c, const uint8_t* ikm, size_t ikm_len, size_t n, const uint8_t* key_info, size_t key_info_len,
                          uint8_t* sks_out) {
  if (!c || ikm_len < 32 || (n && (!ikm || !sks_out)) || (key_info_len && !key_info)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->in_a.reserve(ikm_len * n)); HIPCHK(c, c->in_c.reserve(key_info_len + 1));
  HIPCHK(c, c->out.reserve(32 * n)); HIPCHK(c, c->status.reserve(n));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, ikm, ikm_len * n, hipMemcpyHostToDevice, c->stream));
  if (key_info_len) HIPCHK(c, hipMemcpyAsync(c->in_c.p, key_info, key_info_len, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "keygen", k_keygen, n, (const uint8_t*)c->in_a.p, ikm_len, n, (const uint8_t*)c->in_c.p, key_info_len,
         (uint8_t*)c->out.p, (uint8_t*)c->status.p);
  int bad; int rc = first_bad(c, (const uint8_t*)c->status.p, n, 1, 1, &bad);
  if (rc) return rc;
  if (bad >= 0) return BLSBN254_ERR_SCALAR;
  HIPCHK(c, hipMemcpyAsync(sks_out, c->out.p, 32 * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemsetAsync(c->in_a.p, 0, ikm_len * n, c->stream));     // key material and derived keys do not outlive the call
  HIPCHK(c, hipMemsetAsync(c->out.p, 0, 32 * n, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_hash_to_scalar_batch(blsbn254_ctx* c, const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t* dst, size_t dst_len,
                                  uint8_t* out) {
  if (!c || !off || (n && (!msgs && off[n] != off[0])) || (n && !out) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t dl; int rc = stage_dst(c, dst, dst_len, &dl);
  if (rc) return rc;
  rc = stage_msgs(c, msgs, off, n);
  if (rc) return rc;
  HIPCHK(c, c->out.reserve(32 * n));
  LAUNCH(c, "hash_to_scalar", k_hash_to_scalar, n, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, n, (const uint8_t*)c->dst.p, dl,
         (uint8_t*)c->out.p);
  HIPCHK(c, hipMemcpyAsync(out, c->out.p, 32 * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_pop_prove_batch(blsbn254_ctx* c, const uint8_t* sks, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* proofs_out) {
  if (!c || (n && (!sks || !proofs_out)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t dl; int rc = stage_dst(c, dst, dst_len, &dl);
  if (rc) return rc;
  HIPCHK(c, c->in_a.reserve(32 * n)); HIPCHK(c, c->in_c.reserve(128 * n)); HIPCHK(c, c->in_off.reserve(8 * (n + 1)));
  HIPCHK(c, c->out.reserve(64 * n)); HIPCHK(c, c->status.reserve(n));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, sks, 32 * n, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "sk_to_pk", k_sk_to_pk, n, (const uint8_t*)c->in_a.p, n, (uint8_t*)c->in_c.p, (uint8_t*)c->status.p);
  LAUNCH(c, "iota_off", k_iota_off, n + 1, (uint64_t*)c->in_off.p, n, (uint64_t)128);
  LAUNCH(c, "sign", k_sign, n, (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, n, (const uint8_t*)c->dst.p, dl,
         (uint8_t*)c->out.p, (uint8_t*)c->status.p);
  int bad; rc = first_bad(c, (const uint8_t*)c->status.p, n, 1, 1, &bad);
  if (rc) return rc;
  if (bad >= 0) return BLSBN254_ERR_SCALAR;
  HIPCHK(c, hipMemcpyAsync(proofs_out, c->out.p, 64 * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemsetAsync(c->in_a.p, 0, 32 * n, c->stream));          // the staged secret keys do not outlive the call
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_pop_verify_batch(blsbn254_ctx* c, const uint8_t* pks, const uint8_t* proofs, size_t n, const uint8_t* dst, size_t dst_len,
                              uint8_t* bm) {
  if (!c || (n && (!pks || !proofs || !bm)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  size_t nb = (n + 7) / 8;
  HIPCHK(c, c->in_a.reserve(128 * n)); HIPCHK(c, c->in_b.reserve(64 * n)); HIPCHK(c, c->in_off.reserve(8 * (n + 1)));
  HIPCHK(c, c->bitmap.reserve(nb + 8));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, pks, 128 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, proofs, 64 * n, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "iota_off", k_iota_off, n + 1, (uint64_t*)c->in_off.p, n, (uint64_t)128);
  int rc = blsbn254_verify_batch_dev(c, (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_a.p, (const uint64_t*)c->in_off.p,
                                     (const uint8_t*)c->in_b.p, n, dst, dst_len, (uint8_t*)c->bitmap.p);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(bm, c->bitmap.p, nb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
