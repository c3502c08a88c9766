// host_common.h -- shared by the host-side translation units (host*.hip): the context, device buffers, launch macros and the
// internal helpers one pipeline borrows from another.  Nothing here is exported (BNH = hidden visibility); the C ABI is
// include/blsbn254.h.  There is no CPU fallback anywhere on the host side: every entry point launches kernels or fails.
#pragma once
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
  bool s2_pending = false;           // work was forked onto stream2 and the main stream has not waited for ev_join yet
  uint32_t kd_seed = 0;              // per-context random seed of the key hash table
  bool auto_prepare = true;          // verify_batch: de-duplicate the public keys and prepare each distinct key once (BLSBN254_AUTO_PREPARE=0 disables)
  uint64_t stat_prepared_chunks = 0, stat_exact_chunks = 0, stat_grouped_aggregates = 0, stat_pairwise_aggregates = 0;
  DevBuf q_ws;           // decoded public keys of the two-pairs-per-lane Miller kernel, 72 x lanes limbs
  DevBuf th_x, th_num, th_den, th_glv, th_part, th_part2;   // threshold combine: ids, partial products, GLV halves, window sums
  DevBuf fe_wide_one;    // validity bytes of the wave-per-tuple final exponentiation (mode 0)
  bool wide_fe = true;               // BLSBN254_WIDE_FE=0 disables the wave-per-tuple hard part
  size_t tri_max = 16384;            // launches of wide_fe_max < n <= tri_max tuples run three lanes per tuple (k_tri.hip); BLSBN254_TRI_MAX, 0 = off
  bool tri_miller = true, tri_fe = true;   // BLSBN254_TRI_MILLER=0 / BLSBN254_TRI_FE=0: keep one of the two on the lane-per-tuple kernels (A/B runs)
  DevBuf tri_vals;                   // the named values of the tri hard part, TRI_VALUES x 108 x n limbs
  // Asynchronous blsbn254_verify_batch_dev (host_verify.hip): calls whose pipeline was enqueued on the previous call's key count and
  // whose check (res[1]) has not been read back yet, oldest first
  struct PendingVerify {
    bool active = false; const uint8_t *d_pks = nullptr, *d_msgs = nullptr, *d_sigs = nullptr; const uint64_t* d_off = nullptr;
    size_t n = 0; uint8_t* d_bitmap = nullptr; uint8_t dst[256]; size_t dst_len = 0; hipEvent_t ev = nullptr;
  } pend[4];
  int pend_head = 0, pend_count = 0;
  uint32_t* pend_host = nullptr;     // pinned, 4 slots x (u, ok)
  DevBuf pend_dev;                   // the same on the device
  size_t u_hint = 0;                 // distinct keys of the last chunk that took the prepared path (0: none yet)
  size_t u_max_seen = 0;             // the largest key count a prepared chunk had on this context: the capacity never drops below it (a caller
                                     // alternating between a small and a large key set would otherwise re-run every large batch)
  bool async_verify = true;          // BLSBN254_ASYNC_VERIFY=0: every call reads the key count back before it enqueues the pipeline
  uint64_t stat_async_chunks = 0, stat_async_reruns = 0;
  bool quad_prep = true;             // per-key preparation with four lanes per key while that fits one round of waves (BLSBN254_QUAD_PREP=0: off)
  bool split_easy = true;            // BLSBN254_SPLIT_EASY=0: the one-launch easy part at every size
  size_t wide_fe_max = 2048;         // ... used for launches of at most this many tuples (BLSBN254_WIDE_FE_MAX): two rounds of a wave per tuple
                                     // cost what the three-lanes-per-tuple kernels cost for anything up to 16384 (r03: 4096 wide = 7.8 ms, 4100 on quads = 5.6 ms)
  DevBuf gs_ws[3], gs_ok[3], gs_start, gs_len, gs_pk;   // segmented G2 sums (host_groupops.hip): items / chunk sums (ping-pong), flags, chunk descriptors, the sums' encodings
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

// four lanes per element (tri.h): 256-lane workgroups of 64 elements
#define LAUNCH_TRI(ctx, name, kernel, n, ...) do { ProfScope ps_(ctx, name); \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(((n) + 63) / 64)), dim3(256), 0, (ctx)->stream, __VA_ARGS__); } while (0); HIPCHK(ctx, hipGetLastError())
static const size_t TRI_VALUE_LIMBS = 20 * 108;      // tri.h TRI_VALUES x 108
// G2Prepared::from for u keys (decode, on-curve, psi subgroup test, 88 line triples): four lanes per key (k_keyprep_quad.hip, half
// the latency) while the 8 u lanes fit one round of waves, else one lane per key (k_keyprep.hip).  L = LAUNCH or LAUNCH2.
#define LAUNCH_G2_PREPARE(ctx, L, pks, keys, u, raw, ok, d_u) do { \
    if ((ctx)->quad_prep && 8 * (size_t)(u) <= (ctx)->lanes_per_round) { L(ctx, "g2_prepare", k_g2_prepare_quad, 2 * 256 * (size_t)nblocks(4 * (size_t)(u)), pks, keys, (uint32_t)(u), raw, ok, d_u); } \
    else { L(ctx, "g2_prepare", k_g2_prepare, 2 * 256 * (size_t)nblocks(u), pks, keys, (uint32_t)(u), raw, ok, d_u); } } while (0)

// one workgroup of 128 lanes (two waves) per element: the wave-per-tuple kernels (wide.h)
#define LAUNCH_WIDE(ctx, name, kernel, n, ...) do { ProfScope ps_(ctx, name); \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(n)), dim3(128), 0, (ctx)->stream, __VA_ARGS__); } while (0); HIPCHK(ctx, hipGetLastError())      /* wide.h: WIDE_LANES */

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

// Work forked onto stream2 must never outlive a failing call: a function that forks holds one of these, and an early
// error return (before the main stream has waited for ev_join) then waits for stream2 on the way out, so nothing is
// still writing prep_raw / prep_table when the caller sees the error code.
struct Stream2Guard {
  blsbn254_ctx* c;
  explicit Stream2Guard(blsbn254_ctx* c_) : c(c_) {}
  ~Stream2Guard() { if (c->s2_pending) { (void)hipStreamSynchronize(c->stream2); c->s2_pending = false; } }
};
static inline hipError_t fork_stream2(blsbn254_ctx* c) {
  hipError_t e = hipEventRecord(c->ev_fork, c->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->stream2, c->ev_fork, 0);
  if (e == hipSuccess) c->s2_pending = true;
  return e;
}
static inline hipError_t join_stream2(blsbn254_ctx* c) {
  hipError_t e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
  if (e == hipSuccess) c->s2_pending = false;
  return e;
}

#define BNH __attribute__((visibility("hidden")))
extern "C" {
// every entry point: select the device, then settle what the asynchronous verify path left open (read back its checks, re-run a
// batch whose optimistic choice did not hold) -- results of earlier calls are final before this call touches the context
BNH int resolve_pending(blsbn254_ctx* c, bool blocking);   // host_verify.hip
BNH int blsbn254_internal_verify_batch_dev_sync(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                                                const uint8_t* d_sigs, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* d_bitmap);   // host_verify.hip
#define ENTER(c) do { HIPCHK(c, hipSetDevice((c)->device)); if ((c)->pend_count) { int rc_ = resolve_pending(c, true); if (rc_) return rc_; } } while (0)
extern BNH const uint8_t NEG_G2_BYTES[128];     // -G2gen = (x, p - y) of the generator fp2.rs:305-333, as bytes (host.hip)

// The kernels address limb-major workspaces through a buffer descriptor with a 32-bit scalar byte offset
// (limb index x stride x 4, tower.h `Ws`): a launch may span at most MAX_LANES tuples (107 x 8 Mi x 4 B < 4 GiB).
// Independent-element entry points are processed in chunks of ctx->chunk (4 Mi); the product-type ones reject more.
static const size_t MAX_LANES = (size_t)1 << 23;
#define CHECK_LANES(c, n) do { if ((n) > MAX_LANES) { (c)->last_error = "more than 2^23 elements in one product-type call"; return BLSBN254_E_ARG; } } while (0)
// prepared-key path.  Limits: key ids and table offsets are 32-bit (88 x 54 x 4 B per key): at most PREP_MAX_KEYS keys.
static const size_t PREP_MAX_KEYS = (size_t)1 << 16;
static const size_t PREP_RAW_LIMBS = (size_t)BN_NEG_G2_LINES * 54;       // a key's 88 line triples
static const size_t PREP_KEY_LIMBS = (size_t)BN_NEG_G2_LINES * 162;      // a key's 88 expanded line pairs (key line x -G2gen line)
struct blsbn254_g2prepared { blsbn254_ctx* ctx; size_t u; DevBuf table, raw, ok; };   // pair tables (verify), raw line triples (multi_miller_loop), validity

// ---- internal helpers shared between the units (defined in the unit named on the right)
BNH int stage_dst(blsbn254_ctx* c, const uint8_t* dst, size_t dst_len, uint32_t* out_len);   // host.hip
BNH int check_offsets(const uint64_t* off, size_t n);   // host.hip
BNH int first_bad(blsbn254_ctx* c, const uint8_t* d_status, size_t n, uint8_t mask, uint8_t val, int* out);   // host.hip
BNH int read_status(blsbn254_ctx* c, const uint8_t* d_status, int idx, uint8_t* st);   // host.hip
BNH int run_final_exp(blsbn254_ctx* c, int32_t* f, size_t n, size_t stride, int mode, const uint8_t* flags, const uint8_t* sub_ok,
                         uint8_t* d_bitmap, uint8_t* d_gt, int* d_is_one);   // host.hip
BNH int miller_to_ws(blsbn254_ctx* c, const uint8_t* d_g1, const uint8_t* d_g2, size_t n);   // host.hip
BNH int decode_status_rc(blsbn254_ctx* c, const uint8_t* d_status, size_t n);   // host.hip
BNH int product_tree(blsbn254_ctx* c, size_t n, const int32_t** result, size_t* rs);   // host.hip
BNH int stage_msgs(blsbn254_ctx* c, const uint8_t* msgs, const uint64_t* off, size_t n);   // host.hip
BNH int prepare_keys_async(blsbn254_ctx* c, const uint8_t* d_pks, const uint32_t* d_keys, size_t u, int32_t* table, uint8_t* key_ok, const uint32_t* d_u);   // host_verify.hip
BNH int verify_prepared_dev(blsbn254_ctx* c, const int32_t* table, const uint8_t* key_ok, size_t u, const uint32_t* d_kid, bool hist_done,
                               const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sigs, size_t n, uint32_t dl, uint8_t* d_bitmap, bool join);   // host_verify.hip
BNH int dedup_keys(blsbn254_ctx* c, const uint8_t* d_pks, size_t n, size_t* u_out);   // host_verify.hip
BNH int verify_exact_dev(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                            const uint8_t* d_sigs, size_t n, uint32_t dl, uint8_t* d_bitmap);   // host_verify.hip
BNH int verify_chunk_dev(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                            const uint8_t* d_sigs, size_t n, uint32_t dl, uint8_t* d_bitmap);   // host_verify.hip
BNH int key_sums(blsbn254_ctx* c, const int32_t* pts, const int32_t* pts2, size_t pts_stride, const uint32_t* mark_perm, const uint32_t* pt_perm,
                    const uint32_t* kid, const uint32_t* hist, const uint32_t* run_end, size_t items, size_t u, const int32_t** out, const int32_t** out2);   // host_rlc.hip
BNH int draw_seed(blsbn254_ctx* c, uint8_t out[32]);   // host_rlc.hip
BNH int prepared_round(blsbn254_ctx* c, const uint32_t* perm, const uint32_t* kid, const uint8_t* sigs, const int32_t* h_ws, size_t h_stride,
                          size_t cnt, uint8_t* d_isone);   // host_rlc.hip
BNH int fp12_tree(blsbn254_ctx* c, int32_t* a, size_t cnt, size_t sa, int32_t** res, size_t* rs);   // host_aggregate.hip
BNH int g1_sum_to_bytes(blsbn254_ctx* c, size_t n, uint8_t out[64]);   // host_aggregate.hip
}  // extern "C"
