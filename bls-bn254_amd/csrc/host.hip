// host.hip -- C-ABI host side (include/blsbn254.h), core unit: contexts, profiling, the VALU probe, staging helpers, the final-
// exponentiation pipeline, and the primitive entry points (pairing, Miller loop, hash to curve, point checks).
// The other pipelines: host_verify.hip (verify + G2Prepared), host_rlc.hip (RLC batch verification), host_aggregate.hip
// (aggregate verify, sums, threshold), host_groupops.hip (group operations, FastAggregateVerify), host_misc.hip.
#include "host_common.h"

extern "C" {

// -G2gen = (x, p - y) of the generator fp2.rs:305-333, as bytes
const uint8_t NEG_G2_BYTES[128] = {
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
  if (const char* e = std::getenv("BLSBN254_ASYNC_VERIFY")) c->async_verify = std::atoi(e) != 0;
  if (const char* e = std::getenv("BLSBN254_QUAD_PREP")) c->quad_prep = std::atoi(e) != 0;
  if (const char* e = std::getenv("BLSBN254_SPLIT_EASY")) c->split_easy = std::atoi(e) != 0;
  c->tri_max = c->lanes_per_round / 4;        // four lanes per tuple: one round of waves
  if (const char* e = std::getenv("BLSBN254_TRI_MILLER")) c->tri_miller = std::atoi(e) != 0;
  if (const char* e = std::getenv("BLSBN254_TRI_FE")) c->tri_fe = std::atoi(e) != 0;
  if (const char* e = std::getenv("BLSBN254_TRI_MAX")) { long v = std::atol(e); if (v >= 0 && v <= (1 << 20)) c->tri_max = (size_t)v; }
  if (const char* e = std::getenv("BLSBN254_WIDE_FE_MAX")) { long v = std::atol(e); if (v >= 0 && v <= (1 << 20)) c->wide_fe_max = (size_t)v; }
  *out = c;
  return 0;
}
void blsbn254_ctx_destroy(blsbn254_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& pv : c->pend) if (pv.ev) (void)hipEventDestroy(pv.ev);       // pending checks are dropped: the caller did not ask for their results
  if (c->pend_host) (void)hipHostFree(c->pend_host);
  c->pend_dev.release();
  for (auto& kv : c->prof) for (auto& pr : kv.second.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  DevBuf* bufs[] = {&c->in_a, &c->in_b, &c->in_c, &c->in_off, &c->dst, &c->h_ws, &c->f_ws, &c->f_ws2, &c->flags, &c->sub_ok, &c->status, &c->status_all, &c->bitmap, &c->out, &c->scalars, &c->misc};
  for (DevBuf* b : bufs) b->release();
  for (DevBuf& b : c->fe) b.release();
  c->fe_slots.release(); c->fe_wide_one.release();
  for (DevBuf& b : c->gs_ws) b.release();
  for (DevBuf& b : c->gs_ok) b.release();
  c->gs_start.release(); c->gs_len.release(); c->gs_pk.release(); c->tri_vals.release();
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
int blsbn254_ctx_synchronize(blsbn254_ctx* c) {
  if (!c) return BLSBN254_E_ARG;
  ENTER(c);                                           // settles the asynchronously enqueued verify calls (re-running one whose assumption failed)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
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
  ENTER(c);
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
int stage_dst(blsbn254_ctx* c, const uint8_t* dst, size_t dst_len, uint32_t* out_len) {
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
int check_offsets(const uint64_t* off, size_t n) {
  for (size_t i = 0; i < n; ++i) if (off[i + 1] < off[i]) return BLSBN254_E_ARG;
  return 0;
}
// first index whose status differs from the wanted value, or -1
int first_bad(blsbn254_ctx* c, const uint8_t* d_status, size_t n, uint8_t mask, uint8_t val, int* out) {
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
int read_status(blsbn254_ctx* c, const uint8_t* d_status, int idx, uint8_t* st) {
  HIPCHK(c, hipMemcpyAsync(st, d_status + idx, 1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}


// Final exponentiation of the n Fp12 values at f (limb-major, `stride`), in place for the easy part.
//   mode 0: verify bitmap (flags / sub_ok / d_bitmap)   mode 1: Gt bytes   mode 3: single is_one flag (n == 1)
int run_final_exp(blsbn254_ctx* c, int32_t* f, size_t n, size_t stride, int mode, const uint8_t* flags, const uint8_t* sub_ok,
                         uint8_t* d_bitmap, uint8_t* d_gt, int* d_is_one) {
  CHECK_LANES(c, stride);
  for (DevBuf& b : c->fe) HIPCHK(c, b.reserve(stride * 108 * 4));
  HIPCHK(c, c->fe_slots.reserve(stride * 108 * 4 * 10));
  int32_t* S = (int32_t*)c->fe_slots.p;
  int32_t *X = (int32_t*)c->fe[0].p, *A = (int32_t*)c->fe[1].p, *B = (int32_t*)c->fe[2].p, *C = (int32_t*)c->fe[3].p,
          *B2 = (int32_t*)c->fe[4].p, *D = (int32_t*)c->fe[5].p;
  // t = f^((p^6-1)(p^2+1)), in place.  From a round of waves up, the ONE Fp inversion of the easy part is shared by four tuples
  // per lane (head / inv4 / tail, k_fe_easy.hip; the pieces wait in the phase buffers A and B, which the hard part only fills later)
  if (n >= c->lanes_per_round / 2 && c->split_easy) {
    LAUNCH(c, "fe_easy_head", k_fe_easy_head, n, (const int32_t*)f, n, stride, A, B);
    LAUNCH(c, "fe_inv4", k_fe_inv4, (n + 3) / 4, B, n, stride);
    LAUNCH(c, "fe_easy_tail", k_fe_easy_tail, n, (const int32_t*)f, f, n, stride, (const int32_t*)A, (const int32_t*)B);
  } else {
    LAUNCH(c, "fe_easy", k_fe_easy, n, (const int32_t*)f, f, n, stride);
  }
  // Few tuples: the lane-per-tuple kernels below would be the latency of one lane's chain (4.5 ms for any n <= 65536); the hard
  // part runs with one WAVE per tuple instead (k_fe_wide.hip): ~0.6 ms per round of CUs x 16 tuples.  Same values.
  if (c->wide_fe && n <= c->wide_fe_max) {
    uint8_t* one = nullptr;
    if (mode == 0) { HIPCHK(c, c->fe_wide_one.reserve(n)); one = (uint8_t*)c->fe_wide_one.p; }
    LAUNCH_WIDE(c, "fe_hard_wide", k_fe_hard_wide, n, (const int32_t*)f, n, stride, flags, sub_ok, one, d_gt, d_is_one, mode);
    if (mode == 0) { LAUNCH(c, "pack_bitmap", k_pack_bitmap, n, (const uint8_t*)one, n, d_bitmap); }
    return 0;
  }
  // Between the wave-per-tuple limit and a quarter of a round of lanes: three lanes per tuple (k_tri.hip) -- the lane-per-tuple
  // kernels below would cost one lane's whole chain (5.4 ms) however few tuples there are.  Same values.
  if (n <= c->tri_max && c->tri_fe) {
    HIPCHK(c, c->tri_vals.reserve(n * TRI_VALUE_LIMBS * 4));
    uint8_t* one = nullptr;
    if (mode == 0) { HIPCHK(c, c->fe_wide_one.reserve(n)); one = (uint8_t*)c->fe_wide_one.p; }
    LAUNCH_TRI(c, "fe_tri_hard", k_fe_tri_hard, n, (const int32_t*)f, n, stride, (int32_t*)c->tri_vals.p, flags, sub_ok, one, d_gt, d_is_one, mode);
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
int miller_to_ws(blsbn254_ctx* c, const uint8_t* d_g1, const uint8_t* d_g2, size_t n) {
  CHECK_LANES(c, n);
  HIPCHK(c, c->f_ws.reserve(n * 108 * 4));
  HIPCHK(c, c->status.reserve(n));
  if (c->wide_fe && n <= c->wide_fe_max / 2) {        // few pairs: one wave per pair (lane 0 runs the G2 point arithmetic); the serial part makes the chain ~2 x a prepared one
    LAUNCH_WIDE(c, "miller_wide_1", k_miller_wide_1, n, d_g1, d_g2, n, (int32_t*)c->f_ws.p, n, (uint8_t*)c->status.p);
  } else if (n <= c->tri_max && c->tri_miller) {      // mid-size: a quad of lanes per pair (k_tri.hip: line steps four lanes per point, f three lanes per value)
    LAUNCH_TRI(c, "miller_tri_1", k_miller_tri_1, n, d_g1, d_g2, n, (int32_t*)c->f_ws.p, n, (uint8_t*)c->status.p);
  } else {
    LAUNCH(c, "miller_1", k_miller_1, n, d_g1, d_g2, n, (int32_t*)c->f_ws.p, n, (uint8_t*)c->status.p);
  }
  return 0;
}
int decode_status_rc(blsbn254_ctx* c, const uint8_t* d_status, size_t n) {
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
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
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
int product_tree(blsbn254_ctx* c, size_t n, const int32_t** result, size_t* rs) {
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
  ENTER(c);
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
  ENTER(c);
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
int stage_msgs(blsbn254_ctx* c, const uint8_t* msgs, const uint64_t* off, size_t n) {
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
  ENTER(c);
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
  ENTER(c);
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


}  // extern "C"
