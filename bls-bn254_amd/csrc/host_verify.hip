// host_verify.hip -- batched BLS verify: key de-duplication, per-key line tables (G2Prepared), the table-only and the exact
// per-tuple pipelines, and the explicit G2Prepared API.  Host side of include/blsbn254.h; see host_common.h.
#include "host_common.h"

extern "C" {

// ---------------- verify
// Workspace is ~7.4 KB per tuple (H, f, six final-exponentiation phase buffers, ten chain slots); batches
// larger than ctx->chunk (4 Mi) tuples are processed chunk by chunk so that any n fits the 288 GB of HBM.

// G2Prepared::from for u keys on the second stream (after ev_fork), ev_join recorded behind it.
// keys == nullptr: key k = pks[128 k]; else key k = the public key of tuple keys[k].
// d_u (optional): the key count on the device when u is only a capacity (the asynchronous path)
int prepare_keys_async(blsbn254_ctx* c, const uint8_t* d_pks, const uint32_t* d_keys, size_t u, int32_t* table, uint8_t* key_ok, const uint32_t* d_u) {
  HIPCHK(c, c->prep_raw.reserve(u * PREP_RAW_LIMBS * 4));
  HIPCHK(c, fork_stream2(c));
  LAUNCH_G2_PREPARE(c, LAUNCH2, d_pks, d_keys, u, (int32_t*)c->prep_raw.p, key_ok, d_u);
  LAUNCH2(c, "g2_expand", k_g2_expand, u * (size_t)BN_NEG_G2_LINES, (const int32_t*)c->prep_raw.p, (uint32_t)u, table, d_u);
  HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
  return 0;
}
// Verify n tuples whose keys are given by index into a prepared table (d_kid[i] < u), everything device-resident.
// The caller has put the preparation of the table on stream2 (ev_join) or the table is final (join = false).
int verify_prepared_dev(blsbn254_ctx* c, const int32_t* table, const uint8_t* key_ok, size_t u, const uint32_t* d_kid, bool hist_done,
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
  if (join) HIPCHK(c, join_stream2(c));
  if (c->wide_fe && n <= c->wide_fe_max) {            // few tuples: one wave per tuple (k_miller_wide.hip), same values
    LAUNCH_WIDE(c, "miller_wide_prepared", k_miller_wide_prepared, n, (const uint32_t*)perm, d_kid, d_sigs, (const int32_t*)c->h_ws.p, n, table, key_ok, n,
                (int32_t*)c->f_ws.p, (uint8_t*)c->flags.p);
  } else if (n <= c->tri_max && c->tri_miller) {       // mid-size launches: three lanes per tuple (k_tri.hip), same values
    LAUNCH_TRI(c, "miller_tri_prepared", k_miller_tri_prepared, n, (const uint32_t*)perm, d_kid, d_sigs, (const int32_t*)c->h_ws.p, n, table, key_ok, n,
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
// De-duplicate the public keys of a chunk: kd_rep / kd_kid / kd_keys and the count (kd_cnt) are filled on the device.
static int dedup_enqueue(blsbn254_ctx* c, const uint8_t* d_pks, size_t n) {
  size_t m = 1;
  while (m < 2 * n) m <<= 1;
  // n + 1 entries: aggregate_verify_grouped appends the key of the signature's pair (-G2gen) as entry u, and u can be n
  HIPCHK(c, c->kd_slots.reserve(4 * m)); HIPCHK(c, c->kd_rep.reserve(4 * (n + 1))); HIPCHK(c, c->kd_kid.reserve(4 * (n + 1))); HIPCHK(c, c->kd_keys.reserve(4 * (n + 1)));
  HIPCHK(c, c->kd_hist.reserve(4 * (n + 2))); HIPCHK(c, c->kd_cnt.reserve(64));
  HIPCHK(c, hipMemsetAsync(c->kd_slots.p, 0xff, 4 * m, c->stream));
  HIPCHK(c, hipMemsetAsync(c->kd_cnt.p, 0, 4, c->stream));
  LAUNCH(c, "kd_insert", k_kd_insert, n, d_pks, (uint32_t)n, (uint32_t*)c->kd_slots.p, (uint32_t)(m - 1), c->kd_seed, (uint32_t*)c->kd_rep.p);
  LAUNCH(c, "kd_assign", k_kd_assign, n, (const uint32_t*)c->kd_rep.p, (uint32_t)n, (uint32_t*)c->kd_kid.p, (uint32_t*)c->kd_cnt.p, (uint32_t*)c->kd_keys.p);
  return 0;
}
// ... and *u_out = the number of distinct keys, read back (one 4-byte copy and a stream synchronisation)
int dedup_keys(blsbn254_ctx* c, const uint8_t* d_pks, size_t n, size_t* u_out) {
  int rc = dedup_enqueue(c, d_pks, n);
  if (rc) return rc;
  uint32_t u = 0;
  HIPCHK(c, hipMemcpyAsync(&u, c->kd_cnt.p, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *u_out = u;
  return 0;
}
// the exact per-tuple path: every tuple validates its own key and runs the two-pair Miller loop with a variable-Q pair
int verify_exact_dev(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
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
int verify_chunk_dev(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                            const uint8_t* d_sigs, size_t n, uint32_t dl, uint8_t* d_bitmap) {
  Stream2Guard s2_guard(c);
  // Few distinct keys (a validator set signing many messages): every distinct key is validated and turned into its line
  // table ONCE (G2Prepared), beside hash-to-G1, and the tuples run the table-only Miller loop in key-sorted order.
  // Same bitmap as the exact per-tuple path below, which batches of mostly distinct keys keep taking.
  // Small chunks (at most wide_fe_max tuples) take it whatever their keys: with tables, the Miller loop and the final
  // exponentiation can run one WAVE per tuple (k_miller_wide.hip, k_fe_wide.hip) instead of at the latency of one lane.
  // ... and mid-size chunks (up to tri_max tuples) likewise: three lanes per tuple (k_tri.hip) need the tables too, and a launch of
  // that size is bound by latency, not by the table work (16 384 distinct keys: ~2 ms of preparation against 4 ms saved).
  const bool small = (c->wide_fe && n <= c->wide_fe_max) || (c->tri_miller && c->tri_fe && n <= c->tri_max);
  if (c->auto_prepare && (n >= 1024 || small)) {
    size_t u = 0;
    int rc = dedup_keys(c, d_pks, n, &u);
    if (rc) return rc;
    if ((u * 2 <= n || small) && u <= PREP_MAX_KEYS) {
      HIPCHK(c, c->prep_table.reserve(u * PREP_KEY_LIMBS * 4)); HIPCHK(c, c->prep_ok.reserve(u));
      rc = prepare_keys_async(c, d_pks, (const uint32_t*)c->kd_keys.p, u, (int32_t*)c->prep_table.p, (uint8_t*)c->prep_ok.p, nullptr);
      if (rc) return rc;
      HIPCHK(c, hipMemsetAsync(c->kd_hist.p, 0, 4 * u, c->stream));
      LAUNCH(c, "kd_propagate", k_kd_propagate, n, (const uint32_t*)c->kd_rep.p, (uint32_t)n, (uint32_t)u, (uint32_t*)c->kd_kid.p, (uint32_t*)c->kd_hist.p);
      ++c->stat_prepared_chunks;
      c->u_hint = u ? u : 1;
      if (u > c->u_max_seen) c->u_max_seen = u;
      return verify_prepared_dev(c, (const int32_t*)c->prep_table.p, (const uint8_t*)c->prep_ok.p, u, (const uint32_t*)c->kd_kid.p, true,
                                 d_msgs, d_off, d_sigs, n, dl, d_bitmap, true);
    }
  }
  c->u_hint = 0;                                      // keys did not repeat: the next call counts them first again
  return verify_exact_dev(c, d_pks, d_msgs, d_off, d_sigs, n, dl, d_bitmap);
}
// ---- asynchronous variant: no read-back before the pipeline is enqueued.
// A steady caller's batches repeat their key set: after a chunk has taken the prepared-key path, the NEXT chunk is enqueued on the
// assumption that it does too, with tables reserved for twice the last key count -- de-duplication, a one-lane decision kernel
// (key count <= capacity, and at most half of the keys distinct unless the chunk is small), the per-key preparation bounded by
// the DEVICE-side count, the table-only pipeline -- and the call returns.  Count and decision come back through a pinned
// buffer behind an event; they are read by the next entry point that touches the context (ENTER) or by
// blsbn254_ctx_synchronize.  If the assumption did not hold (a new key set, more keys than reserved), that batch's bitmap is
// not valid yet: it is re-run on the counting path right there, before the caller can see it -- key ids beyond the capacity were
// clamped by k_kd_propagate, so the discarded run indexed nothing out of bounds.  The caller's device buffers must stay untouched
// until blsbn254_ctx_synchronize returns, as the header has always required of the device entry points.
static int verify_chunk_async(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off, const uint8_t* d_sigs, size_t n,
                              uint32_t dl, const uint8_t* dst, size_t dst_len, uint8_t* d_bitmap, bool* took) {
  *took = false;
  const bool small = (c->wide_fe && n <= c->wide_fe_max) || (c->tri_miller && c->tri_fe && n <= c->tri_max);
  if (!c->async_verify || !c->auto_prepare || c->u_hint == 0 || !(n >= 1024 || small) || dst_len > 255) return 0;
  size_t cap = 2 * c->u_hint < 1024 ? 1024 : 2 * c->u_hint;
  if (cap < c->u_max_seen) cap = c->u_max_seen;
  if (cap > PREP_MAX_KEYS) cap = PREP_MAX_KEYS;
  if (cap > n) cap = n;
  if (!small && cap > n / 2) cap = n / 2;
  if (cap == 0) return 0;
  if (c->pend_count == 4) { int rc = resolve_pending(c, true); if (rc) return rc; }
  if (!c->pend_host) {
    HIPCHK(c, hipHostMalloc((void**)&c->pend_host, 4 * 2 * sizeof(uint32_t), hipHostMallocDefault));
    HIPCHK(c, c->pend_dev.reserve(4 * 2 * sizeof(uint32_t)));
    for (auto& pv : c->pend) HIPCHK(c, hipEventCreateWithFlags(&pv.ev, hipEventDisableTiming));
  }
  Stream2Guard s2_guard(c);
  const int slot = (c->pend_head + c->pend_count) & 3;
  uint32_t* d_res = (uint32_t*)c->pend_dev.p + 2 * slot;
  int rc = dedup_enqueue(c, d_pks, n);
  if (rc) return rc;
  hipLaunchKernelGGL(k_kd_decide, dim3(1), dim3(1), 0, c->stream, (const uint32_t*)c->kd_cnt.p, (uint32_t)n, (uint32_t)cap, small ? 1 : 0, d_res);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, c->prep_table.reserve(cap * PREP_KEY_LIMBS * 4)); HIPCHK(c, c->prep_ok.reserve(cap));
  rc = prepare_keys_async(c, d_pks, (const uint32_t*)c->kd_keys.p, cap, (int32_t*)c->prep_table.p, (uint8_t*)c->prep_ok.p, d_res);
  if (rc) return rc;
  HIPCHK(c, hipMemsetAsync(c->kd_hist.p, 0, 4 * cap, c->stream));
  LAUNCH(c, "kd_propagate", k_kd_propagate, n, (const uint32_t*)c->kd_rep.p, (uint32_t)n, (uint32_t)cap, (uint32_t*)c->kd_kid.p, (uint32_t*)c->kd_hist.p);
  rc = verify_prepared_dev(c, (const int32_t*)c->prep_table.p, (const uint8_t*)c->prep_ok.p, cap, (const uint32_t*)c->kd_kid.p, true,
                           d_msgs, d_off, d_sigs, n, dl, d_bitmap, true);
  if (rc) return rc;
  blsbn254_ctx::PendingVerify& pv = c->pend[slot];
  HIPCHK(c, hipMemcpyAsync(c->pend_host + 2 * slot, d_res, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipEventRecord(pv.ev, c->stream));
  pv.active = true; pv.d_pks = d_pks; pv.d_msgs = d_msgs; pv.d_off = d_off; pv.d_sigs = d_sigs; pv.n = n; pv.d_bitmap = d_bitmap;
  pv.dst_len = dst_len;
  if (dst_len) std::memcpy(pv.dst, dst, dst_len);
  ++c->pend_count;
  ++c->stat_async_chunks;
  *took = true;
  return 0;
}
// Read back the checks of the asynchronously enqueued chunks, oldest first (blocking: all of them; else only those whose event has
// fired); a chunk whose optimistic choice did not hold is re-run on the counting path.
int resolve_pending(blsbn254_ctx* c, bool blocking) {
  while (c->pend_count) {
    blsbn254_ctx::PendingVerify& pv = c->pend[c->pend_head];
    if (!blocking && hipEventQuery(pv.ev) != hipSuccess) { (void)hipGetLastError(); break; }
    HIPCHK(c, hipEventSynchronize(pv.ev));
    const uint32_t u = c->pend_host[2 * c->pend_head], ok = c->pend_host[2 * c->pend_head + 1];
    const blsbn254_ctx::PendingVerify done = pv;
    pv.active = false;
    c->pend_head = (c->pend_head + 1) & 3; --c->pend_count;
    if (ok) { c->u_hint = u ? u : 1; if (u > c->u_max_seen) c->u_max_seen = u; ++c->stat_prepared_chunks; continue; }
    ++c->stat_async_reruns;
    c->u_hint = 0;
    uint32_t dl; int rc = stage_dst(c, done.dst, done.dst_len, &dl);
    if (rc) return rc;
    rc = verify_chunk_dev(c, done.d_pks, done.d_msgs, done.d_off, done.d_sigs, done.n, dl, done.d_bitmap);
    if (rc) return rc;
  }
  return 0;
}
static int verify_batch_dev_impl(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                                 const uint8_t* d_sigs, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* d_bitmap, bool allow_async) {
  if (!c || (n && (!d_pks || !d_off || !d_sigs || !d_bitmap)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = resolve_pending(c, !allow_async);           // settle what has already come back (everything, for a caller that reads the result right away)
  if (rc) return rc;
  uint32_t dl; rc = stage_dst(c, dst, dst_len, &dl);   // (a new tag waits for the kernels that still read the old one)
  if (rc) return rc;
  if (allow_async && n <= c->chunk) {
    bool took = false;
    rc = verify_chunk_async(c, d_pks, d_msgs, d_off, d_sigs, n, dl, dst, dst_len, d_bitmap, &took);
    if (rc || took) return rc;
  }
  if (c->pend_count) { rc = resolve_pending(c, true); if (rc) return rc; }     // the counting path reads back: settle the queue first
  for (size_t lo = 0; lo < n; lo += c->chunk) {        // chunk starts are multiples of 8: bitmap bytes do not straddle
    size_t m = n - lo < c->chunk ? n - lo : c->chunk;
    rc = verify_chunk_dev(c, d_pks + 128 * lo, d_msgs, d_off + lo, d_sigs + 64 * lo, m, dl, d_bitmap + lo / 8);
    if (rc) return rc;
  }
  return 0;
}
int blsbn254_verify_batch_dev(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                              const uint8_t* d_sigs, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* d_bitmap) {
  return verify_batch_dev_impl(c, d_pks, d_msgs, d_off, d_sigs, n, dst, dst_len, d_bitmap, true);
}
// for callers inside the library that consume the bitmap on the stream right behind the call (host-pointer entry points, the
// multi-device handle): the counting path, nothing left pending
int blsbn254_internal_verify_batch_dev_sync(blsbn254_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                                            const uint8_t* d_sigs, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* d_bitmap) {
  return verify_batch_dev_impl(c, d_pks, d_msgs, d_off, d_sigs, n, dst, dst_len, d_bitmap, false);
}
int blsbn254_async_stats(blsbn254_ctx* c, uint64_t out[2]) {
  if (!c || !out) return BLSBN254_E_ARG;
  out[0] = c->stat_async_chunks; out[1] = c->stat_async_reruns;
  return 0;
}
int blsbn254_set_async_verify(blsbn254_ctx* c, int on) {
  if (!c) return BLSBN254_E_ARG;
  ENTER(c);
  c->async_verify = on != 0;
  return 0;
}
// ---------------- G2Prepared: explicit API
int blsbn254_g2_prepare_batch(blsbn254_ctx* c, const uint8_t* pks, size_t u, blsbn254_g2prepared** out) {
  if (!c || !out || (u && !pks)) return BLSBN254_E_ARG;
  *out = nullptr;
  if (u + 1 > PREP_MAX_KEYS) { c->last_error = "more than 65535 keys in one prepared table"; return BLSBN254_E_ARG; }
  ENTER(c);
  // owned until handed to the caller: every failure path below releases the three device buffers and the object
  struct Owner {
    blsbn254_g2prepared* p;
    ~Owner() { if (p) { p->table.release(); p->raw.release(); p->ok.release(); delete p; } }
  } own{new blsbn254_g2prepared()};
  blsbn254_g2prepared* p = own.p;
  p->ctx = c; p->u = u;
  // entry u (one past the caller's keys) is -G2gen: the second member of the aggregate signature's pair
  const size_t u1 = u + 1;
  HIPCHK(c, p->table.reserve(u1 * PREP_KEY_LIMBS * 4)); HIPCHK(c, p->ok.reserve(u1)); HIPCHK(c, c->in_a.reserve(128 * u1)); HIPCHK(c, p->raw.reserve(u1 * PREP_RAW_LIMBS * 4));
  if (u) HIPCHK(c, hipMemcpyAsync(c->in_a.p, pks, 128 * u, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync((uint8_t*)c->in_a.p + 128 * u, NEG_G2_BYTES, 128, hipMemcpyHostToDevice, c->stream));
  LAUNCH_G2_PREPARE(c, LAUNCH, (const uint8_t*)c->in_a.p, (const uint32_t*)nullptr, u1, (int32_t*)p->raw.p, (uint8_t*)p->ok.p, (const uint32_t*)nullptr);
  LAUNCH(c, "g2_expand", k_g2_expand, u1 * (size_t)BN_NEG_G2_LINES, (const int32_t*)p->raw.p, (uint32_t)u1, (int32_t*)p->table.p, (const uint32_t*)nullptr);
  hipError_t es = hipStreamSynchronize(c->stream);
  if (es != hipSuccess) { (void)hipDeviceSynchronize(); }       // nothing may still be writing the buffers the owner frees
  HIPCHK(c, es);
  own.p = nullptr;
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
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
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

int blsbn254_verify_batch(blsbn254_ctx* c, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs,
                          size_t n, const uint8_t* dst, size_t dst_len, uint8_t* bm) {
  if (!c || !off || (n && (!pks || !sigs || !bm)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  ENTER(c);
  int rc = stage_msgs(c, msgs, off, n);
  if (rc) return rc;
  size_t nb = (n + 7) / 8;
  HIPCHK(c, c->in_a.reserve(128 * n)); HIPCHK(c, c->in_b.reserve(64 * n)); HIPCHK(c, c->bitmap.reserve(nb + 8));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, pks, 128 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, sigs, 64 * n, hipMemcpyHostToDevice, c->stream));
  rc = blsbn254_internal_verify_batch_dev_sync(c, (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p,
                                 (const uint8_t*)c->in_b.p, n, dst, dst_len, (uint8_t*)c->bitmap.p);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(bm, c->bitmap.p, nb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
