// host_rlc.hip -- random-linear-combination batch verification: per-key sums, the chunk / key rounds over
// repeated keys (k_rlc2.hip) and the distinct-key variant (k_rlc.hip).  Host side of include/blsbn254.h; see host_common.h.
#include "host_common.h"

extern "C" {

// ---------------- sums of G1 points per key
// `items` homogeneous points (limb-major at pts, stride pts_stride; optionally a second array pts2 summed alongside) are grouped
// by key in key order: item j has key kid[mark_perm[j]] and is the point in column pt_perm[j] (or j when pt_perm is NULL); key k
// owns hist[k] consecutive items ending at run_end[k].  Level by level, runs are cut into chunks of at most KEY_SUM_GROUP items,
// every chunk is summed by one lane, and the chunk sums (in key order too) are the items of the next level, until every key has
// ONE sum: *out / *out2 (stride u), indexed by key id.  One host synchronisation per level (the chunk count).
static const size_t KEY_SUM_GROUP = 32;
int key_sums(blsbn254_ctx* c, const int32_t* pts, const int32_t* pts2, size_t pts_stride, const uint32_t* mark_perm, const uint32_t* pt_perm,
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
int draw_seed(blsbn254_ctx* c, uint8_t out[32]) {
  size_t got = 0;
  while (got < 32) {
    ssize_t k = getrandom(out + got, 32 - got, 0);
    if (k <= 0) { c->last_error = "getrandom failed"; return BLSBN254_E_HIP; }
    got += (size_t)k;
  }
  return 0;
}
// One table-only Miller loop + final exponentiation over `cnt` (virtual or real) tuples: is_one bytes to d_isone, flags in ctx->flags.
int prepared_round(blsbn254_ctx* c, const uint32_t* perm, const uint32_t* kid, const uint8_t* sigs, const int32_t* h_ws, size_t h_stride,
                          size_t cnt, uint8_t* d_isone) {
  HIPCHK(c, c->f_ws.reserve(cnt * 108 * 4)); HIPCHK(c, c->flags.reserve(cnt));
  if (c->wide_fe && cnt <= c->wide_fe_max) {
    LAUNCH_WIDE(c, "miller_wide_prepared", k_miller_wide_prepared, cnt, perm, kid, sigs, h_ws, h_stride, (const int32_t*)c->prep_table.p, (const uint8_t*)c->prep_ok.p, cnt,
                (int32_t*)c->f_ws.p, (uint8_t*)c->flags.p);
  } else if (cnt <= c->tri_max && c->tri_miller) {     // the chunk / fallback rounds: three lanes per (virtual) tuple (k_tri.hip)
    LAUNCH_TRI(c, "miller_tri_prepared", k_miller_tri_prepared, cnt, perm, kid, sigs, h_ws, h_stride, (const int32_t*)c->prep_table.p, (const uint8_t*)c->prep_ok.p, cnt,
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
  Stream2Guard s2_guard(c);
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
    // ... and up to tri_max virtual tuples the round runs three lanes per tuple (k_tri.hip: less than half the latency of the
    // lane-per-tuple kernels): a chunk-count bound just above that limit is worth slightly larger chunks (262144 tuples over
    // 1024 keys: G = 18 -> at most 15587 chunks instead of 17408)
    if (c->tri_miller && c->tri_fe && c->tri_max > c->wide_fe_max && n / G + u > c->tri_max)
      for (size_t g = G + 1; g <= 2 * G; ++g)
        if (n / g + u <= c->tri_max) { G = g; break; }
  }
  const size_t nblk = (n + 255) / 256;
  const uint32_t G32 = (uint32_t)G, n32 = (uint32_t)n, u32 = (uint32_t)u;
  HIPCHK(c, c->prep_table.reserve(u * PREP_KEY_LIMBS * 4)); HIPCHK(c, c->prep_ok.reserve(u));
  rc = prepare_keys_async(c, d_pks, (const uint32_t*)c->kd_keys.p, u, (int32_t*)c->prep_table.p, (uint8_t*)c->prep_ok.p, nullptr);
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
  HIPCHK(c, join_stream2(c));                      // the key tables are ready
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
  ENTER(c);
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
  ENTER(c);
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


}  // extern "C"
