// host_aggregate.hip -- aggregate verify (pair by pair, per-key sums, sharded partial / finish), signature aggregation,
// threshold combine and Lagrange coefficients.  Host side of include/blsbn254.h; see host_common.h.
#include "host_common.h"

extern "C" {

// in-place product tree over `cnt` Fp12 values at a (stride sa); result pointer / stride returned
int fp12_tree(blsbn254_ctx* c, int32_t* a, size_t cnt, size_t sa, int32_t** res, size_t* rs) {
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
  Stream2Guard s2_guard(c);
  *all_pks_ok = 1;
  if (sig_ok) *sig_ok = 1;
  const size_t np = n + (extra_sig ? 1 : 0);                 // pairs in the loop
  if (np == 0) { std::memset(ml_out, 0, 384); ml_out[31] = 1; return 0; }
  CHECK_LANES(c, np);
  ENTER(c);
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
      HIPCHK(c, fork_stream2(c));
      LAUNCH_G2_PREPARE(c, LAUNCH2, (const uint8_t*)c->in_a.p, (const uint32_t*)c->kd_keys.p, u, (int32_t*)c->prep_raw.p, (uint8_t*)c->prep_ok.p, (const uint32_t*)nullptr);
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
    HIPCHK(c, join_stream2(c));
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
  ENTER(c);
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
  Stream2Guard s2_guard(c);
  *took = false;
  ENTER(c);
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
  // few pairs: from tables, one wave per pair (or, up to tri_max pairs, a quad of lanes per pair: k_miller_tri_1p), whatever the keys
  const bool small = (c->wide_fe && n + 1 <= c->wide_fe_max) || (c->tri_miller && n + 1 <= c->tri_max);
  if (!((u * 2 <= n || small) && u + 1 <= PREP_MAX_KEYS)) return 0;
  *took = true;
  const size_t np = u + 1, n_lanes = (np + 1) / 2;
  const uint32_t n32 = (uint32_t)n, u32 = (uint32_t)u;
  // the u keys and -G2gen (key id u) become line tables on the second stream, beside the hashing
  const uint32_t last_key = n32;
  HIPCHK(c, hipMemcpyAsync((uint32_t*)c->kd_keys.p + u, &last_key, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));                                   // last_key and the staged copies are consumed
  HIPCHK(c, c->prep_ok.reserve(np)); HIPCHK(c, c->prep_raw.reserve(np * PREP_RAW_LIMBS * 4));
  HIPCHK(c, fork_stream2(c));
  LAUNCH_G2_PREPARE(c, LAUNCH2, (const uint8_t*)c->in_a.p, (const uint32_t*)c->kd_keys.p, np, (int32_t*)c->prep_raw.p, (uint8_t*)c->prep_ok.p, (const uint32_t*)nullptr);
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
  HIPCHK(c, join_stream2(c));
  // few pairs: the launch is the latency of one wave, so one pair per lane (no shared f^2, shorter chain); else two per lane
  const bool one_per_lane = np * 2 <= c->lanes_per_round;
  const size_t f_cnt = one_per_lane ? np : n_lanes;
  HIPCHK(c, c->f_ws.reserve(f_cnt * 108 * 4));
  if (c->wide_fe && np <= c->wide_fe_max) {           // a handful of keys: one WAVE per pair
    LAUNCH_WIDE(c, "miller_wide_1p", k_miller_wide_1p, np, (const int32_t*)h2, np, (const uint32_t*)kid2, (const int32_t*)c->prep_raw.p, (const uint8_t*)c->prep_ok.p, np,
                (int32_t*)c->f_ws.p, np, (uint8_t*)c->flags.p, (const uint8_t*)st);
  } else if (c->tri_miller && np <= c->tri_max) {     // a few thousand pairs: a quad of lanes per pair
    LAUNCH_TRI(c, "miller_tri_1p", k_miller_tri_1p, np, (const int32_t*)h2, np, (const uint32_t*)kid2, (const int32_t*)c->prep_raw.p, (const uint8_t*)c->prep_ok.p, np,
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
int g1_sum_to_bytes(blsbn254_ctx* c, size_t n, uint8_t out[64]) {
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
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
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


}  // extern "C"
