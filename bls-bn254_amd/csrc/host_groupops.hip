// host_groupops.hip -- group operations on caller-supplied points and FastAggregateVerify: Mul<Scalar> for G1 / G2,
// impl Sum for G2Projective (aggregate public keys), and "one message signed by many keys" verified as one pairing equation
// per group over the sum of the group's keys.  Host side of include/blsbn254.h; kernels in k_groupops.hip; see host_common.h.
#include "host_common.h"

extern "C" {

// out_i = [k_i] P_i
static int mul_common(blsbn254_ctx* c, const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t* out, int g2) {
  if (!c || (n && (!pts || !scalars || !out))) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  ENTER(c);
  const size_t sz = g2 ? 128 : 64;
  for (size_t lo = 0; lo < n; lo += c->chunk) {
    const size_t m = n - lo < c->chunk ? n - lo : c->chunk;
    HIPCHK(c, c->in_a.reserve(sz * m)); HIPCHK(c, c->scalars.reserve(32 * m)); HIPCHK(c, c->out.reserve(sz * m)); HIPCHK(c, c->status.reserve(m));
    HIPCHK(c, hipMemcpyAsync(c->in_a.p, pts + sz * lo, sz * m, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->scalars.p, scalars + 32 * lo, 32 * m, hipMemcpyHostToDevice, c->stream));
    if (g2) { LAUNCH(c, "g2_mul", k_g2_mul, m, (const uint8_t*)c->in_a.p, (const uint8_t*)c->scalars.p, m, (uint8_t*)c->out.p, (uint8_t*)c->status.p); }
    else { LAUNCH(c, "g1_mul", k_g1_mul, m, (const uint8_t*)c->in_a.p, (const uint8_t*)c->scalars.p, m, (uint8_t*)c->out.p, (uint8_t*)c->status.p); }
    int bad;
    int rc = first_bad(c, (const uint8_t*)c->status.p, m, 3, 3, &bad);
    if (rc) return rc;
    if (bad >= 0) {
      uint8_t st = 0;
      rc = read_status(c, (const uint8_t*)c->status.p, bad, &st);
      if (rc) return rc;
      c->last_error = std::string(st & 1 ? "scalar not canonical" : "point does not decode or is off the curve") + " at element " + std::to_string(lo + (size_t)bad);
      return (st & 1) ? BLSBN254_ERR_SCALAR : (g2 ? BLSBN254_ERR_G2 : BLSBN254_ERR_G1);
    }
    HIPCHK(c, hipMemcpyAsync(out + sz * lo, c->out.p, sz * m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return 0;
}
int blsbn254_g1_mul_batch(blsbn254_ctx* c, const uint8_t* g1, const uint8_t* scalars, size_t n, uint8_t* out) { return mul_common(c, g1, scalars, n, out, 0); }
int blsbn254_g2_mul_batch(blsbn254_ctx* c, const uint8_t* g2, const uint8_t* scalars, size_t n, uint8_t* out) { return mul_common(c, g2, scalars, n, out, 1); }

// Segmented sums of G2 points: n points already staged at d_pks, groups [goff[g], goff[g + 1]).  Level by level every group is
// cut into chunks of at most G2_SUM_GROUP items, one lane sums a chunk, and the chunk sums (group-major order) are the next
// level's items, until every group is ONE item.  The chunk descriptors of all levels come from the host (the offsets are
// the caller's host array).  On return the sums are at *sum_ws (limb-major, stride = n_groups) with their flags at *sum_ok.
static const size_t G2_SUM_GROUP = 16;
static int g2_group_sums(blsbn254_ctx* c, const uint8_t* d_pks, size_t n, const uint64_t* goff, size_t n_groups, const int32_t** sum_ws, const uint8_t** sum_ok) {
  const size_t G = G2_SUM_GROUP;
  // chunk descriptors of ALL levels first (host arithmetic on the offsets), uploaded in one copy: no synchronisation between levels
  std::vector<uint64_t> cur(goff, goff + n_groups + 1), nxt(n_groups + 1);
  for (size_t g = 0; g <= n_groups; ++g) cur[g] -= goff[0];
  std::vector<uint32_t> start, len;
  std::vector<size_t> level_first, level_count;
  for (int level = 0; ; ++level) {
    if (level > 40) { c->last_error = "internal: group sums do not converge"; return BLSBN254_E_HIP; }
    const size_t first = start.size();
    for (size_t g = 0; g < n_groups; ++g) {
      const uint64_t a = cur[g], b = cur[g + 1];
      nxt[g] = start.size() - first;
      if (a == b) { start.push_back((uint32_t)a); len.push_back(0); }          // empty group: one empty chunk (identity, flag 0)
      for (uint64_t s = a; s < b; s += G) { start.push_back((uint32_t)s); len.push_back((uint32_t)(b - s < G ? b - s : G)); }
    }
    const size_t m = start.size() - first;
    nxt[n_groups] = m;
    level_first.push_back(first); level_count.push_back(m);
    cur.swap(nxt);
    if (m == n_groups) break;
  }
  const size_t total = start.size(), m_max = level_count[0];
  // ping-pong workspaces: the level-0 items (n) and the largest chunk array (the first level's)
  HIPCHK(c, c->gs_ws[0].reserve((n ? n : 1) * 54 * 4)); HIPCHK(c, c->gs_ok[0].reserve(n ? n : 1));
  HIPCHK(c, c->gs_ws[1].reserve(m_max * 54 * 4)); HIPCHK(c, c->gs_ok[1].reserve(m_max));
  HIPCHK(c, c->gs_ws[2].reserve(m_max * 54 * 4)); HIPCHK(c, c->gs_ok[2].reserve(m_max));
  HIPCHK(c, c->gs_start.reserve(4 * total)); HIPCHK(c, c->gs_len.reserve(4 * total));
  HIPCHK(c, hipMemcpyAsync(c->gs_start.p, start.data(), 4 * total, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->gs_len.p, len.data(), 4 * total, hipMemcpyHostToDevice, c->stream));
  if (n) { LAUNCH(c, "g2_load", k_g2_load, n, d_pks, n, (int32_t*)c->gs_ws[0].p, (uint8_t*)c->gs_ok[0].p); }
  int src = 0;
  size_t items = n;
  for (size_t lv = 0; lv < level_count.size(); ++lv) {
    const size_t m = level_count[lv];
    const int dst = src == 1 ? 2 : 1;
    LAUNCH(c, "g2_seg_sum", k_g2_seg_sum, m, (const int32_t*)c->gs_ws[src].p, items ? items : 1, (const uint8_t*)c->gs_ok[src].p,
           (const uint32_t*)c->gs_start.p + level_first[lv], (const uint32_t*)c->gs_len.p + level_first[lv], m, (int32_t*)c->gs_ws[dst].p, m, (uint8_t*)c->gs_ok[dst].p);
    src = dst; items = m;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));                                   // `start` / `len` go out of scope
  *sum_ws = (const int32_t*)c->gs_ws[src].p; *sum_ok = (const uint8_t*)c->gs_ok[src].p;
  return 0;
}

// impl Sum for G2Projective (g2.rs:579-583): out = sum of the n points (the identity encoding for n == 0)
int blsbn254_aggregate_pks(blsbn254_ctx* c, const uint8_t* pks, size_t n, uint8_t out[128]) {
  if (!c || !out || (n && !pks)) return BLSBN254_E_ARG;
  if (n == 0) { std::memset(out, 0, 128); out[127] = 1; return 0; }             // G2Affine::identity: x = 0, y = 1
  if (n > ((size_t)1 << 26)) { c->last_error = "more than 2^26 points in one sum"; return BLSBN254_E_ARG; }
  ENTER(c);
  HIPCHK(c, c->in_a.reserve(128 * n)); HIPCHK(c, c->out.reserve(128));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, pks, 128 * n, hipMemcpyHostToDevice, c->stream));
  const uint64_t goff[2] = {0, (uint64_t)n};
  const int32_t* ws; const uint8_t* ok;
  int rc = g2_group_sums(c, (const uint8_t*)c->in_a.p, n, goff, 1, &ws, &ok);
  if (rc) return rc;
  uint8_t good = 0;
  LAUNCH(c, "g2p_to_bytes", k_g2p_to_bytes, 1, ws, (size_t)1, ok, (size_t)1, (uint8_t*)c->out.p, 0);
  HIPCHK(c, hipMemcpyAsync(out, c->out.p, 128, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&good, ok, 1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!good) { c->last_error = "a point does not decode or is off the curve"; return BLSBN254_ERR_G2; }
  return 0;
}

// FastAggregateVerify per group g: keys pks[key_off[g] .. key_off[g + 1]), ONE message msg_g, ONE signature sig_g:
//   e(sig_g, -G2gen) * e(H(msg_g), sum_k pk_k) == 1
// = CoreVerify under the sum of the group's keys.  The sums are formed on the device (g2_group_sums) and their encodings go
// through the verify pipeline like any public key (KeyValidate on the SUM: not the identity, in the r-torsion).
int blsbn254_fast_aggregate_verify_batch(blsbn254_ctx* c, const uint8_t* pks, const uint64_t* key_off, const uint8_t* msgs, const uint64_t* off,
                                         const uint8_t* sigs, size_t n_groups, const uint8_t* dst, size_t dst_len, uint8_t* valid_bitmap) {
  if (!c || !key_off || !off || (n_groups && (!sigs || !valid_bitmap)) || (dst_len && !dst)) return BLSBN254_E_ARG;
  if (n_groups == 0) return 0;
  for (size_t g = 0; g < n_groups; ++g) if (key_off[g + 1] < key_off[g]) return BLSBN254_E_ARG;
  const size_t n_keys = (size_t)(key_off[n_groups] - key_off[0]);
  if (n_keys && !pks) return BLSBN254_E_ARG;
  if (n_keys > ((size_t)1 << 26) || n_groups > c->chunk) { c->last_error = "more than 2^26 keys or more groups than one launch chunk"; return BLSBN254_E_ARG; }
  ENTER(c);
  uint32_t dl; int rc = stage_dst(c, dst, dst_len, &dl);
  if (rc) return rc;
  rc = stage_msgs(c, msgs, off, n_groups);
  if (rc) return rc;
  const size_t nb = (n_groups + 7) / 8;
  HIPCHK(c, c->in_a.reserve(128 * (n_keys ? n_keys : 1))); HIPCHK(c, c->in_b.reserve(64 * n_groups)); HIPCHK(c, c->gs_pk.reserve(128 * n_groups));
  HIPCHK(c, c->bitmap.reserve(nb + 8));
  if (n_keys) HIPCHK(c, hipMemcpyAsync(c->in_a.p, pks + 128 * (size_t)key_off[0], 128 * n_keys, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, sigs, 64 * n_groups, hipMemcpyHostToDevice, c->stream));
  const int32_t* ws; const uint8_t* ok;
  rc = g2_group_sums(c, (const uint8_t*)c->in_a.p, n_keys, key_off, n_groups, &ws, &ok);
  if (rc) return rc;
  LAUNCH(c, "g2p_to_bytes", k_g2p_to_bytes, n_groups, ws, n_groups, ok, n_groups, (uint8_t*)c->gs_pk.p, 1);
  rc = verify_chunk_dev(c, (const uint8_t*)c->gs_pk.p, (const uint8_t*)c->in_c.p, (const uint64_t*)c->in_off.p, (const uint8_t*)c->in_b.p, n_groups, dl,
                        (uint8_t*)c->bitmap.p);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(valid_bitmap, c->bitmap.p, nb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int blsbn254_fast_aggregate_verify(blsbn254_ctx* c, const uint8_t* pks, size_t n, const uint8_t* msg, size_t msg_len, const uint8_t sig[64],
                                   const uint8_t* dst, size_t dst_len, int* valid) {
  if (!c || !valid || !sig || (n && !pks) || (msg_len && !msg) || (dst_len && !dst)) return BLSBN254_E_ARG;
  *valid = 0;
  if (n == 0) return 0;
  const uint64_t koff[2] = {0, (uint64_t)n}, moff[2] = {0, (uint64_t)msg_len};
  uint8_t bm = 0;
  int rc = blsbn254_fast_aggregate_verify_batch(c, pks, koff, msg, moff, sig, 1, dst, dst_len, &bm);
  if (rc) return rc;
  *valid = bm & 1;
  return 0;
}

}  // extern "C"
