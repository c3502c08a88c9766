// host_misc.hip -- field / tower primitives (debug ABI), Gt group operations, compressed codecs, signing, key derivation and
// proof of possession.  Host side of include/blsbn254.h; see host_common.h.
#include "host_common.h"

extern "C" {

// ---------------- field / tower primitives (debug ABI) and Gt group operations
static size_t field_op_width(int op) { return op < 0 ? 0 : op <= 8 ? 32 : (op >= 16 && op <= 21) ? 64 : (op >= 32 && op <= 35) ? 192 : (op >= 48 && op <= 56) ? 384 : 0; }
static bool field_op_binary(int op) { return op == 0 || op == 3 || op == 4 || op == 16 || op == 32 || op == 48 || op == 56; }
int blsbn254_field_op_batch(blsbn254_ctx* c, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
  const size_t w = field_op_width(op);
  if (!c || w == 0 || (n && (!a || !out || (field_op_binary(op) && !b)))) return BLSBN254_E_ARG;
  if (n == 0) return 0;
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
  size_t nb = (n + 7) / 8;
  HIPCHK(c, c->in_a.reserve(128 * n)); HIPCHK(c, c->in_b.reserve(64 * n)); HIPCHK(c, c->in_off.reserve(8 * (n + 1)));
  HIPCHK(c, c->bitmap.reserve(nb + 8));
  HIPCHK(c, hipMemcpyAsync(c->in_a.p, pks, 128 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_b.p, proofs, 64 * n, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "iota_off", k_iota_off, n + 1, (uint64_t*)c->in_off.p, n, (uint64_t)128);
  int rc = blsbn254_internal_verify_batch_dev_sync(c, (const uint8_t*)c->in_a.p, (const uint8_t*)c->in_a.p, (const uint64_t*)c->in_off.p,
                                     (const uint8_t*)c->in_b.p, n, dst, dst_len, (uint8_t*)c->bitmap.p);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(bm, c->bitmap.p, nb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}


}  // extern "C"
