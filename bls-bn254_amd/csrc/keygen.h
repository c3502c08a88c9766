// keygen.h -- secret-key derivation and hash-to-scalar, one instance per lane.
//   * lane_hash_to_scalar: Scalar::hash<ExpandMsgXmd<Sha256>>(msg, dst)  (scalar.rs:554-563): 48 bytes of
//     expand_message_xmd reduced mod r.
//   * lane_keygen: IETF BLS KeyGen (draft-irtf-cfrg-bls-signature-05 section 2.3) over HKDF-SHA-256
//     (RFC 5869) with the salt the reference names (KEYGEN_SALT, helpers.rs:3) and L = 48.
// The reference declares the salt but has no KeyGen; the procedure is the draft's.
#pragma once
#include "fr29.h"
#include "sha256.h"

namespace bn {

// HMAC-SHA-256 (RFC 2104) with a key of at most 64 bytes; message given as up to three parts.
struct Hmac256 { Sha256 inner; uint8_t kpad[64]; };
BN_FUNC void hmac_init(Hmac256& h, const uint8_t* key, uint32_t key_len) {
  for (uint32_t i = 0; i < 64; ++i) h.kpad[i] = i < key_len ? key[i] : 0;
  sha256_init(h.inner);
  for (int i = 0; i < 64; ++i) sha256_byte(h.inner, h.kpad[i] ^ 0x36);
}
BN_FUNC void hmac_update(Hmac256& h, const uint8_t* p, size_t n) { sha256_update(h.inner, p, n); }
BN_FUNC void hmac_final(Hmac256& h, uint8_t out[32]) {
  uint8_t ih[32];
  sha256_final(h.inner, ih);
  Sha256 o;
  sha256_init(o);
  for (int i = 0; i < 64; ++i) sha256_byte(o, h.kpad[i] ^ 0x5c);
  sha256_update(o, ih, 32);
  sha256_final(o, out);
}

// status: true when a non-zero key was found (always, up to probability 2^-254 per round; the loop is
// bounded so that every lane leaves it).
BN_FUNC Fr lane_keygen(const uint8_t* ikm, size_t ikm_len, const uint8_t* key_info, size_t key_info_len, bool& ok) {
  const uint8_t salt0[20] = {'B', 'L', 'S', '-', 'S', 'I', 'G', '-', 'K', 'E', 'Y', 'G', 'E', 'N', '-', 'S', 'A', 'L', 'T', '-'};
  uint8_t salt[32], prk[32], okm[64];
  Fr sk;
  ok = false;
  for (int round = 0; round < 4 && !ok; ++round) {
    Sha256 s;
    sha256_init(s);
    if (round == 0) sha256_update(s, salt0, 20); else sha256_update(s, salt, 32);
    sha256_final(s, salt);                                            // salt = H(salt)
    Hmac256 h;
    hmac_init(h, salt, 32);                                           // PRK = HKDF-Extract(salt, IKM || 0x00)
    hmac_update(h, ikm, ikm_len);
    const uint8_t zero = 0;
    hmac_update(h, &zero, 1);
    hmac_final(h, prk);
    const uint8_t l2[2] = {0, 48};                                    // info = key_info || I2OSP(L, 2)
    for (uint8_t blk = 1; blk <= 2; ++blk) {                          // OKM = T(1) || T(2)[0..16)
      hmac_init(h, prk, 32);
      if (blk == 2) hmac_update(h, okm, 32);
      hmac_update(h, key_info, key_info_len);
      hmac_update(h, l2, 2);
      hmac_update(h, &blk, 1);
      hmac_final(h, okm + 32 * (blk - 1));
    }
    sk = fr_from_okm(okm);
    ok = !fr_is_zero(sk);
  }
  return sk;
}

BN_FUNC Fr lane_hash_to_scalar(const uint8_t* msg, size_t msg_len, const uint8_t* dst, uint32_t dst_len) {
  uint8_t okm[48];
  expand_message_xmd(okm, 48, msg, msg_len, dst, dst_len);
  return fr_from_okm(okm);
}

}  // namespace bn
