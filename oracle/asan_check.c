/* asan_check.c -- drives the CPU oracle (TEST INFRASTRUCTURE) under AddressSanitizer + UBSan: `make -C oracle asan_check`
 * builds this file together with bn254_oracle.c with -fsanitize=address,undefined and tests/test_oracle_golden.py runs it.
 * Exercises every family of entry points on small inputs, including empty and ragged messages, undecodable points and a
 * multi-threaded batch; exits 0 when the sanitizers stay silent and the algebra holds. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bn254_oracle.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "asan_check: %s failed at line %d\n", #x, __LINE__); return 1; } } while (0)

int main(void) {
  static const uint8_t dst[] = "ASAN_DST";
  uint8_t g1[64], g2[128], gt[384], gt2[384], ml[384];
  oracle_g1_generator(g1); oracle_g2_generator(g2);
  CHECK(oracle_pairing_batch(g1, g2, 1, gt) == 0);
  CHECK(oracle_multi_miller_loop(g1, g2, 1, ml) == 0);
  CHECK(oracle_final_exponentiation(ml, 1, gt2) == 0 && memcmp(gt, gt2, 384) == 0);
  /* sign / verify over ragged messages (lengths 0, 1, 33, 200), 4 threads, one corrupted tuple */
  enum { N = 8 };
  static const size_t lens[N] = {0, 1, 33, 200, 7, 64, 65, 0};
  uint8_t msgs[1024], pks[N * 128], sigs[N * 64], bm[1] = {0}, bm2[1] = {0};
  uint64_t off[N + 1]; off[0] = 0;
  for (int i = 0; i < N; ++i) { off[i + 1] = off[i] + lens[i]; }
  for (size_t k = 0; k < off[N]; ++k) msgs[k] = (uint8_t)(k * 37 + 1);
  for (int i = 0; i < N; ++i) {
    uint8_t sk[32] = {0}; sk[31] = (uint8_t)(i + 2); sk[7] = 0x11;
    CHECK(oracle_sk_to_pk(sk, pks + 128 * i) == 0);
    CHECK(oracle_sign(sk, msgs + off[i], lens[i], dst, sizeof dst - 1, sigs + 64 * i) == 0);
  }
  sigs[64 * 5 + 63] ^= 1;                                   /* off-curve signature */
  memset(pks + 128 * 6, 0xff, 32);                          /* coordinate >= p: undecodable key */
  CHECK(oracle_verify_batch(pks, msgs, off, sigs, N, dst, sizeof dst - 1, bm) == 0);
  CHECK(oracle_verify_batch_mt(pks, msgs, off, sigs, N, dst, sizeof dst - 1, bm2, 4) == 0);
  CHECK(bm[0] == 0x9f && bm2[0] == 0x9f);
  /* hash to curve, point checks, codecs */
  uint8_t h1[N * 64], h2[N * 128], ok[1], c1[32], c2[64], d1[64], d2[128];
  CHECK(oracle_hash_to_g1_batch(msgs, off, N, dst, sizeof dst - 1, h1) == 0);
  CHECK(oracle_hash_to_g2_batch(msgs, off, N, dst, sizeof dst - 1, h2) == 0);
  CHECK(oracle_g1_check_batch(h1, N, ok) == 0 && ok[0] == 0xff);
  CHECK(oracle_g2_check_batch(h2, N, ok) == 0 && ok[0] == 0xff);
  CHECK(oracle_g1_compress(h1, c1) == 0 && oracle_g1_decompress(c1, d1) == 0 && memcmp(d1, h1, 64) == 0);
  CHECK(oracle_g2_compress(h2, c2) == 0 && oracle_g2_decompress(c2, d2) == 0 && memcmp(d2, h2, 128) == 0);
  /* aggregate + threshold */
  uint8_t agg[64]; int valid = 0;
  sigs[64 * 5 + 63] ^= 1;
  { uint8_t sk[32] = {0}; sk[31] = 8; sk[7] = 0x11; CHECK(oracle_sk_to_pk(sk, pks + 128 * 6) == 0); }
  CHECK(oracle_aggregate_sigs(sigs, N, agg) == 0);
  /* messages 0 and 7 are both empty: AggregateVerify is still an algebraic identity here (distinctness is the caller's duty) */
  CHECK(oracle_aggregate_verify(pks, msgs, off, N, agg, dst, sizeof dst - 1, &valid) == 0 && valid == 1);
  uint8_t ids[3 * 32] = {0}, lam[3 * 32], outsig[64];
  ids[31] = 1; ids[63] = 2; ids[95] = 5;
  CHECK(oracle_fr_lagrange_at_zero(ids, 3, lam) == 0);
  CHECK(oracle_threshold_combine(ids, sigs, 3, outsig) == 0);
  ids[63] = 1;                                              /* duplicate id */
  CHECK(oracle_threshold_combine(ids, sigs, 3, outsig) == 1);
  /* the reference-style arithmetic leg */
  CHECK(oracle_verify_batch_refstyle_mt(pks, msgs, off, sigs, 2, dst, sizeof dst - 1, bm, 2) == 0 && (bm[0] & 3) == 3);
  CHECK(oracle_verify_batch(pks, msgs, off, sigs, 2, dst, sizeof dst - 1, bm2) == 0 && (bm2[0] & 3) == 3);
  printf("asan_check ok\n");
  return 0;
}
