/* bn254_oracle.h -- CPU oracle for the BLS-BN254 verification path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's algorithm (mikelodder7/bls-bn254), used only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the *checker*.  The product
 * (bls-bn254_amd/) never links, loads or calls it.
 *
 * Every entry point mirrors one entry point of include/blsbn254.h (same buffers, same byte formats,
 * same return codes) so parity tests can call both with identical arguments.
 *
 * Parity status: PINNED.  Checked (tests/test_oracle_golden.py) against every golden vector the
 * reference's own tests hold for this path: 5+5 G1 and 5+5 G2 hash-to-curve KATs (g1.rs:981-1141,
 * g2.rs:1039-1313), the G2 bad-point fixture (g2.rs:994-1017), the 5*G identities (g1.rs:1144,
 * g2.rs:1031) and the pairing constant Gt::generator() with gt^r == 1 (pairings.rs:387-479,
 * :971-980).  Fr and the Fp6/Fp12 tower have no direct reference test (pinned transitively through
 * the Gt constant).
 */
#ifndef BN254_ORACLE_H
#define BN254_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* return codes: 0 ok; 1..4 = Bn254Error::{InvalidScalarBytes,InvalidG1Bytes,InvalidG2Bytes,InvalidGtBytes}
 * (error.rs:4-10); negative = argument errors. */

int oracle_pairing_batch(const uint8_t* g1, const uint8_t* g2, size_t n, uint8_t* gt);
int oracle_multi_miller_loop(const uint8_t* g1, const uint8_t* g2, size_t n, uint8_t ml_out[384]);
int oracle_miller_loop_batch(const uint8_t* g1, const uint8_t* g2, size_t n, uint8_t* ml_out);
int oracle_final_exponentiation(const uint8_t* ml, size_t n, uint8_t* gt);
int oracle_hash_to_g1_batch(const uint8_t* msgs, const uint64_t* off, size_t n,
                            const uint8_t* dst, size_t dst_len, uint8_t* out);
int oracle_hash_to_g2_batch(const uint8_t* msgs, const uint64_t* off, size_t n,
                            const uint8_t* dst, size_t dst_len, uint8_t* out);
int oracle_encode_to_g1_batch(const uint8_t* msgs, const uint64_t* off, size_t n,
                              const uint8_t* dst, size_t dst_len, uint8_t* out);
int oracle_encode_to_g2_batch(const uint8_t* msgs, const uint64_t* off, size_t n,
                              const uint8_t* dst, size_t dst_len, uint8_t* out);
int oracle_hash_to_field_fp(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len,
                            size_t count, uint8_t* out /* count*32 BE */);
int oracle_g1_check_batch(const uint8_t* g1, size_t n, uint8_t* ok_bitmap);
int oracle_g2_check_batch(const uint8_t* g2, size_t n, uint8_t* ok_bitmap);
/* g2 check with the reference's own method ([r]P == O, g2.rs:733-736) instead of the psi test */
int oracle_g2_check_batch_slow(const uint8_t* g2, size_t n, uint8_t* ok_bitmap);
int oracle_verify_batch(const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs,
                        size_t n, const uint8_t* dst, size_t dst_len, uint8_t* valid_bitmap);
int oracle_verify_batch_mt(const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs,
                           size_t n, const uint8_t* dst, size_t dst_len, uint8_t* valid_bitmap, int nthreads);
/* same bitmap, every Fp product in the reference's own arithmetic (timed baseline only; not re-entrant) */
int oracle_verify_batch_refstyle_mt(const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs,
                           size_t n, const uint8_t* dst, size_t dst_len, uint8_t* valid_bitmap, int nthreads);
int oracle_aggregate_verify(const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, size_t n,
                            const uint8_t agg_sig[64], const uint8_t* dst, size_t dst_len, int* valid);
int oracle_aggregate_sigs(const uint8_t* sigs, size_t n, uint8_t out[64]);
int oracle_threshold_combine(const uint8_t* ids, const uint8_t* partial_sigs, size_t t, uint8_t out_sig[64]);

/* helpers for generating test data (not part of the product ABI) */
int oracle_g1_mul(const uint8_t g1[64], const uint8_t scalar_be[32], uint8_t out[64]);
int oracle_g2_mul(const uint8_t g2[128], const uint8_t scalar_be[32], uint8_t out[128]);
int oracle_g1_add(const uint8_t a[64], const uint8_t b[64], uint8_t out[64]);
int oracle_g2_add(const uint8_t a[128], const uint8_t b[128], uint8_t out[128]);
void oracle_g1_generator(uint8_t out[64]);
void oracle_g2_generator(uint8_t out[128]);
int oracle_sk_to_pk(const uint8_t sk_be[32], uint8_t pk[128]);
int oracle_sign(const uint8_t sk_be[32], const uint8_t* msg, size_t msg_len,
                const uint8_t* dst, size_t dst_len, uint8_t sig[64]);
/* field / tower primitives, element-wise; op codes and byte layouts of include/blsbn254.h (BLSBN254_OP_*); b may be NULL for unary ops */
int oracle_field_op_batch(int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out);
int oracle_gt_pow(const uint8_t gt[384], const uint8_t scalar_be[32], uint8_t out[384]);
int oracle_gt_mul(const uint8_t a[384], const uint8_t b[384], uint8_t out[384]);
int oracle_fr_lagrange_at_zero(const uint8_t* ids, size_t t, uint8_t* out /* t*32 BE */);
int oracle_g1_compress(const uint8_t in[64], uint8_t out[32]);
int oracle_g1_decompress(const uint8_t in[32], uint8_t out[64]);
int oracle_g2_compress(const uint8_t in[128], uint8_t out[64]);
int oracle_g2_decompress(const uint8_t in[64], uint8_t out[128]);
void oracle_sha256(const uint8_t* msg, size_t len, uint8_t out[32]);
/* the reference's canonical-form multiply (fp.rs:404-407 over crypto-bigint's const_rem_wide), for timing only */
void oracle_fp_mul_refstyle(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
double oracle_bench_fp_mul(int refstyle, uint64_t iters);

/* instrumentation: exact Fp multiplication / squaring counts of the calling thread */
void oracle_verify_core_counts(uint64_t out[5]);
void oracle_counters_reset(void);
void oracle_counters_get(uint64_t* fp_mul, uint64_t* fp_sqr);

#ifdef __cplusplus
}
#endif
#endif
