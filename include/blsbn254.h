/* blsbn254.h -- C ABI of the MI355X-native batched BLS-BN254 verification engine.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / HIP types.  Each entry point
 * names the operator of the reference crate (mikelodder7/bls-bn254, /root/reference) whose body a
 * Rust shim would replace with the extern "C" call (INTEGRATION.md shows the binding).  The
 * reference has no FFI and no BLS scheme layer; the verify / aggregate /
 * threshold entry points are the IETF CoreVerify / CoreAggregateVerify composition of its
 * primitives, min-sig variant (signatures in G1, public keys in G2).
 *
 * Byte formats (authoritative, big-endian field elements):
 *   G1  64 B  x || y                          G1Affine::to_uncompressed      g1.rs:297-302
 *   G2 128 B  x.c1 || x.c0 || y.c1 || y.c0    G2Affine::to_uncompressed      g2.rs:292-300
 *   Gt 384 B  c0.c0.c0, c0.c0.c1, c0.c1.c0 .. c1.c2.c1   Gt::to_repr         pairings.rs:499-514
 *   Fr  32 B  big-endian (both directions; the reference's LE to_repr, E11, is not reproduced)
 *   identity: G1 = (0, 1), G2 = (0, 1); on input x == 0 means identity (g1.rs:352-353, g2.rs:377)
 *   Decoding is strict: a coordinate >= p is an error (the reference masks bit 255 of G1
 *   coordinates, g1.rs:346-347; not reproduced).  Compressed encodings are not accepted (the
 *   reference's G1 compressed codec is defective).
 *   msgs = concatenated message bytes, off = n+1 offsets (off[i]..off[i+1] is message i).
 *   bitmaps: ceil(n/8) bytes, bit i of the batch = bit (i & 7) of byte i >> 3.
 *
 * Return codes: 0 ok; 1..4 = Bn254Error::{InvalidScalarBytes, InvalidG1Bytes, InvalidG2Bytes,
 * InvalidGtBytes} in declaration order (error.rs:4-10), returned by the primitive entry points
 * when an operand does not decode (as the reference's TryFrom<&[u8]> does, macros.rs:130-136);
 * negative = BLSBN254_E_*.  In verify_batch a tuple that fails to decode or validate is NOT an
 * error: its bit in the output bitmap is cleared.
 *
 * Threading: a ctx owns one HIP stream and its device workspace on one GPU; calls on one ctx must
 * be externally serialized (the reference is pure and single-threaded, inner_types.rs:33-34).
 * The library never retains caller pointers past return.  There is NO CPU fallback: every entry
 * point fails with BLSBN254_E_NO_DEVICE when no gfx950 device is available.
 */
#ifndef BLSBN254_H
#define BLSBN254_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct blsbn254_ctx blsbn254_ctx;

#define BLSBN254_OK 0
#define BLSBN254_ERR_SCALAR 1 /* Bn254Error::InvalidScalarBytes */
#define BLSBN254_ERR_G1 2     /* Bn254Error::InvalidG1Bytes */
#define BLSBN254_ERR_G2 3     /* Bn254Error::InvalidG2Bytes */
#define BLSBN254_ERR_GT 4     /* Bn254Error::InvalidGtBytes */
#define BLSBN254_E_ARG (-1)
#define BLSBN254_E_HIP (-2)
#define BLSBN254_E_NOMEM (-3)
#define BLSBN254_E_NO_DEVICE (-4)
#define BLSBN254_E_RCCL (-5)

/* One context per GPU (device = HIP ordinal).  Replaces nothing in the reference (it has no
 * state); owns the stream, the workspace and the resident -G2gen line table.  The N-GPU form of SURVEY.md 8b
 * (a ctx over a device list) is blsbn254_multi below, built from these. */
int blsbn254_ctx_create(int device, blsbn254_ctx** out);
void blsbn254_ctx_destroy(blsbn254_ctx* ctx);
const char* blsbn254_strerror(int code);
const char* blsbn254_last_error(blsbn254_ctx* ctx); /* text of the last HIP error on this ctx */

/* ---- primitives, 1:1 with the reference operator API ------------------------------------- */
/* Size limits: entry points whose elements are independent (pairing_batch, miller_loop_batch, final_exponentiation,
 * verify_batch*, pop_verify_batch) accept any n and process it in chunks of 4 Mi elements; the product-type ones
 * (multi_miller_loop, aggregate_*, verify_batch_rlc) take at most 2^23 elements per call (BLSBN254_E_ARG beyond). */
/* pairing(&G1Affine, &G2Affine) -> Gt, pairings.rs:760-802 (pairing::Engine::pairing :685-696).
 * Identity in either slot gives Gt::IDENTITY. */
int blsbn254_pairing_batch(blsbn254_ctx* ctx, const uint8_t* g1, const uint8_t* g2, size_t n, uint8_t* gt);
/* multi_miller_loop(&[(&G1Affine, &G2Prepared)]) -> MillerLoopResult, pairings.rs:808-857
 * (pairing::MultiMillerLoop :706-713).  Pairs with an identity member are skipped.  Output = the
 * 384-byte Fp12 Miller-loop value (before final exponentiation). */
int blsbn254_multi_miller_loop(blsbn254_ctx* ctx, const uint8_t* g1, const uint8_t* g2, size_t n, uint8_t ml_out[384]);
/* per-pair Miller loops (no product): n outputs of 384 B */
int blsbn254_miller_loop_batch(blsbn254_ctx* ctx, const uint8_t* g1, const uint8_t* g2, size_t n, uint8_t* ml_out);
/* MillerLoopResult::final_exponentiation, pairings.rs:50-178 (pairing::MillerLoopResult :698-704) */
int blsbn254_final_exponentiation(blsbn254_ctx* ctx, const uint8_t* ml, size_t n, uint8_t* gt);
/* G1Projective::hash::<ExpandMsgXmd<Sha256>>(msg, dst), g1.rs:910-919; ::encode g1.rs:922-928 */
int blsbn254_hash_to_g1_batch(blsbn254_ctx* ctx, const uint8_t* msgs, const uint64_t* off, size_t n,
                              const uint8_t* dst, size_t dst_len, uint8_t* out);
int blsbn254_encode_to_g1_batch(blsbn254_ctx* ctx, const uint8_t* msgs, const uint64_t* off, size_t n,
                                const uint8_t* dst, size_t dst_len, uint8_t* out);
/* G2Projective::hash / ::encode, g2.rs:919-936 */
int blsbn254_hash_to_g2_batch(blsbn254_ctx* ctx, const uint8_t* msgs, const uint64_t* off, size_t n,
                              const uint8_t* dst, size_t dst_len, uint8_t* out);
int blsbn254_encode_to_g2_batch(blsbn254_ctx* ctx, const uint8_t* msgs, const uint64_t* off, size_t n,
                                const uint8_t* dst, size_t dst_len, uint8_t* out);
/* G1Affine::from_uncompressed + is_on_curve (g1.rs:339-360, :383-391); G1 has cofactor 1 */
int blsbn254_g1_check_batch(blsbn254_ctx* ctx, const uint8_t* g1, size_t n, uint8_t* ok_bitmap);
/* G2Affine::from_uncompressed + is_on_curve + is_torsion_free (g2.rs:350-414, :733-736) */
int blsbn254_g2_check_batch(blsbn254_ctx* ctx, const uint8_t* g2, size_t n, uint8_t* ok_bitmap);

/* ---- BLS layer (build-defined composition; min-sig) ---------------------------------------- */
/* valid_i = sig_i in G1 \ {O}  and  pk_i in G2 \ {O} (on curve, torsion free)  and
 *           e(sig_i, -G2gen) * e(H(msg_i), pk_i) == 1 */
int blsbn254_verify_batch(blsbn254_ctx* ctx, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off,
                          const uint8_t* sigs, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* valid_bitmap);
/* valid = prod_i e(H(msg_i), pk_i) * e(agg_sig, -G2gen) == 1, every pk_i valid, n >= 1.
 * PRECONDITION (rogue-key protection, IETF BLS section 3): this is CoreAggregateVerify -- the algebraic check only.  With a
 * basic-scheme tag (the suggested default DST ends in _NUL_) the CALLER must reject batches with repeated messages;
 * alternatively use a proof-of-possession tag (_POP_) and only keys whose proof passed blsbn254_pop_verify_batch.
 * The library does not detect duplicate messages (the reference has no BLS layer to pin either behaviour). */
int blsbn254_aggregate_verify(blsbn254_ctx* ctx, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, size_t n,
                              const uint8_t agg_sig[64], const uint8_t* dst, size_t dst_len, int* valid);
/* Repeated signers.  blsbn254_verify_batch (and _dev) de-duplicates the public keys of a batch on the GPU (hash table over
 * the 128-byte encodings, full comparison on every hit) and, when at most half of them are distinct (and at most 65536),
 * validates each DISTINCT key once and turns it into its table of 88 line-coefficient triples -- G2Prepared::from,
 * pairings.rs:609-660 (E6: 88 entries, not 68) -- beside hash-to-G1; the tuples then run a table-only Miller loop
 * (multi_miller_loop over prepared terms, pairings.rs:808-857) in key-sorted order.  Same bitmap as the exact
 * per-tuple path, which batches of mostly distinct keys keep taking.  BLSBN254_AUTO_PREPARE=0 in the environment or
 * blsbn254_set_auto_prepare(ctx, 0) forces the exact path; blsbn254_path_stats counts the chunks each path served.
 * Small and mid-size calls.  A launch that does not fill the chip is bound by the latency of one lane's chain, so such calls
 * (verify, pairing, Miller loop, final exponentiation, aggregate verify) run the Miller loop and the hard part of the final
 * exponentiation with one WAVE per tuple up to 2048 tuples and with THREE LANES per tuple (a DPP quad: the three Fp6 products
 * of an Fp12 product side by side) up to 16384 -- same values, same bytes -- and verify chunks of those sizes take the
 * prepared-key path whatever their keys (the tables are what those Miller loops read; the per-key preparation itself runs four
 * lanes per key).  BLSBN254_WIDE_FE=0 / BLSBN254_TRI_MAX=0 switch the two forms off, BLSBN254_WIDE_FE_MAX=<n> /
 * BLSBN254_TRI_MAX=<n> move the limits, BLSBN254_QUAD_PREP=0 keeps the per-key preparation on one lane per key. */
int blsbn254_set_auto_prepare(blsbn254_ctx* ctx, int on);
int blsbn254_path_stats(blsbn254_ctx* ctx, uint64_t out[2] /* prepared, exact */);
/* blsbn254_aggregate_verify over repeated keys (auto-prepare on; at most half of the n >= 1024 keys distinct, or at most 16383 pairs):
 * by bilinearity in the first argument  prod_{i: pk_i = pk} e(H(msg_i), pk) = e(sum_i H(msg_i), pk)  -- exact, no randomness --
 * the H(msg_i) of every distinct key are summed in G1 (n additions) and ONE Miller loop per distinct key runs
 * (multi_miller_loop over u + 1 prepared terms, pairings.rs:808-857).  Same boolean; the Miller value differs from the product
 * of the n per-pair values, which blsbn254_aggregate_partial keeps computing bit-exactly for the sharded API.
 * out[0] = calls served by key sums, out[1] = calls served pair by pair. */
int blsbn254_aggregate_path_stats(blsbn254_ctx* ctx, uint64_t out[2]);
/* The explicit form: prepare u keys once (device-resident, owned by the handle, tied to ctx), then verify any number of
 * batches against them, naming the key of every tuple by its index (key_idx[i] < u, else BLSBN254_E_ARG).  A key that
 * does not decode, is the identity, is off the curve or outside the r-torsion makes its tuples invalid (bit cleared). */
typedef struct blsbn254_g2prepared blsbn254_g2prepared;
int blsbn254_g2_prepare_batch(blsbn254_ctx* ctx, const uint8_t* pks /* u*128 */, size_t u, blsbn254_g2prepared** out);
void blsbn254_g2prepared_destroy(blsbn254_g2prepared* keys);
size_t blsbn254_g2prepared_count(const blsbn254_g2prepared* keys);
int blsbn254_g2prepared_valid(blsbn254_ctx* ctx, const blsbn254_g2prepared* keys, uint8_t* ok_bitmap /* ceil(u/8) */);
int blsbn254_verify_batch_prepared(blsbn254_ctx* ctx, const blsbn254_g2prepared* keys, const uint32_t* key_idx,
                                   const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs, size_t n,
                                   const uint8_t* dst, size_t dst_len, uint8_t* valid_bitmap);
/* multi_miller_loop(&[(&G1Affine, &G2Prepared)]) -> MillerLoopResult (pairings.rs:808-857) with the second members given as
 * prepared keys by index: the 384-byte product of the n Miller values (same bytes as blsbn254_multi_miller_loop on the
 * plain points).  A pair whose G1 member is the identity is skipped, as in the reference; a G1 that does not decode
 * returns BLSBN254_ERR_G1, a referenced key that is not a valid G2 point returns BLSBN254_ERR_G2. */
int blsbn254_multi_miller_loop_prepared(blsbn254_ctx* ctx, const blsbn254_g2prepared* keys, const uint32_t* key_idx,
                                        const uint8_t* g1 /* n*64 */, size_t n, uint8_t ml_out[384]);
/* blsbn254_aggregate_verify with the public keys named by index into a prepared table: only 4 bytes per pair cross the
 * boundary instead of 128, and no key is validated or turned into lines again.  Same preconditions as aggregate_verify. */
int blsbn254_aggregate_verify_prepared(blsbn254_ctx* ctx, const blsbn254_g2prepared* keys, const uint32_t* key_idx,
                                       const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t agg_sig[64],
                                       const uint8_t* dst, size_t dst_len, int* valid);
/* The same check split for sharding over GPUs: every rank reduces ITS (pk_i, msg_i) to one
 * Fp12 partial product prod_i ML(H(msg_i), pk_i) (384 B; n = 0 gives Fp12::ONE) and reports whether all its
 * public keys validated; the partials are exchanged (all-gather of 384-byte records, Fp12 multiplication is
 * not an RCCL reduce op) and any rank finishes: valid = FE(prod partials * ML(agg_sig, -G2gen)) == 1. */
int blsbn254_aggregate_partial(blsbn254_ctx* ctx, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, size_t n,
                               const uint8_t* dst, size_t dst_len, uint8_t ml_out[384], int* all_pks_ok);
int blsbn254_aggregate_finish(blsbn254_ctx* ctx, const uint8_t* partials /* k*384 */, size_t k, const uint8_t agg_sig[64], int* valid);
/* One shard may carry the aggregate signature's pair as well: its partial is then prod_i ML(H(msg_i), pk_i) * ML(agg_sig, -G2gen)
 * (the signature joins the batch as one more pair instead of a one-lane launch of its own), *sig_ok = the signature decodes,
 * is not the identity and is on the curve, and blsbn254_aggregate_finish is called with agg_sig = NULL. */
int blsbn254_aggregate_partial_with_sig(blsbn254_ctx* ctx, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, size_t n,
                                        const uint8_t* dst, size_t dst_len, const uint8_t agg_sig[64], uint8_t ml_out[384],
                                        int* all_pks_ok, int* sig_ok);
/* Same result as blsbn254_verify_batch (same bitmap), computed with random linear combinations (SURVEY.md 8f rank 4;
 * bilinearity as in Gt::mul_by_scalar pairings.rs:585-600, multi_miller_loop over prepared keys pairings.rs:808-857).
 *
 * Batches that repeat keys (at most half as many distinct keys as tuples -- the same rule as verify_batch's prepared-key
 * path): every run of one key, in key-sorted order, is cut into chunks of at most G tuples (default 16,
 * blsbn254_set_rlc_group / BLSBN254_RLC_GROUP), and a chunk C with key pk is checked as ONE virtual tuple
 *   e(sum_{i in C} r_i sig_i, -G2gen) * e(sum_{i in C} r_i H(msg_i), pk) == 1
 * on the prepared-key verify path: 1/G as many Miller loops and final exponentiations.  The eligible tuples of a chunk that
 * fails are re-verified one by one on the exact prepared-key path, so a bit can only differ from verify_batch's when an
 * invalid chunk passes: probability about 2^-64 per chunk for a seed the adversary cannot predict.  Tuples whose
 * signature does not decode / is the identity / is off the curve, or whose key fails its checks, are reported invalid
 * directly and contribute to no sum.
 * Weights: r_i = a_i + b_i lambda (lambda = eigenvalue of the G1 endomorphism (x, y) -> (beta x, y)), with the 32-bit
 * a_i, b_i taken from SHA-256(seed || i || pk_i || sig_i || H(msg_i)); the 2^64 pairs give 2^64 distinct weights mod r.
 * Batches of distinct keys: groups of 16 tuples in the caller's order share the signature-side Miller loop and the final
 * exponentiation (host-pointer entry point; 64-bit weights), or take the exact path (device entry point).
 * seed = NULL (the production setting): the library draws 32 bytes from the OS (getrandom) inside the call, i.e. after
 * the batch is fixed.  A caller-supplied seed exists for reproducible tests; soundness then rests on that seed being
 * fresh and secret -- never reuse one, never derive it from public data. */
int blsbn254_verify_batch_rlc(blsbn254_ctx* ctx, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off,
                              const uint8_t* sigs, size_t n, const uint8_t* dst, size_t dst_len,
                              const uint8_t seed[32], uint8_t* valid_bitmap);
/* The same with every buffer device-resident (the layout of blsbn254_verify_batch_dev); seed and dst are host pointers. */
int blsbn254_verify_batch_rlc_dev(blsbn254_ctx* ctx, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                                  const uint8_t* d_sigs, size_t n, const uint8_t* dst, size_t dst_len,
                                  const uint8_t seed[32], uint8_t* d_valid_bitmap);
/* tuples per chunk of the repeated-key variant: 2 <= group <= 4096 fixes it; 0 = automatic (the default): 16, raised to at
 * most 32 when the larger chunk saves a whole round of waves on the device (chunk count just above a multiple of CUs x 256) */
int blsbn254_set_rlc_group(blsbn254_ctx* ctx, size_t group);
/* The key round (on by default; BLSBN254_RLC_KEY_ROUND=0 or blsbn254_set_rlc_key_round(ctx, 0) skips it): before any chunk is
 * checked, ALL tuples of every key are checked as one virtual tuple per key -- u checks, run with one workgroup (two waves) per check.  A batch
 * without invalid signatures, the usual case, is decided there (262144 tuples over 1024 keys: 9.5 instead of 16.3 ms); if any key
 * fails, only the chunks of the failed keys are checked as described above (the key round then cost about 5 ms extra; a caller
 * whose batches keep failing it does not keep paying: after a failure the next 2, then 4, 8, 16 batches skip it, a pass resets
 * the back-off, and so does this call).  Same weights, same error bound. */
int blsbn254_set_rlc_key_round(blsbn254_ctx* ctx, int on);
/* counters since context creation: out[0] tuples on the repeated-key path, out[1] chunks checked, out[2] tuples re-verified
 * exactly after their chunk failed, out[3] tuples of distinct-key batches (no chunks), out[4] key rounds run, out[5] key rounds
 * that decided their batch (no chunk was checked) */
int blsbn254_rlc_stats(blsbn254_ctx* ctx, uint64_t out[6]);
/* impl Sum for G1Projective, g1.rs:561-565 */
int blsbn254_aggregate_sigs(blsbn254_ctx* ctx, const uint8_t* sigs, size_t n, uint8_t out[64]);
/* impl Sum for G2Projective, g2.rs:579-583: out = pk_0 + ... + pk_(n-1) (uncompressed; the identity encoding for n == 0).
 * Every point must decode and lie on the curve (else BLSBN254_ERR_G2); subgroup membership is not required of the terms
 * (the reference's Sum adds whatever G2Projective values it is given). */
int blsbn254_aggregate_pks(blsbn254_ctx* ctx, const uint8_t* pks /* n*128 */, size_t n, uint8_t out[128]);
/* IETF FastAggregateVerify, min-sig variant (build-defined composition, SURVEY.md 1: the reference has no BLS layer): ONE message
 * signed by n keys, e(sig, -G2gen) * e(H(msg), pk_0 + ... + pk_(n-1)) == 1 -- the sum by Sum for G2Projective (g2.rs:579-583),
 * then the CoreVerify of blsbn254_verify_batch on it (G1Projective::hash g1.rs:910-919, pairing pairings.rs:760-802), including
 * its KeyValidate on the SUM (not the identity, in the r-torsion).  As in the IETF procedure the individual keys are only
 * required to be curve points -- proof of possession (blsbn254_pop_verify_batch) is the caller's precondition; a key that
 * does not decode or is off the curve makes the result invalid, n == 0 likewise.  *valid = 0 / 1. */
int blsbn254_fast_aggregate_verify(blsbn254_ctx* ctx, const uint8_t* pks /* n*128 */, size_t n, const uint8_t* msg, size_t msg_len,
                                   const uint8_t sig[64], const uint8_t* dst, size_t dst_len, int* valid);
/* The same for n_groups independent (key set, message, signature) groups in one call -- the validator workload (thousands of
 * aggregates, each signed by hundreds of keys): group g owns the keys pks[128 * key_off[g] .. 128 * key_off[g + 1]) (key_off:
 * n_groups + 1 non-decreasing element offsets, host array), the message msgs[off[g] .. off[g + 1]) and sigs[64 g ..].  The
 * key sums run as segmented sums on the device (one lane per 16-point chunk, level by level), the n_groups sums then go
 * through the verify_batch pipeline.  Bit g of valid_bitmap (LSB-first) = group g verifies; a group with an undecodable /
 * off-curve key or with no key at all is invalid, never an error. */
int blsbn254_fast_aggregate_verify_batch(blsbn254_ctx* ctx, const uint8_t* pks, const uint64_t* key_off /* n_groups+1 */,
                                         const uint8_t* msgs, const uint64_t* off /* n_groups+1 */, const uint8_t* sigs /* n_groups*64 */,
                                         size_t n_groups, const uint8_t* dst, size_t dst_len, uint8_t* valid_bitmap /* ceil(n_groups/8) */);
/* Mul<Scalar> for G1Projective (g1.rs:518-534, multiply :821-841) and G2Projective (g2.rs:866-886), element-wise:
 * out_i = [k_i] P_i.  Points uncompressed, scalars 32 bytes big-endian (scalar.rs:229-233) and < r.  A point that does not
 * decode or is off the curve returns BLSBN254_ERR_G1 / BLSBN254_ERR_G2, a scalar >= r BLSBN254_ERR_SCALAR (the reference's
 * types make both unrepresentable); the identity and k = 0 give the identity.  Same point as the reference's 255-step
 * double-and-add, computed with 4-bit windows over the complete RCB formulas.  Not constant time. */
int blsbn254_g1_mul_batch(blsbn254_ctx* ctx, const uint8_t* g1 /* n*64 */, const uint8_t* scalars /* n*32 */, size_t n, uint8_t* out /* n*64 */);
int blsbn254_g2_mul_batch(blsbn254_ctx* ctx, const uint8_t* g2 /* n*128 */, const uint8_t* scalars /* n*32 */, size_t n, uint8_t* out /* n*128 */);
/* sum_i lambda_i * sig_i with Lagrange coefficients at 0 for the t distinct non-zero ids
 * (Mul<Scalar> g1.rs:518-534 + Sum; Fr arithmetic scalar.rs:523-548) */
int blsbn254_threshold_combine(blsbn254_ctx* ctx, const uint8_t* ids, const uint8_t* partial_sigs, size_t t, uint8_t out_sig[64]);
/* The Lagrange coefficients alone: out[i] = prod_{j != i} x_j / (x_j - x_i) as 32 bytes big-endian (Scalar mul / invert,
 * scalar.rs:523-548, :216-219).  ids that do not decode (>= r), are zero or repeat return BLSBN254_ERR_SCALAR. */
int blsbn254_lagrange_at_zero(blsbn254_ctx* ctx, const uint8_t* ids, size_t t, uint8_t* out /* t*32 */);

/* ---- signing side (also used to generate large synthetic batches) ----------- */
/* sig_i = [sk_i] H(msg_i): G1Projective::hash (g1.rs:910-919) + Mul<Scalar> (g1.rs:518-534, :821-841).
 * sks = n x 32 B big-endian, each < r (else BLSBN254_ERR_SCALAR).  Not constant time (the reference's
 * ladder is; a verification engine handles public data -- do not use with production secrets).  The staged
 * secret keys are zeroed in device memory before the signing-side entry points return. */
int blsbn254_sign_batch(blsbn254_ctx* ctx, const uint8_t* sks, const uint8_t* msgs, const uint64_t* off, size_t n,
                        const uint8_t* dst, size_t dst_len, uint8_t* sigs_out);
/* pk_i = [sk_i] G2gen: Mul<Scalar> for G2Projective (g2.rs:866-886) */
int blsbn254_sk_to_pk_batch(blsbn254_ctx* ctx, const uint8_t* sks, size_t n, uint8_t* pks_out);
/* sk_i = KeyGen(IKM_i, key_info): IETF BLS KeyGen (draft-irtf-cfrg-bls-signature-05 section 2.3) over
 * HKDF-SHA-256 with the salt the reference names (KEYGEN_SALT, helpers.rs:3) and L = 48; the reference
 * has the constant but no procedure.  ikm = n x ikm_len bytes, ikm_len >= 32 (else BLSBN254_E_ARG);
 * sks_out = n x 32 B big-endian, each in [1, r). */
int blsbn254_keygen_batch(blsbn254_ctx* ctx, const uint8_t* ikm, size_t ikm_len, size_t n,
                          const uint8_t* key_info, size_t key_info_len, uint8_t* sks_out);
/* Scalar::hash<ExpandMsgXmd<Sha256>> (scalar.rs:554-563): OS2IP(expand_message_xmd(msg, dst, 48)) mod r,
 * out = n x 32 B big-endian.  (The reference's Reduce<U384>, scalar.rs:393-402, subtracts r once and
 * truncates; the RFC 9380 hash_to_field value it is meant to produce is what is returned here.) */
int blsbn254_hash_to_scalar_batch(blsbn254_ctx* ctx, const uint8_t* msgs, const uint64_t* off, size_t n,
                                  const uint8_t* dst, size_t dst_len, uint8_t* out);
/* Proof of possession (draft section 3.3.2 / 3.3.3): proof_i = [sk_i] H(pk_i bytes) with the 128-byte
 * uncompressed public key as the message and a caller-supplied POP tag as DST; pop_verify is CoreVerify of
 * (pk_i, pk_i bytes, proof_i), bitmap as in verify_batch. */
int blsbn254_pop_prove_batch(blsbn254_ctx* ctx, const uint8_t* sks, size_t n, const uint8_t* dst, size_t dst_len,
                             uint8_t* proofs_out);
int blsbn254_pop_verify_batch(blsbn254_ctx* ctx, const uint8_t* pks, const uint8_t* proofs, size_t n,
                              const uint8_t* dst, size_t dst_len, uint8_t* valid_bitmap);

/* ---- compressed wire codecs ------------------------------------------------- */
/* G1 32 B: x with bit 255 = parity of y (G1Affine::to_compressed g1.rs:283-288); decompression picks the root
 * whose parity equals the flag -- the corrected rule: the reference's from_compressed (g1.rs:311-328) selects
 * on y.is_high() ^ flag and mis-decodes about half of G1.  G2 64 B: x.c1 || x.c0 with
 * bit 255 = sgn0(y) (g2.rs:274-283, :309-340).  An operand that does not decode (coordinate >= p, or no
 * point with that x) returns BLSBN254_ERR_G1 / _G2.  Decompressed points are on the curve by construction;
 * G2 subgroup membership still has to be checked (blsbn254_g2_check_batch). */
int blsbn254_g1_compress_batch(blsbn254_ctx* ctx, const uint8_t* g1 /* n*64 */, size_t n, uint8_t* out /* n*32 */);
int blsbn254_g1_decompress_batch(blsbn254_ctx* ctx, const uint8_t* in /* n*32 */, size_t n, uint8_t* g1 /* n*64 */);
int blsbn254_g2_compress_batch(blsbn254_ctx* ctx, const uint8_t* g2 /* n*128 */, size_t n, uint8_t* out /* n*64 */);
int blsbn254_g2_decompress_batch(blsbn254_ctx* ctx, const uint8_t* in /* n*64 */, size_t n, uint8_t* g2 /* n*128 */);

/* ---- device-resident variants (plumbing for callers that already hold the batch in HBM) ----- */
/* All d_* pointers are device pointers on ctx's GPU.  Work is enqueued on ctx's stream and is
 * complete after blsbn254_ctx_synchronize().  d_valid_bitmap needs ceil(n/8) bytes.
 * blsbn254_verify_batch_dev does not wait for the device in steady state.  The number of distinct public keys sizes the per-key
 * tables and chooses between the prepared-key and the exact per-tuple pipeline; the FIRST call on a context (and every call
 * after one that took the exact path, and calls of more than one 4 Mi-tuple chunk) reads that 4-byte count back before it enqueues the
 * pipeline.  After a call that took the prepared-key path, the next call is enqueued on the ASSUMPTION that its keys repeat
 * likewise (tables reserved for twice the last count; the device-side count bounds the per-key work) and returns at once; count
 * and a check of the assumption come back behind an event and are read by the next entry point on the context or by
 * blsbn254_ctx_synchronize.  If the assumption failed (a new key set, more keys than reserved), that call is re-run on the
 * counting path there -- before blsbn254_ctx_synchronize returns -- so the bitmap is final after blsbn254_ctx_synchronize as
 * always (a caller that only synchronises the raw stream of blsbn254_ctx_stream must not rely on it).  Up to four calls stay in
 * flight; the caller's device buffers must stay untouched until blsbn254_ctx_synchronize.  BLSBN254_ASYNC_VERIFY=0 or
 * blsbn254_set_async_verify(ctx, 0) makes every call count first; blsbn254_async_stats: out[0] chunks enqueued on the
 * assumption, out[1] of them re-run.  Launch size picks the kernels: up to 2048 tuples one workgroup of two waves per tuple,
 * up to 16384 three lanes per tuple, beyond one lane per tuple -- same values, same bitmap. */
int blsbn254_set_async_verify(blsbn254_ctx* ctx, int on);
int blsbn254_async_stats(blsbn254_ctx* ctx, uint64_t out[2] /* enqueued on the assumption, re-run */);
int blsbn254_verify_batch_dev(blsbn254_ctx* ctx, const uint8_t* d_pks, const uint8_t* d_msgs, const uint64_t* d_off,
                              const uint8_t* d_sigs, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* d_valid_bitmap);
int blsbn254_pairing_batch_dev(blsbn254_ctx* ctx, const uint8_t* d_g1, const uint8_t* d_g2, size_t n, uint8_t* d_gt,
                               uint8_t* d_status /* n bytes or NULL */);
int blsbn254_ctx_synchronize(blsbn254_ctx* ctx);
void* blsbn254_ctx_stream(blsbn254_ctx* ctx); /* the hipStream_t */

/* ---- Gt group operations and the field-primitive debug ABI ------------------------------------------------ */
/* Gt is written multiplicatively: the reference's `Gt + Gt` is the Fp12 product (pairings.rs:245-381 trait glue over
 * fp12.rs:203-210) and Gt::mul_by_scalar (pairings.rs:585-600) is gt^k, k = 32 bytes big-endian (any 256-bit value).
 * Operands that do not decode (a coefficient >= p) return BLSBN254_ERR_GT. */
int blsbn254_gt_mul_batch(blsbn254_ctx* ctx, const uint8_t* a /* n*384 */, const uint8_t* b /* n*384 */, size_t n, uint8_t* out);
int blsbn254_gt_pow_batch(blsbn254_ctx* ctx, const uint8_t* gt /* n*384 */, const uint8_t* scalars /* n*32 */, size_t n, uint8_t* out);
/* One field / tower primitive applied element-wise to n operands: the isolated parity pin of the device arithmetic
 * (the reference's fp6.rs / fp12.rs have no test vectors; tests fuzz this against the CPU oracle).
 * Element bytes are canonical big-endian coefficients: Fp 32 B; Fp2 64 B = c0 || c1 (NOT the c1 || c0 wire order of G2);
 * Fp6 192 B = c0.c0 c0.c1 c1.c0 c1.c1 c2.c0 c2.c1; Fp12 384 B in Gt::to_repr order.  b is read by the binary ops only
 * (pass NULL otherwise); a coefficient >= p returns BLSBN254_ERR_GT. */
#define BLSBN254_OP_FP_MUL 0          /* Fp::multiply        fp.rs:404-407 */
#define BLSBN254_OP_FP_SQR 1          /* Fp::square          fp.rs:409-412 */
#define BLSBN254_OP_FP_INV 2          /* Fp::invert          fp.rs:207-210 (0 -> 0) */
#define BLSBN254_OP_FP_ADD 3          /* fp.rs:388 */
#define BLSBN254_OP_FP_SUB 4          /* fp.rs:396 */
#define BLSBN254_OP_FP_NEG 5          /* fp.rs:400 */
#define BLSBN254_OP_FP_SQRT 6         /* a root of a (sqrt_ratio, fp.rs:212-243: a^((p+1)/4)), or 0 when a is not a square */
#define BLSBN254_OP_FP_IS_SQUARE 7    /* is_square fp.rs:428-431 as the field element 1 / 0 */
#define BLSBN254_OP_FP_MUL_3B 8       /* mul_by_3b = 9 a      fp.rs:414 */
#define BLSBN254_OP_FP2_MUL 16        /* fp2.rs:377-390 */
#define BLSBN254_OP_FP2_SQR 17        /* fp2.rs:392-402 */
#define BLSBN254_OP_FP2_INV 18        /* fp2.rs:161-166 */
#define BLSBN254_OP_FP2_MUL_XI 19     /* times the Fp6 non-residue 9 + u (E1: not fp2.rs:415-420's 1 + u) */
#define BLSBN254_OP_FP2_CONJ 20       /* fp2.rs:428-437 */
#define BLSBN254_OP_FP2_SQRT 21       /* fp2.rs:172-218, a root or 0 */
#define BLSBN254_OP_FP6_MUL 32        /* fp6.rs:225-242 */
#define BLSBN254_OP_FP6_SQR 33        /* a * a (fp6.rs:244-259 is defective, E2) */
#define BLSBN254_OP_FP6_INV 34        /* fp6.rs:261-287, denominator corrected */
#define BLSBN254_OP_FP6_MUL_V 35      /* mul_by_non_residue fp6.rs:146-152 */
#define BLSBN254_OP_FP12_MUL 48       /* fp12.rs:203-210 */
#define BLSBN254_OP_FP12_SQR 49       /* fp12.rs:170-180 */
#define BLSBN254_OP_FP12_INV 50       /* fp12.rs:212-219 */
#define BLSBN254_OP_FP12_CONJ 51      /* fp12.rs:131-137 */
#define BLSBN254_OP_FP12_FROB1 52     /* frobenius_map^1..3 with the xi^((p^k-1)/6) constants (E3) */
#define BLSBN254_OP_FP12_FROB2 53
#define BLSBN254_OP_FP12_FROB3 54
#define BLSBN254_OP_FP12_CYC_SQR 55   /* cyclotomic_square pairings.rs:68-115 (the Granger-Scott formula on any input) */
#define BLSBN254_OP_FP12_MUL_034 56   /* sparse line product, a * (b.c0.c0 + b.c1.c0 w + b.c1.c1 v w)  (E7) */
int blsbn254_field_op_batch(blsbn254_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out);

/* ---- multi-device (SURVEY.md 8b "ctx over a device list", 8e) -------------------------------------------- */
/* A blsbn254_multi owns one ctx (stream + workspace) per entry of `devices` (HIP ordinals; an ordinal may be listed more
 * than once -- two contexts on one GPU -- which is how a single-GPU box exercises this path).  A call runs one host
 * thread per entry.  Verify tuples are independent (the per-term independence of multi_miller_loop,
 * pairings.rs:819-824): device g takes a contiguous range of the batch (boundaries at multiples of 8 tuples) and
 * there is no data-path collective.  Calls on one blsbn254_multi must be externally serialized. */
typedef struct blsbn254_multi blsbn254_multi;
int blsbn254_multi_create(const int* devices, int ndev, blsbn254_multi** out);
void blsbn254_multi_destroy(blsbn254_multi* m);
int blsbn254_multi_device_count(blsbn254_multi* m);
blsbn254_ctx* blsbn254_multi_ctx(blsbn254_multi* m, int i); /* the i-th per-device ctx (owned by m) */
const char* blsbn254_multi_last_error(blsbn254_multi* m);
/* blsbn254_verify_batch over all devices of m: same arguments, same bitmap.  Host pointers in and out; every device
 * copies its slice of the bitmap into valid_bitmap (a host gather of disjoint slices). */
int blsbn254_verify_batch_multi(blsbn254_multi* m, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off,
                                const uint8_t* sigs, size_t n, const uint8_t* dst, size_t dst_len, uint8_t* valid_bitmap);
/* blsbn254_verify_batch_rlc over all devices of m (each device verifies its contiguous shard with its own chunks and weights):
 * same bitmap as blsbn254_verify_batch_multi. */
int blsbn254_verify_batch_rlc_multi(blsbn254_multi* m, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off,
                                    const uint8_t* sigs, size_t n, const uint8_t* dst, size_t dst_len,
                                    const uint8_t seed[32], uint8_t* valid_bitmap);
/* blsbn254_aggregate_verify over all devices of m: per-device blsbn254_aggregate_partial, the 384-byte partials are
 * gathered on the host, one blsbn254_aggregate_finish on the first device. */
int blsbn254_aggregate_verify_multi(blsbn254_multi* m, const uint8_t* pks, const uint8_t* msgs, const uint64_t* off, size_t n,
                                    const uint8_t agg_sig[64], const uint8_t* dst, size_t dst_len, int* valid);
/* Device-resident N-GPU verify with the bitmap exchange of SURVEY.md 8e.  Device g already holds ITS shard in HBM:
 * d_pks[g] (counts[g] x 128), d_msgs[g] with d_off[g] (counts[g] + 1 offsets relative to d_msgs[g]), d_sigs[g]
 * (counts[g] x 64); counts[g] is a multiple of 32 for every g but the last.  Every device writes its bits into a
 * zeroed full-length word array and the arrays are summed by ONE ncclAllReduce(ncclSum, ncclUint32) over xGMI
 * (disjoint bit sets: SUM == OR; RCCL has no bitwise op), so that on return d_full_bitmap[g] (4 * ceil(N / 32) bytes
 * on device g, N = sum of counts) holds the bitmap of the whole batch in global order on EVERY device.  librccl is
 * loaded with dlopen at first use (BLSBN254_E_RCCL if absent or on an RCCL error, e.g. one ordinal listed twice).
 * Status: exercised on hardware with a ONE-rank communicator only (the builder's and the pool's test boxes have one GPU;
 * RCCL rejects a communicator that lists an ordinal twice).  The ndev > 1 branch (ncclCommInitAll over ndev ordinals, grouped
 * all-reduce, shard placement) has a test that runs as soon as two devices are visible
 * (tests/test_gpu_sharded.py::test_c_abi_multi_device_resident_rccl_two_gpus); until it has run, treat ndev > 1 here as
 * unverified and prefer blsbn254_verify_batch_multi (host gather) or one process per GPU with torch.distributed. */
int blsbn254_verify_batch_multi_dev(blsbn254_multi* m, const uint8_t* const* d_pks, const uint8_t* const* d_msgs,
                                    const uint64_t* const* d_off, const uint8_t* const* d_sigs, const size_t* counts,
                                    const uint8_t* dst, size_t dst_len, uint8_t* const* d_full_bitmap);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------ */
/* When enabled, every kernel launch on the ctx is bracketed by HIP events on the ctx stream;
 * profile_read returns, per kernel name, the number of launches and the summed duration in ms
 * since the last reset (it synchronizes the stream). */
int blsbn254_profile_enable(blsbn254_ctx* ctx, int on);
int blsbn254_profile_reset(blsbn254_ctx* ctx);
int blsbn254_profile_read(blsbn254_ctx* ctx, char* names /* max_entries*32 */, uint64_t* launches, double* total_ms, int max_entries);

/* Measured whole-chip v_mad_u64_u32 issue rate (lane-MADs per second): the VALU roofline denominator. */
int blsbn254_valu_peak(blsbn254_ctx* ctx, double* mads_per_s);
/* The whole probe: out[0] v_mad_u64_u32 lane-MADs/s, out[1] plain 32-bit VOP2 lane-ops/s, out[2] / out[3] the shader
 * clock (Hz) the chip held under each of the two probe kernels (in-kernel s_memtime / s_memrealtime), out[4] compute
 * units, out[5] the 4-cycle single-wave issue ceiling at that clock = CUs x 4 SIMDs x 16 lanes x out[2]. */
int blsbn254_valu_probe(blsbn254_ctx* ctx, double out[6]);

#ifdef __cplusplus
}
#endif
#endif
