// k_rlc2.hip -- random-linear-combination batch verification over REPEATED public keys (bilinearity as in
// Gt::mul_by_scalar, pairings.rs:585-600, and multi_miller_loop over prepared keys, pairings.rs:808-857).
//
// A batch that repeats few keys (a validator set signing many messages) is de-duplicated and put in key-sorted order by the
// prepared-key verify path (k_keyprep.hip).  Every run of one key is cut into chunks of at most G tuples; for a chunk C with
// key pk and random weights r_i
//     prod_{i in C} [ e(sig_i, -G2gen) e(H_i, pk) ]^(r_i)  =  e( sum r_i sig_i , -G2gen ) * e( sum r_i H_i , pk )
// which is ONE ordinary verification of the "virtual tuple" (sum r_i sig_i, sum r_i H_i, pk): the chunks go through the same
// table-only Miller loop and final exponentiation as single tuples do (k_miller_prepared, k_fe_*), 1/G as many of them.  A chunk
// that passes makes all its eligible tuples valid (error probability 2^-64 per chunk); the eligible tuples of a chunk that
// fails are re-verified one by one on the exact prepared-key path, so the bitmap is verify_batch's.
//
// Weights: r_i = a_i + b_i * lambda with 32-bit a_i, b_i taken from SHA-256(seed || i || pk_i || sig_i || H(msg_i)), lambda the
// eigenvalue of phi(x, y) = (beta x, y) on G1 (glv.h).  The 2^64 pairs (a, b) give 2^64 DISTINCT residues a + b lambda mod r
// (the lattice {a + b lambda = 0} has no vector shorter than 2^126), so the soundness bound is the one of uniform 64-bit
// weights, while r_i P = a_i P + b_i phi(P) costs a 32-step joint double-and-add instead of 64 steps.
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

namespace {
__device__ inline void store_g1p(int32_t* ws, size_t stride, const G1P& p) {
  store_fp(ws, stride, p.x); store_fp(ws + 9 * stride, stride, p.y); store_fp(ws + 18 * stride, stride, p.z);
}
__device__ inline G1P load_g1p(const int32_t* ws, size_t stride) {
  return {load_fp(ws, stride), load_fp(ws + 9 * stride, stride), load_fp(ws + 18 * stride, stride)};
}
__device__ inline G1P g1_phi(const G1P& p) { return {fp_mul(p.x, fp_const(bnc::GLV_BETA)), p.y, p.z}; }
// the entry of {0, P, phi P, P + phi P} named by the bit pair (a, b)
__device__ inline G1P pick4(bool a, bool b, const G1P& id, const G1P& p, const G1P& pf, const G1P& ps) {
  return proj_select(a, proj_select(b, ps, p), proj_select(b, pf, id));
}
}  // namespace

// Sorted position s -> tuple i = perm[s].  A_s = r_i sig_i and B_s = r_i H_i, both homogeneous, limb-major with stride n;
// sig_ok[s] = the signature decodes, is not the identity and is on the curve (else A_s = B_s = identity: no contribution).
BN_KERNEL k_rlc2_prep(const uint32_t* perm, const uint8_t* pks, const uint8_t* sigs, const int32_t* h_ws, size_t n, const uint8_t* seed,
                      int32_t* a_ws, int32_t* b_ws, uint8_t* sig_ok) {
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const size_t i = perm[s];
  bool oks;
  G1A sig = g1_decode(sigs + 64 * i, oks);
  const bool ok = oks & !sig.inf & g1_on_curve(sig);
  Sha256 sh; sha256_init(sh);
  sha256_update(sh, seed, 32);
  for (int k = 0; k < 8; ++k) sha256_byte(sh, (uint8_t)((uint64_t)i >> (8 * k)));
  sha256_update(sh, pks + 128 * i, 128); sha256_update(sh, sigs + 64 * i, 64);
  for (int k = 0; k < 27; ++k) {                       // the message, through its hash point (canonical limbs of k_hash_to_g1 mode 3)
    uint32_t v = (uint32_t)h_ws[(size_t)k * n + i];
    sha256_byte(sh, (uint8_t)v); sha256_byte(sh, (uint8_t)(v >> 8)); sha256_byte(sh, (uint8_t)(v >> 16)); sha256_byte(sh, (uint8_t)(v >> 24));
  }
  uint8_t dg[32]; sha256_final(sh, dg);
  uint32_t a = ((uint32_t)dg[0] << 24) | ((uint32_t)dg[1] << 16) | ((uint32_t)dg[2] << 8) | dg[3];
  const uint32_t b = ((uint32_t)dg[4] << 24) | ((uint32_t)dg[5] << 16) | ((uint32_t)dg[6] << 8) | dg[7];
  a = (a | b) ? a : 1u;                                 // never the zero weight
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one())); gp.inf = false;
  sig.x = fp_select(ok, sig.x, gp.x); sig.y = fp_select(ok, sig.y, gp.y); sig.inf = false;
  const G1P id = proj_identity<Fp>();
  // P + phi(P) = -phi^2(P) = (beta^2 x, -y) with beta^2 = -1 - beta (lambda^2 + lambda + 1 = 0): no addition needed for the third table entry
  const G1P p1 = proj_from_affine(sig), p1f = g1_phi(p1), p1s = {fp_norm(fp_neg(fp_add(p1.x, p1f.x))), fp_norm(fp_neg(p1.y)), p1.z};
  const G1P p2 = load_g1p(h_ws + i, n), p2f = g1_phi(p2), p2s = {fp_norm(fp_neg(fp_add(p2.x, p2f.x))), fp_norm(fp_neg(p2.y)), p2.z};
  G1P acc1 = id, acc2 = id;
#pragma unroll 1
  for (int j = 31; j >= 0; --j) {                       // the two chains are independent: they fill each other's latency
    const bool ba = (a >> j) & 1, bb = (b >> j) & 1;
    acc1 = proj_dbl(acc1); acc2 = proj_dbl(acc2);
    acc1 = proj_add(acc1, pick4(ba, bb, id, p1, p1f, p1s));
    acc2 = proj_add(acc2, pick4(ba, bb, id, p2, p2f, p2s));
  }
  store_g1p(a_ws + s, n, proj_select(ok, acc1, id));
  store_g1p(b_ws + s, n, proj_select(ok, acc2, id));
  sig_ok[s] = ok ? 1 : 0;
}

// cnt[k] = chunks of key k = ceil(hist[k] / G) for k < u, cnt[u] = 0 (so that the exclusive scan over u + 1 entries ends in the total)
__global__ void k_rlc2_chunk_counts(const uint32_t* hist, uint32_t u, uint32_t G, uint32_t* cnt) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > u) return;
  cnt[k] = k < u ? (hist[k] + G - 1) / G : 0;
}
// Chunk of every sorted position, and the description of every chunk (written by the chunk's first member).
// run_end[k] = one past the last sorted position of key k (the scatter cursor after k_kd_scatter).
__global__ void k_rlc2_mark(const uint32_t* perm, const uint32_t* kid, const uint32_t* hist, const uint32_t* run_end, const uint32_t* chunk_base,
                            uint32_t n, uint32_t G, uint32_t* tuple_chunk, uint32_t* chunk_kid, uint32_t* chunk_start, uint32_t* chunk_len) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t k = kid[perm[s]], cnt = hist[k], pos = s - (run_end[k] - cnt), c = chunk_base[k] + pos / G;
  tuple_chunk[s] = c;
  if (pos % G == 0) { chunk_kid[c] = k; chunk_start[c] = s; chunk_len[c] = cnt - pos < G ? cnt - pos : G; }
}
// One lane per chunk: the sums of the chunk's weighted points (homogeneous, stride m) and the number of eligible members.
BN_KERNEL k_rlc2_sum(const int32_t* a_ws, const int32_t* b_ws, size_t n, const uint8_t* sig_ok, const uint32_t* chunk_start, const uint32_t* chunk_len,
                     size_t m, int32_t* sa_ws, int32_t* sb_ws, uint32_t* elig_out) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m) return;
  const size_t s0 = chunk_start[c];
  const uint32_t len = chunk_len[c];
  G1P A = proj_identity<Fp>(), B = A;
  uint32_t elig = 0;
#pragma unroll 1
  for (uint32_t j = 0; j < len; ++j) {
    A = proj_add(A, load_g1p(a_ws + s0 + j, n));
    B = proj_add(B, load_g1p(b_ws + s0 + j, n));
    elig += sig_ok[s0 + j];
  }
  store_g1p(sa_ws + c, m, A); store_g1p(sb_ws + c, m, B);
  elig_out[c] = elig;
}
// key_elig[k] += eligible members of every chunk of key k (key_elig zeroed by the host)
__global__ void k_rlc2_key_elig(const uint32_t* chunk_kid, const uint32_t* chunk_elig, uint32_t m, uint32_t* key_elig) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < m && chunk_elig[c]) atomicAdd(&key_elig[chunk_kid[c]], chunk_elig[c]);
}
// The virtual tuple of a group (a chunk, or all tuples of a key): sum A as 64 signature bytes, sum B homogeneous (stride cnt).
// state: 0 = no eligible member (nothing to check), 1 = check the virtual tuple, 2 = degenerate sum (treated as failed).
BN_KERNEL k_rlc2_virtual(const int32_t* sa_ws, const int32_t* sb_ws, size_t stride, const uint32_t* elig, const uint32_t* list, size_t cnt,
                         uint8_t* c_sig, int32_t* c_h, uint8_t* c_state) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cnt) return;
  const size_t c = list ? list[j] : j;                 // with a list: only the listed groups, each written at its own index
  G1P A = load_g1p(sa_ws + c, stride), B = load_g1p(sb_ws + c, stride);
  const bool degenerate = fp_is_zero(A.z) | fp_is_zero(B.z);
  const bool live = elig[c] != 0 && !degenerate;
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one())); gp.inf = false;
  A = proj_select(live, A, proj_from_affine(gp));       // placeholders keep every lane on well-formed values
  B = proj_select(live, B, proj_from_affine(gp));
  g1_encode(c_sig + 64 * c, g1_to_affine(A));
  store_g1p(c_h + c, stride, B);
  c_state[c] = elig[c] == 0 ? 0 : degenerate ? 2 : 1;
}
// The key round: *all_pass stays 1 iff every key with eligible tuples and a valid key passed its one check
// (keys that failed their own checks need none: their tuples are invalid whatever the signatures)
__global__ void k_rlc2_keys_pass(const uint8_t* key_ok, const uint8_t* state, const uint8_t* isone, uint32_t u, uint8_t* key_pass, int* all_pass) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= u) return;
  const bool pass = state[k] == 0 || key_ok[k] == 0 || (state[k] == 1 && isone[k] != 0);
  key_pass[k] = pass ? 1 : 0;
  if (!pass) atomicAnd(all_pass, 0);
}
// the chunks that still need their own check: those of the keys that failed the key round.  need[c], and the count per workgroup
// (block_cnt[nblocks] = 0) for the ordered compaction (k_rlc2_compact with perm = the identity gives the list of chunk indices)
__global__ void __launch_bounds__(256) k_rlc2_chunk_need(const uint32_t* chunk_kid, const uint8_t* key_pass, uint32_t m, uint8_t* need, uint32_t* block_cnt) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const bool nd = c < m && key_pass[chunk_kid[c]] == 0;
  if (c < m) need[c] = nd ? 1 : 0;
  const int cnt = __syncthreads_count(nd);
  if (threadIdx.x == 0) { block_cnt[blockIdx.x] = (uint32_t)cnt; if (blockIdx.x == gridDim.x - 1) block_cnt[gridDim.x] = 0; }
}
// results of a chunk round over the listed chunks (position j of the round = chunk list[j], or j without a list)
__global__ void k_rlc2_chunk_pass(const uint32_t* list, const uint8_t* state, const uint8_t* isone, const uint8_t* flags, uint32_t cnt, uint8_t* chunk_pass) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cnt) return;
  const uint32_t c = list ? list[j] : j;
  chunk_pass[c] = (state[c] == 1 && isone[j] != 0 && flags[j] != 0) ? 1 : 0;
}
// ... and then every tuple is decided by its own prechecks
__global__ void k_rlc2_valid_fast(const uint32_t* perm, const uint32_t* kid, const uint8_t* sig_ok, const uint8_t* key_ok, uint32_t n, uint8_t* valid) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t i = perm[s];
  valid[i] = (sig_ok[s] != 0 && key_ok[kid[i]] != 0) ? 1 : 0;
}
__global__ void k_iota_u32(uint32_t* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}
// After the chunk round: valid[i] for every tuple that is decided, need[s] = 1 where the exact path must decide
// (eligible tuple of a failed chunk), block_cnt[b] = number of such positions in workgroup b (block_cnt[nblocks] = 0).
__global__ void __launch_bounds__(256) k_rlc2_resolve(const uint32_t* perm, const uint32_t* kid, const uint32_t* tuple_chunk, const uint8_t* sig_ok,
                                                      const uint8_t* key_ok, const uint8_t* chunk_pass, uint32_t n, uint8_t* valid, uint8_t* need, uint32_t* block_cnt) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  bool nd = false;
  if (s < n) {
    const uint32_t i = perm[s];
    const bool elig = sig_ok[s] != 0 && key_ok[kid[i]] != 0;
    const bool pass = chunk_pass[tuple_chunk[s]] != 0;           // its chunk passed, or its whole key did in the key round
    valid[i] = (elig && pass) ? 1 : 0;
    nd = elig && !pass;
    need[s] = nd ? 1 : 0;
  }
  const int cnt = __syncthreads_count(nd);
  if (threadIdx.x == 0) { block_cnt[blockIdx.x] = (uint32_t)cnt; if (blockIdx.x == gridDim.x - 1) block_cnt[gridDim.x] = 0; }
}
// Ordered compaction: list[block_base[b] + rank within the workgroup] = perm[s] for every s with need[s] (key-sorted order kept).
__global__ void __launch_bounds__(256) k_rlc2_compact(const uint8_t* need, const uint32_t* perm, uint32_t n, const uint32_t* block_base, uint32_t* list) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const bool nd = s < n && need[s] != 0;
  const unsigned long long bal = __ballot(nd);
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[w] = (uint32_t)__popcll(bal);
  __syncthreads();
  uint32_t base = block_base[blockIdx.x];
  for (uint32_t k = 0; k < w; ++k) base += wave_cnt[k];
  if (nd) list[base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = perm[s];
}

// ---- sums of G1 points per key (aggregate verify over repeated keys: prod_{i in S_k} e(H_i, pk_k) = e(sum_{i in S_k} H_i, pk_k))
// One lane per chunk: out[c] = sum of the chunk's members.  Member s (a position in key-sorted order) is the point in column
// perm[s] (or s when perm is NULL) of the limb-major homogeneous array in_ws.
BN_KERNEL k_g1_seg_sum(const int32_t* in_ws, size_t in_stride, const uint32_t* perm, const uint32_t* chunk_start, const uint32_t* chunk_len, size_t m,
                       int32_t* out_ws, size_t out_stride) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m) return;
  const size_t s0 = chunk_start[c];
  const uint32_t len = chunk_len[c];
  G1P acc = proj_identity<Fp>();
#pragma unroll 1
  for (uint32_t j = 0; j < len; ++j) acc = proj_add(acc, load_g1p(in_ws + (perm ? perm[s0 + j] : s0 + j), in_stride));
  store_g1p(out_ws + c, out_stride, acc);
}
// u sums (homogeneous) -> affine limbs in slots 0..u-1 of an H workspace (18 x h_stride); status[k] bit 1 = the sum is the identity
// (that pair contributes 1 to the Miller product; the generator stands in for it), bit 0 always set.
BN_KERNEL k_g1p_to_h_affine(const int32_t* in_ws, size_t in_stride, size_t u, int32_t* h_ws, size_t h_stride, uint8_t* status) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= u) return;
  G1P p = load_g1p(in_ws + k, in_stride);
  const bool inf = fp_is_zero(p.z);
  G1A gp; gp.x = fp_one(); gp.y = fp_norm(fp_add(fp_one(), fp_one())); gp.inf = false;
  const G1A a = g1_to_affine(proj_select(inf, proj_from_affine(gp), p));
  store_fp(h_ws + k, h_stride, a.x); store_fp(h_ws + 9 * h_stride + k, h_stride, a.y);
  status[k] = (uint8_t)(1 | (inf ? 2 : 0));
}
