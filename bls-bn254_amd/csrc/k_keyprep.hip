// k_keyprep.hip -- public-key side of the prepared-key verify path (G2Prepared, pairings.rs:609-660):
//   * de-duplication of the batch's public keys (a batch of validator signatures repeats few keys: the synthetic workload of
//     SURVEY.md 8d draws 262144 tuples from a pool of 1024): open-addressing hash table over the 128-byte encodings with a
//     FULL comparison on every hit (a hash collision can never merge two different keys), per-key ids, key-sorted order;
//   * k_g2_prepare: once per DISTINCT key -- decode, on-curve, psi subgroup test, and the 88 line coefficient triples of the
//     optimal ate loop written as a table (g2_prepare_lines, pairing.h);
//   * the small kernels that carry validity back from key-sorted order to the caller's order.
// Own translation unit, tower / curve functions force-inlined.
#define BN_WANT_LINE_TABLE
#define BN_LINE_TABLE_QUAL static __device__ const
#include "lane_ops.h"
#include "kernels.h"
using namespace bn;

namespace {
__device__ inline uint32_t load_u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
__device__ inline uint32_t key_hash(const uint8_t* pk, uint32_t seed) {
  uint32_t h = seed ^ 0x9e3779b9u;
  for (int k = 0; k < 32; ++k) { h ^= load_u32(pk + 4 * k); h *= 0x01000193u; h = (h << 13) | (h >> 19); }
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ inline bool key_equal(const uint8_t* a, const uint8_t* b) {
  uint32_t d = 0;
  for (int k = 0; k < 32; ++k) d |= load_u32(a + 4 * k) ^ load_u32(b + 4 * k);
  return d == 0;
}
}  // namespace

// rep[i] = index of the first-inserted tuple whose public key has the same 128 bytes.  slots: M = 2^log2m entries, all 0xffffffff.
__global__ void k_kd_insert(const uint8_t* pks, uint32_t n, uint32_t* slots, uint32_t mask, uint32_t seed, uint32_t* rep) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* mine = pks + 128 * (size_t)i;
  uint32_t s = key_hash(mine, seed) & mask;
  for (;;) {                                             // terminates: the table has at least 2n slots
    // read first: with few distinct keys almost every probe finds its slot taken, and a million CAS operations on a handful
    // of addresses serialise in the L2 atomic units (1.5 ms for 2^20 tuples over 9 keys); a slot never changes once taken
    uint32_t old = __atomic_load_n(&slots[s], __ATOMIC_RELAXED);
    if (old == 0xffffffffu) old = atomicCAS(&slots[s], 0xffffffffu, i);
    if (old == 0xffffffffu) { rep[i] = i; return; }
    if (key_equal(mine, pks + 128 * (size_t)old)) { rep[i] = old; return; }
    s = (s + 1) & mask;
  }
}
// representatives take consecutive key ids (order of arrival; results do not depend on it); keys[id] = representative tuple
__global__ void k_kd_assign(const uint32_t* rep, uint32_t n, uint32_t* kid, uint32_t* counter, uint32_t* keys) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || rep[i] != i) return;
  uint32_t id = atomicAdd(counter, 1u);
  kid[i] = id; keys[id] = i;
}
// kid of every tuple; hist (optional, for the key-sorted order): tuples per key.  The counts of a workgroup are first
// gathered in LDS when the key ids fit (u <= 8192), so that a batch over a handful of keys does not send every tuple's
// increment to the same few global addresses.
__global__ void __launch_bounds__(256) k_kd_propagate(const uint32_t* rep, uint32_t n, uint32_t u, uint32_t* kid, uint32_t* hist) {
  __shared__ uint32_t lh[8192];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool use_lds = hist != nullptr && u <= 8192u;
  if (use_lds) {
    for (uint32_t k = threadIdx.x; k < u; k += blockDim.x) lh[k] = 0;
    __syncthreads();
  }
  if (i < n) {
    // id < u always on the synchronous paths (u is the counted number of keys).  The asynchronous verify path passes a CAPACITY for
    // u before the count is known on the host: ids beyond it are clamped, so that no later kernel indexes past the tables (that
    // batch's results are then discarded and the batch re-run, host_verify.hip)
    const uint32_t id0 = kid[rep[i]], id = id0 < u ? id0 : 0u;
    kid[i] = id;                                       // counted under the clamped id as well: histogram, key ids and the sorted order stay consistent
    if (use_lds) atomicAdd(&lh[id], 1u);
    else if (hist) atomicAdd(&hist[id], 1u);
  }
  if (use_lds) {
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < u; k += blockDim.x) { uint32_t v = lh[k]; if (v) atomicAdd(&hist[k], v); }
  }
}
// histogram over explicit key indices (the G2Prepared API: the caller names the key of every tuple); bad = an index >= u
__global__ void k_kd_hist(const uint32_t* kid, uint32_t n, uint32_t u, uint32_t* hist, int* bad) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (kid[i] >= u) { atomicMin(bad, (int)(i > 0x7ffffffeu ? 0x7ffffffeu : i)); return; }
  atomicAdd(&hist[kid[i]], 1u);
}
// exclusive prefix sum of hist[0..u) into cursor[0..u): one workgroup, each thread a contiguous slice
__global__ void __launch_bounds__(1024) k_scan_excl(const uint32_t* hist, uint32_t u, uint32_t* cursor) {
  __shared__ uint32_t part[1024];
  const uint32_t t = threadIdx.x, per = (u + 1023) / 1024, lo = t * per, hi = (lo + per < u) ? lo + per : u;
  uint32_t s = 0;
  for (uint32_t k = lo; k < hi; ++k) s += hist[k];
  part[t] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = t ? part[t - 1] : 0;
  for (uint32_t k = lo; k < hi; ++k) { cursor[k] = run; run += hist[k]; }
}
// perm: the tuples in key-sorted order (order within a key: arbitrary).  When the key ids fit (u <= 8192) a workgroup first
// ranks its tuples per key in LDS and reserves each key's range with ONE global atomic, so that a batch over a handful of
// keys does not serialise a million increments on a few addresses (2.4 ms for 2^20 tuples over 8 keys before).
__global__ void __launch_bounds__(256) k_kd_scatter(const uint32_t* kid, uint32_t n, uint32_t u, uint32_t* cursor, uint32_t* perm) {
  __shared__ uint32_t cnt[8192];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n && kid[i] < u;
  if (u > 8192u) {
    if (live) perm[atomicAdd(&cursor[kid[i]], 1u)] = i;
    return;
  }
  for (uint32_t k = threadIdx.x; k < u; k += blockDim.x) cnt[k] = 0;
  __syncthreads();
  const uint32_t k = live ? kid[i] : 0;
  uint32_t rank = 0;
  if (live) rank = atomicAdd(&cnt[k], 1u);
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < u; j += blockDim.x) { const uint32_t v = cnt[j]; if (v) cnt[j] = atomicAdd(&cursor[j], v); }   // count -> base
  __syncthreads();
  if (live) perm[cnt[k] + rank] = i;
}
// lines + validity of u keys.  keys == NULL: key k is pks[128 k]; else key k is the public key of tuple keys[k].
// table: u x 88 x 54 limbs, key-major contiguous.
// The two halves of the work -- the psi subgroup test and the 88 line steps -- are independent chains of about the same
// length, and a launch over a few thousand keys is the latency of one lane: they run in DIFFERENT workgroups (the first
// ceil(u / 256) workgroups compute lines, the next ceil(u / 256) validity), side by side instead of one after the other.
// Launch with 2 * ceil(u / 256) workgroups of 256.
// d_u (optional): the number of keys as counted on the device; the launch is then sized for a capacity u and keys beyond *d_u are skipped
BN_KERNEL k_g2_prepare(const uint8_t* pks, const uint32_t* keys, uint32_t u, int32_t* table, uint8_t* key_ok, const uint32_t* d_u) {
  const uint32_t nb = (u + 255u) / 256u;
  const bool check_role = blockIdx.x >= nb;
  const uint32_t k = (check_role ? blockIdx.x - nb : blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= u || (d_u && k >= *d_u)) return;
  const uint8_t* b = pks + 128 * (size_t)(keys ? keys[k] : k);
  bool okd;
  G2A q = g2_decode(b, okd);
  const bool curve_ok = okd & !q.inf & g2_on_curve(q);
  if (check_role) {
    key_ok[k] = (curve_ok & g2_torsion_free(q)) ? 1 : 0;
  } else {
    // a point that fails decoding / the curve equation is replaced by the generator (well-formed arithmetic; its table is never
    // used: key_ok is 0); a point on the curve outside the subgroup runs the line steps as it is (likewise unused)
    q.x = fp2_select(curve_ok, q.x, fp2_const(bnc::G2_GEN_X)); q.y = fp2_select(curve_ok, q.y, fp2_const(bnc::G2_GEN_Y)); q.inf = false;
    g2_prepare_lines(q, Ws{table, 1, k * (uint32_t)(BN_NEG_G2_LINES * 54 * 4), true});
  }
}
// The pair tables: lane (key k, step t) multiplies the key's line t with the fixed -G2gen line t (pairing.h line_pair_expand):
// raw (u x 88 x 54 limbs) -> expanded (u x 88 x 162 limbs).  u x 88 lanes: the part of the preparation that is not sequential.
// A lane's entry is 648 consecutive bytes, so a lane-by-lane store instruction would touch 64 different cache lines with 4 bytes
// each (the kernel was bound by exactly that: 233 MB of partial-line writes per 4096 keys).  The entries of a workgroup are one
// contiguous block of 256 x 648 bytes: every product goes to LDS first ([lane][18 limbs]) and is written out by the workgroup
// in runs of 18 consecutive dwords; the raw triples come in the same way.
BN_KERNEL k_g2_expand(const int32_t* raw, uint32_t u, int32_t* expanded, const uint32_t* d_u) {
  __shared__ int32_t stage[256 * 54];                       // the workgroup's raw triples, then one Fp2 product per lane at a time
  const uint32_t e0 = blockIdx.x * blockDim.x, e = e0 + threadIdx.x;
  uint32_t total = u * (uint32_t)BN_NEG_G2_LINES;
  if (d_u && *d_u * (uint32_t)BN_NEG_G2_LINES < total) total = *d_u * (uint32_t)BN_NEG_G2_LINES;
  if (e0 >= total) return;                                  // whole workgroups leave together (barriers below)
  const uint32_t here = total - e0 < 256u ? total - e0 : 256u;
  for (uint32_t x = threadIdx.x; x < here * 54u; x += 256u) stage[x] = raw[(size_t)e0 * 54u + x];      // coalesced: the block's triples are contiguous
  __syncthreads();
  const bool live = threadIdx.x < here;
  const uint32_t t = e % (uint32_t)BN_NEG_G2_LINES;
  const Line a = line_from_table(BN_NEG_G2_LINE_TABLE[t]);
  const Line b = line_load_limbs(Ws{stage, 1, (live ? threadIdx.x : 0u) * (uint32_t)(54 * 4), false});
  __syncthreads();
  BN_UNROLL for (int j = 0; j < 9; ++j) {                   // T0 .. T8 in the order of line_pair_expand
    const Fp2 p = j == 0 ? fp2_mul(a.c0, b.c0) : j == 1 ? fp2_mul_xi(fp2_mul(a.c2, b.c2)) : j == 2 ? fp2_mul(a.c1, b.c1) : j == 3 ? fp2_mul(a.c1, b.c2)
                : j == 4 ? fp2_mul(a.c2, b.c1) : j == 5 ? fp2_mul(a.c0, b.c1) : j == 6 ? fp2_mul(a.c1, b.c0) : j == 7 ? fp2_mul(a.c0, b.c2) : fp2_mul(a.c2, b.c0);
    fp2_store_limbs_lazy(Ws{stage, 1, threadIdx.x * (uint32_t)(18 * 4), false}, p);
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < here * 18u; x += 256u) {
      const uint32_t lane = x / 18u, limb = x - 18u * lane;
      expanded[((size_t)e0 + lane) * 162u + 18u * (uint32_t)j + limb] = stage[x];
    }
    __syncthreads();
  }
}
// valid (caller's order, one byte per tuple) from the key-sorted results: is_one[s] & flags[s] of sorted position s = perm^-1
__global__ void k_prep_unsort(const uint8_t* is_one, const uint8_t* flags, const uint32_t* perm, uint32_t n, uint8_t* valid) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  valid[perm[s]] = (is_one[s] != 0 && flags[s] != 0) ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_pack_bitmap(const uint8_t* valid, size_t n, uint8_t* bitmap) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  write_ballot(bitmap, n, i, i < n && valid[i] != 0);
}

// asynchronous verify path: res[0] = the counted number of distinct keys, res[1] = 1 when the optimistic choice holds (the keys fit
// the capacity the tables were reserved for, and -- unless the chunk is small -- at most half of the tuples' keys are distinct)
__global__ void k_kd_decide(const uint32_t* cnt, uint32_t n, uint32_t cap, int small, uint32_t* res) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint32_t u = *cnt;
  res[0] = u;
  res[1] = (u <= cap && (small || 2u * u <= n)) ? 1u : 0u;
}
