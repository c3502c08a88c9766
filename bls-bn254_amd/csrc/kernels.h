// kernels.h -- declarations of the phase kernels (defined in k_miller.hip, k_finalexp.hip, k_misc.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#ifndef BN_WAVES_PER_SIMD
#define BN_WAVES_PER_SIMD 1
#endif
#define BN_KERNEL __global__ void __launch_bounds__(256, BN_WAVES_PER_SIMD)

#if defined(__HIPCC__)
// One validity bit per tuple, written as the wave's 64-bit ballot: lanes 0..7 store one byte each (LSB-first bitmap).
// Every lane of the wave must call it (lanes past n pass bit = false).
__device__ inline void write_ballot(uint8_t* bitmap, size_t n, size_t i, bool bit) {
  unsigned long long m = __ballot(bit);
  unsigned lane = threadIdx.x & 63;
  size_t base = (i - lane) >> 3;                       // first byte of this wave's 64 tuples
  size_t nbytes = (n + 7) >> 3;
  if (lane < 8 && base + lane < nbytes) bitmap[base + lane] = (uint8_t)(m >> (8 * lane));
}
#endif

BN_KERNEL k_hash_to_g1(const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t* dst, uint32_t dst_len,
                       int32_t* h_ws, size_t h_stride, uint8_t* out_bytes, int mode);
BN_KERNEL k_hash_to_g2(const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t* dst, uint32_t dst_len,
                       uint8_t* out_bytes, int ro);
BN_KERNEL k_g1_check(const uint8_t* g1, size_t n, uint8_t* bitmap);
BN_KERNEL k_g2_check(const uint8_t* g2, size_t n, uint8_t* ok_bytes, uint8_t* bitmap);
BN_KERNEL k_miller_1(const uint8_t* g1, const uint8_t* g2, size_t n, int32_t* f_ws, size_t f_stride, uint8_t* status);
BN_KERNEL k_miller_hpk(const int32_t* h_ws, const uint8_t* pks, size_t n, int32_t* f_ws, size_t f_stride, uint8_t* flags);
BN_KERNEL k_miller_hpk2(const int32_t* h_ws, const uint8_t* pks, size_t n, int32_t* q_ws, int32_t* f_ws, size_t f_stride, uint8_t* flags);
BN_KERNEL k_miller_hpk2p(const int32_t* h_ws, size_t h_stride, const uint32_t* kid, const int32_t* table, const uint8_t* key_ok, size_t n,
                         int32_t* f_ws, size_t f_stride, uint8_t* flags, const uint8_t* skip);
BN_KERNEL k_miller_hpk1p(const int32_t* h_ws, size_t h_stride, const uint32_t* kid, const int32_t* table, const uint8_t* key_ok, size_t n,
                         int32_t* f_ws, size_t f_stride, uint8_t* flags, const uint8_t* skip);
BN_KERNEL k_g1_to_ws_batch(const uint8_t* g1, size_t n, int32_t* h_ws, uint8_t* status);
BN_KERNEL k_g1_to_ws(const uint8_t* g1, int32_t* h_ws, size_t slot, size_t stride, uint8_t* ok);
BN_KERNEL k_miller_verify(const uint8_t* pks, const uint8_t* sigs, const int32_t* h_ws, size_t n, int32_t* f_ws, uint8_t* flags);
BN_KERNEL k_fe_easy(const int32_t* in, int32_t* out, size_t n, size_t stride);
BN_KERNEL k_fe_easy_head(const int32_t* in, size_t n, size_t stride, int32_t* head, int32_t* nu);
BN_KERNEL k_fe_inv4(int32_t* nu, size_t n, size_t stride);
BN_KERNEL k_fe_easy_tail(const int32_t* in, int32_t* out, size_t n, size_t stride, const int32_t* head, const int32_t* nu);
BN_KERNEL k_fe_expx(const int32_t* in, int32_t* out, int32_t* slots, size_t n, size_t stride);
BN_KERNEL k_fe_expx_h1(const int32_t* in, int32_t* slots, size_t n, size_t stride, int32_t* a_out, int32_t* b_out);
BN_KERNEL k_fe_expx_h2(const int32_t* in, int32_t* slots, size_t n, size_t stride, int32_t* b_in, int32_t* c_out, int32_t* b2_out, int32_t* d2_out);
BN_KERNEL k_fe_h3(const int32_t* t, const int32_t* a, const int32_t* c, const int32_t* b2, const int32_t* x0, int32_t* tmp, size_t n, size_t stride,
                  const uint8_t* flags, const uint8_t* sub_ok, uint8_t* bitmap, uint8_t* gt_bytes, int* is_one, int mode);
__global__ void __launch_bounds__(128) k_miller_wide_prepared(const uint32_t* perm, const uint32_t* kid, const uint8_t* sigs, const int32_t* h_ws, size_t h_stride,
                                                             const int32_t* table, const uint8_t* key_ok, size_t n, int32_t* f_ws, uint8_t* flags);
__global__ void __launch_bounds__(128) k_miller_wide_1p(const int32_t* h_ws, size_t h_stride, const uint32_t* kid, const int32_t* table, const uint8_t* key_ok, size_t n,
                                                       int32_t* f_ws, size_t f_stride, uint8_t* flags, const uint8_t* skip);
__global__ void __launch_bounds__(128) k_miller_wide_1(const uint8_t* g1, const uint8_t* g2, size_t n, int32_t* f_ws, size_t f_stride, uint8_t* status);
__global__ void __launch_bounds__(128) k_fe_hard_wide(const int32_t* t_ws, size_t n, size_t stride, const uint8_t* flags, const uint8_t* sub_ok,
                                                     uint8_t* one, uint8_t* gt_bytes, int* is_one, int mode);
BN_KERNEL k_fp12_from_bytes(const uint8_t* in, size_t n, int32_t* f_ws, size_t f_stride, uint8_t* status);
BN_KERNEL k_fp12_to_bytes(const int32_t* f_ws, size_t n, size_t stride, uint8_t* out);
BN_KERNEL k_fp12_mul_pairs(const int32_t* in, size_t n_in, size_t in_stride, int32_t* out, size_t out_stride);
BN_KERNEL k_g1_load(const uint8_t* g1, size_t n, int32_t* ws, uint8_t* status);
BN_KERNEL k_g1_add_pairs(const int32_t* in, size_t n_in, size_t in_stride, int32_t* out, size_t out_stride);
BN_KERNEL k_g1_to_bytes(const int32_t* ws, size_t stride, uint8_t* out);
BN_KERNEL k_sign(const uint8_t* sks, const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t* dst, uint32_t dst_len,
                 uint8_t* sigs, uint8_t* status);
BN_KERNEL k_sk_to_pk(const uint8_t* sks, size_t n, uint8_t* pks, uint8_t* status);
BN_KERNEL k_g1_mul(const uint8_t* g1, const uint8_t* scalars, size_t n, uint8_t* out, uint8_t* status);
BN_KERNEL k_g2_mul(const uint8_t* g2, const uint8_t* scalars, size_t n, uint8_t* out, uint8_t* status);
BN_KERNEL k_g2_load(const uint8_t* g2, size_t n, int32_t* ws, uint8_t* ok);
BN_KERNEL k_g2_seg_sum(const int32_t* in_ws, size_t in_stride, const uint8_t* ok_in, const uint32_t* chunk_start, const uint32_t* chunk_len, size_t m,
                       int32_t* out_ws, size_t out_stride, uint8_t* ok_out);
BN_KERNEL k_g2p_to_bytes(const int32_t* ws, size_t stride, const uint8_t* ok, size_t m, uint8_t* out, int poison);
BN_KERNEL k_keygen(const uint8_t* ikm, size_t ikm_len, size_t n, const uint8_t* key_info, size_t key_info_len,
                   uint8_t* sks, uint8_t* status);
BN_KERNEL k_hash_to_scalar(const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t* dst, uint32_t dst_len, uint8_t* out);
BN_KERNEL k_iota_off(uint64_t* off, size_t n, uint64_t step);
BN_KERNEL k_g1_codec(const uint8_t* in, size_t n, uint8_t* out, uint8_t* status, int mode);
BN_KERNEL k_g2_codec(const uint8_t* in, size_t n, uint8_t* out, uint8_t* status, int mode);
BN_KERNEL k_rlc_prep(const uint8_t* pks, const uint8_t* sigs, const int32_t* h_ws, const uint8_t* sub_ok, const uint8_t* seed,
                     size_t n, size_t n_pad, int32_t* a_ws, int32_t* b_ws, uint8_t* elig);
BN_KERNEL k_fp12_mask_one(int32_t* f_ws, size_t stride, const uint8_t* elig, size_t n_pad);
BN_KERNEL k_fp12_mul_elem(int32_t* a, size_t sa, const int32_t* b, size_t sb, size_t m);
BN_KERNEL k_g1p_to_bytes(const int32_t* ws, size_t stride, size_t m, uint8_t* out);
BN_KERNEL k_rlc_gather(const uint32_t* idx, size_t m, const uint8_t* pks, const uint8_t* sigs, const int32_t* h_ws, size_t n,
                       const uint8_t* sub_ok, uint8_t* c_pks, uint8_t* c_sigs, int32_t* c_h, uint8_t* c_sub);
BN_KERNEL k_rlc2_prep(const uint32_t* perm, const uint8_t* pks, const uint8_t* sigs, const int32_t* h_ws, size_t n, const uint8_t* seed,
                      int32_t* a_ws, int32_t* b_ws, uint8_t* sig_ok);
__global__ void k_rlc2_chunk_counts(const uint32_t* hist, uint32_t u, uint32_t G, uint32_t* cnt);
__global__ void k_rlc2_mark(const uint32_t* perm, const uint32_t* kid, const uint32_t* hist, const uint32_t* run_end, const uint32_t* chunk_base,
                            uint32_t n, uint32_t G, uint32_t* tuple_chunk, uint32_t* chunk_kid, uint32_t* chunk_start, uint32_t* chunk_len);
BN_KERNEL k_rlc2_sum(const int32_t* a_ws, const int32_t* b_ws, size_t n, const uint8_t* sig_ok, const uint32_t* chunk_start, const uint32_t* chunk_len,
                     size_t m, int32_t* sa_ws, int32_t* sb_ws, uint32_t* elig_out);
__global__ void k_rlc2_key_elig(const uint32_t* chunk_kid, const uint32_t* chunk_elig, uint32_t m, uint32_t* key_elig);
BN_KERNEL k_rlc2_virtual(const int32_t* sa_ws, const int32_t* sb_ws, size_t stride, const uint32_t* elig, const uint32_t* list, size_t cnt,
                         uint8_t* c_sig, int32_t* c_h, uint8_t* c_state);
__global__ void k_rlc2_keys_pass(const uint8_t* key_ok, const uint8_t* state, const uint8_t* isone, uint32_t u, uint8_t* key_pass, int* all_pass);
__global__ void __launch_bounds__(256) k_rlc2_chunk_need(const uint32_t* chunk_kid, const uint8_t* key_pass, uint32_t m, uint8_t* need, uint32_t* block_cnt);
__global__ void k_rlc2_chunk_pass(const uint32_t* list, const uint8_t* state, const uint8_t* isone, const uint8_t* flags, uint32_t cnt, uint8_t* chunk_pass);
__global__ void k_rlc2_valid_fast(const uint32_t* perm, const uint32_t* kid, const uint8_t* sig_ok, const uint8_t* key_ok, uint32_t n, uint8_t* valid);
__global__ void k_iota_u32(uint32_t* out, uint32_t n);
__global__ void __launch_bounds__(256) k_rlc2_resolve(const uint32_t* perm, const uint32_t* kid, const uint32_t* tuple_chunk, const uint8_t* sig_ok,
                                                      const uint8_t* key_ok, const uint8_t* chunk_pass, uint32_t n, uint8_t* valid, uint8_t* need, uint32_t* block_cnt);
__global__ void __launch_bounds__(256) k_rlc2_compact(const uint8_t* need, const uint32_t* perm, uint32_t n, const uint32_t* block_base, uint32_t* list);
BN_KERNEL k_g1_seg_sum(const int32_t* in_ws, size_t in_stride, const uint32_t* perm, const uint32_t* chunk_start, const uint32_t* chunk_len, size_t m,
                       int32_t* out_ws, size_t out_stride);
BN_KERNEL k_g1p_to_h_affine(const int32_t* in_ws, size_t in_stride, size_t u, int32_t* h_ws, size_t h_stride, uint8_t* status);
BN_KERNEL k_field_op(int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status);
BN_KERNEL k_gt_pow(const uint8_t* gt, const uint8_t* scalars, size_t n, uint8_t* out, uint8_t* status);
BN_KERNEL k_fr_decode(const uint8_t* ids, size_t t, int32_t* x_ws, uint8_t* status);
__global__ void k_kd_insert(const uint8_t* pks, uint32_t n, uint32_t* slots, uint32_t mask, uint32_t seed, uint32_t* rep);
__global__ void k_kd_assign(const uint32_t* rep, uint32_t n, uint32_t* kid, uint32_t* counter, uint32_t* keys);
__global__ void __launch_bounds__(256) k_kd_propagate(const uint32_t* rep, uint32_t n, uint32_t u, uint32_t* kid, uint32_t* hist);
__global__ void k_kd_hist(const uint32_t* kid, uint32_t n, uint32_t u, uint32_t* hist, int* bad);
__global__ void __launch_bounds__(1024) k_scan_excl(const uint32_t* hist, uint32_t u, uint32_t* cursor);
__global__ void __launch_bounds__(256) k_kd_scatter(const uint32_t* kid, uint32_t n, uint32_t u, uint32_t* cursor, uint32_t* perm);
BN_KERNEL k_g2_prepare(const uint8_t* pks, const uint32_t* keys, uint32_t u, int32_t* table, uint8_t* key_ok, const uint32_t* d_u);
BN_KERNEL k_g2_prepare_quad(const uint8_t* pks, const uint32_t* keys, uint32_t u, int32_t* table, uint8_t* key_ok, const uint32_t* d_u);
BN_KERNEL k_g2_expand(const int32_t* raw, uint32_t u, int32_t* expanded, const uint32_t* d_u);
__global__ void k_kd_decide(const uint32_t* cnt, uint32_t n, uint32_t cap, int small, uint32_t* res);
__global__ void k_prep_unsort(const uint8_t* is_one, const uint8_t* flags, const uint32_t* perm, uint32_t n, uint8_t* valid);
__global__ void __launch_bounds__(256) k_pack_bitmap(const uint8_t* valid, size_t n, uint8_t* bitmap);
BN_KERNEL k_miller_prepared(const uint32_t* perm, const uint32_t* kid, const uint8_t* sigs, const int32_t* h_ws, size_t h_stride,
                            const int32_t* table, const uint8_t* key_ok, size_t n, int32_t* f_ws, uint8_t* flags);
BN_KERNEL k_miller_tri_prepared(const uint32_t* perm, const uint32_t* kid, const uint8_t* sigs, const int32_t* h_ws, size_t h_stride,
                                const int32_t* table, const uint8_t* key_ok, size_t n, int32_t* f_ws, uint8_t* flags);
BN_KERNEL k_miller_tri_1(const uint8_t* g1, const uint8_t* g2, size_t n, int32_t* f_ws, size_t f_stride, uint8_t* status);
BN_KERNEL k_miller_tri_1p(const int32_t* h_ws, size_t h_stride, const uint32_t* kid, const int32_t* table, const uint8_t* key_ok, size_t n,
                          int32_t* f_ws, size_t f_stride, uint8_t* flags, const uint8_t* skip);
BN_KERNEL k_fe_tri_hard(const int32_t* t_ws, size_t n, size_t stride, int32_t* vals, const uint8_t* flags, const uint8_t* sub_ok,
                        uint8_t* one, uint8_t* gt_bytes, int* is_one, int mode);
BN_KERNEL k_lagrange_partial(const int32_t* x_ws, size_t t, size_t J, int32_t* pnum, int32_t* pden, uint8_t* dup);
BN_KERNEL k_lagrange_finish(const int32_t* pnum, const int32_t* pden, size_t t, size_t S, uint8_t* scalars, uint32_t* glv_ws);
BN_KERNEL k_msm_window(const uint8_t* g1, const uint32_t* glv_ws, size_t t, int32_t* part, uint8_t* status);
__global__ void __launch_bounds__(64) k_msm_finish(const int32_t* part, size_t n_chunks, uint8_t* out);
BN_KERNEL k_msm_fold(const int32_t* in, size_t n_in, int32_t* out);
__global__ void __launch_bounds__(256) k_valu_peak(uint32_t* out, uint32_t seed, int iters, int kind, uint64_t* stamps);
__global__ void k_status_reduce(const uint8_t* status, size_t n, uint8_t want_mask, uint8_t want_val, int* first_bad);
__global__ void k_and_reduce(const uint8_t* flags, const uint8_t* sub_ok, size_t n, int* all_ok);
