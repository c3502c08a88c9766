// k_keygen.hip -- key derivation side: IETF KeyGen over HKDF-SHA-256, hash-to-scalar
// (Scalar::hash, scalar.rs:554-563), and the offset table that lets the proof-of-possession entry
// points sign / verify the 128-byte public keys themselves as messages.
#include "keygen.h"
#include "kernels.h"
using namespace bn;

BN_KERNEL k_keygen(const uint8_t* ikm, size_t ikm_len, size_t n, const uint8_t* key_info, size_t key_info_len,
                   uint8_t* sks, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok;
  Fr sk = lane_keygen(ikm + ikm_len * i, ikm_len, key_info, key_info_len, ok);
  fr_to_be(sks + 32 * i, sk);
  status[i] = ok ? 1 : 0;
}
BN_KERNEL k_hash_to_scalar(const uint8_t* msgs, const uint64_t* off, size_t n, const uint8_t* dst, uint32_t dst_len, uint8_t* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fr_to_be(out + 32 * i, lane_hash_to_scalar(msgs + off[i], (size_t)(off[i + 1] - off[i]), dst, dst_len));
}
BN_KERNEL k_iota_off(uint64_t* off, size_t n, uint64_t step) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) off[i] = step * i;
}
