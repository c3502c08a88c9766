//! `src/inner_types/gpu.rs` — MI355X engine behind the crate's pairing front-end (feature `mi355x`).
//!
//! This file is NOT built in the engine's repository (its image has no Rust toolchain); it is the
//! reference-side half of the drop-in boundary described in INTEGRATION.md.  It depends on nothing but
//! the crate's own public byte codecs (`to_uncompressed` / `from_uncompressed`, `Gt::from_repr`) and on
//! `ffi.rs`, which is generated from `include/blsbn254.h`.
//!
//! Wiring (three edits in the crate):
//!   * `Cargo.toml`:  `[features] mi355x = ["std"]`
//!   * `build.rs`:    see `build.rs` beside this file (adds the library search path)
//!   * `src/inner_types.rs`:  `#[cfg(feature = "mi355x")] mod ffi; #[cfg(feature = "mi355x")] pub mod gpu;`
//!     and `pub use pairings::*;` (the module is private today, so `pairing`, `Gt`, `Bn254` are unreachable)
//! after which `pairings.rs` forwards its four entry points:
//!   `pairing(p, q)`                       -> `gpu::pairing(p, q)`
//!   `multi_miller_loop(terms)`            -> `gpu::multi_miller_loop(terms)`   (with `G2Prepared` = `gpu::G2Prepared`)
//!   `MillerLoopResult::final_exponentiation(&self)` -> `gpu::final_exponentiation(self)`
//!   `G1Projective::hash::<ExpandMsgXmd<Sha256>>`    -> `gpu::hash_to_g1(msg, dst)`
use super::ffi::{self, Ctx};
use super::{G1Affine, G2Affine, Gt, MillerLoopResult};
use crate::Bn254Error;
use core::ffi::c_int;
use std::sync::{Mutex, OnceLock};
use std::vec::Vec;

/// One engine context per process (GPU 0, or `BLSBN254_DEVICE`).  Calls on a context are serialized,
/// which matches the reference: it has no threads of its own and its types are `Send + Sync` markers.
struct Engine(*mut Ctx);
unsafe impl Send for Engine {}

fn engine() -> &'static Mutex<Engine> {
    static ENGINE: OnceLock<Mutex<Engine>> = OnceLock::new();
    ENGINE.get_or_init(|| {
        let device = std::env::var("BLSBN254_DEVICE").ok().and_then(|s| s.parse::<c_int>().ok()).unwrap_or(0);
        let mut ctx: *mut Ctx = core::ptr::null_mut();
        let rc = unsafe { ffi::blsbn254_ctx_create(device, &mut ctx) };
        // no CPU fallback by design: without a gfx950 device the engine refuses to start
        assert!(rc == 0 && !ctx.is_null(), "blsbn254_ctx_create failed with code {rc}");
        Mutex::new(Engine(ctx))
    })
}

/// Return codes 1..4 are the crate's own error enum in declaration order (`error.rs:4-10`).
fn check(rc: c_int) -> Result<(), Bn254Error> {
    match rc {
        0 => Ok(()),
        1 => Err(Bn254Error::InvalidScalarBytes),
        2 => Err(Bn254Error::InvalidG1Bytes),
        3 => Err(Bn254Error::InvalidG2Bytes),
        4 => Err(Bn254Error::InvalidGtBytes),
        other => panic!("blsbn254 device error {other}"),
    }
}

fn with_ctx<T>(f: impl FnOnce(*mut Ctx) -> T) -> T {
    let guard = engine().lock().expect("engine mutex poisoned");
    f(guard.0)
}

/// Concatenate messages and build the `n + 1` offsets the batch ABI takes.
fn pack(msgs: &[&[u8]]) -> (Vec<u8>, Vec<u64>) {
    let mut data = Vec::with_capacity(msgs.iter().map(|m| m.len()).sum());
    let mut off = Vec::with_capacity(msgs.len() + 1);
    off.push(0u64);
    for m in msgs {
        data.extend_from_slice(m);
        off.push(data.len() as u64);
    }
    (data, off)
}

fn bits(bitmap: &[u8], n: usize) -> Vec<bool> {
    (0..n).map(|i| (bitmap[i >> 3] >> (i & 7)) & 1 == 1).collect()
}

// ---------------------------------------------------------------- reference operator API

/// Body of `pairing` (`pairings.rs:760`), `Engine::pairing` (`:685-696`) and `pairing_with` (`:662-678`).
pub fn pairing(p: &G1Affine, q: &G2Affine) -> Gt {
    pairing_batch(core::slice::from_ref(p), core::slice::from_ref(q)).pop().expect("one pairing")
}

/// `n` independent pairings in one launch sequence.
pub fn pairing_batch(ps: &[G1Affine], qs: &[G2Affine]) -> Vec<Gt> {
    assert_eq!(ps.len(), qs.len());
    let n = ps.len();
    let g1: Vec<u8> = ps.iter().flat_map(|p| p.to_uncompressed()).collect();
    let g2: Vec<u8> = qs.iter().flat_map(|q| q.to_uncompressed()).collect();
    let mut out = vec![0u8; Gt::BYTES * n];
    with_ctx(|c| check(unsafe { ffi::blsbn254_pairing_batch(c, g1.as_ptr(), g2.as_ptr(), n, out.as_mut_ptr()) }))
        .expect("points produced by the crate's own encoders decode");
    out.chunks_exact(Gt::BYTES)
        .map(|b| Option::<Gt>::from(Gt::from_repr(b.try_into().expect("384 bytes"))).expect("engine output is canonical"))
        .collect()
}

/// Replacement for `G2Prepared` (`pairings.rs:609-660`, whose `From<G2Affine>` panics): the engine walks
/// the line steps itself, so the prepared form is the affine point.
#[derive(Copy, Clone, Debug)]
pub struct G2Prepared(pub G2Affine);

impl From<G2Affine> for G2Prepared {
    fn from(q: G2Affine) -> Self {
        Self(q)
    }
}

/// Body of `multi_miller_loop` (`pairings.rs:808-857`) and `MultiMillerLoop::multi_miller_loop` (`:706-713`):
/// the product of the Miller values of all terms (identity terms contribute 1, as `:820-823`).
pub fn multi_miller_loop(terms: &[(&G1Affine, &G2Prepared)]) -> MillerLoopResult {
    let g1: Vec<u8> = terms.iter().flat_map(|t| t.0.to_uncompressed()).collect();
    let g2: Vec<u8> = terms.iter().flat_map(|t| (t.1).0.to_uncompressed()).collect();
    let mut out = [0u8; Gt::BYTES];
    with_ctx(|c| check(unsafe { ffi::blsbn254_multi_miller_loop(c, g1.as_ptr(), g2.as_ptr(), terms.len(), out.as_mut_ptr()) }))
        .expect("points produced by the crate's own encoders decode");
    // same 12 x 32-byte layout as Gt::to_repr (pairings.rs:499-514); Gt::from_repr (:516-579) only decodes
    // the twelve coordinates, it makes no subgroup claim, so it also carries a raw Miller value
    MillerLoopResult(Option::<Gt>::from(Gt::from_repr(&out)).expect("engine output is canonical").0)
}

/// Body of `MillerLoopResult::final_exponentiation` (`pairings.rs:50-178`, trait impl `:698-704`).
pub fn final_exponentiation(f: &MillerLoopResult) -> Gt {
    let input = Gt(f.0).to_repr();
    let mut out = [0u8; Gt::BYTES];
    with_ctx(|c| check(unsafe { ffi::blsbn254_final_exponentiation(c, input.as_ptr(), 1, out.as_mut_ptr()) }))
        .expect("a field element encodes canonically");
    Option::<Gt>::from(Gt::from_repr(&out)).expect("engine output is canonical")
}

/// Body of `G1Projective::hash::<ExpandMsgXmd<Sha256>>` (`g1.rs:910-919`); `encode` (`:922-928`) is the
/// same call on `blsbn254_encode_to_g1_batch`.
pub fn hash_to_g1(msg: &[u8], dst: &[u8]) -> G1Affine {
    let off = [0u64, msg.len() as u64];
    let mut out = [0u8; G1Affine::UNCOMPRESSED_BYTES];
    with_ctx(|c| check(unsafe {
        ffi::blsbn254_hash_to_g1_batch(c, msg.as_ptr(), off.as_ptr(), 1, dst.as_ptr(), dst.len(), out.as_mut_ptr())
    }))
    .expect("hashing cannot fail");
    Option::<G1Affine>::from(G1Affine::from_uncompressed(&out)).expect("engine output is on the curve")
}

/// Body of `G2Projective::hash::<ExpandMsgXmd<Sha256>>` (`g2.rs:919-927`).
pub fn hash_to_g2(msg: &[u8], dst: &[u8]) -> G2Affine {
    let off = [0u64, msg.len() as u64];
    let mut out = [0u8; G2Affine::UNCOMPRESSED_BYTES];
    with_ctx(|c| check(unsafe {
        ffi::blsbn254_hash_to_g2_batch(c, msg.as_ptr(), off.as_ptr(), 1, dst.as_ptr(), dst.len(), out.as_mut_ptr())
    }))
    .expect("hashing cannot fail");
    Option::<G2Affine>::from(G2Affine::from_uncompressed(&out)).expect("engine output is in the subgroup")
}

/// `G2Affine::is_on_curve & is_torsion_free` (`g2.rs:404-414`, `:733-746`) for a batch of encoded keys.
pub fn g2_check_batch(pks: &[[u8; 128]]) -> Vec<bool> {
    let n = pks.len();
    let flat: Vec<u8> = pks.iter().flatten().copied().collect();
    let mut bm = vec![0u8; (n + 7) / 8];
    with_ctx(|c| check(unsafe { ffi::blsbn254_g2_check_batch(c, flat.as_ptr(), n, bm.as_mut_ptr()) })).expect("no decode errors here");
    bits(&bm, n)
}

// ---------------------------------------------------------------- the BLS layer the crate lacks

/// CoreVerify of `n` independent (public key, message, signature) tuples; signatures in G1, keys in G2.
/// Malformed or invalid tuples clear their bit; they are not errors.
pub fn verify_batch(pks: &[[u8; 128]], msgs: &[&[u8]], sigs: &[[u8; 64]], dst: &[u8]) -> Vec<bool> {
    assert!(pks.len() == msgs.len() && msgs.len() == sigs.len());
    let n = pks.len();
    let (data, off) = pack(msgs);
    let pk: Vec<u8> = pks.iter().flatten().copied().collect();
    let sg: Vec<u8> = sigs.iter().flatten().copied().collect();
    let mut bm = vec![0u8; (n + 7) / 8];
    with_ctx(|c| check(unsafe {
        ffi::blsbn254_verify_batch(c, pk.as_ptr(), data.as_ptr(), off.as_ptr(), sg.as_ptr(), n, dst.as_ptr(), dst.len(), bm.as_mut_ptr())
    }))
    .expect("per-tuple failures are reported in the bitmap");
    bits(&bm, n)
}

/// Batch verification by random linear combination: the same answers as `verify_batch` (an invalid chunk of 16 tuples passes
/// with probability about 2^-64), several times faster on batches that repeat public keys.  The weights are drawn inside the
/// library (OS randomness, after the batch is fixed).
pub fn verify_batch_rlc(pks: &[[u8; 128]], msgs: &[&[u8]], sigs: &[[u8; 64]], dst: &[u8]) -> Vec<bool> {
    assert!(pks.len() == msgs.len() && msgs.len() == sigs.len());
    let n = pks.len();
    let (data, off) = pack(msgs);
    let pk: Vec<u8> = pks.iter().flatten().copied().collect();
    let sg: Vec<u8> = sigs.iter().flatten().copied().collect();
    let mut bm = vec![0u8; (n + 7) / 8];
    with_ctx(|c| check(unsafe {
        ffi::blsbn254_verify_batch_rlc(c, pk.as_ptr(), data.as_ptr(), off.as_ptr(), sg.as_ptr(), n, dst.as_ptr(), dst.len(),
                                       std::ptr::null(), bm.as_mut_ptr())
    }))
    .expect("per-tuple failures are reported in the bitmap");
    bits(&bm, n)
}

/// CoreAggregateVerify: one aggregate signature over `n` (public key, message) pairs.
pub fn aggregate_verify(pks: &[[u8; 128]], msgs: &[&[u8]], agg_sig: &[u8; 64], dst: &[u8]) -> bool {
    assert_eq!(pks.len(), msgs.len());
    let (data, off) = pack(msgs);
    let pk: Vec<u8> = pks.iter().flatten().copied().collect();
    let mut valid: c_int = 0;
    with_ctx(|c| check(unsafe {
        ffi::blsbn254_aggregate_verify(c, pk.as_ptr(), data.as_ptr(), off.as_ptr(), pks.len(), agg_sig.as_ptr(), dst.as_ptr(), dst.len(), &mut valid)
    }))
    .expect("invalid inputs yield valid = 0");
    valid == 1
}

/// `impl Sum for G1Projective` (`g1.rs:561-565`) over encoded signatures.
pub fn aggregate_sigs(sigs: &[[u8; 64]]) -> Result<[u8; 64], Bn254Error> {
    let flat: Vec<u8> = sigs.iter().flatten().copied().collect();
    let mut out = [0u8; 64];
    with_ctx(|c| check(unsafe { ffi::blsbn254_aggregate_sigs(c, flat.as_ptr(), sigs.len(), out.as_mut_ptr()) }))?;
    Ok(out)
}

/// Lagrange interpolation at zero of `t` partial signatures (`Mul<Scalar>` `g1.rs:518-534` + `Sum`);
/// `ids` are the 32-byte big-endian participant identifiers.
pub fn threshold_combine(ids: &[[u8; 32]], partials: &[[u8; 64]]) -> Result<[u8; 64], Bn254Error> {
    assert_eq!(ids.len(), partials.len());
    let id: Vec<u8> = ids.iter().flatten().copied().collect();
    let ps: Vec<u8> = partials.iter().flatten().copied().collect();
    let mut out = [0u8; 64];
    with_ctx(|c| check(unsafe { ffi::blsbn254_threshold_combine(c, id.as_ptr(), ps.as_ptr(), ids.len(), out.as_mut_ptr()) }))?;
    Ok(out)
}

// ---------------------------------------------------------------- group operators on the crate's own types

/// Body of `impl Mul<Scalar> for G1Projective` (`g1.rs:518-534`, `multiply :821-841`), element-wise over a slice:
/// `out[i] = scalars[i] * points[i]`.  Scalars are the 32-byte big-endian encoding (`scalar.rs:229-233`).
pub fn g1_mul_batch(points: &[G1Affine], scalars: &[[u8; 32]]) -> Result<Vec<G1Affine>, Bn254Error> {
    assert_eq!(points.len(), scalars.len());
    let n = points.len();
    let p: Vec<u8> = points.iter().flat_map(|x| x.to_uncompressed()).collect();
    let k: Vec<u8> = scalars.iter().flatten().copied().collect();
    let mut out = vec![0u8; 64 * n];
    with_ctx(|c| check(unsafe { ffi::blsbn254_g1_mul_batch(c, p.as_ptr(), k.as_ptr(), n, out.as_mut_ptr()) }))?;
    Ok(out.chunks_exact(64).map(|b| Option::<G1Affine>::from(G1Affine::from_uncompressed(b.try_into().unwrap())).expect("engine output is canonical")).collect())
}

/// Body of `impl Mul<Scalar> for G2Projective` (`g2.rs:866-886`), element-wise.
pub fn g2_mul_batch(points: &[G2Affine], scalars: &[[u8; 32]]) -> Result<Vec<G2Affine>, Bn254Error> {
    assert_eq!(points.len(), scalars.len());
    let n = points.len();
    let p: Vec<u8> = points.iter().flat_map(|x| x.to_uncompressed()).collect();
    let k: Vec<u8> = scalars.iter().flatten().copied().collect();
    let mut out = vec![0u8; 128 * n];
    with_ctx(|c| check(unsafe { ffi::blsbn254_g2_mul_batch(c, p.as_ptr(), k.as_ptr(), n, out.as_mut_ptr()) }))?;
    Ok(out.chunks_exact(128).map(|b| Option::<G2Affine>::from(G2Affine::from_uncompressed(b.try_into().unwrap())).expect("engine output is canonical")).collect())
}

/// Body of `impl Sum for G2Projective` (`g2.rs:579-583`) over affine public keys: the aggregate public key.
pub fn aggregate_pks(pks: &[[u8; 128]]) -> Result<[u8; 128], Bn254Error> {
    let flat: Vec<u8> = pks.iter().flatten().copied().collect();
    let mut out = [0u8; 128];
    with_ctx(|c| check(unsafe { ffi::blsbn254_aggregate_pks(c, flat.as_ptr(), pks.len(), out.as_mut_ptr()) }))?;
    Ok(out)
}

/// IETF FastAggregateVerify (min-sig): one message signed by all of `pks` (proof of possession is the caller's precondition).
pub fn fast_aggregate_verify(pks: &[[u8; 128]], msg: &[u8], sig: &[u8; 64], dst: &[u8]) -> bool {
    let flat: Vec<u8> = pks.iter().flatten().copied().collect();
    let mut valid: c_int = 0;
    with_ctx(|c| check(unsafe {
        ffi::blsbn254_fast_aggregate_verify(c, flat.as_ptr(), pks.len(), msg.as_ptr(), msg.len(), sig.as_ptr(), dst.as_ptr(), dst.len(), &mut valid)
    }))
    .expect("invalid inputs yield valid = 0");
    valid == 1
}

/// The same for many (key set, message, signature) groups in one call: `groups[g]` = the keys of group g.
pub fn fast_aggregate_verify_batch(groups: &[&[[u8; 128]]], msgs: &[&[u8]], sigs: &[[u8; 64]], dst: &[u8]) -> Vec<bool> {
    assert!(groups.len() == msgs.len() && msgs.len() == sigs.len());
    let n = groups.len();
    let mut koff = Vec::with_capacity(n + 1);
    koff.push(0u64);
    let mut flat: Vec<u8> = Vec::new();
    for g in groups {
        for k in g.iter() { flat.extend_from_slice(k); }
        koff.push((flat.len() / 128) as u64);
    }
    let (data, off) = pack(msgs);
    let sg: Vec<u8> = sigs.iter().flatten().copied().collect();
    let mut bm = vec![0u8; (n + 7) / 8];
    with_ctx(|c| check(unsafe {
        ffi::blsbn254_fast_aggregate_verify_batch(c, flat.as_ptr(), koff.as_ptr(), data.as_ptr(), off.as_ptr(), sg.as_ptr(), n, dst.as_ptr(), dst.len(), bm.as_mut_ptr())
    }))
    .expect("per-group failures are reported in the bitmap");
    bits(&bm, n)
}

// ---------------------------------------------------------------- repeated signers: keys prepared once (G2Prepared, batched)

/// The line tables of a set of public keys, resident on the GPU (`blsbn254_g2prepared`): what `G2Prepared::from`
/// (`pairings.rs:614-660`) computes per key, for a whole validator set at once.  Tied to the process-wide engine context.
pub struct PreparedKeys(*mut ffi::PreparedKeys);
unsafe impl Send for PreparedKeys {}

impl PreparedKeys {
    pub fn new(pks: &[[u8; 128]]) -> Self {
        let flat: Vec<u8> = pks.iter().flatten().copied().collect();
        let mut h: *mut ffi::PreparedKeys = core::ptr::null_mut();
        with_ctx(|c| check(unsafe { ffi::blsbn254_g2_prepare_batch(c, flat.as_ptr(), pks.len(), &mut h) })).expect("device error");
        PreparedKeys(h)
    }
    pub fn len(&self) -> usize { unsafe { ffi::blsbn254_g2prepared_count(self.0) } }
}
impl Drop for PreparedKeys {
    fn drop(&mut self) { let _g = engine().lock(); unsafe { ffi::blsbn254_g2prepared_destroy(self.0) } }
}

/// `verify_batch` against prepared keys: tuple i was signed under key `key_idx[i]`.
pub fn verify_batch_prepared(keys: &PreparedKeys, key_idx: &[u32], msgs: &[&[u8]], sigs: &[[u8; 64]], dst: &[u8]) -> Vec<bool> {
    assert!(key_idx.len() == msgs.len() && msgs.len() == sigs.len());
    let n = msgs.len();
    let (data, off) = pack(msgs);
    let sg: Vec<u8> = sigs.iter().flatten().copied().collect();
    let mut bm = vec![0u8; (n + 7) / 8];
    with_ctx(|c| check(unsafe {
        ffi::blsbn254_verify_batch_prepared(c, keys.0, key_idx.as_ptr(), data.as_ptr(), off.as_ptr(), sg.as_ptr(), n, dst.as_ptr(), dst.len(), bm.as_mut_ptr())
    }))
    .expect("key index out of range");
    bits(&bm, n)
}

/// `multi_miller_loop` (`pairings.rs:808-857`) over (G1 point, prepared key) terms.
pub fn multi_miller_loop_prepared(keys: &PreparedKeys, terms: &[(&G1Affine, u32)]) -> Result<MillerLoopResult, Bn254Error> {
    let g1: Vec<u8> = terms.iter().flat_map(|t| t.0.to_uncompressed()).collect();
    let idx: Vec<u32> = terms.iter().map(|t| t.1).collect();
    let mut out = [0u8; 384];
    with_ctx(|c| check(unsafe { ffi::blsbn254_multi_miller_loop_prepared(c, keys.0, idx.as_ptr(), g1.as_ptr(), terms.len(), out.as_mut_ptr()) }))?;
    Ok(MillerLoopResult(Option::<Gt>::from(Gt::from_repr(&out)).expect("engine output is canonical").0))
}

// ---------------------------------------------------------------- N GPUs of one node

/// All GPUs named in `BLSBN254_DEVICES` (comma-separated HIP ordinals, default "0"): one context and one host thread per
/// GPU inside the library, contiguous shards, no data-path collective (`blsbn254_multi`).
struct MultiEngine(*mut ffi::Multi);
unsafe impl Send for MultiEngine {}

fn multi_engine() -> &'static Mutex<MultiEngine> {
    static ENGINE: OnceLock<Mutex<MultiEngine>> = OnceLock::new();
    ENGINE.get_or_init(|| {
        let devs: Vec<c_int> = std::env::var("BLSBN254_DEVICES").unwrap_or_else(|_| "0".into())
            .split(',').filter_map(|s| s.trim().parse::<c_int>().ok()).collect();
        let mut m: *mut ffi::Multi = core::ptr::null_mut();
        let rc = unsafe { ffi::blsbn254_multi_create(devs.as_ptr(), devs.len() as c_int, &mut m) };
        assert!(rc == 0 && !m.is_null(), "blsbn254_multi_create failed with code {rc}");
        Mutex::new(MultiEngine(m))
    })
}

/// `verify_batch` sharded over every configured GPU; same result as the single-GPU call.
pub fn verify_batch_multi(pks: &[[u8; 128]], msgs: &[&[u8]], sigs: &[[u8; 64]], dst: &[u8]) -> Vec<bool> {
    assert!(pks.len() == msgs.len() && msgs.len() == sigs.len());
    let n = pks.len();
    let (data, off) = pack(msgs);
    let pk: Vec<u8> = pks.iter().flatten().copied().collect();
    let sg: Vec<u8> = sigs.iter().flatten().copied().collect();
    let mut bm = vec![0u8; (n + 7) / 8];
    let guard = multi_engine().lock().expect("engine mutex poisoned");
    check(unsafe {
        ffi::blsbn254_verify_batch_multi(guard.0, pk.as_ptr(), data.as_ptr(), off.as_ptr(), sg.as_ptr(), n, dst.as_ptr(), dst.len(), bm.as_mut_ptr())
    })
    .expect("per-tuple failures are reported in the bitmap");
    bits(&bm, n)
}

/// `aggregate_verify` sharded over every configured GPU (per-GPU Fp12 partial products, one final exponentiation).
pub fn aggregate_verify_multi(pks: &[[u8; 128]], msgs: &[&[u8]], agg_sig: &[u8; 64], dst: &[u8]) -> bool {
    assert_eq!(pks.len(), msgs.len());
    let (data, off) = pack(msgs);
    let pk: Vec<u8> = pks.iter().flatten().copied().collect();
    let mut valid: c_int = 0;
    let guard = multi_engine().lock().expect("engine mutex poisoned");
    check(unsafe {
        ffi::blsbn254_aggregate_verify_multi(guard.0, pk.as_ptr(), data.as_ptr(), off.as_ptr(), pks.len(), agg_sig.as_ptr(), dst.as_ptr(), dst.len(), &mut valid)
    })
    .expect("invalid inputs yield valid = 0");
    valid == 1
}

#[cfg(test)]
mod tests {
    use super::*;
    use elliptic_curve::Group;

    /// The crate's own `pairing_test` (`pairings.rs:971-980`), which fails on the CPU path today.
    #[test]
    fn pairing_of_generators_is_gt_generator() {
        let gt = pairing(&G1Affine::generator(), &G2Affine::generator());
        assert_eq!(gt, Gt::generator());
        let r_gt = final_exponentiation(&multi_miller_loop(&[(&G1Affine::generator(), &G2Prepared(G2Affine::generator()))]));
        assert_eq!(r_gt, gt);
    }
}
