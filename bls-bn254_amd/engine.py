"""ctypes binding of the C ABI (include/blsbn254.h) with the reference crate's operator names.

Mirrors, for the hot path, the interface a user of mikelodder7/bls-bn254 sees:
pairing / multi_miller_loop / final_exponentiation (pairings.rs:760-857, :50-178),
G1Projective::hash / encode, G2Projective::hash / encode (g1.rs:910-928, g2.rs:919-936),
is_on_curve / is_torsion_free (g1.rs:383-391, g2.rs:409-414, :733-736), Sum (g1.rs:561-565), and the
Bn254Error variants (error.rs:4-10) as exceptions.  Everything runs on the GPU through
libblsbn254_hip.so; if the library or a gfx950 device is missing the calls raise -- there is no CPU
path in this package.
"""
import ctypes
import threading
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_DST = b"BLS_SIG_BN254G1_XMD:SHA-256_SVDW_RO_NUL_"
POP_DST = b"BLS_POP_BN254G1_XMD:SHA-256_SVDW_RO_POP_"
_u8p = ctypes.POINTER(ctypes.c_uint8)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_lib = None


class Bn254Error(Exception):
    """Mirror of the reference's error enum (error.rs:4-10) plus device errors (negative codes)."""
    code = None

    def __init__(self, code, detail=""):
        self.code = code
        msg = load_library().blsbn254_strerror(code).decode()
        super().__init__("%s (code %d)%s" % (msg, code, (": " + detail) if detail else ""))


class InvalidScalarBytes(Bn254Error):
    pass


class InvalidG1Bytes(Bn254Error):
    pass


class InvalidG2Bytes(Bn254Error):
    pass


class InvalidGtBytes(Bn254Error):
    pass


_ERR = {1: InvalidScalarBytes, 2: InvalidG1Bytes, 3: InvalidG2Bytes, 4: InvalidGtBytes}


def library_path():
    # BLSBN254_LIB selects another build of the SAME HIP library (A/B of compile options); never a fallback
    return os.environ.get("BLSBN254_LIB") or os.path.join(HERE, "libblsbn254_hip.so")


def load_library():
    """Loads the HIP extension.  Fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError("HIP extension %s is missing: run `python -m bls_bn254_amd.build` "
                               "(or __graft_entry__.build()); there is no CPU fallback" % path)
        # torch ships its own HIP runtime with the same SONAME as /opt/rocm's: whichever is loaded first
        # serves the whole process, and torch only finds its GPUs through its own copy.  Load torch's
        # first when it is installed so that the extension and torch share one runtime (plumbing only).
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        lib = ctypes.CDLL(path)
        lib.blsbn254_strerror.restype = ctypes.c_char_p
        lib.blsbn254_last_error.restype = ctypes.c_char_p
        lib.blsbn254_ctx_stream.restype = ctypes.c_void_p
        _lib = lib
    return _lib


def _inbuf(b, expect=None):
    if isinstance(b, np.ndarray):
        a = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1)
    else:
        a = np.frombuffer(bytes(b), dtype=np.uint8)
    if expect is not None and a.size != expect:
        raise ValueError("expected %d bytes, got %d" % (expect, a.size))
    if a.size == 0:
        a = np.zeros(1, dtype=np.uint8)
    return a, a.ctypes.data_as(_u8p)


_BIG_OUT = threading.local()


def _outbuf(n):
    """Host buffer for a call's output.  Every caller copies its result out (tobytes) before returning, so large outputs share
    one grow-only buffer per thread: a fresh multi-megabyte allocation per call is untouched mmap'ed memory, and the device-to-host
    copy into it then pays the page faults and the pinning (measured: pairing_batch(8192) 5 ms -> 24 ms depending on the
    allocator's history)."""
    if n >= (1 << 18):
        a = getattr(_BIG_OUT, "buf", None)
        if a is None or a.size < n:
            a = np.zeros(n + n // 4, dtype=np.uint8)
            _BIG_OUT.buf = a
        return a, a.ctypes.data_as(_u8p)
    a = np.zeros(max(n, 1), dtype=np.uint8)
    return a, a.ctypes.data_as(_u8p)


def pack_messages(msgs):
    """list of bytes -> (concatenated bytes, n+1 uint64 offsets)"""
    off = np.zeros(len(msgs) + 1, dtype=np.uint64)
    if msgs:
        off[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
    return b"".join(msgs), off


class Engine:
    """One context = one GPU (blsbn254_ctx): stream, workspace, resident -G2gen line table."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._ctx = ctypes.c_void_p()
        rc = self._lib.blsbn254_ctx_create(ctypes.c_int(device), ctypes.byref(self._ctx))
        if rc != 0:
            self._ctx = None
            raise Bn254Error(rc)
        self.device = device

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.blsbn254_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            detail = self._lib.blsbn254_last_error(self._ctx).decode() if rc == -2 else ""
            raise _ERR.get(rc, Bn254Error)(rc, detail)

    # ---- primitives (reference operator API)
    def pairing_batch(self, g1, g2, n):
        a, pa = _inbuf(g1, 64 * n); b, pb = _inbuf(g2, 128 * n); o, po = _outbuf(384 * n)
        self._chk(self._lib.blsbn254_pairing_batch(self._ctx, pa, pb, ctypes.c_size_t(n), po))
        return o[:384 * n].tobytes()

    def pairing(self, g1, g2):
        return self.pairing_batch(g1, g2, 1)

    def miller_loop_batch(self, g1, g2, n):
        a, pa = _inbuf(g1, 64 * n); b, pb = _inbuf(g2, 128 * n); o, po = _outbuf(384 * n)
        self._chk(self._lib.blsbn254_miller_loop_batch(self._ctx, pa, pb, ctypes.c_size_t(n), po))
        return o[:384 * n].tobytes()

    def multi_miller_loop(self, g1, g2, n):
        a, pa = _inbuf(g1, 64 * n); b, pb = _inbuf(g2, 128 * n); o, po = _outbuf(384)
        self._chk(self._lib.blsbn254_multi_miller_loop(self._ctx, pa, pb, ctypes.c_size_t(n), po))
        return o.tobytes()

    def final_exponentiation(self, ml, n=1):
        a, pa = _inbuf(ml, 384 * n); o, po = _outbuf(384 * n)
        self._chk(self._lib.blsbn254_final_exponentiation(self._ctx, pa, ctypes.c_size_t(n), po))
        return o[:384 * n].tobytes()

    def _h2c(self, fn, msgs, dst, size):
        data, off = pack_messages(msgs)
        a, pa = _inbuf(data); d, pd = _inbuf(dst); o, po = _outbuf(size * len(msgs))
        self._chk(fn(self._ctx, pa, off.ctypes.data_as(_u64p), ctypes.c_size_t(len(msgs)), pd, ctypes.c_size_t(len(dst)), po))
        return o[:size * len(msgs)].tobytes()

    def hash_to_g1_batch(self, msgs, dst): return self._h2c(self._lib.blsbn254_hash_to_g1_batch, msgs, dst, 64)
    def encode_to_g1_batch(self, msgs, dst): return self._h2c(self._lib.blsbn254_encode_to_g1_batch, msgs, dst, 64)
    def hash_to_g2_batch(self, msgs, dst): return self._h2c(self._lib.blsbn254_hash_to_g2_batch, msgs, dst, 128)
    def encode_to_g2_batch(self, msgs, dst): return self._h2c(self._lib.blsbn254_encode_to_g2_batch, msgs, dst, 128)

    def g1_check_batch(self, g1, n):
        a, pa = _inbuf(g1, 64 * n); o, po = _outbuf((n + 7) // 8)
        self._chk(self._lib.blsbn254_g1_check_batch(self._ctx, pa, ctypes.c_size_t(n), po))
        return o[:(n + 7) // 8].tobytes()

    def g2_check_batch(self, g2, n):
        a, pa = _inbuf(g2, 128 * n); o, po = _outbuf((n + 7) // 8)
        self._chk(self._lib.blsbn254_g2_check_batch(self._ctx, pa, ctypes.c_size_t(n), po))
        return o[:(n + 7) // 8].tobytes()

    # ---- BLS layer
    def verify_batch(self, pks, msgs, sigs, dst=DEFAULT_DST):
        n = len(msgs)
        data, off = pack_messages(msgs)
        a, pa = _inbuf(pks, 128 * n); m, pm = _inbuf(data); s, ps = _inbuf(sigs, 64 * n); d, pd = _inbuf(dst)
        o, po = _outbuf((n + 7) // 8)
        self._chk(self._lib.blsbn254_verify_batch(self._ctx, pa, pm, off.ctypes.data_as(_u64p), ps, ctypes.c_size_t(n), pd,
                                                  ctypes.c_size_t(len(dst)), po))
        return o[:(n + 7) // 8].tobytes()

    def set_async_verify(self, on):
        """verify_batch_dev enqueues on the previous call's key count without reading the new one back first (default on)"""
        self._chk(self._lib.blsbn254_set_async_verify(self._ctx, ctypes.c_int(1 if on else 0)))

    def async_stats(self):
        """(chunks enqueued on the assumption that the key set repeats, how many of them had to be re-run)"""
        o = (ctypes.c_uint64 * 2)()
        self._chk(self._lib.blsbn254_async_stats(self._ctx, o))
        return int(o[0]), int(o[1])

    def set_auto_prepare(self, on):
        """verify_batch's automatic key de-duplication + per-key preparation (default on); off = exact per-tuple path."""
        self._chk(self._lib.blsbn254_set_auto_prepare(self._ctx, ctypes.c_int(1 if on else 0)))

    def path_stats(self):
        """(chunks served by the prepared-key path, chunks served by the exact per-tuple path)"""
        o = (ctypes.c_uint64 * 2)()
        self._chk(self._lib.blsbn254_path_stats(self._ctx, o))
        return int(o[0]), int(o[1])

    def g2_prepare_batch(self, pks, u):
        """G2Prepared::from (pairings.rs:609-660) for u keys: a device-resident table handle (PreparedKeys)."""
        return PreparedKeys(self, pks, u)

    def verify_batch_prepared(self, keys, key_idx, msgs, sigs, dst=DEFAULT_DST):
        n = len(msgs)
        data, off = pack_messages(msgs)
        idx = np.ascontiguousarray(np.asarray(key_idx, dtype=np.uint32))
        if idx.size != n:
            raise ValueError("one key index per tuple")
        if idx.size == 0:
            idx = np.zeros(1, dtype=np.uint32)
        m, pm = _inbuf(data); s, ps = _inbuf(sigs, 64 * n); d, pd = _inbuf(dst); o, po = _outbuf((n + 7) // 8)
        self._chk(self._lib.blsbn254_verify_batch_prepared(self._ctx, keys._h, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), pm,
                                                           off.ctypes.data_as(_u64p), ps, ctypes.c_size_t(n), pd, ctypes.c_size_t(len(dst)), po))
        return o[:(n + 7) // 8].tobytes()

    def multi_miller_loop_prepared(self, keys, key_idx, g1, n):
        """multi_miller_loop over (G1 point, prepared key) terms (pairings.rs:808-857): 384-byte Miller product."""
        idx = np.ascontiguousarray(np.asarray(key_idx, dtype=np.uint32))
        if idx.size != n:
            raise ValueError("one key index per pair")
        if idx.size == 0:
            idx = np.zeros(1, dtype=np.uint32)
        a, pa = _inbuf(g1, 64 * n); o, po = _outbuf(384)
        self._chk(self._lib.blsbn254_multi_miller_loop_prepared(self._ctx, keys._h, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), pa,
                                                                ctypes.c_size_t(n), po))
        return o.tobytes()

    def aggregate_verify_prepared(self, keys, key_idx, msgs, agg_sig, dst=DEFAULT_DST):
        n = len(msgs)
        data, off = pack_messages(msgs)
        idx = np.ascontiguousarray(np.asarray(key_idx, dtype=np.uint32))
        if idx.size != n:
            raise ValueError("one key index per pair")
        if idx.size == 0:
            idx = np.zeros(1, dtype=np.uint32)
        m, pm = _inbuf(data); s, ps = _inbuf(agg_sig, 64); d, pd = _inbuf(dst)
        valid = ctypes.c_int(0)
        self._chk(self._lib.blsbn254_aggregate_verify_prepared(self._ctx, keys._h, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), pm,
                                                               off.ctypes.data_as(_u64p), ctypes.c_size_t(n), ps, pd, ctypes.c_size_t(len(dst)),
                                                               ctypes.byref(valid)))
        return bool(valid.value)

    def verify_batch_rlc(self, pks, msgs, sigs, dst=DEFAULT_DST, seed=None):
        """Same bitmap as verify_batch, via random linear combinations (one final exponentiation per 16 tuples,
        exact re-verification of failing groups).  seed = None: the library draws it from the OS inside the call
        (production); a caller seed is for reproducible tests only."""
        n = len(msgs)
        data, off = pack_messages(msgs)
        a, pa = _inbuf(pks, 128 * n); m, pm = _inbuf(data); s, ps = _inbuf(sigs, 64 * n); d, pd = _inbuf(dst)
        if seed is None:
            sd, psd = None, ctypes.cast(None, _u8p)
        else:
            sd, psd = _inbuf(seed, 32)
        o, po = _outbuf((n + 7) // 8)
        self._chk(self._lib.blsbn254_verify_batch_rlc(self._ctx, pa, pm, off.ctypes.data_as(_u64p), ps, ctypes.c_size_t(n), pd,
                                                      ctypes.c_size_t(len(dst)), psd, po))
        return o[:(n + 7) // 8].tobytes()

    def aggregate_path_stats(self):
        """(aggregate_verify calls served by per-key sums, calls served pair by pair)"""
        o = (ctypes.c_uint64 * 2)()
        self._chk(self._lib.blsbn254_aggregate_path_stats(self._ctx, o))
        return int(o[0]), int(o[1])

    def set_rlc_group(self, group):
        """tuples per chunk of the repeated-key RLC path; 0 = automatic (16, raised when that saves a round of waves)"""
        self._chk(self._lib.blsbn254_set_rlc_group(self._ctx, ctypes.c_size_t(group)))

    def rlc_stats(self):
        """dict: tuples verified through chunks, chunks checked, tuples re-verified exactly, tuples of distinct-key batches"""
        o = (ctypes.c_uint64 * 6)()
        self._chk(self._lib.blsbn254_rlc_stats(self._ctx, o))
        return {"chunked_tuples": int(o[0]), "chunks": int(o[1]), "fallback_tuples": int(o[2]), "distinct_key_tuples": int(o[3]),
                "key_rounds": int(o[4]), "key_rounds_passed": int(o[5])}

    def set_rlc_key_round(self, on):
        """RLC: check every key's whole run as one virtual tuple first (default on)"""
        self._chk(self._lib.blsbn254_set_rlc_key_round(self._ctx, ctypes.c_int(1 if on else 0)))

    def aggregate_verify(self, pks, msgs, agg_sig, dst=DEFAULT_DST):
        n = len(msgs)
        data, off = pack_messages(msgs)
        a, pa = _inbuf(pks, 128 * n); m, pm = _inbuf(data); s, ps = _inbuf(agg_sig, 64); d, pd = _inbuf(dst)
        valid = ctypes.c_int(0)
        self._chk(self._lib.blsbn254_aggregate_verify(self._ctx, pa, pm, off.ctypes.data_as(_u64p), ctypes.c_size_t(n), ps, pd,
                                                      ctypes.c_size_t(len(dst)), ctypes.byref(valid)))
        return bool(valid.value)

    def aggregate_partial(self, pks, msgs, dst=DEFAULT_DST):
        """(384-byte Fp12 partial product of ML(H(msg_i), pk_i), all public keys valid?) for one shard."""
        n = len(msgs)
        data, off = pack_messages(msgs)
        a, pa = _inbuf(pks, 128 * n); m, pm = _inbuf(data); d, pd = _inbuf(dst); o, po = _outbuf(384)
        ok = ctypes.c_int(0)
        self._chk(self._lib.blsbn254_aggregate_partial(self._ctx, pa, pm, off.ctypes.data_as(_u64p), ctypes.c_size_t(n), pd,
                                                       ctypes.c_size_t(len(dst)), po, ctypes.byref(ok)))
        return o.tobytes(), bool(ok.value)

    def aggregate_partial_with_sig(self, pks, msgs, agg_sig, dst=DEFAULT_DST):
        """(partial product times ML(agg_sig, -G2gen), all public keys valid?, signature valid?): the shard that carries
        the aggregate signature's pair; finish with aggregate_finish(partials, k, None)."""
        n = len(msgs)
        data, off = pack_messages(msgs)
        a, pa = _inbuf(pks, 128 * n); m, pm = _inbuf(data); d, pd = _inbuf(dst); s, ps = _inbuf(agg_sig, 64); o, po = _outbuf(384)
        ok = ctypes.c_int(0); sok = ctypes.c_int(0)
        self._chk(self._lib.blsbn254_aggregate_partial_with_sig(self._ctx, pa, pm, off.ctypes.data_as(_u64p), ctypes.c_size_t(n), pd,
                                                                ctypes.c_size_t(len(dst)), ps, po, ctypes.byref(ok), ctypes.byref(sok)))
        return o.tobytes(), bool(ok.value), bool(sok.value)

    def aggregate_finish(self, partials, k, agg_sig):
        a, pa = _inbuf(partials, 384 * k)
        if agg_sig is None:
            s, ps = None, ctypes.cast(None, _u8p)
        else:
            s, ps = _inbuf(agg_sig, 64)
        valid = ctypes.c_int(0)
        self._chk(self._lib.blsbn254_aggregate_finish(self._ctx, pa, ctypes.c_size_t(k), ps, ctypes.byref(valid)))
        return bool(valid.value)

    def aggregate_sigs(self, sigs, n):
        a, pa = _inbuf(sigs, 64 * n); o, po = _outbuf(64)
        self._chk(self._lib.blsbn254_aggregate_sigs(self._ctx, pa, ctypes.c_size_t(n), po))
        return o.tobytes()

    def aggregate_pks(self, pks, n):
        """impl Sum for G2Projective (g2.rs:579-583): the sum of n public keys, 128 bytes."""
        a, pa = _inbuf(pks, 128 * n); o, po = _outbuf(128)
        self._chk(self._lib.blsbn254_aggregate_pks(self._ctx, pa, ctypes.c_size_t(n), po))
        return o[:128].tobytes()

    def fast_aggregate_verify(self, pks, n, msg, sig, dst=DEFAULT_DST):
        """One message signed by n keys: e(sig, -G2gen) * e(H(msg), sum pk_i) == 1."""
        a, pa = _inbuf(pks, 128 * n); m, pm = _inbuf(msg); s, ps = _inbuf(sig, 64); d, pd = _inbuf(dst)
        valid = ctypes.c_int(0)
        self._chk(self._lib.blsbn254_fast_aggregate_verify(self._ctx, pa, ctypes.c_size_t(n), pm, ctypes.c_size_t(len(msg)), ps, pd,
                                                           ctypes.c_size_t(len(dst)), ctypes.byref(valid)))
        return bool(valid.value)

    def fast_aggregate_verify_batch(self, key_sets, msgs, sigs, dst=DEFAULT_DST):
        """key_sets: list of byte strings (each a multiple of 128 bytes: the keys of one group); msgs: one message per group;
        sigs: 64 bytes per group.  Returns the LSB-first bitmap over the groups."""
        g = len(msgs)
        if len(key_sets) != g:
            raise ValueError("one key set per message")
        koff = np.zeros(g + 1, dtype=np.uint64)
        for i, ks in enumerate(key_sets):
            if len(ks) % 128:
                raise ValueError("a key set must be a multiple of 128 bytes")
            koff[i + 1] = koff[i] + len(ks) // 128
        allk = b"".join(key_sets)
        data, off = pack_messages(msgs)
        a, pa = _inbuf(allk); m, pm = _inbuf(data); s, ps = _inbuf(sigs, 64 * g); d, pd = _inbuf(dst); o, po = _outbuf((g + 7) // 8)
        self._chk(self._lib.blsbn254_fast_aggregate_verify_batch(self._ctx, pa, koff.ctypes.data_as(_u64p), pm, off.ctypes.data_as(_u64p), ps,
                                                                 ctypes.c_size_t(g), pd, ctypes.c_size_t(len(dst)), po))
        return o[:(g + 7) // 8].tobytes()

    def g1_mul_batch(self, g1, scalars, n):
        """Mul<Scalar> for G1Projective (g1.rs:518-534), element-wise: [k_i] P_i."""
        a, pa = _inbuf(g1, 64 * n); k, pk = _inbuf(scalars, 32 * n); o, po = _outbuf(64 * n)
        self._chk(self._lib.blsbn254_g1_mul_batch(self._ctx, pa, pk, ctypes.c_size_t(n), po))
        return o[:64 * n].tobytes()

    def g2_mul_batch(self, g2, scalars, n):
        """Mul<Scalar> for G2Projective (g2.rs:866-886), element-wise: [k_i] Q_i."""
        a, pa = _inbuf(g2, 128 * n); k, pk = _inbuf(scalars, 32 * n); o, po = _outbuf(128 * n)
        self._chk(self._lib.blsbn254_g2_mul_batch(self._ctx, pa, pk, ctypes.c_size_t(n), po))
        return o[:128 * n].tobytes()

    def threshold_combine(self, ids, partial_sigs, t):
        a, pa = _inbuf(ids, 32 * t); s, ps = _inbuf(partial_sigs, 64 * t); o, po = _outbuf(64)
        self._chk(self._lib.blsbn254_threshold_combine(self._ctx, pa, ps, ctypes.c_size_t(t), po))
        return o.tobytes()

    def lagrange_at_zero(self, ids, t):
        """t x 32 bytes: lambda_i = prod_{j != i} x_j / (x_j - x_i) (big-endian Fr)."""
        a, pa = _inbuf(ids, 32 * t); o, po = _outbuf(32 * t)
        self._chk(self._lib.blsbn254_lagrange_at_zero(self._ctx, pa, ctypes.c_size_t(t), po))
        return o[:32 * t].tobytes()

    # ---- Gt group operations and the field-primitive debug ABI
    FIELD_OP_WIDTH = {**{k: 32 for k in range(0, 9)}, **{k: 64 for k in range(16, 22)}, **{k: 192 for k in range(32, 36)},
                      **{k: 384 for k in range(48, 57)}}
    FIELD_OP_BINARY = (0, 3, 4, 16, 32, 48, 56)

    def field_op_batch(self, op, a, b, n):
        """One field / tower primitive element-wise (op = BLSBN254_OP_* of the header); n * width bytes back."""
        w = self.FIELD_OP_WIDTH[op]
        x, px = _inbuf(a, w * n); o, po = _outbuf(w * n)
        if op in self.FIELD_OP_BINARY:
            y, py = _inbuf(b, w * n)
        else:
            y, py = None, ctypes.cast(None, _u8p)
        self._chk(self._lib.blsbn254_field_op_batch(self._ctx, ctypes.c_int(op), px, py, ctypes.c_size_t(n), po))
        return o[:w * n].tobytes()

    def gt_mul_batch(self, a, b, n):
        x, px = _inbuf(a, 384 * n); y, py = _inbuf(b, 384 * n); o, po = _outbuf(384 * n)
        self._chk(self._lib.blsbn254_gt_mul_batch(self._ctx, px, py, ctypes.c_size_t(n), po))
        return o[:384 * n].tobytes()

    def gt_pow_batch(self, gt, scalars, n):
        """Gt::mul_by_scalar (pairings.rs:585-600): gt_i ^ k_i, k_i = 32 bytes big-endian."""
        x, px = _inbuf(gt, 384 * n); k, pk = _inbuf(scalars, 32 * n); o, po = _outbuf(384 * n)
        self._chk(self._lib.blsbn254_gt_pow_batch(self._ctx, px, pk, ctypes.c_size_t(n), po))
        return o[:384 * n].tobytes()

    # ---- compressed codecs
    def _codec(self, fn, data, n, isz, osz):
        a, pa = _inbuf(data, isz * n); o, po = _outbuf(osz * n)
        self._chk(fn(self._ctx, pa, ctypes.c_size_t(n), po))
        return o[:osz * n].tobytes()

    def g1_compress_batch(self, g1, n): return self._codec(self._lib.blsbn254_g1_compress_batch, g1, n, 64, 32)
    def g1_decompress_batch(self, c, n): return self._codec(self._lib.blsbn254_g1_decompress_batch, c, n, 32, 64)
    def g2_compress_batch(self, g2, n): return self._codec(self._lib.blsbn254_g2_compress_batch, g2, n, 128, 64)
    def g2_decompress_batch(self, c, n): return self._codec(self._lib.blsbn254_g2_decompress_batch, c, n, 64, 128)

    # ---- signing side (also used to generate large synthetic batches)
    def sign_batch(self, sks, msgs, dst=DEFAULT_DST):
        n = len(msgs)
        data, off = pack_messages(msgs)
        k, pk = _inbuf(sks, 32 * n); m, pm = _inbuf(data); d, pd = _inbuf(dst); o, po = _outbuf(64 * n)
        self._chk(self._lib.blsbn254_sign_batch(self._ctx, pk, pm, off.ctypes.data_as(_u64p), ctypes.c_size_t(n), pd,
                                                ctypes.c_size_t(len(dst)), po))
        return o[:64 * n].tobytes()

    def sk_to_pk_batch(self, sks, n):
        k, pk = _inbuf(sks, 32 * n); o, po = _outbuf(128 * n)
        self._chk(self._lib.blsbn254_sk_to_pk_batch(self._ctx, pk, ctypes.c_size_t(n), po))
        return o[:128 * n].tobytes()

    def keygen_batch(self, ikm, n, key_info=b""):
        """IETF KeyGen over HKDF-SHA-256 (salt KEYGEN_SALT, helpers.rs:3); ikm = n equal-length seeds >= 32 B."""
        if n == 0:
            return b""
        if len(ikm) % n:
            raise ValueError("ikm must hold n equal-length seeds")
        k, pk = _inbuf(ikm); ki, pki = _inbuf(key_info); o, po = _outbuf(32 * n)
        self._chk(self._lib.blsbn254_keygen_batch(self._ctx, pk, ctypes.c_size_t(len(ikm) // n), ctypes.c_size_t(n), pki,
                                                  ctypes.c_size_t(len(key_info)), po))
        return o[:32 * n].tobytes()

    def hash_to_scalar_batch(self, msgs, dst): return self._h2c(self._lib.blsbn254_hash_to_scalar_batch, msgs, dst, 32)

    def pop_prove_batch(self, sks, n, dst=POP_DST):
        k, pk = _inbuf(sks, 32 * n); d, pd = _inbuf(dst); o, po = _outbuf(64 * n)
        self._chk(self._lib.blsbn254_pop_prove_batch(self._ctx, pk, ctypes.c_size_t(n), pd, ctypes.c_size_t(len(dst)), po))
        return o[:64 * n].tobytes()

    def pop_verify_batch(self, pks, proofs, n, dst=POP_DST):
        a, pa = _inbuf(pks, 128 * n); b, pb = _inbuf(proofs, 64 * n); d, pd = _inbuf(dst); o, po = _outbuf((n + 7) // 8)
        self._chk(self._lib.blsbn254_pop_verify_batch(self._ctx, pa, pb, ctypes.c_size_t(n), pd, ctypes.c_size_t(len(dst)), po))
        return o[:(n + 7) // 8].tobytes()

    # ---- device-resident variants (raw device pointers, e.g. torch tensor .data_ptr())
    def verify_batch_dev(self, d_pks, d_msgs, d_off, d_sigs, n, d_bitmap, dst=DEFAULT_DST):
        d, pd = _inbuf(dst)
        self._chk(self._lib.blsbn254_verify_batch_dev(self._ctx, ctypes.c_void_p(d_pks), ctypes.c_void_p(d_msgs), ctypes.c_void_p(d_off),
                                                      ctypes.c_void_p(d_sigs), ctypes.c_size_t(n), pd, ctypes.c_size_t(len(dst)),
                                                      ctypes.c_void_p(d_bitmap)))

    def verify_batch_rlc_dev(self, d_pks, d_msgs, d_off, d_sigs, n, d_bitmap, dst=DEFAULT_DST, seed=None):
        d, pd = _inbuf(dst)
        if seed is None:
            sd, psd = None, ctypes.cast(None, _u8p)
        else:
            sd, psd = _inbuf(seed, 32)
        self._chk(self._lib.blsbn254_verify_batch_rlc_dev(self._ctx, ctypes.c_void_p(d_pks), ctypes.c_void_p(d_msgs), ctypes.c_void_p(d_off),
                                                          ctypes.c_void_p(d_sigs), ctypes.c_size_t(n), pd, ctypes.c_size_t(len(dst)), psd,
                                                          ctypes.c_void_p(d_bitmap)))

    def pairing_batch_dev(self, d_g1, d_g2, n, d_gt, d_status=0):
        self._chk(self._lib.blsbn254_pairing_batch_dev(self._ctx, ctypes.c_void_p(d_g1), ctypes.c_void_p(d_g2), ctypes.c_size_t(n),
                                                       ctypes.c_void_p(d_gt), ctypes.c_void_p(d_status)))

    def synchronize(self):
        self._chk(self._lib.blsbn254_ctx_synchronize(self._ctx))

    @property
    def stream(self):
        return self._lib.blsbn254_ctx_stream(self._ctx)

    # ---- measurement hooks
    def profile_enable(self, on=True):
        self._chk(self._lib.blsbn254_profile_enable(self._ctx, ctypes.c_int(1 if on else 0)))

    def profile_reset(self):
        self._chk(self._lib.blsbn254_profile_reset(self._ctx))

    def valu_peak(self):
        """Measured whole-chip v_mad_u64_u32 rate (lane-MADs/s)."""
        v = ctypes.c_double(0)
        self._chk(self._lib.blsbn254_valu_peak(self._ctx, ctypes.byref(v)))
        return v.value

    def valu_probe(self):
        """The whole VALU probe (blsbn254_valu_probe): MAD and plain-VOP2 rates, the clock held under each probe
        kernel, and the 4-cycle issue ceiling at that clock."""
        o = (ctypes.c_double * 6)()
        self._chk(self._lib.blsbn254_valu_probe(self._ctx, o))
        return {"mad_per_s": o[0], "vop2_per_s": o[1], "clock_hz_mad": o[2], "clock_hz_vop2": o[3], "cus": int(o[4]),
                "issue_ceiling_per_s": o[5]}

    def profile_read(self):
        names = ctypes.create_string_buffer(32 * 64)
        launches = (ctypes.c_uint64 * 64)()
        ms = (ctypes.c_double * 64)()
        k = self._lib.blsbn254_profile_read(self._ctx, names, launches, ms, ctypes.c_int(64))
        if k < 0:
            self._chk(k)
        out = {}
        for i in range(k):
            nm = names.raw[32 * i:32 * i + 32].split(b"\0")[0].decode()
            out[nm] = {"launches": int(launches[i]), "total_ms": float(ms[i])}
        return out


class PreparedKeys:
    """blsbn254_g2prepared: the line tables of u public keys, resident on the engine's GPU."""

    def __init__(self, engine, pks, u):
        self._eng = engine
        self._lib = engine._lib
        self._lib.blsbn254_g2prepared_count.restype = ctypes.c_size_t
        self._h = ctypes.c_void_p()
        a, pa = _inbuf(pks, 128 * u)
        engine._chk(self._lib.blsbn254_g2_prepare_batch(engine._ctx, pa, ctypes.c_size_t(u), ctypes.byref(self._h)))

    def count(self):
        return int(self._lib.blsbn254_g2prepared_count(self._h))

    def valid_bitmap(self):
        u = self.count()
        o, po = _outbuf((u + 7) // 8)
        self._eng._chk(self._lib.blsbn254_g2prepared_valid(self._eng._ctx, self._h, po))
        return o[:(u + 7) // 8].tobytes()

    def close(self):
        if getattr(self, "_h", None) and getattr(self._eng, "_ctx", None):
            self._lib.blsbn254_g2prepared_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiEngine:
    """blsbn254_multi: one context per listed GPU, one host thread per context per call.
    An ordinal may be listed twice (two contexts on one GPU)."""

    def __init__(self, devices):
        self._lib = load_library()
        self._lib.blsbn254_multi_last_error.restype = ctypes.c_char_p
        self._lib.blsbn254_multi_ctx.restype = ctypes.c_void_p
        self._m = ctypes.c_void_p()
        devs = (ctypes.c_int * len(devices))(*devices)
        rc = self._lib.blsbn254_multi_create(devs, ctypes.c_int(len(devices)), ctypes.byref(self._m))
        if rc != 0:
            self._m = None
            raise Bn254Error(rc)
        self.devices = list(devices)

    def close(self):
        if getattr(self, "_m", None):
            self._lib.blsbn254_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            detail = self._lib.blsbn254_multi_last_error(self._m).decode() if rc < 0 else ""
            raise _ERR.get(rc, Bn254Error)(rc, detail)

    def device_count(self):
        return int(self._lib.blsbn254_multi_device_count(self._m))

    def verify_batch(self, pks, msgs, sigs, dst=DEFAULT_DST):
        n = len(msgs)
        data, off = pack_messages(msgs)
        a, pa = _inbuf(pks, 128 * n); m, pm = _inbuf(data); s, ps = _inbuf(sigs, 64 * n); d, pd = _inbuf(dst)
        o, po = _outbuf((n + 7) // 8)
        self._chk(self._lib.blsbn254_verify_batch_multi(self._m, pa, pm, off.ctypes.data_as(_u64p), ps, ctypes.c_size_t(n), pd,
                                                        ctypes.c_size_t(len(dst)), po))
        return o[:(n + 7) // 8].tobytes()

    def verify_batch_rlc(self, pks, msgs, sigs, dst=DEFAULT_DST, seed=None):
        n = len(msgs)
        data, off = pack_messages(msgs)
        a, pa = _inbuf(pks, 128 * n); m, pm = _inbuf(data); s, ps = _inbuf(sigs, 64 * n); d, pd = _inbuf(dst)
        sd, psd = (None, ctypes.cast(None, _u8p)) if seed is None else _inbuf(seed, 32)
        o, po = _outbuf((n + 7) // 8)
        self._chk(self._lib.blsbn254_verify_batch_rlc_multi(self._m, pa, pm, off.ctypes.data_as(_u64p), ps, ctypes.c_size_t(n), pd,
                                                            ctypes.c_size_t(len(dst)), psd, po))
        return o[:(n + 7) // 8].tobytes()

    def aggregate_verify(self, pks, msgs, agg_sig, dst=DEFAULT_DST):
        n = len(msgs)
        data, off = pack_messages(msgs)
        a, pa = _inbuf(pks, 128 * n); m, pm = _inbuf(data); s, ps = _inbuf(agg_sig, 64); d, pd = _inbuf(dst)
        valid = ctypes.c_int(0)
        self._chk(self._lib.blsbn254_aggregate_verify_multi(self._m, pa, pm, off.ctypes.data_as(_u64p), ctypes.c_size_t(n), ps, pd,
                                                            ctypes.c_size_t(len(dst)), ctypes.byref(valid)))
        return bool(valid.value)

    def verify_batch_dev(self, d_pks, d_msgs, d_off, d_sigs, counts, d_full_bitmaps, dst=DEFAULT_DST):
        """Device-resident shards (lists of raw device pointers, one per device) -> the full bitmap on every device
        (RCCL all-reduce of the disjoint word arrays)."""
        g = len(self.devices)
        vp = lambda xs: (ctypes.c_void_p * g)(*xs)
        cnt = (ctypes.c_size_t * g)(*counts)
        d, pd = _inbuf(dst)
        self._chk(self._lib.blsbn254_verify_batch_multi_dev(self._m, vp(d_pks), vp(d_msgs), vp(d_off), vp(d_sigs), cnt, pd,
                                                            ctypes.c_size_t(len(dst)), vp(d_full_bitmaps)))
