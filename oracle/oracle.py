"""ctypes front-end of the C oracle (oracle/bn254_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (bls-bn254_amd/) never does.  Function names and argument order mirror include/blsbn254.h.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbn254_oracle.so")
_lib = None

u8p = ctypes.POINTER(ctypes.c_uint8)
u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "bn254_oracle.c")
    if force or not os.path.exists(_SO) or (os.path.exists(src) and os.path.getmtime(_SO) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _buf(b):
    """bytes / bytearray / np.uint8 array -> (keepalive, pointer)"""
    if isinstance(b, np.ndarray):
        a = np.ascontiguousarray(b, dtype=np.uint8)
    else:
        a = np.frombuffer(bytes(b), dtype=np.uint8)
    if a.size == 0:
        a = np.zeros(1, dtype=np.uint8)
    return a, a.ctypes.data_as(u8p)


def _out(n):
    a = np.zeros(max(n, 1), dtype=np.uint8)
    return a, a.ctypes.data_as(u8p)


def pack_msgs(msgs):
    off = np.zeros(len(msgs) + 1, dtype=np.uint64)
    for i, m in enumerate(msgs):
        off[i + 1] = off[i] + len(m)
    return b"".join(msgs), off


class OracleError(Exception):
    def __init__(self, rc):
        super().__init__("oracle rc=%d" % rc)
        self.rc = rc


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


def pairing_batch(g1, g2, n):
    a, pa = _buf(g1); b, pb = _buf(g2); o, po = _out(384 * n)
    _chk(lib().oracle_pairing_batch(pa, pb, ctypes.c_size_t(n), po))
    return o[:384 * n].tobytes()


def miller_loop_batch(g1, g2, n):
    a, pa = _buf(g1); b, pb = _buf(g2); o, po = _out(384 * n)
    _chk(lib().oracle_miller_loop_batch(pa, pb, ctypes.c_size_t(n), po))
    return o[:384 * n].tobytes()


def multi_miller_loop(g1, g2, n):
    a, pa = _buf(g1); b, pb = _buf(g2); o, po = _out(384)
    _chk(lib().oracle_multi_miller_loop(pa, pb, ctypes.c_size_t(n), po))
    return o.tobytes()


def final_exponentiation(ml, n):
    a, pa = _buf(ml); o, po = _out(384 * n)
    _chk(lib().oracle_final_exponentiation(pa, ctypes.c_size_t(n), po))
    return o[:384 * n].tobytes()


def _h2c(fn, msgs, dst, sz):
    data, off = pack_msgs(msgs)
    a, pa = _buf(data); d, pd = _buf(dst); o, po = _out(sz * len(msgs))
    _chk(fn(pa, off.ctypes.data_as(u64p), ctypes.c_size_t(len(msgs)), pd, ctypes.c_size_t(len(dst)), po))
    return o[:sz * len(msgs)].tobytes()


def hash_to_g1_batch(msgs, dst): return _h2c(lib().oracle_hash_to_g1_batch, msgs, dst, 64)
def hash_to_g2_batch(msgs, dst): return _h2c(lib().oracle_hash_to_g2_batch, msgs, dst, 128)
def encode_to_g1_batch(msgs, dst): return _h2c(lib().oracle_encode_to_g1_batch, msgs, dst, 64)
def encode_to_g2_batch(msgs, dst): return _h2c(lib().oracle_encode_to_g2_batch, msgs, dst, 128)


def hash_to_field_fp(msg, dst, count):
    a, pa = _buf(msg); d, pd = _buf(dst); o, po = _out(32 * count)
    _chk(lib().oracle_hash_to_field_fp(pa, ctypes.c_size_t(len(msg)), pd, ctypes.c_size_t(len(dst)), ctypes.c_size_t(count), po))
    return [int.from_bytes(o[32 * i:32 * i + 32].tobytes(), "big") for i in range(count)]


def _check(fn, pts, n):
    a, pa = _buf(pts); o, po = _out((n + 7) // 8)
    _chk(fn(pa, ctypes.c_size_t(n), po))
    return o[:(n + 7) // 8].tobytes()


def g1_check_batch(g1, n): return _check(lib().oracle_g1_check_batch, g1, n)
def g2_check_batch(g2, n): return _check(lib().oracle_g2_check_batch, g2, n)
def g2_check_batch_slow(g2, n): return _check(lib().oracle_g2_check_batch_slow, g2, n)


def verify_batch(pks, msgs, sigs, dst, nthreads=0):
    n = len(msgs)
    data, off = pack_msgs(msgs)
    a, pa = _buf(pks); m, pm = _buf(data); s, ps = _buf(sigs); d, pd = _buf(dst); o, po = _out((n + 7) // 8)
    if nthreads:
        _chk(lib().oracle_verify_batch_mt(pa, pm, off.ctypes.data_as(u64p), ps, ctypes.c_size_t(n), pd,
                                          ctypes.c_size_t(len(dst)), po, ctypes.c_int(nthreads)))
    else:
        _chk(lib().oracle_verify_batch(pa, pm, off.ctypes.data_as(u64p), ps, ctypes.c_size_t(n), pd,
                                       ctypes.c_size_t(len(dst)), po))
    return o[:(n + 7) // 8].tobytes()


def aggregate_verify(pks, msgs, agg_sig, dst):
    n = len(msgs)
    data, off = pack_msgs(msgs)
    a, pa = _buf(pks); m, pm = _buf(data); s, ps = _buf(agg_sig); d, pd = _buf(dst)
    valid = ctypes.c_int(0)
    _chk(lib().oracle_aggregate_verify(pa, pm, off.ctypes.data_as(u64p), ctypes.c_size_t(n), ps, pd,
                                       ctypes.c_size_t(len(dst)), ctypes.byref(valid)))
    return bool(valid.value)


def aggregate_sigs(sigs, n):
    a, pa = _buf(sigs); o, po = _out(64)
    _chk(lib().oracle_aggregate_sigs(pa, ctypes.c_size_t(n), po))
    return o.tobytes()


def aggregate_pks(pks, n):
    """impl Sum for G2Projective (g2.rs:579-583)"""
    a, pa = _buf(pks); o, po = _out(128)
    _chk(lib().oracle_aggregate_pks(pa, ctypes.c_size_t(n), po))
    return o.tobytes()


def fast_aggregate_verify(pks, n, msg, sig, dst):
    a, pa = _buf(pks); m, pm = _buf(msg); s, ps = _buf(sig); d, pd = _buf(dst)
    v = ctypes.c_int(0)
    _chk(lib().oracle_fast_aggregate_verify(pa, ctypes.c_size_t(n), pm, ctypes.c_size_t(len(msg)), ps, pd, ctypes.c_size_t(len(dst)), ctypes.byref(v)))
    return bool(v.value)


def threshold_combine(ids, sigs, t):
    a, pa = _buf(ids); s, ps = _buf(sigs); o, po = _out(64)
    _chk(lib().oracle_threshold_combine(pa, ps, ctypes.c_size_t(t), po))
    return o.tobytes()


def fr_lagrange_at_zero(ids, t):
    a, pa = _buf(ids); o, po = _out(32 * t)
    _chk(lib().oracle_fr_lagrange_at_zero(pa, ctypes.c_size_t(t), po))
    return o[:32 * t].tobytes()


def g1_generator():
    o, po = _out(64); lib().oracle_g1_generator(po); return o.tobytes()


def g2_generator():
    o, po = _out(128); lib().oracle_g2_generator(po); return o.tobytes()


def _sc(k):
    return (k if isinstance(k, (bytes, bytearray)) else int(k).to_bytes(32, "big"))


def g1_mul(pt, k):
    a, pa = _buf(pt); s, ps = _buf(_sc(k)); o, po = _out(64)
    _chk(lib().oracle_g1_mul(pa, ps, po)); return o.tobytes()


def g2_mul(pt, k):
    a, pa = _buf(pt); s, ps = _buf(_sc(k)); o, po = _out(128)
    _chk(lib().oracle_g2_mul(pa, ps, po)); return o.tobytes()


def g1_add(p, q):
    a, pa = _buf(p); b, pb = _buf(q); o, po = _out(64)
    _chk(lib().oracle_g1_add(pa, pb, po)); return o.tobytes()


def g2_add(p, q):
    a, pa = _buf(p); b, pb = _buf(q); o, po = _out(128)
    _chk(lib().oracle_g2_add(pa, pb, po)); return o.tobytes()


def sk_to_pk(sk):
    s, ps = _buf(_sc(sk)); o, po = _out(128)
    _chk(lib().oracle_sk_to_pk(ps, po)); return o.tobytes()


def sign(sk, msg, dst):
    s, ps = _buf(_sc(sk)); m, pm = _buf(msg); d, pd = _buf(dst); o, po = _out(64)
    _chk(lib().oracle_sign(ps, pm, ctypes.c_size_t(len(msg)), pd, ctypes.c_size_t(len(dst)), po)); return o.tobytes()


def gt_pow(gt, k):
    a, pa = _buf(gt); s, ps = _buf(_sc(k)); o, po = _out(384)
    _chk(lib().oracle_gt_pow(pa, ps, po)); return o.tobytes()


def gt_mul(x, y):
    a, pa = _buf(x); b, pb = _buf(y); o, po = _out(384)
    _chk(lib().oracle_gt_mul(pa, pb, po)); return o.tobytes()


def sha256(msg):
    m, pm = _buf(msg); o, po = _out(32)
    lib().oracle_sha256(pm, ctypes.c_size_t(len(msg)), po); return o.tobytes()


def counters_reset():
    lib().oracle_counters_reset()


def counters_get():
    a = ctypes.c_uint64(0); b = ctypes.c_uint64(0)
    lib().oracle_counters_get(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def verify_batch_refstyle(pks, msgs, sigs, dst, nthreads=1):
    """verify_batch with every Fp product done in the reference's own arithmetic (fp.rs:404-407, fp2.rs:377-390):
    ~100x slower, same bitmap.  Timed CPU baseline only."""
    n = len(msgs)
    data, off = pack_msgs(msgs)
    a, pa = _buf(pks); m, pm = _buf(data); s, ps = _buf(sigs); d, pd = _buf(dst); o, po = _out((n + 7) // 8)
    _chk(lib().oracle_verify_batch_refstyle_mt(pa, pm, off.ctypes.data_as(u64p), ps, ctypes.c_size_t(n), pd,
                                               ctypes.c_size_t(len(dst)), po, ctypes.c_int(max(1, nthreads))))
    return o[:(n + 7) // 8].tobytes()


FIELD_OP_WIDTH = {**{k: 32 for k in range(0, 9)}, **{k: 64 for k in range(16, 22)}, **{k: 192 for k in range(32, 36)},
                  **{k: 384 for k in range(48, 57)}}
FIELD_OP_BINARY = (0, 3, 4, 16, 32, 48, 56)


def field_op_batch(op, a, b, n):
    """Element-wise field / tower primitive (op codes of include/blsbn254.h); returns n * width bytes."""
    w = FIELD_OP_WIDTH[op]
    x, px = _buf(a); o, po = _out(w * n)
    if op in FIELD_OP_BINARY:
        y, py = _buf(b)
    else:
        y, py = None, ctypes.cast(None, u8p)
    _chk(lib().oracle_field_op_batch(ctypes.c_int(op), px, py, ctypes.c_size_t(n), po))
    return o[:w * n].tobytes()


def verify_core_counts():
    """Exact Fp mul+sqr counts of the algorithmic unit: (variable-Q Miller pair, fixed-Q pair from table, final exp, table
    entries, the loop's shared squarings of f)."""
    a = (ctypes.c_uint64 * 5)()
    lib().oracle_verify_core_counts(a)
    return tuple(int(v) for v in a)


def _codec(fn, data, n_in, n_out):
    a, pa = _buf(data); o, po = _out(n_out)
    rc = fn(pa, po)
    return (o[:n_out].tobytes() if rc == 0 else None)


def g1_compress(pt): return _codec(lib().oracle_g1_compress, pt, 64, 32)
def g1_decompress(c): return _codec(lib().oracle_g1_decompress, c, 32, 64)
def g2_compress(pt): return _codec(lib().oracle_g2_compress, pt, 128, 64)
def g2_decompress(c): return _codec(lib().oracle_g2_decompress, c, 64, 128)


def fp_mul_refstyle(a, b):
    """Fp::multiply as the reference computes it (fp.rs:404-407 over const_rem_wide); timing aid only."""
    x, px = _buf(a.to_bytes(32, "big")); y, py = _buf(b.to_bytes(32, "big")); o, po = _out(32)
    lib().oracle_fp_mul_refstyle(px, py, po); return int.from_bytes(o.tobytes(), "big")


def bench_fp_mul(refstyle, iters):
    f = lib().oracle_bench_fp_mul; f.restype = ctypes.c_double
    return f(ctypes.c_int(1 if refstyle else 0), ctypes.c_uint64(iters))
