"""Latency of small calls: verify_batch / pairing_batch / aggregate_verify at n = 1 ... 8192 through the host-pointer C ABI
(ctypes): with the wave-per-tuple (n <= 4096) and three-lanes-per-tuple (n <= 16384) kernels (default), with the former only
(BLSBN254_TRI_MAX=0) and with neither (BLSBN254_WIDE_FE=0, BLSBN254_TRI_MAX=0).
Usage: python scripts/bench_small.py [sizes...]  -> JSON on stdout"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth

dst = M.DEFAULT_DST
sizes = [int(x) for x in sys.argv[1:]] or [1, 16, 256, 1024, 4096, 4100, 8192, 16384, 32768]
out = {}
e0 = M.Engine(0)
nmax = max(sizes)
pks, msgs, sigs, exp = synth.make_batch_gpu(e0, O, nmax, dst, pool=64, invalid_every=0, spot=10)
g2 = O.g2_generator()
for label, env, tri in (("wave_and_tri", None, None), ("wave_only", None, "0"), ("lane_per_tuple", "0", "0")):
    for k, v in (("BLSBN254_WIDE_FE", env), ("BLSBN254_TRI_MAX", tri)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    e = M.Engine(0)
    res = {}
    for n in sizes:
        a, b, c = pks[:128 * n], msgs[:n], sigs[:64 * n]
        want = synth.bitmap_of(exp[:n])
        assert e.verify_batch(a, b, c, dst) == want
        reps = 5
        t = time.perf_counter()
        for _ in range(reps):
            e.verify_batch(a, b, c, dst)
        tv = (time.perf_counter() - t) / reps
        g1 = c                                        # signatures are G1 points
        q = g2 * n
        e.pairing_batch(g1, q, n)
        t = time.perf_counter()
        for _ in range(reps):
            e.pairing_batch(g1, q, n)
        tp = (time.perf_counter() - t) / reps
        agg = e.aggregate_sigs(c, n)
        assert e.aggregate_verify(a, b, agg, dst) is True
        t = time.perf_counter()
        for _ in range(reps):
            e.aggregate_verify(a, b, agg, dst)
        ta = (time.perf_counter() - t) / reps
        res[str(n)] = {"verify_batch_ms": round(tv * 1e3, 3), "verifies_per_s": round(n / tv), "pairing_batch_ms": round(tp * 1e3, 3),
                       "aggregate_verify_ms": round(ta * 1e3, 3)}
    out[label] = res
    e.close()
print(json.dumps(out, indent=1))
