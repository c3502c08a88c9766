"""Mid-size verify_batch over ALL-DISTINCT public keys: the prepared-key path with three lanes per tuple (default) against the
exact per-tuple path (BLSBN254_TRI_MAX=0).  Usage: python scripts/bench_distinct_mid.py -> JSON"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth
dst = M.DEFAULT_DST
out = {}
e0 = M.Engine(0)
nmax = 16384
pks, msgs, sigs, exp = synth.make_batch_gpu(e0, O, nmax, dst, pool=nmax, invalid_every=7, spot=10)
e0.close()
for label, tri in (("prepared_tri", None), ("exact_lane", "0")):
    if tri is None:
        os.environ.pop("BLSBN254_TRI_MAX", None)
    else:
        os.environ["BLSBN254_TRI_MAX"] = tri
    e = M.Engine(0)
    res = {}
    for n in (4100, 8192, 16384):
        a, b, c = pks[:128 * n], msgs[:n], sigs[:64 * n]
        assert e.verify_batch(a, b, c, dst) == synth.bitmap_of(exp[:n])
        e.profile_enable(True); e.profile_reset()
        t = time.perf_counter()
        for _ in range(3):
            e.verify_batch(a, b, c, dst)
        dt = (time.perf_counter() - t) / 3
        pr = e.profile_read(); e.profile_enable(False)
        res[str(n)] = {"verify_batch_ms": round(dt * 1e3, 3), "kernel_ms": {k: round(v["total_ms"] / 3, 3) for k, v in pr.items() if v["total_ms"] / 3 > 0.05}}
    out[label] = res
    e.close()
print(json.dumps(out, indent=1))
