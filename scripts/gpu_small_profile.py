"""Per-kernel times (engine profile) and wall time of small host-pointer calls: verify_batch and pairing_batch at n = 1 ... 1024.
Usage: python scripts/gpu_small_profile.py [sizes...]  -> JSON on stdout"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth

sizes = [int(x) for x in sys.argv[1:]] or [1, 256, 1024]
dst = M.DEFAULT_DST
e = M.Engine(0)
pks, msgs, sigs, exp = synth.make_batch_gpu(e, O, max(sizes), dst, pool=64, invalid_every=0, spot=4)
g2 = O.g2_generator()


def run(f, reps=5):
    f()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    wall = (time.perf_counter() - t) / reps
    e.profile_enable(True); e.profile_reset()
    f()
    prof = e.profile_read()
    e.profile_enable(False)
    return {"wall_ms": round(wall * 1e3, 3), "kernel_sum_ms": round(sum(v["total_ms"] for v in prof.values()), 3),
            "kernels": {k: round(v["total_ms"], 3) for k, v in prof.items()}}


out = {}
for n in sizes:
    a, b, c = pks[:128 * n], msgs[:n], sigs[:64 * n]
    q = g2 * n
    out[str(n)] = {"verify_batch": run(lambda: e.verify_batch(a, b, c, dst)), "pairing_batch": run(lambda: e.pairing_batch(c, q, n))}
print(json.dumps(out, indent=1))
