"""Per-kernel times of pairing_batch at mid sizes (engine profile): python scripts/gpu_pairing_mid.py [sizes...]"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O

sizes = [int(x) for x in sys.argv[1:]] or [4096, 8192, 12288, 16384]
e = M.Engine(0)
g1 = O.g1_generator()
g2 = O.g2_generator()
out = {}
for n in sizes:
    a, b = g1 * n, g2 * n
    e.pairing_batch(a, b, n)
    e.profile_enable(True); e.profile_reset()
    t = time.perf_counter()
    for _ in range(3):
        e.pairing_batch(a, b, n)
    dt = (time.perf_counter() - t) / 3
    prof = e.profile_read()
    e.profile_enable(False)
    out[str(n)] = {"wall_ms": round(dt * 1e3, 3), "kernels": {k: [v["launches"], round(v["total_ms"] / max(v["launches"], 1), 3)] for k, v in prof.items()}}
print(json.dumps(out, indent=1))
