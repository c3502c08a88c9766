"""RLC vs exact verify at the bench workload (host-pointer entry points, wall time incl. PCIe staging)."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth
e = M.Engine(0); dst = M.DEFAULT_DST
n = 262144
out = {}
for name, inv in (("1/64 invalid (bench workload)", 64), ("all valid", 0), ("1/1024 invalid", 1024)):
    pks, msgs, sigs, exp = synth.make_batch_gpu(e, O, n, dst, pool=1024, invalid_every=inv, spot=50)
    want = synth.bitmap_of(exp)
    res = {}
    for label, fn in (("exact", lambda: e.verify_batch(pks, msgs, sigs, dst)), ("rlc", lambda: e.verify_batch_rlc(pks, msgs, sigs, dst))):
        assert fn() == want
        e.profile_enable(True); e.profile_reset()
        t = time.perf_counter(); reps = 3
        for _ in range(reps): fn()
        dt = (time.perf_counter() - t) / reps
        p = e.profile_read(); e.profile_enable(False)
        kms = sum(v["total_ms"] for v in p.values()) / reps
        res[label] = {"wall_ms": round(dt * 1e3, 2), "kernel_ms": round(kms, 2), "verifies_per_s_kernels": round(n / (kms * 1e-3)),
                      "kernels": {k: round(v["total_ms"] / reps, 2) for k, v in p.items() if v["total_ms"] / reps > 0.3}}
    out[name] = res
print(json.dumps(out, indent=1))
