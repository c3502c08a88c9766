"""FastAggregateVerify at the validator shape: G groups of K keys each (one message and one aggregate signature per group), host-pointer
entry point blsbn254_fast_aggregate_verify_batch; kernel times from the engine's HIP-event profile.
Usage: python scripts/bench_fast_aggregate.py -> JSON"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from tests import synth
dst = M.DEFAULT_DST
e = M.Engine(0)
R = synth.R
pool = 4096
sks = [synth.sk_of(k) for k in range(pool)]
pk_pool = e.sk_to_pk_batch(b"".join(s.to_bytes(32, "big") for s in sks), pool)
out = {}
for g, per in ((4096, 64), (1024, 512), (16384, 16)):
    key_sets, msgs, agg = [], [], []
    for i in range(g):
        lo = (i * 37) % (pool - per)
        key_sets.append(pk_pool[128 * lo:128 * (lo + per)])
        agg.append(sum(sks[lo:lo + per]) % R)
        msgs.append(synth.msg_of(20000 + i))
    sigs = e.sign_batch(b"".join(s.to_bytes(32, "big") for s in agg), msgs, dst)
    bm = e.fast_aggregate_verify_batch(key_sets, msgs, sigs, dst)
    assert bm == synth.bitmap_of([True] * g)
    e.profile_enable(True); e.profile_reset()
    t = time.perf_counter()
    for _ in range(3):
        e.fast_aggregate_verify_batch(key_sets, msgs, sigs, dst)
    dt = (time.perf_counter() - t) / 3
    pr = e.profile_read(); e.profile_enable(False)
    k = {n: round(v["total_ms"] / 3, 3) for n, v in pr.items() if v["total_ms"] / 3 > 0.03}
    out["%d groups x %d keys" % (g, per)] = {"wall_ms": round(dt * 1e3, 2), "keys": g * per, "keys_per_s_wall": round(g * per / dt),
                                              "kernel_ms": k, "kernel_ms_sum_main_stream": round(sum(v for n, v in k.items() if n not in ("g2_prepare", "g2_expand")), 3)}
print(json.dumps(out, indent=1))
