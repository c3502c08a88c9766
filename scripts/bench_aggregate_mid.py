"""aggregate_verify over n DISTINCT keys, mid-size: the table path on quads (default) against the pairwise lane path (BLSBN254_TRI_MAX=0).
Usage: python scripts/bench_aggregate_mid.py -> JSON"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from tests import synth
dst = M.DEFAULT_DST
out = {}
e0 = M.Engine(0)
nmax = 8192
sks = [synth.sk_of(40000 + k) for k in range(nmax)]
skb = b"".join(x.to_bytes(32, "big") for x in sks)
pk = e0.sk_to_pk_batch(skb, nmax)
ms = [synth.msg_of(50000 + i) for i in range(nmax)]
sg = e0.sign_batch(skb, ms, dst)
e0.close()
for label, tri in (("tables_on_quads", None), ("pairwise_lanes", "0")):
    if tri is None:
        os.environ.pop("BLSBN254_TRI_MAX", None)
    else:
        os.environ["BLSBN254_TRI_MAX"] = tri
    e = M.Engine(0)
    res = {}
    for n in (4100, 8192):
        agg = e.aggregate_sigs(sg[:64 * n], n)
        assert e.aggregate_verify(pk[:128 * n], ms[:n], agg, dst) is True
        e.profile_enable(True); e.profile_reset()
        t = time.perf_counter()
        for _ in range(3):
            e.aggregate_verify(pk[:128 * n], ms[:n], agg, dst)
        dt = (time.perf_counter() - t) / 3
        pr = e.profile_read(); e.profile_enable(False)
        res[str(n)] = {"aggregate_verify_ms": round(dt * 1e3, 3), "kernel_ms": {k: round(v["total_ms"] / 3, 3) for k, v in pr.items() if v["total_ms"] / 3 > 0.05}}
    out[label] = res
    e.close()
print(json.dumps(out, indent=1))
