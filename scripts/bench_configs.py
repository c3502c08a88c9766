"""Secondary measurements for the BASELINE configs that are not the headline bench line:
pairing_batch (262144), aggregate_verify (1M pairs, configs[2]), threshold_combine (1000-of-2000, configs[4]).
Host-pointer entry points (includes PCIe staging); kernel times from the engine's HIP-event profile."""
import json, os, random, sys, time
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from oracle.pyref import bn254 as B
from tests import synth

e = M.Engine(0); dst = M.DEFAULT_DST
out = {}

def timed(fn, reps=3):
    fn(); e.profile_enable(True); e.profile_reset()
    t = time.perf_counter()
    for _ in range(reps): r = fn()
    dt = (time.perf_counter() - t) / reps
    p = e.profile_read(); e.profile_enable(False)
    return dt, {k: round(v["total_ms"] / reps, 3) for k, v in p.items()}, r

# pairing_batch
n = 262144
rnd = random.Random(5)
G1, G2 = O.g1_generator(), O.g2_generator()
g1 = b"".join(O.g1_mul(G1, rnd.randrange(1, B.R)) for _ in range(64)) * (n // 64)
g2 = b"".join(O.g2_mul(G2, rnd.randrange(1, B.R)) for _ in range(64)) * (n // 64)
dt, k, r = timed(lambda: e.pairing_batch(g1, g2, n))
assert r[:384 * 64] == O.pairing_batch(g1[:64 * 64], g2[:128 * 64], 64)
out["pairing_batch_262144"] = {"wall_s": round(dt, 4), "pairings_per_s_wall": round(n / dt), "kernel_ms": k,
                               "pairings_per_s_kernels": round(n / (sum(k.values()) * 1e-3))}
# aggregate verify 1M
n, uniq = 1 << 20, 64
sks = [synth.sk_of(i) for i in range(8)]; pkp = [O.sk_to_pk(s) for s in sks]
base = [(pkp[i % 8], synth.msg_of(i)) for i in range(uniq)]
sig64 = O.aggregate_sigs(b"".join(O.sign(sks[i % 8], base[i][1], dst) for i in range(uniq)), uniq)
agg = O.g1_mul(sig64, n // uniq)
pks = b"".join(b[0] for b in base) * (n // uniq); msgs = [b[1] for b in base] * (n // uniq)
dt, k, r = timed(lambda: e.aggregate_verify(pks, msgs, agg, dst), reps=2)
assert r is True
out["aggregate_verify_1M"] = {"wall_s": round(dt, 4), "kernel_ms": k, "miller_loops_per_s_kernels": round(n / (sum(k.values()) * 1e-3)),
                              "miller_loops_per_s_wall": round(n / dt)}
# threshold 1000 of 2000
t, total = 1000, 2000
coeffs = [rnd.randrange(1, B.R) for _ in range(t)]; ids = rnd.sample(range(1, total + 1), t)
def f(x):
    a = 0
    for c in reversed(coeffs): a = (a * x + c) % B.R
    return a
h = O.hash_to_g1_batch([b"thr"], dst)
parts = b"".join(O.g1_mul(h, f(i)) for i in ids); idb = b"".join(i.to_bytes(32, "big") for i in ids)
dt, k, r = timed(lambda: e.threshold_combine(idb, parts, t))
assert r == O.sign(coeffs[0], b"thr", dst)
out["threshold_combine_1000_of_2000"] = {"wall_s": round(dt, 5), "kernel_ms": k}
print(json.dumps(out, indent=1))
