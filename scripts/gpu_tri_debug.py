"""step-by-step run of the three-lanes-per-tuple paths with progress lines (debugging aid)"""
import os, sys, time, random
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from oracle.pyref import bn254 as B
from tests import synth
def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)
dst = M.DEFAULT_DST
e = M.Engine(0)
log("engine up")
pks, msgs, sigs, exp = synth.make_batch_gpu(e, O, 16385, dst, pool=37, invalid_every=5, spot=5)
log("batch made")
for n in (4097, 6000, 16384, 16385, 32768):
    if n > len(exp):
        pks, msgs, sigs, exp = synth.make_batch_gpu(e, O, n, dst, pool=37, invalid_every=5, spot=5)
    ok = e.verify_batch(pks[:128 * n], msgs[:n], sigs[:64 * n], dst) == synth.bitmap_of(exp[:n])
    e.profile_enable(True); e.profile_reset()
    t = time.perf_counter()
    for _ in range(3):
        e.verify_batch(pks[:128 * n], msgs[:n], sigs[:64 * n], dst)
    dt = (time.perf_counter() - t) / 3
    pr = e.profile_read(); e.profile_enable(False)
    log("verify_batch", n, ok, "%.2f ms" % (dt * 1e3), {k: round(v["total_ms"] / 3, 3) for k, v in pr.items() if v["total_ms"] / 3 > 0.05})
n = 9000
log("rlc", e.verify_batch_rlc(pks[:128 * n], msgs[:n], sigs[:64 * n], dst) == synth.bitmap_of(exp[:n]))
rnd = random.Random(4242)
G1, G2 = O.g1_generator(), O.g2_generator()
m = 4099
ks = b"".join(rnd.randrange(1, B.R).to_bytes(32, "big") for _ in range(m))
g1 = e.g1_mul_batch(G1 * m, ks, m); log("g1 mul")
ks2 = b"".join(rnd.randrange(1, B.R).to_bytes(32, "big") for _ in range(m))
g2 = e.g2_mul_batch(G2 * m, ks2, m); log("g2 mul")
gt = e.pairing_batch(g1, g2, m); log("pairing_batch")
log("oracle", gt[:384] == O.pairing_batch(g1[:64], g2[:128], 1))
ml = e.miller_loop_batch(g1, g2, m); log("miller")
log("fe", e.final_exponentiation(ml, m) == gt)
for m in (8192, 16384):
    ks = b"".join(rnd.randrange(1, B.R).to_bytes(32, "big") for _ in range(m))
    g1 = e.g1_mul_batch(G1 * m, ks, m); g2 = G2 * m
    e.pairing_batch(g1, g2, m)
    e.profile_enable(True); e.profile_reset()
    t = time.perf_counter()
    for _ in range(3):
        e.pairing_batch(g1, g2, m)
    dt = (time.perf_counter() - t) / 3
    pr = e.profile_read(); e.profile_enable(False)
    log("pairing_batch", m, "%.2f ms" % (dt * 1e3), {k: round(v["total_ms"] / 3, 3) for k, v in pr.items() if v["total_ms"] / 3 > 0.05})
