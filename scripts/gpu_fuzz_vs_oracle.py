"""One-off differential run: N random tuples (random keys, random-length messages, 1/8 corrupted in the five ways of
tests/synth.py) verified on the GPU and by the C oracle (all host threads); also pairing_batch vs the oracle on
random points.  Usage (on the GPU box): python scripts/gpu_fuzz_vs_oracle.py [N]"""
import os, random, sys, time
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
e = M.Engine(0); dst = M.DEFAULT_DST
t = time.time()
pks, msgs, sigs, exp = synth.make_batch_gpu(e, O, n, dst, pool=4096, invalid_every=8, spot=200)
print("generated", n, "tuples in %.1f s" % (time.time() - t), flush=True)
t = time.time(); gb = e.verify_batch(pks, msgs, sigs, dst); tg = time.time() - t
t = time.time(); ob = O.verify_batch(pks, msgs, sigs, dst, nthreads=os.cpu_count()); to = time.time() - t
print("gpu %.2f s  oracle %.1f s  equal=%s  expected=%s" % (tg, to, gb == ob, gb == synth.bitmap_of(exp)), flush=True)
assert gb == ob == synth.bitmap_of(exp)
rnd = random.Random(5); m = 2000
G1, G2 = O.g1_generator(), O.g2_generator()
g1 = b"".join(O.g1_mul(G1, rnd.randrange(1, 2**200)) for _ in range(m)); g2 = b"".join(O.g2_mul(G2, rnd.randrange(1, 2**200)) for _ in range(m))
assert e.pairing_batch(g1, g2, m) == O.pairing_batch(g1, g2, m)
assert e.miller_loop_batch(g1, g2, m) == O.miller_loop_batch(g1, g2, m)
print("pairing_batch / miller_loop_batch bit-exact on", m, "random pairs")
