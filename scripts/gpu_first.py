import sys, time, json, os
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth
t0=time.time()
e = M.Engine(0); print("ctx ok", time.time()-t0, flush=True)
e.profile_enable(True)
G1, G2 = O.g1_generator(), O.g2_generator()
kats = json.load(open("tests/golden/reference_kats.json"))
t=time.time(); gt = e.pairing_batch(G1, G2, 1); print("pairing 1:", time.time()-t, flush=True)
print("golden:", gt.hex() == kats["constants"]["gt_generator_bytes_hex"], flush=True)
ml = e.miller_loop_batch(G1, G2, 1); print("ML bit-exact:", ml == O.miller_loop_batch(G1, G2, 1), flush=True)
dst = M.DEFAULT_DST
msgs=[b"", b"abc", os.urandom(100)]
print("h2g1:", e.hash_to_g1_batch(msgs, dst) == O.hash_to_g1_batch(msgs, dst), flush=True)
print("h2g2:", e.hash_to_g2_batch(msgs, dst) == O.hash_to_g2_batch(msgs, dst), flush=True)
n=512
pks, ms, sigs, exp = synth.make_batch(O, n, dst, invalid_every=8, uniq=64)
t=time.time(); bm = e.verify_batch(pks, ms, sigs, dst); dt=time.time()-t
print("verify %d: %.3fs match=%s"%(n, dt, bm == synth.bitmap_of(exp)), flush=True)
print(json.dumps(e.profile_read()), flush=True)
e.profile_reset()
n=16384
pks, ms, sigs, exp = synth.make_batch(O, n, dst, invalid_every=64, uniq=64)
t=time.time(); bm = e.verify_batch(pks, ms, sigs, dst); dt=time.time()-t
print("verify %d: %.3fs (%.0f/s) match=%s"%(n, dt, n/dt, bm == synth.bitmap_of(exp)), flush=True)
print(json.dumps(e.profile_read()), flush=True)
