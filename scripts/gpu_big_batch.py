"""Sanity run above the per-launch chunk (4 Mi): verify_batch on n = 4.5 M synthetic tuples (two chunks), bitmap
checked against the closed-form expectation of tests/synth.py.  Usage: python scripts/gpu_big_batch.py [n]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_500_000
e = M.Engine(0); dst = M.DEFAULT_DST
t = time.time()
pks, msgs, sigs, exp = synth.make_batch_gpu(e, O, n, dst, pool=1024, invalid_every=64, spot=200)
print("generated %d tuples in %.1f s" % (n, time.time() - t), flush=True)
t = time.time(); bm = e.verify_batch(pks, msgs, sigs, dst); dt = time.time() - t
print("verify_batch: %.2f s wall (%.2f M/s incl. PCIe and packing), bitmap ok = %s" % (dt, n / dt / 1e6, bm == synth.bitmap_of(exp)), flush=True)
assert bm == synth.bitmap_of(exp)
