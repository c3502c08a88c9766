"""One RLC batch verification of the bench workload (262144 tuples, 1024-key pool, 1/64 invalid), device entry point: the
profiling target for the RLC kernels (k_rlc2_*, the three-lanes-per-tuple rounds).  Usage: python scripts/run_rlc_once.py [steps]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 262144
e = M.Engine(0); dst = M.DEFAULT_DST
dev = torch.device("cuda", 0)
pks, msgs, sigs, exp = synth.make_batch_gpu(e, O, n, dst, pool=1024, invalid_every=64, spot=20)
data, off = M.engine.pack_messages(msgs)
t_pk = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
t_sg = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev)
t_ms = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
t_bm = torch.zeros((n + 7) // 8, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
e.set_rlc_key_round(False)            # every step takes the chunk round (16 384 virtual tuples at most) and the fallback round
for _ in range(steps):
    e.verify_batch_rlc_dev(t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), n, t_bm.data_ptr(), dst)
    e.synchronize()
assert bytes(t_bm.cpu().numpy()) == synth.bitmap_of(exp)
print("rlc ok", e.rlc_stats())
