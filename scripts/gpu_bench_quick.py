import sys, time, json, os
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth
import numpy as np, torch
e = M.Engine(0); dst = M.DEFAULT_DST
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
pks, msgs, sigs, exp = synth.make_batch(O, n, dst, invalid_every=64, uniq=64)
data, off = M.engine.pack_messages(msgs)
dev = torch.device("cuda:0")
t_pk = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
t_sg = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev)
t_ms = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
t_bm = torch.zeros((n + 7) // 8, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
def step():
    e.verify_batch_dev(t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), n, t_bm.data_ptr(), dst)
step(); e.synchronize()
print("match", bytes(t_bm.cpu().numpy()) == synth.bitmap_of(exp), flush=True)
e.profile_enable(True); e.profile_reset()
K = 3
t = time.time()
for _ in range(K): step()
e.synchronize(); dt = (time.time() - t) / K
print("n=%d  %.2f ms/step  %.0f verifies/s" % (n, dt * 1e3, n / dt), flush=True)
p = e.profile_read()
for k, v in sorted(p.items(), key=lambda kv: -kv[1]["total_ms"]): print("  %-16s %8.3f ms/launch" % (k, v["total_ms"] / v["launches"]))
