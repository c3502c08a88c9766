"""Host-pointer entry point (blsbn254_verify_batch: 224 B/verify H2D + bitmap D2H) against the device-resident one, with the
caller's buffers in pageable memory and in pinned (page-locked) memory.  Usage: python scripts/bench_host_api.py [n] -> JSON"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
e = M.Engine(0); dst = M.DEFAULT_DST
lib = e._lib
pks, msgs, sigs, exp = synth.make_batch_gpu(e, O, n, dst, pool=1024, invalid_every=64, spot=20)
want = synth.bitmap_of(exp)
data, off = M.engine.pack_messages(msgs)
u8p, u64p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint64)


def run(label, t_pk, t_ms, t_off, t_sg, t_bm, reps=5):
    d = (ctypes.c_uint8 * len(dst)).from_buffer_copy(dst)
    def call():
        rc = lib.blsbn254_verify_batch(e._ctx, ctypes.cast(t_pk.data_ptr(), u8p), ctypes.cast(t_ms.data_ptr(), u8p), ctypes.cast(t_off.data_ptr(), u64p),
                                       ctypes.cast(t_sg.data_ptr(), u8p), ctypes.c_size_t(n), d, ctypes.c_size_t(len(dst)), ctypes.cast(t_bm.data_ptr(), u8p))
        assert rc == 0, rc
    call()
    assert bytes(t_bm.numpy()) == want
    t = time.perf_counter()
    for _ in range(reps):
        call()
    return round((time.perf_counter() - t) / reps * 1e3, 2)


mk = lambda b, dt=torch.uint8: torch.frombuffer(bytearray(b), dtype=dt)
host = [mk(pks), mk(data), torch.from_numpy(off.astype(np.int64)), mk(sigs), torch.zeros((n + 7) // 8, dtype=torch.uint8)]
out = {"n": n, "bytes_in": len(pks) + len(data) + 8 * (n + 1) + len(sigs)}
out["pageable_ms"] = run("pageable", *host)
pinned = [t.pin_memory() for t in host]
out["pinned_ms"] = run("pinned", *pinned)
dev = [t.cuda() for t in host]
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    e.verify_batch_dev(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), n, dev[4].data_ptr(), dst)
    e.synchronize()
out["device_resident_ms"] = round((time.perf_counter() - t) / 5 * 1e3, 2)
assert bytes(dev[4].cpu().numpy()) == want
print(json.dumps(out))
