"""RLC batch verification vs the exact path at the bench workload: inputs resident in HBM (device entry points), kernel
time by HIP events and wall time per step; three invalid rates, several chunk sizes.
Usage: python scripts/bench_rlc.py [n] [groups, comma separated; 0 = automatic] [contexts in flight, comma separated]  ->  JSON on stdout"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
groups = [int(g) for g in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16]
pipeline = [int(k) for k in sys.argv[3].split(",")] if len(sys.argv) > 3 else []      # contexts in flight, e.g. 2,3
e = M.Engine(0); dst = M.DEFAULT_DST
dev = torch.device("cuda", 0)
out = {"n": n, "pool": 1024}
# the bench workload's invalid tuples (every 64th, key = index mod 1024) all fall on 16 of the 1024 keys; every 61st spreads them over all keys
for name, inv in (("1/64 invalid (bench workload: invalid tuples on 16 keys)", 64), ("1/61 invalid (spread over all keys)", 61), ("all valid", 0),
                  ("1/1021 invalid (spread over all keys)", 1021)):
    pks, msgs, sigs, exp = synth.make_batch_gpu(e, O, n, dst, pool=1024, invalid_every=inv, spot=50)
    want = synth.bitmap_of(exp)
    data, off = M.engine.pack_messages(msgs)
    t_pk = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
    t_sg = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev)
    t_ms = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    t_bm = torch.zeros((n + 7) // 8, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    args = (t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), n, t_bm.data_ptr(), dst)
    res = {}
    runs = [("exact", None)] + [("rlc_g%d" % g if g else "rlc_auto", g) for g in groups]      # group 0 = automatic
    for label, g in runs:
        if g is not None:
            e.set_rlc_group(g)
        fn = (lambda: e.verify_batch_rlc_dev(*args)) if g is not None else (lambda: e.verify_batch_dev(*args))
        t_bm.zero_()
        e.set_rlc_key_round(True)                   # fresh back-off state for every measurement
        fn(); e.synchronize()
        assert bytes(t_bm.cpu().numpy()) == want, label
        s0 = e.rlc_stats()
        e.profile_enable(True); e.profile_reset()
        t = time.perf_counter(); reps = 5
        for _ in range(reps):
            fn()
            e.synchronize()
        dt = (time.perf_counter() - t) / reps
        p = e.profile_read(); e.profile_enable(False)
        s1 = e.rlc_stats()
        kms = sum(v["total_ms"] for v in p.values()) / reps
        res[label] = {"wall_ms": round(dt * 1e3, 2), "verifies_per_s": round(n / dt), "kernel_ms_sum": round(kms, 2),
                      "kernels": {k: round(v["total_ms"] / reps, 2) for k, v in p.items() if v["total_ms"] / reps > 0.05}}
        if g is not None:
            res[label]["chunks_per_step"] = (s1["chunks"] - s0["chunks"]) // reps
            res[label]["fallback_tuples_per_step"] = (s1["fallback_tuples"] - s0["fallback_tuples"]) // reps
            res[label]["key_rounds_that_decided"] = "%d of %d" % (s1["key_rounds_passed"] - s0["key_rounds_passed"], s1["key_rounds"] - s0["key_rounds"])
    # Throughput with several batches in flight: K contexts (own stream + workspace each) driven by K host threads.  A single
    # RLC call is latency-bound in its chunk round (a quarter of the SIMDs busy); with two or three calls in flight the
    # rounds of one overlap the hashing / weighting of another.  The exact path fills the chip by itself (shown for contrast).
    if pipeline:
        import threading
        for label, rlc in (("exact", False), ("rlc_auto", True)):
            for K in pipeline:
                engs = [e] + [M.Engine(0) for _ in range(K - 1)]
                e.set_rlc_key_round(True)
                bms = [torch.zeros((n + 7) // 8, dtype=torch.uint8, device=dev) for _ in range(K)]
                torch.cuda.synchronize()
                reps = 6

                def work(j, count):
                    fn = engs[j].verify_batch_rlc_dev if rlc else engs[j].verify_batch_dev
                    for _ in range(count):
                        fn(t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), n, bms[j].data_ptr(), dst)
                        engs[j].synchronize()
                for warm in (1, reps):
                    th = [threading.Thread(target=work, args=(j, warm)) for j in range(K)]
                    t = time.perf_counter()
                    for x in th: x.start()
                    for x in th: x.join()
                    dt = time.perf_counter() - t
                for j in range(K):
                    assert bytes(bms[j].cpu().numpy()) == want, (label, K, j)
                res[label]["in_flight_%d" % K] = {"ms_per_batch": round(dt / (K * reps) * 1e3, 2), "verifies_per_s": round(n * K * reps / dt)}
                for x in engs[1:]:
                    x.close()
    out[name] = res
e.set_rlc_group(0)
print(json.dumps(out, indent=1))
