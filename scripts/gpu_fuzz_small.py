"""Differential run for the SMALL-call paths (wave-per-tuple kernels, always-prepared keys): random batch sizes 1 ... 5000 with
random keys of random multiplicity, ragged messages, 1/5 corrupted in the five ways of tests/synth.py; verify_batch,
verify_batch_rlc and aggregate_verify on the GPU against the C oracle.  Usage (GPU box): python scripts/gpu_fuzz_small.py [rounds]"""
import os, random, sys, time
sys.path.insert(0, os.getcwd())
import blsbn254_loader; M = blsbn254_loader.load()
from oracle import oracle as O
from tests import synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
e = M.Engine(0); dst = M.DEFAULT_DST
rnd = random.Random(2026)
total = 0
for r in range(rounds):
    n = rnd.choice([1, 2, 3, 17, 64, 65, 300, 1023, 1024, 1500, 4095, 4096, 4097, 5000])
    pool = rnd.choice([1, 2, 7, n, max(1, n // 2), max(1, n // 3)])
    pks, msgs, sigs, exp = synth.make_batch_gpu(e, O, n, dst, pool=pool, invalid_every=rnd.choice([0, 5, 2]), spot=5, seed=r)
    msgs = [m[:rnd.randrange(0, 33)] if i % 3 == 0 else m for i, m in enumerate(msgs)]          # ragged: some messages change -> their tuples become invalid
    want = O.verify_batch(pks, msgs, sigs, dst, nthreads=os.cpu_count())
    assert e.verify_batch(pks, msgs, sigs, dst) == want, ("verify", n, pool)
    assert e.verify_batch_rlc(pks, msgs, sigs, dst) == want, ("rlc", n, pool)
    # aggregate verify over the valid tuples only, then with one invalid tuple mixed in
    ok_idx = [i for i in range(n) if want[i >> 3] >> (i & 7) & 1]
    if ok_idx:
        a = b"".join(pks[128 * i:128 * i + 128] for i in ok_idx); mm = [msgs[i] for i in ok_idx]; ss = b"".join(sigs[64 * i:64 * i + 64] for i in ok_idx)
        agg = e.aggregate_sigs(ss, len(ok_idx))
        assert e.aggregate_verify(a, mm, agg, dst) is True, ("agg", n, pool)
        mm[len(mm) // 2] = mm[len(mm) // 2] + b"!"
        assert e.aggregate_verify(a, mm, agg, dst) is False, ("agg-bad", n, pool)
    total += n
    print("round", r, "n", n, "pool", pool, "valid", len(ok_idx), "ok", flush=True)
print("all", rounds, "rounds,", total, "tuples: GPU == oracle")
