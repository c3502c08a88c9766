"""CPU-only, world_size 2 over gloo: the N > 1 path (contiguous sharding + bitmap all-reduce).
The per-shard verifier here is the CPU oracle (test stand-in for the GPU engine); what is under test is
bls-bn254_amd/sharded.py, the only code that differs between 1 and N GPUs."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import blsbn254_loader
    blsbn254_loader.load()
    sharded = __import__("bls_bn254_amd.sharded", fromlist=["x"])
    from oracle import oracle as O
    from tests import synth
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dst = b"TEST_DST"
    pks, msgs, sigs, exp = synth.make_batch(O, n, dst, invalid_every=3, uniq=12)

    def verify_local(lo, hi):
        bm = O.verify_batch(pks[128 * lo:128 * hi], msgs[lo:hi], sigs[64 * lo:64 * hi], dst)
        return torch.frombuffer(bytearray(bm), dtype=torch.uint8)
    words = sharded.verify_batch_sharded(verify_local, n, rank, world, dist, torch, torch.device("cpu"))
    got = sharded.words_to_bitmap_bytes(words.numpy(), n)
    np.save(os.path.join(outdir, "r%d.npy" % rank), np.frombuffer(got, dtype=np.uint8))
    dist.barrier()
    dist.destroy_process_group()


def _agg_worker(rank, world, port, n, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import blsbn254_loader
    blsbn254_loader.load()
    sharded = __import__("bls_bn254_amd.sharded", fromlist=["x"])
    from oracle import oracle as O
    from oracle.pyref import bn254 as B
    from tests import synth
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dst = b"TEST_DST"
    sks = [synth.sk_of(k) for k in range(n)]
    pks = [O.sk_to_pk(s) for s in sks]
    msgs = [synth.msg_of(i) for i in range(n)]
    agg = O.aggregate_sigs(b"".join(O.sign(s, m, dst) for s, m in zip(sks, msgs)), n)
    neg_g2 = B.g2_to_bytes(B.g2_neg(B.G2_GEN))
    one = (1).to_bytes(32, "big") + bytes(352)

    def partial_local(lo, hi):                      # oracle stand-in for Engine.aggregate_partial
        if hi == lo:
            return one, True
        h = O.hash_to_g1_batch(msgs[lo:hi], dst)
        ok = O.g2_check_batch(b"".join(pks[lo:hi]), hi - lo) == synth.bitmap_of([True] * (hi - lo))
        return O.multi_miller_loop(h, b"".join(pks[lo:hi]), hi - lo), ok

    def finish(partials, k):                        # oracle stand-in for Engine.aggregate_finish
        acc = O.miller_loop_batch(agg, neg_g2, 1)
        for i in range(k):
            acc = O.gt_mul(acc, partials[384 * i:384 * i + 384])
        return O.final_exponentiation(acc, 1) == one
    res = []
    res.append(sharded.aggregate_verify_sharded(partial_local, finish, n, rank, world, dist, torch, torch.device("cpu")))
    msgs[1] = b"tampered"
    res.append(sharded.aggregate_verify_sharded(partial_local, finish, n, rank, world, dist, torch, torch.device("cpu")))
    np.save(os.path.join(outdir, "a%d.npy" % rank), np.array(res, dtype=np.uint8))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_aggregate_allgather(tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    O.build()
    mp.spawn(_agg_worker, args=(2, _free_port(), 5, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(tmp_path / ("a%d.npy" % r)).tolist() == [1, 0]


@pytest.mark.parametrize("n", [37, 64])
def test_world2_bitmap_allreduce(tmp_path, n):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from tests import synth
    O.build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    _, _, _, exp = synth.make_batch(O, n, b"TEST_DST", invalid_every=3, uniq=12)
    want = synth.bitmap_of(exp)
    for r in range(2):
        assert np.load(tmp_path / ("r%d.npy" % r)).tobytes() == want


def test_shard_ranges_and_bit_packing():
    sys.path.insert(0, ROOT)
    import torch
    import blsbn254_loader
    blsbn254_loader.load()
    sharded = __import__("bls_bn254_amd.sharded", fromlist=["x"])
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            r = [sharded.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2, size=77).astype(np.uint8)
    bm = torch.from_numpy(np.packbits(bits, bitorder="little"))
    words = sharded.allreduce_bitmap(bm, 13, 77, 200, None, torch)
    full = np.zeros(200, dtype=np.uint8); full[13:90] = bits
    assert sharded.words_to_bitmap_bytes(words.numpy(), 200) == np.packbits(full, bitorder="little").tobytes()
