"""GPU (MI355X): the N > 1 path with the REAL engine.  Two child processes share device 0 (the GPU box has one
card), rendezvous over gloo on 127.0.0.1, and run bls-bn254_amd/sharded.py with Engine.verify_batch_dev /
aggregate_partial / aggregate_finish as the per-shard workers; results are compared with the CPU oracle on the same
inputs.  Also: `bench.py --gpus 2` started as a plain process must spawn its two ranks itself and report n_gpus 2."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _verify_worker(rank, world, port, n, outdir, rlc=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import blsbn254_loader
    M = blsbn254_loader.load()
    sharded = __import__("bls_bn254_amd.sharded", fromlist=["x"])
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    eng = M.Engine(0)
    dst = b"TEST_DST"
    pks, msgs, sigs, exp = synth.make_batch(O, n, dst, invalid_every=3, uniq=12)       # same batch on both ranks

    def verify_local(lo, hi):
        data, off = M.engine.pack_messages(msgs[lo:hi])
        m = hi - lo
        t_pk = torch.frombuffer(bytearray(pks[128 * lo:128 * hi]), dtype=torch.uint8).to(dev)
        t_sg = torch.frombuffer(bytearray(sigs[64 * lo:64 * hi]), dtype=torch.uint8).to(dev)
        t_ms = torch.frombuffer(bytearray(data) or bytearray(1), dtype=torch.uint8).to(dev)
        t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
        t_bm = torch.zeros((m + 7) // 8, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        fn = eng.verify_batch_rlc_dev if rlc else eng.verify_batch_dev       # same bitmap either way
        fn(t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), m, t_bm.data_ptr(), dst)
        eng.synchronize()
        return t_bm.cpu()          # gloo reduces host tensors
    words = sharded.verify_batch_sharded(verify_local, n, rank, world, dist, torch, torch.device("cpu"))
    got = sharded.words_to_bitmap_bytes(words.numpy(), n)
    np.save(os.path.join(outdir, "v%d.npy" % rank), np.frombuffer(got, dtype=np.uint8))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def _agg_worker(rank, world, port, n, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import blsbn254_loader
    M = blsbn254_loader.load()
    sharded = __import__("bls_bn254_amd.sharded", fromlist=["x"])
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    eng = M.Engine(0)
    dst = b"TEST_DST"
    sks = [synth.sk_of(k) for k in range(n)]
    pks = [O.sk_to_pk(s) for s in sks]
    msgs = [synth.msg_of(i) for i in range(n)]
    agg = O.aggregate_sigs(b"".join(O.sign(s, m, dst) for s, m in zip(sks, msgs)), n)

    def partial_local(lo, hi):
        return eng.aggregate_partial(b"".join(pks[lo:hi]), msgs[lo:hi], dst)

    def finish(partials, k):
        return eng.aggregate_finish(partials, k, agg)
    res = [sharded.aggregate_verify_sharded(partial_local, finish, n, rank, world, dist, torch, torch.device("cpu"))]
    msgs[n - 1] = b"tampered"                       # lives in the last rank's shard
    res.append(sharded.aggregate_verify_sharded(partial_local, finish, n, rank, world, dist, torch, torch.device("cpu")))
    np.save(os.path.join(outdir, "a%d.npy" % rank), np.array(res, dtype=np.uint8))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,rlc", [(37, False), (256, False), (256, True), (2 * 65536, False)])
def test_world2_verify_batch_sharded_with_engine(tmp_path, oracle, n, rlc):
    """n = 2 x 65536: every rank runs a full round of waves on the lane-per-tuple kernels (the shape of a real rank), not the
    wave-per-tuple small-call path; the expectation there is the closed form (the oracle would need minutes)."""
    import torch.multiprocessing as mp
    mp.spawn(_verify_worker, args=(2, _free_port(), n, str(tmp_path), rlc), nprocs=2, join=True)
    pks, msgs, sigs, exp = synth.make_batch(oracle, n, b"TEST_DST", invalid_every=3, uniq=12)
    want = synth.bitmap_of(exp)
    if n <= 4096:
        assert oracle.verify_batch(pks, msgs, sigs, b"TEST_DST", nthreads=4) == want
    for r in range(2):
        assert np.load(tmp_path / ("v%d.npy" % r)).tobytes() == want


def test_world2_aggregate_verify_sharded_with_engine(tmp_path, oracle):
    import torch.multiprocessing as mp
    mp.spawn(_agg_worker, args=(2, _free_port(), 9, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(tmp_path / ("a%d.npy" % r)).tolist() == [1, 0]


def test_c_abi_multi_device_equals_single(oracle):
    """blsbn254_multi over [0, 0] (two contexts, two host threads on the one GPU of the box) == blsbn254_verify_batch and
    == the oracle; aggregate verify likewise (SURVEY.md 8b signature block, 8e)."""
    import blsbn254_loader
    M = blsbn254_loader.load()
    dst = b"TEST_DST"
    with M.Engine(0) as eng, M.MultiEngine([0, 0]) as me, M.MultiEngine([0, 0, 0]) as me3:
        assert me.device_count() == 2
        for n in (1, 7, 8, 9, 37, 600):
            pks, msgs, sigs, exp = synth.make_batch(oracle, n, dst, invalid_every=3, uniq=12)
            want = eng.verify_batch(pks, msgs, sigs, dst)
            assert want == synth.bitmap_of(exp)
            assert me.verify_batch(pks, msgs, sigs, dst) == want
            assert me3.verify_batch(pks, msgs, sigs, dst) == want
            assert me.verify_batch_rlc(pks, msgs, sigs, dst) == want and me3.verify_batch_rlc(pks, msgs, sigs, dst, seed=bytes(32)) == want
        assert me.verify_batch(b"", [], b"", dst) == b""
        n = 11
        sks = [synth.sk_of(k) for k in range(n)]
        pks = b"".join(oracle.sk_to_pk(s) for s in sks)
        msgs = [synth.msg_of(i) + bytes(i) for i in range(n)]           # ragged lengths: shard offsets are relative
        agg = oracle.aggregate_sigs(b"".join(oracle.sign(s, m, dst) for s, m in zip(sks, msgs)), n)
        assert eng.aggregate_verify(pks, msgs, agg, dst) is True
        assert me.aggregate_verify(pks, msgs, agg, dst) is True and me3.aggregate_verify(pks, msgs, agg, dst) is True
        bad = list(msgs); bad[n - 1] = b"tampered"
        assert me.aggregate_verify(pks, bad, agg, dst) is False
        # a public key outside the subgroup in the second shard
        pk_bad = pks[:128 * (n - 1)] + synth.NON_SUBGROUP_PK
        assert me.aggregate_verify(pk_bad, msgs, agg, dst) is False
        # an undecodable tuple is not an error in verify_batch (bit cleared) on either path
        pks2, msgs2, sigs2, _ = synth.make_batch(oracle, 16, dst, uniq=4)
        sigs2 = sigs2[:64 * 9] + b"\xff" * 64 + sigs2[64 * 10:]
        assert me.verify_batch(pks2, msgs2, sigs2, dst) == eng.verify_batch(pks2, msgs2, sigs2, dst)


def test_c_abi_multi_device_resident_rccl(oracle):
    """Device-resident entry point with the RCCL all-reduce of the bitmap words, on the one device of the box (a
    one-rank communicator; the N-rank form is the same code path with ncclCommInitAll over N ordinals).  Listing an
    ordinal twice is an RCCL error, reported as BLSBN254_E_RCCL -- never a silent fallback."""
    import torch
    import blsbn254_loader
    M = blsbn254_loader.load()
    dst = b"TEST_DST"
    dev = torch.device("cuda", 0)
    n = 200
    pks, msgs, sigs, exp = synth.make_batch(oracle, n, dst, invalid_every=3, uniq=12)
    data, off = M.engine.pack_messages(msgs)
    t_pk = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
    t_sg = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev)
    t_ms = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    nwords = (n + 31) // 32
    t_full = torch.full((4 * nwords,), 0xAA, dtype=torch.uint8, device=dev)      # must be overwritten, not accumulated into
    torch.cuda.synchronize()
    with M.MultiEngine([0]) as me:
        me.verify_batch_dev([t_pk.data_ptr()], [t_ms.data_ptr()], [t_off.data_ptr()], [t_sg.data_ptr()], [n], [t_full.data_ptr()], dst)
        got = bytes(t_full.cpu().numpy())
        assert got[:(n + 7) // 8] == synth.bitmap_of(exp) and not any(got[(n + 7) // 8:])
    with M.MultiEngine([0, 0]) as me2:
        t_full2 = torch.zeros(4 * nwords, dtype=torch.uint8, device=dev)
        half = 96
        o2 = (off[half:] - off[half]).astype(np.int64)
        t_off2 = torch.from_numpy(o2).to(dev)
        with pytest.raises(M.Bn254Error) as e:
            me2.verify_batch_dev([t_pk.data_ptr(), t_pk.data_ptr() + 128 * half], [t_ms.data_ptr(), t_ms.data_ptr() + int(off[half])],
                                 [t_off.data_ptr(), t_off2.data_ptr()], [t_sg.data_ptr(), t_sg.data_ptr() + 64 * half], [half, n - half],
                                 [t_full.data_ptr(), t_full2.data_ptr()], dst)
        assert e.value.code == -5


def test_c_abi_multi_device_resident_rccl_two_gpus(oracle):
    """The N > 1 form of the device-resident entry point: two real ordinals, ncclCommInitAll over both, the grouped
    ncclAllReduce of the bitmap words and the shard offsets of k_place_bitmap.  Skipped on a one-GPU box (the pool's test boxes
    have one card; the path is then covered by the one-rank communicator above and by the torch.distributed ranks of bench.py)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL rejects a communicator that lists one ordinal twice")
    import blsbn254_loader
    M = blsbn254_loader.load()
    dst = b"TEST_DST"
    n, half = 608, 320                                   # shard boundary at a multiple of 32
    pks, msgs, sigs, exp = synth.make_batch(oracle, n, dst, invalid_every=3, uniq=12)
    data, off = M.engine.pack_messages(msgs)
    nwords = (n + 31) // 32
    bufs = []
    for g, (lo, hi) in enumerate(((0, half), (half, n))):
        dev = torch.device("cuda", g)
        o = (off[lo:hi + 1] - off[lo]).astype(np.int64)
        bufs.append((torch.frombuffer(bytearray(pks[128 * lo:128 * hi]), dtype=torch.uint8).to(dev),
                     torch.frombuffer(bytearray(data[int(off[lo]):int(off[hi])]), dtype=torch.uint8).to(dev),
                     torch.from_numpy(o).to(dev),
                     torch.frombuffer(bytearray(sigs[64 * lo:64 * hi]), dtype=torch.uint8).to(dev),
                     torch.full((4 * nwords,), 0x55, dtype=torch.uint8, device=dev)))
    for g in range(2):
        torch.cuda.synchronize(g)
    with M.MultiEngine([0, 1]) as me:
        me.verify_batch_dev([b[0].data_ptr() for b in bufs], [b[1].data_ptr() for b in bufs], [b[2].data_ptr() for b in bufs],
                            [b[3].data_ptr() for b in bufs], [half, n - half], [b[4].data_ptr() for b in bufs], dst)
        for b in bufs:                                   # the whole bitmap, in global order, on EVERY device
            got = bytes(b[4].cpu().numpy())
            assert got[:(n + 7) // 8] == synth.bitmap_of(exp) and not any(got[(n + 7) // 8:])


def test_bench_spawns_its_own_ranks():
    """plain `python bench.py --gpus 2` (WORLD_SIZE unset): the parent starts two ranks before touching the GPU and relays
    rank 0's line; both ranks rehearse on device 0 over gloo (the real run is one rank per GPU over RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--backend", "gloo", "--all-on-device0", "--tuples-per-gpu", "8192"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["tuples_total"] == 2 * 8192 and j["value"] > 0
    assert "cpu_baseline" not in j                       # rank 0 at N = 1 only


