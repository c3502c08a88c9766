"""GPU parity tests (MI355X): every C-ABI entry point of include/blsbn254.h, called through the
ctypes binding, against the CPU oracle on the same seeded inputs and against the reference's golden
vectors.  Bit-exact (integer/byte work)."""
import json
import os
import random

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu
H = lambda s: int(s, 16)
ONE_GT = (1).to_bytes(32, "big") + bytes(352)
IDENT1 = bytes(32) + (1).to_bytes(32, "big")
IDENT2 = bytes(64) + bytes(32) + bytes(31) + b"\x01"        # x = 0, y = (c1 = 0, c0 = 1): G2Affine::identity, g2.rs


@pytest.fixture(scope="module")
def eng():
    import blsbn254_loader
    M = blsbn254_loader.load()
    e = M.Engine(0)           # raises when the HIP extension or the GPU is missing: no fallback
    yield e
    e.close()


@pytest.fixture(scope="module")
def M():
    import blsbn254_loader
    return blsbn254_loader.load()


def test_pairing_golden(eng, oracle, kats):
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    assert eng.pairing_batch(G1, G2, 1).hex() == kats["constants"]["gt_generator_bytes_hex"]   # pairings.rs:971-976


def test_pairing_batch_vs_oracle(eng, oracle, pyref):
    rnd = random.Random(1)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    n = 70
    g1 = b"".join(oracle.g1_mul(G1, rnd.randrange(1, pyref.R)) for _ in range(n))
    g2 = b"".join(oracle.g2_mul(G2, rnd.randrange(1, pyref.R)) for _ in range(n))
    # identity handling (pairings.rs:789-800): slots 3 and 5
    g1 = g1[:64 * 3] + IDENT1 + g1[64 * 4:]
    g2 = g2[:128 * 5] + IDENT2 + g2[128 * 6:]
    got = eng.pairing_batch(g1, g2, n)
    assert got == oracle.pairing_batch(g1, g2, n)
    assert got[384 * 3:384 * 4] == ONE_GT and got[384 * 5:384 * 6] == ONE_GT
    assert eng.miller_loop_batch(g1, g2, n) == oracle.miller_loop_batch(g1, g2, n)        # bit-exact Miller values
    ml = eng.multi_miller_loop(g1, g2, n)
    assert ml == oracle.multi_miller_loop(g1, g2, n)
    assert eng.final_exponentiation(ml, 1) == oracle.final_exponentiation(ml, 1)
    assert eng.multi_miller_loop(b"", b"", 0) == ONE_GT


def test_bilinearity_on_gpu(eng, oracle, pyref):
    rnd = random.Random(2)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    a, b = rnd.randrange(1, pyref.R), rnd.randrange(1, pyref.R)
    lhs = eng.pairing_batch(oracle.g1_mul(G1, a), oracle.g2_mul(G2, b), 1)
    assert lhs == oracle.gt_pow(eng.pairing_batch(G1, G2, 1), a * b % pyref.R)
    assert oracle.gt_pow(lhs, pyref.R) == ONE_GT


def test_error_codes(eng, oracle, pyref, M):
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    bad1 = pyref.P.to_bytes(32, "big") + G1[32:]
    with pytest.raises(M.InvalidG1Bytes):
        eng.pairing_batch(G1 + bad1, G2 + G2, 2)
    with pytest.raises(M.InvalidG2Bytes):
        eng.pairing_batch(G1, pyref.P.to_bytes(32, "big") + G2[32:], 1)
    with pytest.raises(M.InvalidGtBytes):
        eng.final_exponentiation(pyref.P.to_bytes(32, "big") + bytes(352), 1)
    with pytest.raises(M.InvalidG1Bytes):
        eng.aggregate_sigs(G1 + bad1, 2)


def test_hash_to_curve_kats(eng, kats):
    g1, g2 = kats["g1"], kats["g2"]
    out = eng.hash_to_g1_batch([v["msg"].encode() for v in g1["hash"]], g1["hash_dst"].encode())
    for i, v in enumerate(g1["hash"]):
        assert out[64 * i:64 * i + 64].hex() == v["p_x"] + v["p_y"]                        # g1.rs:1052-1141
    out = eng.encode_to_g1_batch([v["msg"].encode() for v in g1["encode"]], g1["encode_dst"].encode())
    for i, v in enumerate(g1["encode"]):
        assert out[64 * i:64 * i + 64].hex() == v["p_x"] + v["p_y"]                        # g1.rs:981-1049
    for name, fn in (("hash", eng.hash_to_g2_batch), ("encode", eng.encode_to_g2_batch)):
        out = fn([v["msg"].encode() for v in g2[name]], g2[name + "_dst"].encode())
        for i, v in enumerate(g2[name]):
            assert out[128 * i:128 * i + 128].hex() == v["x_c1"] + v["x_c0"] + v["y_c1"] + v["y_c0"]   # g2.rs:1039-1313


def test_hash_to_curve_ragged(eng, oracle):
    rnd = random.Random(3)
    msgs = [b"", b"a", bytes(55), bytes(56), bytes(63), bytes(64), bytes(65), os.urandom(119), os.urandom(120), os.urandom(300)]
    msgs += [bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 90))) for _ in range(60)]
    for dst in (b"QUUX-V01-CS02-with-BN254G1_XMD:SHA-256_SVDW_RO_", b"d", b"x" * 255, b"y" * 300):
        assert eng.hash_to_g1_batch(msgs, dst) == oracle.hash_to_g1_batch(msgs, dst)
    dst = b"QUUX-V01-CS02-with-BN254G2_XMD:SHA-256_SVDW_RO_"
    assert eng.hash_to_g2_batch(msgs[:20], dst) == oracle.hash_to_g2_batch(msgs[:20], dst)
    assert eng.hash_to_g1_batch([], dst) == b""


def test_point_checks(eng, oracle, pyref, kats):
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    bp = kats["g2"]["bad_point"]
    bad = bytes.fromhex(bp["x_c1"] + bp["x_c0"] + bp["y_c1"] + bp["y_c0"])                  # g2.rs:994-1017
    rnd = random.Random(4)
    pts2 = [bad, G2, IDENT2, synth.NON_SUBGROUP_PK, pyref.P.to_bytes(32, "big") + G2[32:]]
    pts2 += [oracle.g2_mul(G2, rnd.randrange(1, pyref.R)) for _ in range(68)]
    buf = b"".join(pts2)
    want = oracle.g2_check_batch(buf, len(pts2))
    assert want[0] & 0x1f == 0b00110
    assert eng.g2_check_batch(buf, len(pts2)) == want == oracle.g2_check_batch_slow(buf, len(pts2))
    pts1 = [G1, IDENT1, G1[:32] + (3).to_bytes(32, "big"), pyref.P.to_bytes(32, "big") + G1[32:], G1[:32] + pyref.P.to_bytes(32, "big")]
    pts1 += [oracle.g1_mul(G1, rnd.randrange(1, pyref.R)) for _ in range(70)]
    buf = b"".join(pts1)
    assert eng.g1_check_batch(buf, len(pts1)) == oracle.g1_check_batch(buf, len(pts1))
    assert oracle.g1_check_batch(buf, len(pts1))[0] & 0x1f == 0b00011


@pytest.mark.parametrize("n,invalid_every", [(1, 0), (63, 4), (64, 0), (65, 5), (1000, 8)])
def test_verify_batch(eng, oracle, M, n, invalid_every):
    dst = M.DEFAULT_DST
    pks, msgs, sigs, exp = synth.make_batch(oracle, n, dst, invalid_every=invalid_every, uniq=40)
    got = eng.verify_batch(pks, msgs, sigs, dst)
    assert got == synth.bitmap_of(exp)
    if n <= 65:
        assert got == oracle.verify_batch(pks, msgs, sigs, dst)


def test_verify_negative_cases(eng, oracle, M):
    dst = M.DEFAULT_DST
    sks = [synth.sk_of(k) for k in range(2)]
    pks = [oracle.sk_to_pk(s) for s in sks]
    msgs = [b"m0", bytes(range(200))]
    sigs = [oracle.sign(s, m, dst) for s, m in zip(sks, msgs)]
    G1 = oracle.g1_generator()
    cases = [
        (pks[0], b"x", sigs[0]), (pks[1], msgs[0], sigs[0]), (pks[0], msgs[0], oracle.g1_add(sigs[0], G1)),
        (pks[0], msgs[0], sigs[0][:32] + (5).to_bytes(32, "big")), (pks[0], msgs[0], IDENT1), (IDENT2, msgs[0], sigs[0]),
        (synth.NON_SUBGROUP_PK, msgs[0], sigs[0]), (b"\xff" * 128, msgs[0], sigs[0]), (pks[0], msgs[0], b"\xff" * 64),
        (pks[0], msgs[0], sigs[0]), (pks[1], msgs[1], sigs[1]),
    ]
    pk_b = b"".join(c[0] for c in cases); ms = [c[1] for c in cases]; sg = b"".join(c[2] for c in cases)
    got = eng.verify_batch(pk_b, ms, sg, dst)
    assert got == oracle.verify_batch(pk_b, ms, sg, dst) == bytes([0, 0b110])
    assert eng.verify_batch(b"", [], b"", dst) == b""
    # a different ciphersuite tag changes H(msg): valid tuples become invalid
    assert eng.verify_batch(pks[0], [msgs[0]], sigs[0], b"OTHER_DST") == b"\x00"


def test_aggregate(eng, oracle, M):
    dst = M.DEFAULT_DST
    n = 37
    sks = [synth.sk_of(k) for k in range(n)]
    pks = b"".join(oracle.sk_to_pk(s) for s in sks)
    msgs = [synth.msg_of(i) for i in range(n)]
    sigs = b"".join(oracle.sign(s, m, dst) for s, m in zip(sks, msgs))
    agg = eng.aggregate_sigs(sigs, n)
    assert agg == oracle.aggregate_sigs(sigs, n)                                          # impl Sum, g1.rs:561-565
    assert eng.aggregate_sigs(b"", 0) == IDENT1
    assert eng.aggregate_verify(pks, msgs, agg, dst) is True
    bad = list(msgs); bad[17] = b"tampered"
    assert eng.aggregate_verify(pks, bad, agg, dst) is False
    assert eng.aggregate_verify(pks[:128] + synth.NON_SUBGROUP_PK + pks[256:], msgs, agg, dst) is False
    assert eng.aggregate_verify(pks, msgs, IDENT1, dst) is False
    assert eng.aggregate_verify(b"", [], agg, dst) is False
    assert oracle.aggregate_verify(pks, msgs, agg, dst) and not oracle.aggregate_verify(pks, bad, agg, dst)


def test_threshold(eng, oracle, pyref, M):
    dst = M.DEFAULT_DST
    rnd = random.Random(9)
    t, total = 20, 40
    coeffs = [rnd.randrange(1, pyref.R) for _ in range(t)]
    f = lambda x: sum(c * pow(x, i, pyref.R) for i, c in enumerate(coeffs)) % pyref.R
    ids = rnd.sample(range(1, total + 1), t)
    msg = b"threshold message"
    h = oracle.hash_to_g1_batch([msg], dst)
    parts = b"".join(oracle.g1_mul(h, f(i)) for i in ids)
    idb = b"".join(i.to_bytes(32, "big") for i in ids)
    sig = eng.threshold_combine(idb, parts, t)
    assert sig == oracle.threshold_combine(idb, parts, t) == oracle.sign(coeffs[0], msg, dst)
    assert eng.verify_batch(oracle.sk_to_pk(coeffs[0]), [msg], sig, dst) == b"\x01"
    with pytest.raises(M.InvalidScalarBytes):
        eng.threshold_combine(idb[:32] + idb[:32] + idb[64:], parts, t)                   # duplicate id
    with pytest.raises(M.InvalidScalarBytes):
        eng.threshold_combine(pyref.R.to_bytes(32, "big") + idb[32:], parts, t)           # id >= r
    with pytest.raises(M.InvalidScalarBytes):
        eng.threshold_combine(bytes(32) + idb[32:], parts, t)                             # id == 0


def test_full_size_properties(eng, oracle, M):
    """BASELINE config 2 size (262144): the oracle cannot sign that many, so tile a small signed set and
    check size-independent properties: the bitmap equals the closed-form pattern, and flipping one
    message flips exactly one bit."""
    dst = M.DEFAULT_DST
    n = 262144
    pks, msgs, sigs, exp = synth.make_batch(oracle, n, dst, invalid_every=64, uniq=64)
    got = eng.verify_batch(pks, msgs, sigs, dst)
    want = synth.bitmap_of(exp)
    assert got == want
    assert sum(exp) == n - n // 64
    msgs2 = list(msgs); msgs2[123456] = b"flipped"
    got2 = eng.verify_batch(pks, msgs2, sigs, dst)
    diff = np.unpackbits(np.frombuffer(got, dtype=np.uint8) ^ np.frombuffer(got2, dtype=np.uint8), bitorder="little")
    assert diff.sum() == 1 and diff[123456] == 1


def test_device_resident_entry_points(eng, oracle, M):
    torch = pytest.importorskip("torch")
    dst = M.DEFAULT_DST
    n = 300
    pks, msgs, sigs, exp = synth.make_batch(oracle, n, dst, invalid_every=7, uniq=30)
    data, off = M.engine.pack_messages(msgs)
    dev = torch.device("cuda:0")
    t_pk = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
    t_sg = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev)
    t_ms = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    t_bm = torch.zeros((n + 7) // 8, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng.verify_batch_dev(t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), n, t_bm.data_ptr(), dst)
    eng.synchronize()
    assert bytes(t_bm.cpu().numpy()) == synth.bitmap_of(exp)


def test_aggregate_verify_1m(eng, oracle, pyref, M):
    """BASELINE configs[2]: one aggregated G1 signature over 1 048 576 (pk_i, msg_i).  64 signed tuples
    are tiled (the oracle cannot sign a million); the aggregate of the tiled set is (n/64) * sum_64, so
    the expected answer is known: valid, and invalid after corrupting one message."""
    dst = M.DEFAULT_DST
    n, uniq = 1 << 20, 64
    sks = [synth.sk_of(k) for k in range(8)]
    pk_pool = [oracle.sk_to_pk(s) for s in sks]
    base = [(pk_pool[i % 8], synth.msg_of(i)) for i in range(uniq)]
    sig64 = oracle.aggregate_sigs(b"".join(oracle.sign(sks[i % 8], base[i][1], dst) for i in range(uniq)), uniq)
    agg = oracle.g1_mul(sig64, n // uniq)
    pks = b"".join(b[0] for b in base) * (n // uniq)
    msgs = [b[1] for b in base] * (n // uniq)
    assert eng.aggregate_verify(pks, msgs, agg, dst) is True
    msgs[777777] = b"corrupted"
    assert eng.aggregate_verify(pks, msgs, agg, dst) is False


def test_threshold_1000_of_2000(eng, oracle, pyref, M):
    """BASELINE configs[4]: combine 1000 partial signatures (ids drawn from 1..2000) and verify under f(0)*G2."""
    dst = M.DEFAULT_DST
    rnd = random.Random(2000)
    t, total = 1000, 2000
    coeffs = [rnd.randrange(1, pyref.R) for _ in range(t)]
    ids = rnd.sample(range(1, total + 1), t)

    def f(x):
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * x + c) % pyref.R
        return acc
    msg = b"threshold-1000-of-2000"
    h = oracle.hash_to_g1_batch([msg], dst)
    parts = b"".join(oracle.g1_mul(h, f(i)) for i in ids)
    idb = b"".join(i.to_bytes(32, "big") for i in ids)
    sig = eng.threshold_combine(idb, parts, t)
    assert sig == oracle.sign(coeffs[0], msg, dst)
    assert eng.verify_batch(oracle.sk_to_pk(coeffs[0]), [msg], sig, dst) == b"\x01"


def test_sign_and_keygen_kernels(eng, oracle, pyref, M):
    """[sk]H(msg) and [sk]G2gen on the GPU, bit-exact vs the oracle."""
    dst = M.DEFAULT_DST
    rnd = random.Random(77)
    n = 70
    sks = [rnd.randrange(1, pyref.R) for _ in range(n - 3)] + [1, 2, pyref.R - 1]
    msgs = [bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 50))) for _ in range(n)]
    skb = b"".join(s.to_bytes(32, "big") for s in sks)
    sigs = eng.sign_batch(skb, msgs, dst)
    pks = eng.sk_to_pk_batch(skb, n)
    for i in range(n):
        assert sigs[64 * i:64 * i + 64] == oracle.sign(sks[i], msgs[i], dst)
        assert pks[128 * i:128 * i + 128] == oracle.sk_to_pk(sks[i])
    assert eng.verify_batch(pks, msgs, sigs, dst) == synth.bitmap_of([True] * n)
    with pytest.raises(M.InvalidScalarBytes):
        eng.sign_batch(pyref.R.to_bytes(32, "big"), [b"m"], dst)
    with pytest.raises(M.InvalidScalarBytes):
        eng.sk_to_pk_batch(b"\xff" * 32, 1)


def test_keygen_hash_to_scalar_pop(eng, oracle, pyref, M):
    """IETF KeyGen (HKDF, KEYGEN_SALT helpers.rs:3), Scalar::hash (scalar.rs:554-563) and
    proof of possession, against the stdlib-hmac / big-int restatement and the oracle's sign / verify.
    Parity unpinned by the reference (it holds no vectors for these); the checker is independent code."""
    rnd = random.Random(78)
    n = 67
    for ikm_len, info in ((32, b""), (48, b"info")):
        ikm = rnd.randbytes(ikm_len * n)
        sks = eng.keygen_batch(ikm, n, info)
        for i in range(n):
            assert int.from_bytes(sks[32 * i:32 * i + 32], "big") == pyref.keygen(ikm[ikm_len * i:ikm_len * (i + 1)], info)
    with pytest.raises(M.Bn254Error):
        eng.keygen_batch(bytes(31 * 2), 2)                        # IKM shorter than 32 bytes
    dst = b"QUUX-V01-CS02-with-BN254FR_XMD:SHA-256"
    msgs = [b"", b"abc"] + [rnd.randbytes(rnd.randrange(0, 300)) for _ in range(n - 2)]
    hs = eng.hash_to_scalar_batch(msgs, dst)
    for i in range(n):
        assert int.from_bytes(hs[32 * i:32 * i + 32], "big") == pyref.hash_to_scalar(msgs[i], dst)
    # proof of possession: prove == oracle.sign(sk, pk bytes, POP tag); verify accepts those and nothing else
    pks = eng.sk_to_pk_batch(sks, n)
    proofs = eng.pop_prove_batch(sks, n)
    for i in range(0, n, 7):
        sk = int.from_bytes(sks[32 * i:32 * i + 32], "big")
        assert proofs[64 * i:64 * i + 64] == oracle.sign(sk, pks[128 * i:128 * i + 128], M.POP_DST)
    assert eng.pop_verify_batch(pks, proofs, n) == synth.bitmap_of([True] * n)
    swapped = proofs[64:128] + proofs[:64] + proofs[128:]
    assert eng.pop_verify_batch(pks, swapped, n) == synth.bitmap_of([False, False] + [True] * (n - 2))
    sigs_as_pop = eng.sign_batch(sks, [pks[128 * i:128 * i + 128] for i in range(n)], M.DEFAULT_DST)
    assert eng.pop_verify_batch(pks, sigs_as_pop, n) == synth.bitmap_of([False] * n)      # domain separation
    assert eng.keygen_batch(b"", 0) == b"" and eng.pop_prove_batch(b"", 0) == b"" and eng.pop_verify_batch(b"", b"", 0) == b""


def test_abi_edge_cases(eng, oracle, M):
    """Argument errors and empty batches straight at the C ABI (no Python conveniences in between)."""
    import ctypes
    lib = M.load_library()
    ctx = eng._ctx
    z = ctypes.c_size_t(0)
    buf = (ctypes.c_uint8 * 512)()
    off = (ctypes.c_uint64 * 2)(0, 0)
    valid = ctypes.c_int(7)
    assert lib.blsbn254_pairing_batch(ctx, None, None, z, None) == 0                       # n = 0 is a no-op
    assert lib.blsbn254_pairing_batch(ctx, None, buf, ctypes.c_size_t(1), buf) == -1       # NULL operand
    assert lib.blsbn254_verify_batch(ctx, None, None, off, None, z, None, z, None) == 0
    assert lib.blsbn254_verify_batch(ctx, buf, buf, None, buf, ctypes.c_size_t(1), buf, z, buf) == -1
    assert lib.blsbn254_aggregate_verify(ctx, None, None, off, z, buf, None, z, ctypes.byref(valid)) == 0 and valid.value == 0
    assert lib.blsbn254_final_exponentiation(None, buf, ctypes.c_size_t(1), buf) == -1     # NULL ctx
    bad_off = (ctypes.c_uint64 * 2)(5, 2)                                                   # decreasing offsets
    assert lib.blsbn254_hash_to_g1_batch(ctx, buf, bad_off, ctypes.c_size_t(1), buf, ctypes.c_size_t(3), buf) == -1
    # messages window that does not start at offset 0
    msgs = b"XXXXhello"
    o2 = (ctypes.c_uint64 * 2)(4, 9)
    out = (ctypes.c_uint8 * 64)()
    dst = b"dst"
    assert lib.blsbn254_hash_to_g1_batch(ctx, msgs, o2, ctypes.c_size_t(1), dst, ctypes.c_size_t(3), out) == 0
    assert bytes(out) == oracle.hash_to_g1_batch([b"hello"], dst)
    # a second context on the same device works independently
    e2 = M.Engine(0)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    assert e2.pairing_batch(G1, G2, 1) == eng.pairing_batch(G1, G2, 1)
    e2.close()


def test_aggregate_partial_finish(eng, oracle, M):
    """The sharded form of aggregate verify: two partial products + finish == the one-call result."""
    dst = M.DEFAULT_DST
    n = 21
    sks = [synth.sk_of(k) for k in range(n)]
    pks = b"".join(oracle.sk_to_pk(s) for s in sks)
    msgs = [synth.msg_of(i) for i in range(n)]
    agg = oracle.aggregate_sigs(b"".join(oracle.sign(s, m, dst) for s, m in zip(sks, msgs)), n)
    p0, ok0 = eng.aggregate_partial(pks[:128 * 8], msgs[:8], dst)
    p1, ok1 = eng.aggregate_partial(pks[128 * 8:], msgs[8:], dst)
    pe, oke = eng.aggregate_partial(b"", [], dst)
    assert ok0 and ok1 and oke and pe == ONE_GT
    h = oracle.hash_to_g1_batch(msgs[:8], dst)
    assert p0 == oracle.multi_miller_loop(h, pks[:128 * 8], 8)                  # partial == oracle Miller product
    assert eng.aggregate_finish(p0 + p1 + pe, 3, agg) is True
    assert eng.aggregate_finish(p0, 1, agg) is False
    assert eng.aggregate_partial(synth.NON_SUBGROUP_PK, [b"m"], dst)[1] is False
    # every split of the 21 pairs (odd and even shard sizes: the two-pairs-per-lane kernel pads the odd ones with the
    # constant line 1) gives the oracle's Miller product bit for bit
    hall = oracle.hash_to_g1_batch(msgs, dst)
    for lo, hi in ((0, 1), (0, 2), (3, 6), (0, 21), (5, 21), (20, 21)):
        assert eng.aggregate_partial(pks[128 * lo:128 * hi], msgs[lo:hi], dst)[0] == oracle.multi_miller_loop(hall[64 * lo:64 * hi], pks[128 * lo:128 * hi], hi - lo)
    # the shard that carries the signature's pair: partial == oracle product over the n + 1 pairs (sig, -G2gen) included
    from oracle.pyref import bn254 as B
    neg_g2 = B.g2_to_bytes(B.g2_neg(B.G2_GEN))
    ps, oks, sok = eng.aggregate_partial_with_sig(pks[:128 * 8], msgs[:8], agg, dst)
    assert oks and sok and ps == oracle.multi_miller_loop(h + agg, pks[:128 * 8] + neg_g2, 9)
    assert eng.aggregate_finish(ps + p1, 2, None) is True
    assert eng.aggregate_finish(ps, 1, None) is False
    assert eng.aggregate_partial_with_sig(pks[:128 * 8], msgs[:8], IDENT1, dst)[2] is False          # identity signature
    assert eng.aggregate_partial_with_sig(pks[:128 * 8], msgs[:8], agg[:63] + bytes([agg[63] ^ 1]), dst)[2] is False   # off curve


def test_randomized_differential(eng, oracle, pyref, M):
    """Larger seeded differential run: 1500 random pairings (random multiples of the generators, identities and
    negations mixed in), 1500 random-length messages hashed to G1, 600 point encodings of which a third are
    malformed, all compared byte for byte with the oracle."""
    rnd = random.Random(20261004)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    n = 1500
    base1 = [oracle.g1_mul(G1, rnd.randrange(1, pyref.R)) for _ in range(40)] + [IDENT1]
    base2 = [oracle.g2_mul(G2, rnd.randrange(1, pyref.R)) for _ in range(40)] + [IDENT2]
    # more distinct points cheaply: sums of pairs
    pts1 = [oracle.g1_add(rnd.choice(base1), rnd.choice(base1)) for _ in range(200)] + base1
    pts2 = [oracle.g2_add(rnd.choice(base2), rnd.choice(base2)) for _ in range(200)] + base2
    g1 = b"".join(rnd.choice(pts1) for _ in range(n))
    g2 = b"".join(rnd.choice(pts2) for _ in range(n))
    assert eng.pairing_batch(g1, g2, n) == oracle.pairing_batch(g1, g2, n)
    assert eng.multi_miller_loop(g1[:64 * 300], g2[:128 * 300], 300) == oracle.multi_miller_loop(g1[:64 * 300], g2[:128 * 300], 300)
    msgs = [bytes(rnd.getrandbits(8) for _ in range(rnd.choice([0, 1, 31, 32, 33, 55, 56, 64, 100, 200]))) for _ in range(n)]
    dst = bytes(rnd.getrandbits(8) for _ in range(43))
    assert eng.hash_to_g1_batch(msgs, dst) == oracle.hash_to_g1_batch(msgs, dst)
    # encodings: valid points, off-curve points, non-canonical coordinates
    enc1, enc2 = [], []
    for i in range(600):
        a, b = bytearray(rnd.choice(pts1)), bytearray(rnd.choice(pts2))
        k = i % 3
        if k == 1:
            a[rnd.randrange(64)] ^= 1 << rnd.randrange(8); b[rnd.randrange(128)] ^= 1 << rnd.randrange(8)
        elif k == 2:
            a[0] |= 0x80; b[64] |= 0xc0
        enc1.append(bytes(a)); enc2.append(bytes(b))
    e1, e2 = b"".join(enc1), b"".join(enc2)
    assert eng.g1_check_batch(e1, 600) == oracle.g1_check_batch(e1, 600)
    assert eng.g2_check_batch(e2, 600) == oracle.g2_check_batch(e2, 600)


def test_compressed_codecs_gpu(eng, oracle, pyref, M):
    rnd = random.Random(12)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    n = 80
    p1 = [oracle.g1_mul(G1, rnd.randrange(1, pyref.R)) for _ in range(n - 2)] + [G1, IDENT1]
    p2 = [oracle.g2_mul(G2, rnd.randrange(1, pyref.R)) for _ in range(n - 2)] + [G2, IDENT2]
    c1 = eng.g1_compress_batch(b"".join(p1), n); c2 = eng.g2_compress_batch(b"".join(p2), n)
    assert c1 == b"".join(oracle.g1_compress(p) for p in p1) and c2 == b"".join(oracle.g2_compress(p) for p in p2)
    assert eng.g1_decompress_batch(c1, n) == b"".join(p1) and eng.g2_decompress_batch(c2, n) == b"".join(p2)
    x = 1
    while pyref.fp_sqrt((x ** 3 + 3) % pyref.P) is not None:
        x += 1
    with pytest.raises(M.InvalidG1Bytes):
        eng.g1_decompress_batch(c1[:32] + x.to_bytes(32, "big"), 2)             # no point with that x
    with pytest.raises(M.InvalidG2Bytes):
        eng.g2_decompress_batch(b"\x7f" + b"\xff" * 63, 1)                      # coordinate >= p


@pytest.mark.parametrize("n,invalid_every", [(1, 0), (15, 0), (16, 3), (17, 0), (1000, 8), (4096, 64), (777, 1)])
def test_verify_batch_rlc_matches_exact_path(eng, oracle, M, n, invalid_every):
    """the RLC entry point must return the SAME bitmap as the exact per-tuple path."""
    dst = M.DEFAULT_DST
    pks, msgs, sigs, exp = synth.make_batch(oracle, n, dst, invalid_every=invalid_every, uniq=40)
    want = synth.bitmap_of(exp)
    assert eng.verify_batch(pks, msgs, sigs, dst) == want
    assert eng.verify_batch_rlc(pks, msgs, sigs, dst, seed=bytes(range(32))) == want
    assert eng.verify_batch_rlc(pks, msgs, sigs, dst) == want                       # fresh random seed


def test_verify_batch_rlc_adversarial_pairs(eng, oracle, pyref, M):
    """Two signatures that are individually wrong but whose errors cancel in an UNWEIGHTED product
    (sig_0 + D, sig_1 - D): the random weights must catch them."""
    dst = M.DEFAULT_DST
    sks = [synth.sk_of(k) for k in range(2)]
    pks = b"".join(oracle.sk_to_pk(s) for s in sks)
    msgs = [b"a", b"b"]
    s0, s1 = (oracle.sign(s, m, dst) for s, m in zip(sks, msgs))
    D = oracle.g1_mul(oracle.g1_generator(), 12345)
    negD = D[:32] + (pyref.P - int.from_bytes(D[32:], "big")).to_bytes(32, "big")
    sigs = oracle.g1_add(s0, D) + oracle.g1_add(s1, negD)
    assert eng.verify_batch(pks, msgs, sigs, dst) == b"\x00"
    for t in range(4):
        assert eng.verify_batch_rlc(pks, msgs, sigs, dst, seed=bytes([t]) * 32) == b"\x00"


def test_verify_batch_rlc_repeated_keys_cancelling_errors_in_one_chunk(eng, oracle, pyref, M):
    """Repeated-key RLC path: two signatures under the SAME key whose errors cancel in an unweighted sum land in the same
    chunk; the random weights must catch them, and the honest tuples of that chunk must still come out valid (fallback)."""
    dst = M.DEFAULT_DST
    sk = synth.sk_of(0)
    pk = oracle.sk_to_pk(sk)
    msgs = [b"m%d" % i for i in range(6)]
    good = [oracle.sign(sk, m, dst) for m in msgs]
    D = oracle.g1_mul(oracle.g1_generator(), 777)
    negD = D[:32] + (pyref.P - int.from_bytes(D[32:], "big")).to_bytes(32, "big")
    sigs = oracle.g1_add(good[0], D) + oracle.g1_add(good[1], negD) + b"".join(good[2:])
    want = bytes([0b111100])
    assert eng.verify_batch(pk * 6, msgs, sigs, dst) == want
    before = eng.rlc_stats()
    for t in range(4):
        assert eng.verify_batch_rlc(pk * 6, msgs, sigs, dst, seed=bytes([t + 1]) * 32) == want
    after = eng.rlc_stats()
    assert after["chunked_tuples"] - before["chunked_tuples"] == 24          # the repeated-key path ran ...
    assert after["fallback_tuples"] - before["fallback_tuples"] == 24        # ... and every chunk failed: all six re-verified exactly


@pytest.mark.parametrize("group", [2, 3, 16, 100])
def test_verify_batch_rlc_repeated_keys_group_sizes(eng, oracle, M, group):
    """Chunk sizes that do and do not divide the runs; keys of very different multiplicity (1 ... hundreds), one invalid key,
    invalid tuples of every kind; bitmap == exact path == closed-form expectation, and the counters add up."""
    dst = M.DEFAULT_DST
    n = 1500
    pks, msgs, sigs, exp = synth.make_batch(oracle, n, dst, pool=7, invalid_every=9, uniq=60)
    # skew the multiplicities: the last 300 tuples all use the key of tuple 0 (most of their signatures become wrong); the
    # expectation for that tail comes from the CPU oracle
    pks = pks[:128 * (n - 300)] + pks[:128] * 300
    tail = oracle.verify_batch(pks[128 * (n - 300):], msgs[n - 300:], sigs[64 * (n - 300):], dst, nthreads=8)
    want = synth.bitmap_of(exp[:n - 300]) + tail
    assert 0 < sum(bin(b).count("1") for b in tail) < 300
    assert eng.verify_batch(pks, msgs, sigs, dst) == want
    eng.set_rlc_group(group)
    try:
        before = eng.rlc_stats()
        assert eng.verify_batch_rlc(pks, msgs, sigs, dst) == want
        after = eng.rlc_stats()
    finally:
        eng.set_rlc_group(0)                      # back to automatic
    assert after["chunked_tuples"] - before["chunked_tuples"] == n
    assert 0 < after["chunks"] - before["chunks"] <= n // group + 9      # the chunks of the keys that failed the key round; at most one partial chunk per key
    assert 0 < after["fallback_tuples"] - before["fallback_tuples"] <= n


def test_verify_batch_rlc_all_valid_has_no_fallback(eng, oracle, M):
    dst = M.DEFAULT_DST
    n = 2048
    pks, msgs, sigs, exp = synth.make_batch_gpu(eng, oracle, n, dst, pool=16, invalid_every=0, spot=20)
    eng.set_rlc_key_round(True)                               # (also clears the back-off earlier failing batches left behind)
    before = eng.rlc_stats()
    assert eng.verify_batch_rlc(pks, msgs, sigs, dst) == synth.bitmap_of(exp) == b"\xff" * (n // 8)
    after = eng.rlc_stats()
    assert after["chunked_tuples"] - before["chunked_tuples"] == n
    # the key round (16 checks, one per key) decides an all-valid batch: no chunk is looked at
    assert (after["key_rounds"] - before["key_rounds"], after["key_rounds_passed"] - before["key_rounds_passed"]) == (1, 1)
    assert after["chunks"] == before["chunks"] and after["fallback_tuples"] == before["fallback_tuples"]
    eng.set_rlc_key_round(False)                              # without it: every chunk is checked, none fails
    try:
        assert eng.verify_batch_rlc(pks, msgs, sigs, dst) == b"\xff" * (n // 8)
    finally:
        eng.set_rlc_key_round(True)
    last = eng.rlc_stats()
    assert last["chunks"] - after["chunks"] == n // 16 and last["fallback_tuples"] == after["fallback_tuples"] and last["key_rounds"] == after["key_rounds"]
    # one bad signature: its key fails the key round, its chunk fails the chunk round, 16 tuples are re-verified
    bad = bytearray(sigs); bad[64 * 100:64 * 101] = sigs[64 * 101:64 * 102]
    want = bytearray(b"\xff" * (n // 8)); want[100 // 8] &= ~(1 << (100 % 8)) & 0xff
    assert eng.verify_batch_rlc(pks, msgs, bytes(bad), dst) == bytes(want)
    end = eng.rlc_stats()
    assert (end["key_rounds"] - last["key_rounds"], end["key_rounds_passed"] - last["key_rounds_passed"]) == (1, 0)
    assert end["chunks"] - last["chunks"] == n // 16 // 16 and end["fallback_tuples"] - last["fallback_tuples"] == 16    # only the 8 chunks of that key
    # back-off: after that failure the next two batches skip the key round (all chunks are checked), the third runs it again
    for expect_round in (0, 0, 1):
        s0 = eng.rlc_stats()
        assert eng.verify_batch_rlc(pks, msgs, sigs, dst) == b"\xff" * (n // 8)
        s1 = eng.rlc_stats()
        assert s1["key_rounds"] - s0["key_rounds"] == expect_round and s1["chunks"] - s0["chunks"] == (0 if expect_round else n // 16)


@pytest.mark.parametrize("n,pool", [(3000, 2), (40000, 3)])
def test_verify_batch_rlc_key_round_over_long_runs(eng, oracle, M, n, pool):
    """keys with hundreds / thousands of chunks each: the key round sums them level by level (key_sums); a valid batch is
    decided by `pool` checks, one wrong signature sends exactly its key's chunks to the chunk round"""
    dst = M.DEFAULT_DST
    pks, msgs, sigs, exp = synth.make_batch_gpu(eng, oracle, n, dst, pool=pool, invalid_every=0, spot=10)
    full = synth.bitmap_of(exp)
    eng.set_rlc_key_round(True)
    s0 = eng.rlc_stats()
    assert eng.verify_batch_rlc(pks, msgs, sigs, dst) == full
    s1 = eng.rlc_stats()
    assert s1["key_rounds_passed"] - s0["key_rounds_passed"] == 1 and s1["chunks"] == s0["chunks"]
    j = n // 2 + 1
    bad = bytearray(sigs); bad[64 * j:64 * j + 64] = sigs[64 * (j - 1):64 * j]
    want = bytearray(full); want[j // 8] &= ~(1 << (j % 8)) & 0xff
    assert eng.verify_batch_rlc(pks, msgs, bytes(bad), dst) == bytes(want)
    s2 = eng.rlc_stats()
    run = len(range(j % pool, n, pool))                        # tuples of the bad tuple's key
    assert s2["chunks"] - s1["chunks"] == (run + 15) // 16 and s2["fallback_tuples"] - s1["fallback_tuples"] in (15, 16)
    eng.set_rlc_key_round(True)


def test_verify_batch_rlc_dev_full_size(eng, oracle, M):
    """BASELINE configs[1] size through the device entry point: same bitmap as verify_batch_dev and as the closed form."""
    import torch
    dst = M.DEFAULT_DST
    n = 262144
    pks, msgs, sigs, exp = synth.make_batch_gpu(eng, oracle, n, dst, pool=1024, invalid_every=64, spot=50)
    data, off = M.engine.pack_messages(msgs)
    dev = torch.device("cuda", 0)
    t_pk = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
    t_sg = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev)
    t_ms = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    t_a = torch.zeros((n + 7) // 8, dtype=torch.uint8, device=dev)
    t_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng.verify_batch_dev(t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), n, t_a.data_ptr(), dst)
    eng.verify_batch_rlc_dev(t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), n, t_b.data_ptr(), dst)
    eng.synchronize()
    want = synth.bitmap_of(exp)
    assert bytes(t_a.cpu().numpy()) == want
    assert bytes(t_b.cpu().numpy()) == want


def test_verify_batch_dev_1m_per_gpu_shape(eng, oracle, M):
    """BASELINE configs[3]'s per-GPU shape: 1 048 576 unique GPU-signed tuples (1024-key pool, 1/64 invalid) through the device
    entry point, on the prepared-key path and with BLSBN254_AUTO_PREPARE=0 (the exact per-tuple path): both bitmaps equal the
    closed form.  (The 8-GPU run itself is the driver's; this is what every rank of it executes.)"""
    import torch
    dst = M.DEFAULT_DST
    n = 1 << 20
    pks, msgs, sigs, exp = synth.make_batch_gpu(eng, oracle, n, dst, pool=1024, invalid_every=64, spot=64)
    want = synth.bitmap_of(exp)
    assert want == synth.bitmap_of(synth.expected_bits(n, 64))
    data, off = M.engine.pack_messages(msgs)
    dev = torch.device("cuda", 0)
    t_pk = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
    t_sg = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev)
    t_ms = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    del pks, sigs, data, msgs
    t_bm = torch.zeros((n + 7) // 8, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    p0, e0 = eng.path_stats()
    eng.verify_batch_dev(t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), n, t_bm.data_ptr(), dst)
    eng.synchronize()
    assert bytes(t_bm.cpu().numpy()) == want
    assert eng.path_stats() == (p0 + 1, e0)
    t_bm.zero_()
    eng.set_auto_prepare(False)
    try:
        eng.verify_batch_dev(t_pk.data_ptr(), t_ms.data_ptr(), t_off.data_ptr(), t_sg.data_ptr(), n, t_bm.data_ptr(), dst)
        eng.synchronize()
    finally:
        eng.set_auto_prepare(True)
    assert bytes(t_bm.cpu().numpy()) == want
    assert eng.path_stats() == (p0 + 1, e0 + 1)


def test_aggregate_verify_all_distinct_after_small_verify(oracle, M):
    """Regression: verify_batch of 64 tuples sizes the key-id buffers for 64 entries; an aggregate_verify over 136
    DISTINCT keys on the same context then appends the signature's key (-G2gen) as entry u = n of kd_keys -- one past 4 n bytes."""
    dst = M.DEFAULT_DST
    e = M.Engine(0)
    try:
        pks, msgs, sigs, exp = synth.make_batch(oracle, 64, dst, invalid_every=5, uniq=16)
        assert e.verify_batch(pks, msgs, sigs, dst) == synth.bitmap_of(exp)
        n = 136
        sks = [synth.sk_of(1000 + k) for k in range(n)]
        skb = b"".join(x.to_bytes(32, "big") for x in sks)
        pk = e.sk_to_pk_batch(skb, n)
        assert pk[:128] == oracle.sk_to_pk(sks[0]) and pk[-128:] == oracle.sk_to_pk(sks[-1])
        ms = [synth.msg_of(7000 + i) for i in range(n)]
        sg = e.sign_batch(skb, ms, dst)
        agg = e.aggregate_sigs(sg, n)
        assert agg == oracle.aggregate_sigs(sg, n)
        assert e.aggregate_verify(pk, ms, agg, dst) is True
        ms[n - 1] = b"tampered"
        assert e.aggregate_verify(pk, ms, agg, dst) is False
    finally:
        e.close()


def test_verify_batch_rlc_edge_cases(eng, oracle, pyref, M, monkeypatch):
    """Repeated-key RLC path on awkward batches: a key none of whose tuples is eligible (every signature off the curve), a key
    outside the subgroup carrying many tuples, the distinct / repeated switch-over (n = 2 u), identity signatures, and a batch
    beyond one launch chunk (chunk forced to 64: every chunk de-duplicates and weighs on its own)."""
    dst = M.DEFAULT_DST
    sks = [synth.sk_of(k) for k in range(8)]
    pkp = [oracle.sk_to_pk(s) for s in sks]
    n = 96
    msgs = [synth.msg_of(500 + i) for i in range(n)]
    kidx = [i % 4 for i in range(n)]
    sigs = [oracle.sign(sks[k], m, dst) for k, m in zip(kidx, msgs)]
    pks = [pkp[k] for k in kidx]
    exp = [True] * n
    for i in range(n):
        if kidx[i] == 1:                                   # every signature of key 1: off the curve
            sigs[i] = sigs[i][:32] + (int.from_bytes(sigs[i][32:], "big") ^ 1).to_bytes(32, "big"); exp[i] = False
        elif kidx[i] == 2:                                 # key 2 replaced by a point outside the subgroup
            pks[i] = synth.NON_SUBGROUP_PK; exp[i] = False
        elif i in (3, 7):                                  # identity signatures under key 3
            sigs[i] = IDENT1; exp[i] = False
    want = synth.bitmap_of(exp)
    for group in (0, 2, 5):
        eng.set_rlc_group(group)
        assert eng.verify_batch_rlc(b"".join(pks), msgs, b"".join(sigs), dst) == want
    eng.set_rlc_group(0)
    assert eng.verify_batch(b"".join(pks), msgs, b"".join(sigs), dst) == want
    # switch-over: 8 distinct keys; 16 tuples take the chunked path, 15 the distinct-key variant -- same bitmaps
    for m in (15, 16):
        p2, m2, s2, e2 = synth.make_batch(oracle, m, dst, pool=8, invalid_every=4, uniq=m)
        before = eng.rlc_stats()
        assert eng.verify_batch_rlc(p2, m2, s2, dst) == synth.bitmap_of(e2)
        after = eng.rlc_stats()
        assert (after["chunked_tuples"] - before["chunked_tuples"], after["distinct_key_tuples"] - before["distinct_key_tuples"]) == ((16, 0) if m == 16 else (0, 15))
    monkeypatch.setenv("BLSBN254_CHUNK_LANES", "64")
    e = M.Engine(0)
    try:
        n = 64 * 3 + 17
        p3, m3, s3, e3 = synth.make_batch(oracle, n, dst, invalid_every=5, uniq=16)
        assert e.verify_batch_rlc(p3, m3, s3, dst) == synth.bitmap_of(e3)
        st = e.rlc_stats()                       # the 17-tuple tail holds 9 distinct keys: it takes the exact path
        assert st["chunked_tuples"] == 192 and st["chunked_tuples"] + st["distinct_key_tuples"] == n
    finally:
        e.close()


def test_wide_final_exponentiation_equals_serial(eng, oracle, pyref, M, monkeypatch):
    """Launches of at most 2048 tuples run the hard part of the final exponentiation with one WAVE per tuple
    (k_fe_wide.hip); larger ones three lanes per tuple, or with BLSBN254_WIDE_FE=0 one lane per tuple: same Gt bytes, same bitmaps, at the
    switch-over sizes and through every mode of the finishing step (Gt bytes, verify bitmap, single flag)."""
    rnd = random.Random(4096)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    base = 24
    p1 = [oracle.g1_mul(G1, rnd.randrange(1, pyref.R)) for _ in range(base)]
    p2 = [oracle.g2_mul(G2, rnd.randrange(1, pyref.R)) for _ in range(base)]
    want = oracle.pairing_batch(b"".join(p1), b"".join(p2), base)
    monkeypatch.setenv("BLSBN254_WIDE_FE", "0")
    serial = M.Engine(0)
    monkeypatch.delenv("BLSBN254_WIDE_FE")
    try:
        for n in (1, 63, 2048, 2049, 4097):
            g1 = b"".join(p1[i % base] for i in range(n)); g2 = b"".join(p2[i % base] for i in range(n))
            exp = b"".join(want[384 * (i % base):384 * (i % base) + 384] for i in range(n))
            assert eng.pairing_batch(g1, g2, n) == exp                   # wide up to 2048, three lanes per tuple beyond
            if n <= 63:
                assert serial.pairing_batch(g1, g2, n) == exp            # serial kernels on the small sizes too
        dst = M.DEFAULT_DST
        for n in (5, 300):
            pks, msgs, sigs, e = synth.make_batch(oracle, n, dst, invalid_every=4, uniq=20)
            assert eng.verify_batch(pks, msgs, sigs, dst) == serial.verify_batch(pks, msgs, sigs, dst) == synth.bitmap_of(e)
        sks = [synth.sk_of(k) for k in range(5)]
        pk5 = b"".join(oracle.sk_to_pk(s) for s in sks); m5 = [synth.msg_of(i) for i in range(5)]
        agg = oracle.aggregate_sigs(b"".join(oracle.sign(s, m, dst) for s, m in zip(sks, m5)), 5)
        for e_ in (eng, serial):
            assert e_.aggregate_verify(pk5, m5, agg, dst) is True
            assert e_.aggregate_verify(pk5, m5[:4] + [b"x"], agg, dst) is False
    finally:
        serial.close()


def test_chunked_entry_points(oracle, pyref, M, monkeypatch):
    """Batches larger than the per-launch chunk (4 Mi tuples in production; forced to 64 here through the
    BLSBN254_CHUNK_LANES test knob) are processed chunk by chunk: same results as the one-launch path."""
    monkeypatch.setenv("BLSBN254_CHUNK_LANES", "64")
    e = M.Engine(0)
    try:
        n = 64 * 3 + 17
        pks, msgs, sigs, exp = synth.make_batch(oracle, n, M.DEFAULT_DST, invalid_every=5, uniq=16)
        assert e.verify_batch(pks, msgs, sigs, M.DEFAULT_DST) == synth.bitmap_of(exp)
        rnd = random.Random(90)
        G1, G2 = oracle.g1_generator(), oracle.g2_generator()
        m = 64 * 2 + 5
        g1 = b"".join(oracle.g1_mul(G1, rnd.randrange(1, pyref.R)) for _ in range(m))
        g2 = b"".join(oracle.g2_mul(G2, rnd.randrange(1, pyref.R)) for _ in range(m))
        gt = e.pairing_batch(g1, g2, m)
        assert gt == oracle.pairing_batch(g1, g2, m)
        ml = e.miller_loop_batch(g1, g2, m)
        assert ml == oracle.miller_loop_batch(g1, g2, m)
        assert e.final_exponentiation(ml, m) == gt
        bad = g1[:64 * 130] + b"\xff" * 32 + g1[64 * 130 + 32:]         # element 130 (third chunk): x >= p does not decode
        with pytest.raises(M.InvalidG1Bytes):
            e.pairing_batch(bad, g2, m)
    finally:
        e.close()


def test_chunked_prepared_paths(oracle, M, monkeypatch):
    """Chunked calls on the prepared-key paths (chunk forced to 2048): every chunk de-duplicates and prepares its own keys;
    the explicit G2Prepared form walks the same key table chunk by chunk.  Same bitmap as the unchunked engine and the oracle."""
    dst = M.DEFAULT_DST
    n = 2048 * 2 + 900
    pks, msgs, sigs, exp = _mixed_batch(oracle, n, dst, pool=5, seed=4242)
    want = synth.bitmap_of(exp)
    monkeypatch.setenv("BLSBN254_CHUNK_LANES", "2048")
    e = M.Engine(0)
    try:
        p0, e0 = e.path_stats()
        assert e.verify_batch(pks, msgs, sigs, dst) == want
        # the two full chunks take the prepared path in either mode; the 900-tuple tail takes it as a small chunk (one wave per
        # tuple, or three lanes per tuple) only while those kernels are on -- with BLSBN254_WIDE_FE=0 and BLSBN254_TRI_MAX=0
        # (documented production knobs) a chunk below 1024 tuples goes down the exact per-tuple path
        wide = os.environ.get("BLSBN254_WIDE_FE", "1") != "0" and int(os.environ.get("BLSBN254_WIDE_FE_MAX", "2048")) >= 900
        tri = (int(os.environ.get("BLSBN254_TRI_MAX", "16384")) >= 900 and os.environ.get("BLSBN254_TRI_MILLER", "1") != "0"
               and os.environ.get("BLSBN254_TRI_FE", "1") != "0")          # three lanes per tuple read the tables as well
        assert e.path_stats() == ((p0 + 3, e0) if (wide or tri) else (p0 + 2, e0 + 1))
        keys = sorted(set(pks[128 * i:128 * i + 128] for i in range(n)))
        index = {k: j for j, k in enumerate(keys)}
        prep = e.g2_prepare_batch(b"".join(keys), len(keys))
        assert e.verify_batch_prepared(prep, [index[pks[128 * i:128 * i + 128]] for i in range(n)], msgs, sigs, dst) == want
        prep.close()
    finally:
        e.close()


def _rand_coeffs(seed, n, per):
    """n elements of `per` canonical 32-byte big-endian coefficients (top byte < 0x30 keeps every value below p); the first
    elements carry the edge values 0, 1, p - 1."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n * per, 32), dtype=np.uint8)
    a[:, 0] %= 0x30
    edge = [0, 1, synth.P - 1, 2, synth.P - 2, (synth.P + 1) // 2, (synth.P - 1) // 2, 1 << 29, (1 << 29) - 1, 1 << 58, 1 << 232, (1 << 232) - 1,
            1 << 253, (1 << 253) - 1, synth.P // 3, 3]            # limb boundaries of the 9 x 29-bit form, long carry runs
    for k, v in enumerate(edge[:min(len(edge), n * per)]):
        a[k] = np.frombuffer(v.to_bytes(32, "big"), dtype=np.uint8)
    return a.tobytes()


@pytest.mark.parametrize("op,per,n", [
    (0, 1, 1 << 20), (1, 1, 1 << 20), (3, 1, 1 << 20), (4, 1, 1 << 20), (5, 1, 1 << 20), (8, 1, 1 << 20),       # Fp mul sqr add sub neg x9
    (2, 1, 1 << 17), (6, 1, 20000), (7, 1, 20000),                                                                 # Fp inv (divstep recurrence: 609 steps must do for every input) sqrt is_square
    (16, 2, 1 << 18), (17, 2, 1 << 18), (19, 2, 1 << 18), (20, 2, 1 << 18), (18, 2, 20000), (21, 2, 2000),         # Fp2
    (32, 6, 1 << 17), (33, 6, 1 << 17), (35, 6, 1 << 17), (34, 6, 10000),                                          # Fp6
    (48, 12, 1 << 16), (49, 12, 1 << 16), (51, 12, 1 << 16), (52, 12, 1 << 16), (53, 12, 1 << 16), (54, 12, 1 << 16),
    (55, 12, 1 << 16), (56, 12, 1 << 16), (50, 12, 10000)])                                                        # Fp12
def test_field_primitives_fuzz_vs_oracle(eng, oracle, op, per, n):
    """Primitive-level parity: every Fp / Fp2 / Fp6 / Fp12 operation of the device arithmetic,
    element-wise on random operands (up to 2^20 wide), bit-exact against the CPU oracle.  fp6.rs / fp12.rs hold no reference
    vectors, so this -- with the oracle itself checked against the independent Python model on CPU -- is their isolated pin."""
    a = _rand_coeffs(1000 + op, n, per)
    b = _rand_coeffs(2000 + op, n, per) if op in eng.FIELD_OP_BINARY else None
    got = eng.field_op_batch(op, a, b, n)
    want = oracle.field_op_batch(op, a, b, n)
    if got != want:
        w = 32 * per
        bad = [i for i in range(n) if got[w * i:w * i + w] != want[w * i:w * i + w]]
        raise AssertionError("op %d: %d of %d elements differ, first at %d" % (op, len(bad), n, bad[0]))


def test_field_op_abi_errors(eng, M):
    a = (synth.P).to_bytes(32, "big")                       # not canonical
    with pytest.raises(M.InvalidGtBytes):
        eng.field_op_batch(1, a, None, 1)
    with pytest.raises(M.Bn254Error) as e:
        eng._chk(eng._lib.blsbn254_field_op_batch(eng._ctx, 99, None, None, 1, None))
    assert e.value.code == -1
    assert eng.field_op_batch(0, b"", b"", 0) == b""


def test_gt_group_ops(eng, oracle, pyref):
    """Gt multiply and Gt::mul_by_scalar (pairings.rs:585-600) against the oracle; gt^r == 1 (pairings.rs:977-979)."""
    rnd = random.Random(5)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    n = 24
    g1 = b"".join(oracle.g1_mul(G1, rnd.randrange(1, pyref.R)) for _ in range(n))
    g2 = b"".join(oracle.g2_mul(G2, rnd.randrange(1, pyref.R)) for _ in range(n))
    gts = eng.pairing_batch(g1, g2, n)
    ks = [0, 1, 2, pyref.R, pyref.R - 1, (1 << 256) - 1] + [rnd.randrange(1 << 256) for _ in range(n - 6)]
    kb = b"".join(k.to_bytes(32, "big") for k in ks)
    got = eng.gt_pow_batch(gts, kb, n)
    want = b"".join(oracle.gt_pow(gts[384 * i:384 * i + 384], ks[i]) for i in range(n))
    assert got == want
    assert got[:384] == ONE_GT and got[384:768] == gts[384:768] and got[384 * 3:384 * 4] == ONE_GT      # k = 0, 1, r
    rot = gts[384:] + gts[:384]
    assert eng.gt_mul_batch(gts, rot, n) == b"".join(oracle.gt_mul(gts[384 * i:384 * i + 384], rot[384 * i:384 * i + 384]) for i in range(n))
    # bilinearity through the new entry points: e(aP, Q) = e(P, Q)^a
    a = rnd.randrange(1, pyref.R)
    assert eng.pairing_batch(oracle.g1_mul(G1, a), G2, 1) == eng.gt_pow_batch(eng.pairing_batch(G1, G2, 1), a.to_bytes(32, "big"), 1)


@pytest.mark.parametrize("t", [1, 2, 3, 63, 64, 65, 257, 1000, 5000])
def test_threshold_combine_sizes_and_lagrange(eng, oracle, pyref, M, t):
    """The chip-filling threshold path (Lagrange over t x sqrt(t) lanes, GLV-split windowed MSM) against the oracle's
    t x 255-step ladders: same bytes, for sizes that exercise one lane, partial waves, several chunks and the chunk fold
    (t = 5000: 40 chunks).  Random 254-bit ids (not small integers) and random points incl. the identity and repeats."""
    rnd = random.Random(100 + t)
    R = pyref.R
    ids = rnd.sample(range(1, 1 << 20), t) if t > 300 else [rnd.randrange(1, R) for _ in range(t)]
    idb = b"".join(i.to_bytes(32, "big") for i in ids)
    G1 = oracle.g1_generator()
    pool = [oracle.g1_mul(G1, rnd.randrange(1, R)) for _ in range(min(t, 24))] + [IDENT1]
    parts = b"".join(pool[rnd.randrange(len(pool))] for _ in range(t))
    lam = eng.lagrange_at_zero(idb, t)
    if t <= 1000:
        assert lam == oracle.fr_lagrange_at_zero(idb, t)
    else:       # O(t^2) on the CPU: spot-check 40 coefficients with Python integers
        for i in rnd.sample(range(t), 40):
            num = den = 1
            for j in range(t):
                if j != i:
                    num = num * ids[j] % R; den = den * (ids[j] - ids[i]) % R
            assert int.from_bytes(lam[32 * i:32 * i + 32], "big") == num * pow(den, R - 2, R) % R
    got = eng.threshold_combine(idb, parts, t)
    if t <= 1000:
        assert got == oracle.threshold_combine(idb, parts, t)
    else:       # sum_i lambda_i sigma_i with the coefficients already checked: group the equal points
        acc = {}
        for i in range(t):
            p = parts[64 * i:64 * i + 64]
            acc[p] = (acc.get(p, 0) + int.from_bytes(lam[32 * i:32 * i + 32], "big")) % R
        want = IDENT1
        for p, k in acc.items():
            if p != IDENT1 and k:
                want = oracle.g1_add(want, oracle.g1_mul(p, k))
        assert got == want
    if t >= 3:
        with pytest.raises(M.InvalidScalarBytes):
            eng.threshold_combine(idb[:32 * (t - 1)] + idb[:32], parts, t)              # last id repeats the first
        with pytest.raises(M.InvalidG1Bytes):
            eng.threshold_combine(idb, parts[:64 * (t - 1)] + b"\xff" * 64, t)          # undecodable partial signature


def _mixed_batch(oracle, n, dst, pool, seed):
    """n tuples over `pool` honest keys plus bad keys (outside the subgroup, undecodable, identity) and every kind of bad
    signature / message; returns (pks, msgs, sigs, expected bools)."""
    rnd = random.Random(seed)
    sks = [synth.sk_of(k) for k in range(pool)]
    pkp = [oracle.sk_to_pk(s) for s in sks]
    uniq = min(n, 96)
    base = []
    for i in range(uniq):
        m = synth.msg_of(1000 * seed + i) + bytes(i % 5)
        base.append((i % pool, m, oracle.sign(sks[i % pool], m, dst)))
    g1 = oracle.g1_generator()
    pks, msgs, sigs, exp = [], [], [], []
    for i in range(n):
        k, m, s = base[rnd.randrange(uniq)]
        pk, ok, kind = pkp[k], True, rnd.randrange(24)
        if kind == 0: m, ok = m + b"!", False
        elif kind == 1: s, ok = oracle.g1_add(s, g1), False
        elif kind == 2: pk, ok = pkp[(k + 1) % pool], pool == 1
        elif kind == 3: s, ok = s[:63] + bytes([s[63] ^ 1]), False
        elif kind == 4: pk, ok = synth.NON_SUBGROUP_PK, False
        elif kind == 5: pk, ok = b"\xff" * 32 + pk[32:], False
        elif kind == 6: pk, ok = IDENT2, False
        elif kind == 7: s, ok = IDENT1, False
        pks.append(pk); msgs.append(m); sigs.append(s); exp.append(ok)
    return b"".join(pks), msgs, b"".join(sigs), exp


@pytest.mark.parametrize("n,pool", [(1024, 1), (1500, 7), (4096, 64), (20000, 300)])
def test_prepared_key_path_equals_exact_path(eng, oracle, M, n, pool):
    """verify_batch with the automatic key de-duplication + per-key line tables (G2Prepared, pairings.rs:609-660) against
    the exact per-tuple path and the oracle: same bitmap on batches that mix honest repeated keys with keys outside the
    subgroup, undecodable keys, the identity key, and every kind of bad signature and message."""
    dst = M.DEFAULT_DST
    pks, msgs, sigs, exp = _mixed_batch(oracle, n, dst, pool, seed=n)
    want = synth.bitmap_of(exp)
    p0, e0 = eng.path_stats()
    eng.set_auto_prepare(True)
    got_prep = eng.verify_batch(pks, msgs, sigs, dst)
    p1, e1 = eng.path_stats()
    eng.set_auto_prepare(False)
    got_exact = eng.verify_batch(pks, msgs, sigs, dst)
    p2, e2 = eng.path_stats()
    eng.set_auto_prepare(True)
    assert p1 == p0 + 1 and e1 == e0 and p2 == p1 and e2 == e1 + 1          # each call really took the path it names
    assert got_exact == want and got_prep == want
    assert oracle.verify_batch(pks, msgs, sigs, dst, nthreads=16) == want       # ~3 s of CPU at n = 20000


@pytest.mark.parametrize("pool,prepared", [(8448, True), (8449, False)])
def test_prepared_path_threshold(eng, oracle, M, pool, prepared):
    """the switch-over: a chunk of more than 16384 tuples takes the prepared path exactly when at most half of its public keys
    are distinct (smaller chunks always do: their Miller loops and final exponentiations run one wave / three lanes per tuple off
    the tables, k_miller_wide.hip / k_tri.hip)"""
    dst = M.DEFAULT_DST
    n = 16896
    skb = b"".join(synth.sk_of(k).to_bytes(32, "big") for k in range(pool))
    pkp = eng.sk_to_pk_batch(skb, pool)
    msgs = [synth.msg_of(90000 + i) for i in range(n)]
    idx = [i % pool for i in range(n)]
    pks = b"".join(pkp[128 * k:128 * k + 128] for k in idx)
    sigs = bytearray(eng.sign_batch(b"".join(skb[32 * k:32 * k + 32] for k in idx), msgs, dst))
    sigs[64 * 7 + 63] ^= 1                                       # one bad signature
    p0, e0 = eng.path_stats()
    bm = eng.verify_batch(pks, msgs, bytes(sigs), dst)
    assert eng.path_stats() == ((p0 + 1, e0) if prepared else (p0, e0 + 1))
    assert bm == synth.bitmap_of([i != 7 for i in range(n)])


def test_prepared_path_not_taken_for_distinct_keys(eng, oracle, M):
    """mostly distinct keys in a chunk beyond the three-lanes-per-tuple limit: the de-duplication finds more than n / 2 of them and
    the exact path runs (up to 16384 tuples the table path is taken whatever the keys: test_three_lanes_per_tuple_...)"""
    dst = M.DEFAULT_DST
    n = 16500
    sks = [synth.sk_of(k) for k in range(n)]
    pks = eng.sk_to_pk_batch(b"".join(s.to_bytes(32, "big") for s in sks), n)
    msgs = [synth.msg_of(i) for i in range(n)]
    sigs = eng.sign_batch(b"".join(s.to_bytes(32, "big") for s in sks), msgs, dst)
    p0, e0 = eng.path_stats()
    bm = eng.verify_batch(pks, msgs, sigs, dst)
    assert eng.path_stats() == (p0, e0 + 1) and bm == synth.bitmap_of([True] * n)


def test_explicit_g2prepared_api(eng, oracle, M):
    """blsbn254_g2_prepare_batch + blsbn254_verify_batch_prepared: keys prepared once, batches verified by key index"""
    dst = M.DEFAULT_DST
    pool = 5
    sks = [synth.sk_of(k) for k in range(pool)]
    keys = [oracle.sk_to_pk(s) for s in sks] + [synth.NON_SUBGROUP_PK, IDENT2, b"\xff" * 128]
    prep = eng.g2_prepare_batch(b"".join(keys), len(keys))
    assert prep.count() == 8 and prep.valid_bitmap() == bytes([0b00011111])
    rnd = random.Random(3)
    n = 333
    idx, msgs, sigs, exp = [], [], [], []
    for i in range(n):
        k = rnd.randrange(len(keys))
        m = synth.msg_of(i)
        signer = k if k < pool else 0
        s = oracle.sign(sks[signer], m, dst) if i < 40 or i % 7 == 0 else None
        ok = k < pool
        if s is None:                       # reuse an early tuple as it is (valid or not)
            j = rnd.randrange(min(i, 40))
            k, m, s, ok = idx[j], msgs[j], sigs[j], exp[j]
        if i % 11 == 3:
            m, ok = m + b"x", False
        idx.append(k); msgs.append(m); sigs.append(s); exp.append(ok)
    got = eng.verify_batch_prepared(prep, idx, msgs, b"".join(sigs), dst)
    assert got == synth.bitmap_of(exp)
    assert got == eng.verify_batch(b"".join(keys[k] for k in idx), msgs, b"".join(sigs), dst)
    assert eng.verify_batch_prepared(prep, [], [], b"", dst) == b""
    with pytest.raises(M.Bn254Error) as e:
        eng.verify_batch_prepared(prep, [0, 8], msgs[:2], b"".join(sigs[:2]), dst)          # key index out of range
    assert e.value.code == -1
    prep.close()


@pytest.mark.parametrize("n,pool", [(1023, 3), (1024, 3), (1501, 40)])
def test_aggregate_with_repeated_keys_prepared_path(eng, oracle, M, n, pool):
    """Aggregate verify over a batch that repeats few public keys: from 1024 pairs on the keys are de-duplicated and prepared
    once and the two-pairs-per-lane loop reads both keys' lines from tables (k_miller_hpk2p).  The shard's partial product
    equals the oracle's multi_miller_loop bit for bit on either path, with and without the signature's pair."""
    dst = M.DEFAULT_DST
    sks = [synth.sk_of(k) for k in range(pool)]
    pkp = [oracle.sk_to_pk(s) for s in sks]
    msgs = [synth.msg_of(7000 + i) for i in range(n)]
    pks = b"".join(pkp[i % pool] for i in range(n))
    h = eng.hash_to_g1_batch(msgs, dst)
    # aggregate signature = sum_k sk_k * (sum of the H(msg_i) signed by key k)
    agg = IDENT1
    for k in range(pool):
        acc = IDENT1
        for i in range(k, n, pool):
            acc = oracle.g1_add(acc, h[64 * i:64 * i + 64])
        agg = oracle.g1_add(agg, oracle.g1_mul(acc, sks[k]))
    want = oracle.multi_miller_loop(h, pks, n)
    # the explicit form: keys prepared once, pairs name their key by index (4 bytes per pair instead of 128)
    prep = eng.g2_prepare_batch(b"".join(pkp) + synth.NON_SUBGROUP_PK, pool + 1)
    kidx = [i % pool for i in range(n)]
    assert eng.aggregate_verify_prepared(prep, kidx, msgs, agg, dst) is True
    badm = list(msgs); badm[n // 3] = b"tampered"
    assert eng.aggregate_verify_prepared(prep, kidx, badm, agg, dst) is False
    assert eng.aggregate_verify_prepared(prep, kidx[:-1] + [pool], msgs, agg, dst) is False        # a key outside the subgroup
    assert eng.aggregate_verify_prepared(prep, kidx, msgs, IDENT1, dst) is False                    # identity signature
    assert eng.aggregate_verify_prepared(prep, [], [], agg, dst) is False
    with pytest.raises(M.Bn254Error) as e:
        eng.aggregate_verify_prepared(prep, kidx[:-1] + [pool + 1], msgs, agg, dst)                 # index of the table's own -G2gen entry / out of range
    assert e.value.code == -1
    prep.close()
    for auto in (True, False):
        eng.set_auto_prepare(auto)
        part, ok = eng.aggregate_partial(pks, msgs, dst)
        assert ok and part == want
        assert eng.aggregate_verify(pks, msgs, agg, dst) is True
        bad = list(msgs); bad[n // 2] = b"tampered"
        assert eng.aggregate_verify(pks, bad, agg, dst) is False
        assert eng.aggregate_verify(pks[:128 * (n - 1)] + synth.NON_SUBGROUP_PK, msgs, agg, dst) is False
    eng.set_auto_prepare(True)


@pytest.mark.parametrize("n,pool", [(1024, 1), (5000, 3), (40000, 2), (3000, 700), (66200, 33050)])      # the last: two pairs per lane
def test_aggregate_verify_sums_per_key(eng, oracle, pyref, M, n, pool):
    """aggregate_verify over repeated keys sums the H(msg_i) of every distinct key and runs ONE Miller loop per key
    (bilinearity in the first argument; one to three levels of chunk sums here, keys of multiplicity 1 included; one pair per
    lane while the launch is latency-bound, two per lane beyond 32768 pairs): same
    boolean as the pair-by-pair path on valid, tampered, wrong-signature and invalid-key batches."""
    dst = M.DEFAULT_DST
    sks = [synth.sk_of(k) for k in range(pool)]
    pkp = eng.sk_to_pk_batch(b"".join(s.to_bytes(32, "big") for s in sks), pool)
    pkp = [pkp[128 * k:128 * k + 128] for k in range(pool)]
    msgs = [synth.msg_of(90000 + i) for i in range(n)]
    # skewed multiplicities: key 0 takes every tuple whose index is not a multiple of 7, the others share the rest
    # (the many-key case spreads evenly: two tuples per key, 33051 pairs)
    kidx = [i % pool if pool > 1000 else 0 if (i % 7 or pool == 1) else 1 + (i // 7) % (pool - 1) for i in range(n)]
    pks = b"".join(pkp[k] for k in kidx)
    skb = b"".join(sks[k].to_bytes(32, "big") for k in kidx)
    agg = eng.aggregate_sigs(eng.sign_batch(skb, msgs, dst), n)
    g0, p0 = eng.aggregate_path_stats()
    assert eng.aggregate_verify(pks, msgs, agg, dst) is True
    bad = list(msgs); bad[n // 2] = b"tampered"
    assert eng.aggregate_verify(pks, bad, agg, dst) is False
    assert eng.aggregate_verify(pks, msgs, oracle.g1_add(agg, oracle.g1_generator()), dst) is False
    assert eng.aggregate_verify(pks, msgs, IDENT1, dst) is False
    assert eng.aggregate_verify(pks[:128 * (n - 1)] + synth.NON_SUBGROUP_PK, msgs, agg, dst) is False
    g1, p1 = eng.aggregate_path_stats()
    assert (g1 - g0, p1 - p0) == (5, 0)
    eng.set_auto_prepare(False)                           # the pair-by-pair path agrees
    try:
        assert eng.aggregate_verify(pks, msgs, agg, dst) is True
        assert eng.aggregate_verify(pks, bad, agg, dst) is False
    finally:
        eng.set_auto_prepare(True)
    assert eng.aggregate_path_stats() == (g1, p1 + 2)


def test_multi_miller_loop_over_prepared_keys(eng, oracle, pyref, M):
    """multi_miller_loop(&[(&G1Affine, &G2Prepared)]) with the G2 side prepared once (G2Prepared::from, pairings.rs:609-660,
    which panics in the reference: E6): same 384 bytes as the plain multi_miller_loop of the oracle; identity G1 terms are
    skipped; odd and even term counts."""
    rnd = random.Random(77)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    keys = [oracle.g2_mul(G2, rnd.randrange(1, pyref.R)) for _ in range(6)]
    prep = eng.g2_prepare_batch(b"".join(keys), len(keys))
    for n in (1, 2, 5, 64, 333):
        idx = [rnd.randrange(len(keys)) for _ in range(n)]
        pts = [oracle.g1_mul(G1, rnd.randrange(1, pyref.R)) for _ in range(min(n, 12))]
        g1 = [pts[rnd.randrange(len(pts))] for _ in range(n)]
        if n >= 5:
            g1[3] = IDENT1                                            # skipped term
        got = eng.multi_miller_loop_prepared(prep, idx, b"".join(g1), n)
        assert got == oracle.multi_miller_loop(b"".join(g1), b"".join(keys[k] for k in idx), n)
        assert got == eng.multi_miller_loop(b"".join(g1), b"".join(keys[k] for k in idx), n)
    assert eng.multi_miller_loop_prepared(prep, [], b"", 0) == ONE_GT
    bad = eng.g2_prepare_batch(keys[0] + synth.NON_SUBGROUP_PK, 2)
    with pytest.raises(M.InvalidG2Bytes):
        eng.multi_miller_loop_prepared(bad, [0, 1], g1[0] + g1[0], 2)
    with pytest.raises(M.InvalidG1Bytes):
        eng.multi_miller_loop_prepared(prep, [0], b"\xff" * 64, 1)
    prep.close(); bad.close()


# ---- group operations on caller-supplied points and FastAggregateVerify -------------------------
def _neg_g2(pk):
    """(x, -y) of an uncompressed G2 point"""
    yc1 = int.from_bytes(pk[64:96], "big"); yc0 = int.from_bytes(pk[96:128], "big")
    return pk[:64] + ((synth.P - yc1) % synth.P).to_bytes(32, "big") + ((synth.P - yc0) % synth.P).to_bytes(32, "big")


def test_g1_g2_mul_batch_vs_oracle(eng, oracle, pyref, M):
    """Mul<Scalar> (g1.rs:518-534, g2.rs:866-886) element-wise == the oracle's double-and-add: random points and scalars, the
    scalars 0, 1, r - 1, the identity as input, a point of E'(Fp2) outside the r-torsion; errors for off-curve points and
    scalars >= r."""
    rnd = random.Random(4100)
    R = pyref.R
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    n = 70
    ks = [0, 1, R - 1, 2, (1 << 252) + 5] + [rnd.randrange(R) for _ in range(n - 5)]
    p1 = [oracle.g1_mul(G1, rnd.randrange(1, R)) for _ in range(n)]
    p2 = [oracle.g2_mul(G2, rnd.randrange(1, R)) for _ in range(n)]
    p1[7] = IDENT1; p2[7] = IDENT2
    p2[9] = synth.NON_SUBGROUP_PK                       # Mul works on any curve point
    kb = b"".join(k.to_bytes(32, "big") for k in ks)
    got1 = eng.g1_mul_batch(b"".join(p1), kb, n)
    got2 = eng.g2_mul_batch(b"".join(p2), kb, n)
    for i in range(n):
        assert got1[64 * i:64 * i + 64] == oracle.g1_mul(p1[i], ks[i]), i
        assert got2[128 * i:128 * i + 128] == oracle.g2_mul(p2[i], ks[i]), i
    assert got1[:64] == IDENT1 and got2[:128] == IDENT2                       # k = 0
    assert got1[64 * 7:64 * 8] == IDENT1 and got2[128 * 7:128 * 8] == IDENT2  # identity in
    assert eng.g1_mul_batch(b"", b"", 0) == b"" and eng.g2_mul_batch(b"", b"", 0) == b""
    off1 = G1[:32] + (3).to_bytes(32, "big")
    with pytest.raises(M.InvalidG1Bytes):
        eng.g1_mul_batch(p1[0] + off1, kb[:64], 2)
    off2 = G2[:96] + (int.from_bytes(G2[96:], "big") ^ 1).to_bytes(32, "big")
    with pytest.raises(M.InvalidG2Bytes):
        eng.g2_mul_batch(off2, kb[32:64], 1)
    with pytest.raises(M.InvalidScalarBytes):
        eng.g1_mul_batch(p1[0], R.to_bytes(32, "big"), 1)
    with pytest.raises(M.InvalidScalarBytes):
        eng.g2_mul_batch(p2[0], b"\xff" * 32, 1)
    # sk_to_pk is the same operator on the generator
    assert eng.g2_mul_batch(G2 * 4, kb[32 * 5:32 * 9], 4) == eng.sk_to_pk_batch(kb[32 * 5:32 * 9], 4)


def test_aggregate_pks_vs_oracle(eng, oracle, pyref, M):
    """impl Sum for G2Projective (g2.rs:579-583) over n = 0 ... 5000 points (every chunk-boundary case of the 16-point levels):
    == the oracle's running sum, == [sum sk] G2gen; identity and repeated terms, a term outside the subgroup; off-curve -> error."""
    R = pyref.R
    n_max = 5000
    sks = [synth.sk_of(3000 + k) for k in range(n_max)]
    pk_all = eng.sk_to_pk_batch(b"".join(s.to_bytes(32, "big") for s in sks), n_max)
    assert eng.aggregate_pks(b"", 0) == IDENT2
    for n in (1, 2, 15, 16, 17, 255, 256, 257, 4097, n_max):
        got = eng.aggregate_pks(pk_all[:128 * n], n)
        assert got == oracle.sk_to_pk(sum(sks[:n]) % R), n
        if n <= 257:
            assert got == oracle.aggregate_pks(pk_all[:128 * n], n)
    pk0, pk1 = pk_all[:128], pk_all[128:256]
    mix = pk0 + IDENT2 + pk1 + pk0 + synth.NON_SUBGROUP_PK + _neg_g2(pk1)
    assert eng.aggregate_pks(mix, 6) == oracle.aggregate_pks(mix, 6)
    assert eng.aggregate_pks(pk0 + _neg_g2(pk0), 2) == IDENT2
    bad = pk0 + pk1[:96] + (int.from_bytes(pk1[96:], "big") ^ 1).to_bytes(32, "big")
    with pytest.raises(M.InvalidG2Bytes):
        eng.aggregate_pks(bad, 2)
    with pytest.raises(M.InvalidG2Bytes):
        eng.aggregate_pks(pk0 * 20 + b"\xff" * 128 + pk1 * 20, 41)


def test_fast_aggregate_verify_vs_oracle(eng, oracle, pyref, M):
    """One message signed by many keys, group by group == the oracle's FastAggregateVerify: group sizes around the chunk
    boundaries, tampered message, a foreign key, an off-curve key, a key outside the subgroup, an identity key among valid
    ones, keys that cancel to the identity, a group without keys, a repeated key (signed twice)."""
    dst = M.DEFAULT_DST
    R = pyref.R
    sizes = [1, 2, 16, 17, 33, 300]
    groups = []                                       # (key bytes, message, signature)
    base = 0
    for gi, sz in enumerate(sizes):
        sks = [synth.sk_of(5000 + base + k) for k in range(sz)]
        base += sz
        skb = b"".join(s.to_bytes(32, "big") for s in sks)
        pkb = eng.sk_to_pk_batch(skb, sz)
        msg = synth.msg_of(9000 + gi) + bytes(gi)
        sig = eng.aggregate_sigs(eng.sign_batch(skb, [msg] * sz, dst), sz)
        assert sig == oracle.sign(sum(sks) % R, msg, dst)
        groups.append((pkb, msg, sig))
    pk17, m17, s17 = groups[3]
    pk2, m2, s2 = groups[1]
    stranger = oracle.sk_to_pk(synth.sk_of(1))
    cases = list(groups)
    cases.append((pk17, b"tampered", s17))
    cases.append((pk17[:128 * 16] + stranger, m17, s17))                                   # a foreign key in place of the last
    cases.append((pk17[:128 * 5] + pk17[128 * 5:128 * 6 - 1] + bytes([pk17[128 * 6 - 1] ^ 1]) + pk17[128 * 6:], m17, s17))   # off-curve key
    cases.append((pk17 + synth.NON_SUBGROUP_PK, m17, s17))                                 # sum leaves the r-torsion
    cases.append((pk17[:128 * 9] + IDENT2 + pk17[128 * 9:], m17, s17))                    # an identity term does not change the sum
    cases.append((pk2[:128] + _neg_g2(pk2[:128]), m2, s2))                                 # keys cancel: the sum is the identity
    cases.append((b"", m2, s2))                                                            # no key at all
    sk_a = synth.sk_of(77)
    pk_a = oracle.sk_to_pk(sk_a)
    cases.append((pk_a + pk_a, b"twice", oracle.sign(2 * sk_a % R, b"twice", dst)))         # a repeated key counts twice
    cases.append((pk17, m17, IDENT1))                                                      # identity signature
    want = [oracle.fast_aggregate_verify(k, len(k) // 128, m, s, dst) for k, m, s in cases]
    assert want == [True] * 6 + [False, False, False, False, True, False, False, True, False]
    got = eng.fast_aggregate_verify_batch([k for k, _, _ in cases], [m for _, m, _ in cases], b"".join(s for _, _, s in cases), dst)
    assert got == synth.bitmap_of(want)
    for (k, m, s), w in zip(cases, want):
        assert eng.fast_aggregate_verify(k, len(k) // 128, m, s, dst) is w
    assert eng.fast_aggregate_verify_batch([], [], b"", dst) == b""


def test_fast_aggregate_verify_batch_validator_shape(eng, oracle, pyref, M):
    """4096 aggregates of 64 keys each (262 144 keys): GPU-signed, every 7th group's message tampered; bitmap == closed form,
    spot groups == the oracle."""
    dst = M.DEFAULT_DST
    R = pyref.R
    g, per = 4096, 64
    pool = 1024
    sks = [synth.sk_of(k) for k in range(pool)]
    skb = b"".join(s.to_bytes(32, "big") for s in sks)
    pk_pool = eng.sk_to_pk_batch(skb, pool)
    key_sets, msgs, sig_list, exp = [], [], [], []
    agg_sk = []
    for i in range(g):
        lo = (i * 37) % (pool - per)
        key_sets.append(pk_pool[128 * lo:128 * (lo + per)])
        agg_sk.append(sum(sks[lo:lo + per]) % R)
        msgs.append(synth.msg_of(20000 + i))
    sigs = bytearray(eng.sign_batch(b"".join(s.to_bytes(32, "big") for s in agg_sk), msgs, dst))       # [sum sk] H(msg) = sum of the members' signatures
    for i in range(g):
        ok = i % 7 != 6
        if not ok:
            msgs[i] = bytes([msgs[i][0] ^ 1]) + msgs[i][1:]
        exp.append(ok)
    got = eng.fast_aggregate_verify_batch(key_sets, msgs, bytes(sigs), dst)
    assert got == synth.bitmap_of(exp)
    for i in (0, 6, 4095):
        assert oracle.fast_aggregate_verify(key_sets[i], per, msgs[i], bytes(sigs[64 * i:64 * i + 64]), dst) is exp[i]
    # 6000 groups of 8 keys: the per-group sums go through the mid-size (three lanes per tuple) verify pipeline
    g2n, per2 = 6000, 8
    ks2, ms2, ag2, ex2 = [], [], [], []
    for i in range(g2n):
        lo = (i * 13) % (pool - per2)
        ks2.append(pk_pool[128 * lo:128 * (lo + per2)])
        ag2.append(sum(sks[lo:lo + per2]) % R)
        ms2.append(synth.msg_of(40000 + i))
    sg2 = eng.sign_batch(b"".join(s.to_bytes(32, "big") for s in ag2), ms2, dst)
    for i in range(g2n):
        ok = i % 11 != 10
        if not ok:
            ms2[i] = bytes([ms2[i][0] ^ 1]) + ms2[i][1:]
        ex2.append(ok)
    assert eng.fast_aggregate_verify_batch(ks2, ms2, sg2, dst) == synth.bitmap_of(ex2)


def test_three_lanes_per_tuple_kernels_equal_lane_per_tuple(oracle, pyref, M, monkeypatch):
    """k_tri.hip (three lanes per tuple: launches of 4097 ... 16384 tuples) == the lane-per-tuple kernels (BLSBN254_TRI_MAX=0) == the
    expectation: verify_batch on the prepared-key path (Miller loop + hard part on quads), the RLC path (its chunk and fallback
    rounds), and pairing_batch (the final exponentiation alone on quads, Gt bytes) at sizes around both switch-overs."""
    if os.environ.get("BLSBN254_TRI_MAX", "16384") == "0":
        pytest.skip("the three-lane kernels are switched off by the environment (BLSBN254_TRI_MAX=0): nothing to compare")
    dst = M.DEFAULT_DST
    e_tri = M.Engine(0)
    monkeypatch.setenv("BLSBN254_TRI_MAX", "0")
    e_ref = M.Engine(0)
    monkeypatch.delenv("BLSBN254_TRI_MAX")
    try:
        n_max = 16385
        pks, msgs, sigs, exp = synth.make_batch_gpu(e_ref, oracle, n_max, dst, pool=37, invalid_every=5, spot=20)
        for n in (4097, 6000, 16384, 16385):
            want = synth.bitmap_of(exp[:n])
            a, b, c = pks[:128 * n], msgs[:n], sigs[:64 * n]
            assert e_tri.verify_batch(a, b, c, dst) == want, n
            assert e_ref.verify_batch(a, b, c, dst) == want, n
        n = 9000
        assert e_tri.verify_batch_rlc(pks[:128 * n], msgs[:n], sigs[:64 * n], dst) == synth.bitmap_of(exp[:n])
        rnd = random.Random(4242)
        G1, G2 = oracle.g1_generator(), oracle.g2_generator()
        m = 4099
        ks = b"".join(rnd.randrange(1, pyref.R).to_bytes(32, "big") for _ in range(m))
        g1 = e_ref.g1_mul_batch(G1 * m, ks, m)
        ks2 = b"".join(rnd.randrange(1, pyref.R).to_bytes(32, "big") for _ in range(m))
        g2 = e_ref.g2_mul_batch(G2 * m, ks2, m)
        gt = e_tri.pairing_batch(g1, g2, m)
        assert gt == e_ref.pairing_batch(g1, g2, m)
        for i in (0, 1, m - 1):
            assert gt[384 * i:384 * i + 384] == oracle.pairing_batch(g1[64 * i:64 * i + 64], g2[128 * i:128 * i + 128], 1)
        ml = e_ref.miller_loop_batch(g1, g2, m)
        assert e_tri.final_exponentiation(ml, m) == gt
        # the variable-Q Miller loop on quads (k_miller_tri_1): per-pair Miller values, byte for byte, incl. an identity and an
        # undecodable operand in the middle of the batch
        ml_tri = e_tri.miller_loop_batch(g1, g2, m)
        assert ml_tri == ml
        assert ml_tri[:384 * 2] == oracle.miller_loop_batch(g1[:128], g2[:256], 2)
        g1b = g1[:64 * 5] + IDENT1 + g1[64 * 6:]
        assert e_tri.pairing_batch(g1b, g2, m) == e_ref.pairing_batch(g1b, g2, m)
        assert e_tri.pairing_batch(g1b, g2, m)[384 * 5:384 * 6] == ONE_GT
        g2b = g2[:128 * 7] + b"\xff" * 128 + g2[128 * 8:]
        with pytest.raises(M.InvalidG2Bytes):
            e_tri.pairing_batch(g1, g2b, m)
    finally:
        e_tri.close(); e_ref.close()


def test_aggregate_verify_mid_size_distinct_keys_on_quads(oracle, pyref, M, monkeypatch):
    """aggregate_verify over a few thousand DISTINCT keys: every key is prepared (four lanes per key) and every pair runs on a quad
    of lanes off its key's table (k_miller_tri_1p), one product tree, one final exponentiation == the pairwise lane path
    (BLSBN254_TRI_MAX=0) == the expectation; tampered message, a key outside the subgroup."""
    if os.environ.get("BLSBN254_TRI_MAX", "16384") == "0" or os.environ.get("BLSBN254_TRI_MILLER", "1") == "0" or os.environ.get("BLSBN254_QUAD_PREP", "1") == "0":
        pytest.skip("the quad kernels are switched off by the environment (production knobs): nothing to compare")
    dst = M.DEFAULT_DST
    e_tri = M.Engine(0)
    monkeypatch.setenv("BLSBN254_TRI_MAX", "0")
    e_ref = M.Engine(0)
    monkeypatch.delenv("BLSBN254_TRI_MAX")
    try:
        n = 5000
        sks = [synth.sk_of(40000 + k) for k in range(n)]
        skb = b"".join(x.to_bytes(32, "big") for x in sks)
        pk = e_ref.sk_to_pk_batch(skb, n)
        ms = [synth.msg_of(50000 + i) for i in range(n)]
        agg = e_ref.aggregate_sigs(e_ref.sign_batch(skb, ms, dst), n)
        g0, p0 = e_tri.aggregate_path_stats()
        assert e_tri.aggregate_verify(pk, ms, agg, dst) is True
        assert e_tri.aggregate_path_stats() == (g0 + 1, p0)                  # the table path, although no key repeats
        assert e_ref.aggregate_verify(pk, ms, agg, dst) is True
        bad = list(ms); bad[n - 3] = b"tampered"
        assert e_tri.aggregate_verify(pk, bad, agg, dst) is False and e_ref.aggregate_verify(pk, bad, agg, dst) is False
        pk_bad = pk[:128 * 77] + synth.NON_SUBGROUP_PK + pk[128 * 78:]
        assert e_tri.aggregate_verify(pk_bad, ms, agg, dst) is False
        assert e_tri.aggregate_verify(pk, ms, IDENT1, dst) is False
    finally:
        e_tri.close(); e_ref.close()


def _dev_batch(M, torch, pks, msgs, sigs):
    data, off = M.engine.pack_messages(msgs)
    dev = torch.device("cuda", 0)
    t = [torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev), torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev),
         torch.from_numpy(off.astype(np.int64)).to(dev), torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev),
         torch.full(((len(msgs) + 7) // 8,), 0x5a, dtype=torch.uint8, device=dev)]
    torch.cuda.synchronize()
    return t


def test_verify_batch_dev_asynchronous_path(oracle, M):
    """blsbn254_verify_batch_dev in steady state enqueues on the previous call's key count and returns; the check comes back at
    blsbn254_ctx_synchronize.  Same bitmaps as the counting path: repeated calls, two calls in flight, a new domain-separation tag
    between calls, and the cases where the assumption FAILS and the call is re-run -- more keys than reserved, a key set that no
    longer repeats (exact path), then back to few keys."""
    if os.environ.get("BLSBN254_ASYNC_VERIFY", "1") == "0":
        pytest.skip("the optimistic path is switched off by the environment (BLSBN254_ASYNC_VERIFY=0)")
    import torch
    dst = M.DEFAULT_DST
    e = M.Engine(0)
    try:
        def run(t, n, tag=dst):
            e.verify_batch_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), n, t[4].data_ptr(), tag)
        nA = 20000
        pA, mA, sA, xA = synth.make_batch_gpu(e, oracle, nA, dst, pool=40, invalid_every=9, spot=10)
        tA = _dev_batch(M, torch, pA, mA, sA); wantA = synth.bitmap_of(xA)
        run(tA, nA); e.synchronize()
        assert bytes(tA[4].cpu().numpy()) == wantA and e.async_stats() == (0, 0)          # the first call counts
        tA[4].fill_(0x5a); run(tA, nA); e.synchronize()
        assert bytes(tA[4].cpu().numpy()) == wantA and e.async_stats() == (1, 0)          # the second does not
        # two calls in flight (different output buffers), then one synchronisation
        tA2 = _dev_batch(M, torch, pA[:128 * 9000], mA[:9000], sA[:64 * 9000])
        tA[4].fill_(0x5a)
        run(tA, nA); run(tA2, 9000); e.synchronize()
        assert bytes(tA[4].cpu().numpy()) == wantA and bytes(tA2[4].cpu().numpy()) == synth.bitmap_of(xA[:9000])
        assert e.async_stats() == (3, 0)
        # a different tag between two calls in flight: every signature is wrong under it
        tA[4].fill_(0x5a); tA2[4].fill_(0x5a)
        run(tA, nA); run(tA2, 9000, b"ANOTHER_TAG"); e.synchronize()
        assert bytes(tA[4].cpu().numpy()) == wantA and not any(bytes(tA2[4].cpu().numpy()))
        a0, r0 = e.async_stats()
        # more keys than the tables were reserved for (40 keys -> capacity 1024; now 5000 keys): re-run, then the larger capacity holds
        nC = 40000
        pC, mC, sC, xC = synth.make_batch_gpu(e, oracle, nC, dst, pool=5000, invalid_every=11, spot=10)
        tC = _dev_batch(M, torch, pC, mC, sC); wantC = synth.bitmap_of(xC)
        run(tC, nC); e.synchronize()
        assert bytes(tC[4].cpu().numpy()) == wantC and e.async_stats() == (a0 + 1, r0 + 1)
        tC[4].fill_(0x5a); run(tC, nC); e.synchronize()                                   # (the re-run counted the keys: the next call is asynchronous again)
        assert bytes(tC[4].cpu().numpy()) == wantC and e.async_stats() == (a0 + 2, r0 + 1)
        # a key set that does not repeat any more (beyond the mid-size limit: the exact path): the assumption fails, the re-run takes the exact path
        nD = 17000
        pD, mD, sD, xD = synth.make_batch_gpu(e, oracle, nD, dst, pool=nD, invalid_every=13, spot=10)
        tD = _dev_batch(M, torch, pD, mD, sD); wantD = synth.bitmap_of(xD)
        p0, x0 = e.path_stats()
        run(tD, nD); e.synchronize()
        assert bytes(tD[4].cpu().numpy()) == wantD and e.async_stats() == (a0 + 3, r0 + 2)
        assert e.path_stats() == (p0, x0 + 1)
        tD[4].fill_(0x5a); run(tD, nD); e.synchronize()                                   # after an exact chunk every call counts first
        assert bytes(tD[4].cpu().numpy()) == wantD and e.async_stats() == (a0 + 3, r0 + 2)
        # back to few keys; another entry point between a pending call and its synchronisation settles it
        tA[4].fill_(0x5a); run(tA, nA); e.synchronize()
        tA[4].fill_(0x5a); run(tA, nA)
        assert e.g1_check_batch(oracle.g1_generator(), 1) == b"\x01"
        assert bytes(tA[4].cpu().numpy()) == wantA
        # alternating between a small and a large key set: the capacity remembers the largest set seen, no call is re-run
        a2, r2 = e.async_stats()
        for _ in range(2):
            tC[4].fill_(0x5a); run(tC, nC); e.synchronize()
            assert bytes(tC[4].cpu().numpy()) == wantC
            tA[4].fill_(0x5a); run(tA, nA); e.synchronize()
            assert bytes(tA[4].cpu().numpy()) == wantA
        assert e.async_stats() == (a2 + 4, r2)
        # switched off: every call counts
        e.set_async_verify(False)
        a1, r1 = e.async_stats()
        tA[4].fill_(0x5a); run(tA, nA); e.synchronize()
        assert bytes(tA[4].cpu().numpy()) == wantA and e.async_stats() == (a1, r1)
    finally:
        e.close()
