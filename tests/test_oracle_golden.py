"""CPU-only: pins both oracles (C and pure Python) to the reference's own golden vectors.

Vectors: tests/golden/reference_kats.json (transcribed from the reference's test modules by
tests/golden/make_fixtures.py): g1.rs:981-1141, g2.rs:994-1017,1039-1313, pairings.rs:387-479,971-980.
"""
import hashlib
import os
import random

import pytest

H = lambda s: int(s, 16)
ONE_GT = (1).to_bytes(32, "big") + bytes(352)


def test_sha256_matches_hashlib(oracle):
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 128, 1000):
        m = os.urandom(n)
        assert oracle.sha256(m) == hashlib.sha256(m).digest()


def test_g1_encode_kats(oracle, pyref, kats):
    g1 = kats["g1"]
    dst = g1["encode_dst"].encode()
    for v in g1["encode"]:
        msg = v["msg"].encode()
        assert oracle.hash_to_field_fp(msg, dst, 1) == [H(v["u"])]
        assert pyref.hash_to_fp(msg, dst, 1) == [H(v["u"])]
        assert oracle.encode_to_g1_batch([msg], dst).hex() == v["p_x"] + v["p_y"]
        assert pyref.encode_to_g1(msg, dst) == (H(v["p_x"]), H(v["p_y"]))


def test_g1_hash_kats(oracle, pyref, kats):
    g1 = kats["g1"]
    dst = g1["hash_dst"].encode()
    msgs = [v["msg"].encode() for v in g1["hash"]]
    out = oracle.hash_to_g1_batch(msgs, dst)
    for i, v in enumerate(g1["hash"]):
        assert oracle.hash_to_field_fp(msgs[i], dst, 2) == [H(v["u0"]), H(v["u1"])]
        assert out[64 * i:64 * i + 64].hex() == v["p_x"] + v["p_y"]
        u0, u1 = pyref.hash_to_fp(msgs[i], dst, 2)
        assert pyref.svdw_g1(u0) == (H(v["q0_x"]), H(v["q0_y"]))
        assert pyref.svdw_g1(u1) == (H(v["q1_x"]), H(v["q1_y"]))
        assert pyref.hash_to_g1(msgs[i], dst) == (H(v["p_x"]), H(v["p_y"]))
    # the KAT outputs are on the curve (g1.rs:1137-1139)
    assert oracle.g1_check_batch(out, 5) == bytes([0x1f])


def test_g2_encode_and_hash_kats(oracle, pyref, kats):
    g2 = kats["g2"]
    for name, fn, pfn in (("encode", oracle.encode_to_g2_batch, pyref.encode_to_g2),
                          ("hash", oracle.hash_to_g2_batch, pyref.hash_to_g2)):
        dst = g2[name + "_dst"].encode()
        msgs = [v["msg"].encode() for v in g2[name]]
        out = fn(msgs, dst)
        for i, v in enumerate(g2[name]):
            assert out[128 * i:128 * i + 128].hex() == v["x_c1"] + v["x_c0"] + v["y_c1"] + v["y_c0"]
            assert pfn(msgs[i], dst) == ((H(v["x_c0"]), H(v["x_c1"])), (H(v["y_c0"]), H(v["y_c1"])))
        assert oracle.g2_check_batch(out, 5) == bytes([0x1f])         # in the r-torsion
        assert oracle.g2_check_batch_slow(out, 5) == bytes([0x1f])


def test_constants(oracle, pyref, kats):
    c = kats["constants"]
    g2 = oracle.g2_generator()
    assert g2.hex() == c["fp2_gen_x"]["c1"] + c["fp2_gen_x"]["c0"] + c["fp2_gen_y"]["c1"] + c["fp2_gen_y"]["c0"]
    assert pyref.g2_to_bytes(pyref.G2_GEN) == g2
    assert pyref.B2 == (H(c["fp2_b"]["c0"]), H(c["fp2_b"]["c1"]))
    assert pyref.PSI_X == (H(c["psi_endo_u"]["c0"]), H(c["psi_endo_u"]["c1"]))
    assert pyref.PSI_Y == (H(c["psi_endo_v"]["c0"]), H(c["psi_endo_v"]["c1"]))
    assert oracle.g1_generator() == (1).to_bytes(32, "big") + (2).to_bytes(32, "big")


def test_g2_bad_point_and_identity(oracle, kats):
    # g2.rs:994-1027: bad point is neither on the curve nor torsion free; generator and identity pass
    bp = kats["g2"]["bad_point"]
    bad = bytes.fromhex(bp["x_c1"] + bp["x_c0"] + bp["y_c1"] + bp["y_c0"])
    ident = bytes(64) + bytes(32) + bytes(31) + b"\x01"          # x = 0, y = (c1 = 0, c0 = 1)
    pts = bad + oracle.g2_generator() + ident
    assert oracle.g2_check_batch(pts, 3) == bytes([0b110])
    assert oracle.g2_check_batch_slow(pts, 3) == bytes([0b110])
    g1_ident = bytes(32) + (1).to_bytes(32, "big")
    assert oracle.g1_check_batch(oracle.g1_generator() + g1_ident, 2) == bytes([0b11])   # g1.rs:968-978


def test_five_g(oracle):
    # g1.rs:1144-1149, g2.rs:1031-1036: 4G + G == 5*G
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    d = oracle.g1_add(G1, G1); q = oracle.g1_add(d, d)
    assert oracle.g1_add(q, G1) == oracle.g1_mul(G1, 5)
    d = oracle.g2_add(G2, G2); q = oracle.g2_add(d, d)
    assert oracle.g2_add(q, G2) == oracle.g2_mul(G2, 5)


def test_pairing_golden(oracle, pyref, kats):
    # pairings.rs:971-980: pairing(g1, g2) == Gt::generator(); gt * r == identity
    gold = kats["constants"]["gt_generator_bytes_hex"]
    gt = oracle.pairing_batch(oracle.g1_generator(), oracle.g2_generator(), 1)
    assert gt.hex() == gold
    assert oracle.gt_pow(gt, pyref.R) == ONE_GT
    pg = pyref.pairing(pyref.G1_GEN, pyref.G2_GEN)
    assert pyref.f12_to_bytes(pg).hex() == gold
    assert pyref.f12_pow(pg, pyref.R) == pyref.F12_ONE
    assert pyref.final_exponentiation_slow(pyref.miller_loop(pyref.G1_GEN, pyref.G2_GEN)) == pg


def test_pairing_properties(oracle, pyref):
    rnd = random.Random(7)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    gt = oracle.pairing_batch(G1, G2, 1)
    a, b = rnd.randrange(1, pyref.R), rnd.randrange(1, pyref.R)
    aP, bQ = oracle.g1_mul(G1, a), oracle.g2_mul(G2, b)
    assert oracle.pairing_batch(aP, bQ, 1) == oracle.gt_pow(gt, a * b % pyref.R)      # bilinearity
    # identity handling (pairings.rs:789-800)
    ident1 = bytes(32) + (1).to_bytes(32, "big")
    assert oracle.pairing_batch(ident1, G2, 1) == ONE_GT
    # multi-pairing == product of pairings; ML then FE == pairing
    ml = oracle.multi_miller_loop(aP + G1, G2 + bQ, 2)
    prod = oracle.gt_mul(oracle.pairing_batch(aP, G2, 1), oracle.pairing_batch(G1, bQ, 1))
    assert oracle.final_exponentiation(ml, 1) == prod
    # e(aP, Q) * e(-P, aQ) == 1
    negG1 = G1[:32] + (pyref.P - 2).to_bytes(32, "big")
    ml = oracle.multi_miller_loop(aP + negG1, G2 + oracle.g2_mul(G2, a), 2)
    assert oracle.final_exponentiation(ml, 1) == ONE_GT
    # the pure-Python Miller loop differs only by a subfield factor
    pm = pyref.f12_to_bytes(pyref.miller_loop(pyref.G1_GEN, pyref.G2_GEN))
    assert oracle.final_exponentiation(pm, 1) == gt


def test_subgroup_test_equivalence(oracle, pyref):
    """psi-based membership == [r]P on E'(Fp2), including points of small order and mixtures."""
    rnd = random.Random(11)
    P = pyref.P
    h2 = 2 * P - pyref.R

    def rand_twist():
        while True:
            x = (rnd.randrange(P), rnd.randrange(P))
            y = pyref.f2_sqrt(pyref.f2_add(pyref.f2_mul(pyref.f2_sqr(x), x), pyref.B2))
            if y is not None:
                return (x, y)
    pts, expect = [], []
    for _ in range(3):
        t = rand_twist()
        pts.append(t); expect.append(False)
        pts.append(pyref.g2_mul(t, h2)); expect.append(True)
    for q in (10069, 5864401):
        s = None
        while s is None:
            s = pyref.g2_mul(rand_twist(), pyref.R * h2 // q)
        pts.append(s); expect.append(False)
        pts.append(pyref.g2_add(s, pyref.G2_GEN)); expect.append(False)
    buf = b"".join(pyref.g2_to_bytes(p) for p in pts)
    n = len(pts)
    want = sum(1 << i for i, e in enumerate(expect) if e).to_bytes((n + 7) // 8, "little")
    assert oracle.g2_check_batch(buf, n) == want
    assert oracle.g2_check_batch_slow(buf, n) == want
    for p, e in zip(pts, expect):
        assert pyref.g2_in_subgroup_fast(p) == e


def test_decode_strictness(oracle, pyref):
    P = pyref.P
    G1 = oracle.g1_generator()
    bad_x = P.to_bytes(32, "big") + G1[32:]
    bad_y = G1[:32] + (P + 1).to_bytes(32, "big")
    off_curve = G1[:32] + (3).to_bytes(32, "big")
    assert oracle.g1_check_batch(bad_x + bad_y + off_curve + G1, 4) == bytes([0b1000])
    with pytest.raises(oracle.OracleError) as e:
        oracle.pairing_batch(bad_x, oracle.g2_generator(), 1)
    assert e.value.rc == 2                                        # InvalidG1Bytes (error.rs:4-10)
    with pytest.raises(oracle.OracleError) as e:
        oracle.pairing_batch(G1, P.to_bytes(32, "big") + bytes(96), 1)
    assert e.value.rc == 3                                        # InvalidG2Bytes
    with pytest.raises(oracle.OracleError) as e:
        oracle.final_exponentiation(P.to_bytes(32, "big") + bytes(352), 1)
    assert e.value.rc == 4                                        # InvalidGtBytes


def test_bls_verify_and_negative_cases(oracle, pyref):
    dst = pyref.DEFAULT_DST
    rnd = random.Random(3)
    sks = [rnd.randrange(1, pyref.R) for _ in range(3)]
    pks = [oracle.sk_to_pk(s) for s in sks]
    msgs = [b"", b"abc", bytes(range(200))]
    sigs = [oracle.sign(s, m, dst) for s, m in zip(sks, msgs)]
    assert pks[0] == pyref.g2_to_bytes(pyref.sk_to_pk(sks[0]))
    assert sigs[1] == pyref.g1_to_bytes(pyref.sign(sks[1], msgs[1], dst))
    assert pyref.verify_bytes(pks[1], msgs[1], sigs[1], dst)
    assert oracle.verify_batch(b"".join(pks), msgs, b"".join(sigs), dst) == bytes([0b111])
    assert oracle.verify_batch(b"".join(pks), msgs, b"".join(sigs), dst, nthreads=3) == bytes([0b111])
    G1 = oracle.g1_generator()
    ident1 = bytes(32) + (1).to_bytes(32, "big")
    ident2 = bytes(64) + bytes(32) + bytes(31) + b"\x01"
    cases = [
        (pks[0], b"x", sigs[0]),                                   # wrong message
        (pks[1], msgs[0], sigs[0]),                                # wrong key
        (pks[0], msgs[0], oracle.g1_add(sigs[0], G1)),             # wrong signature
        (pks[0], msgs[0], sigs[0][:32] + (5).to_bytes(32, "big")),  # sig off curve
        (pks[0], msgs[0], ident1),                                 # identity signature
        (ident2, msgs[0], sigs[0]),                                # identity public key
        (pks[0], msgs[0], sigs[0]),                                # control: valid
    ]
    bm = oracle.verify_batch(b"".join(c[0] for c in cases), [c[1] for c in cases], b"".join(c[2] for c in cases), dst)
    assert bm == bytes([0b1000000])
    assert not pyref.verify_bytes(*cases[0], dst) and not pyref.verify_bytes(*cases[3], dst)


def test_aggregate_and_threshold(oracle, pyref):
    dst = pyref.DEFAULT_DST
    rnd = random.Random(5)
    n = 4
    sks = [rnd.randrange(1, pyref.R) for _ in range(n)]
    pks = b"".join(oracle.sk_to_pk(s) for s in sks)
    msgs = [b"msg-%d" % i for i in range(n)]
    sigs = b"".join(oracle.sign(s, m, dst) for s, m in zip(sks, msgs))
    agg = oracle.aggregate_sigs(sigs, n)
    assert oracle.aggregate_verify(pks, msgs, agg, dst)
    assert pyref.aggregate_verify_bytes([pks[128 * i:128 * i + 128] for i in range(n)], msgs, agg, dst)
    assert not oracle.aggregate_verify(pks, [b"msg-0", b"msg-1", b"msg-2", b"oops"], agg, dst)
    assert not oracle.aggregate_verify(b"", [], agg, dst)
    # threshold: 3-of-5 Shamir shares of sk, combine partial signatures, verify under f(0)*G2
    coeffs = [rnd.randrange(1, pyref.R) for _ in range(3)]
    f = lambda x: sum(c * pow(x, i, pyref.R) for i, c in enumerate(coeffs)) % pyref.R
    ids = [2, 5, 3]
    msg = b"threshold"
    parts = b"".join(oracle.sign(f(i), msg, dst) for i in ids)
    idb = b"".join(i.to_bytes(32, "big") for i in ids)
    lam = oracle.fr_lagrange_at_zero(idb, 3)
    assert [int.from_bytes(lam[32 * i:32 * i + 32], "big") for i in range(3)] == pyref.lagrange_at_zero(ids)
    sig = oracle.threshold_combine(idb, parts, 3)
    assert sig == oracle.sign(coeffs[0], msg, dst)
    assert oracle.verify_batch(oracle.sk_to_pk(coeffs[0]), [msg], sig, dst) == b"\x01"
    with pytest.raises(oracle.OracleError) as e:
        oracle.threshold_combine(idb[:32] * 3, parts, 3)           # duplicate ids
    assert e.value.rc == 1


def test_compressed_codecs(oracle, pyref):
    """Corrected G1 rule (flag = parity of y) round-trips EVERY point, unlike the reference's from_compressed
    (g1.rs:320, E8); the generator vector of the reference's own round-trip test (g1.rs:938-953) is kept."""
    rnd = random.Random(6)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    assert oracle.g1_compress(G1) == (1).to_bytes(32, "big")                    # y = 2 is even: flag clear (g1.rs:956-965)
    assert oracle.g1_decompress(oracle.g1_compress(G1)) == G1
    assert oracle.g2_decompress(oracle.g2_compress(G2)) == G2                   # g2.rs:963-978
    parities = set()
    for _ in range(20):
        p = oracle.g1_mul(G1, rnd.randrange(1, pyref.R)); q = oracle.g2_mul(G2, rnd.randrange(1, pyref.R))
        c = oracle.g1_compress(p)
        parities.add((c[0] >> 7, int.from_bytes(p[32:], "big") > (pyref.P - 1) // 2))
        assert oracle.g1_decompress(c) == p and oracle.g2_decompress(oracle.g2_compress(q)) == q
        x = int.from_bytes(p[:32], "big"); y = int.from_bytes(p[32:], "big")
        assert (c[0] >> 7) == (y & 1) and int.from_bytes(bytes([c[0] & 0x7f]) + c[1:], "big") == x
    assert len(parities) >= 3           # (odd,low) / (even,high) points exist in the sample: the cases E8 breaks
    ident1 = bytes(32) + (1).to_bytes(32, "big")
    assert oracle.g1_decompress(oracle.g1_compress(ident1)) == ident1
    assert oracle.g1_decompress(pyref.P.to_bytes(32, "big")) is None            # x >= p
    # an x with no point on the curve
    x = 1
    while pyref.fp_sqrt((x ** 3 + 3) % pyref.P) is not None:
        x += 1
    assert oracle.g1_decompress(x.to_bytes(32, "big")) is None


def test_reference_style_multiply(oracle, pyref):
    """The timing-only restatement of Fp::multiply (fp.rs:404-407, wide product + const_rem_wide) is a correct
    multiply, so the cost ratio bench.py reports is between two implementations of the same function."""
    rnd = random.Random(9)
    P = pyref.P
    for a, b in [(0, 0), (1, P - 1), (P - 1, P - 1), (2**253, 2**253)] + [(rnd.randrange(P), rnd.randrange(P)) for _ in range(100)]:
        assert oracle.fp_mul_refstyle(a, b) == a * b % P
    assert oracle.bench_fp_mul(True, 1000) > 0 and oracle.bench_fp_mul(False, 1000) > 0


def test_oracle_under_address_and_ub_sanitizers():
    """The C oracle driven by oracle/asan_check.c, built with -fsanitize=address,undefined (CPU build only: GPU ASan is
    not available on the pool).  Every entry-point family on small, ragged and malformed inputs."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", "asan_check"], stderr=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(root, "oracle", "_build", "asan_check")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert out.returncode == 0 and "asan_check ok" in out.stdout, out.stderr[-3000:]


def _rand_fp_bytes(rnd, n, P):
    vals = [rnd.randrange(P) for _ in range(n)]
    for k, v in enumerate((0, 1, P - 1, 2, P - 2)):
        if k < n:
            vals[k] = v
    return b"".join(v.to_bytes(32, "big") for v in vals), vals


def test_oracle_field_ops_vs_independent_model(oracle, pyref):
    """oracle_field_op_batch (the CPU side of the device's primitive-parity ABI) against the structurally different pure-
    Python model: Fp / Fp2 by integer arithmetic, Fp12 by the dense degree-6 polynomial model.  The reference has no Fp6 /
    Fp12 vectors, so these two agreeing is what the GPU fuzz tests lean on."""
    B, P = pyref, pyref.P
    rnd = random.Random(77)
    n = 24
    a, av = _rand_fp_bytes(rnd, n, P)
    b, bv = _rand_fp_bytes(random.Random(78), n, P)
    i2b = lambda vs: b"".join(v.to_bytes(32, "big") for v in vs)
    assert oracle.field_op_batch(0, a, b, n) == i2b([x * y % P for x, y in zip(av, bv)])
    assert oracle.field_op_batch(1, a, None, n) == i2b([x * x % P for x in av])
    assert oracle.field_op_batch(2, a, None, n) == i2b([pow(x, P - 2, P) for x in av])
    assert oracle.field_op_batch(3, a, b, n) == i2b([(x + y) % P for x, y in zip(av, bv)])
    assert oracle.field_op_batch(4, a, b, n) == i2b([(x - y) % P for x, y in zip(av, bv)])
    assert oracle.field_op_batch(5, a, None, n) == i2b([(-x) % P for x in av])
    assert oracle.field_op_batch(8, a, None, n) == i2b([9 * x % P for x in av])
    sq = oracle.field_op_batch(6, a, None, n); isq = oracle.field_op_batch(7, a, None, n)
    for k, x in enumerate(av):
        is_sq = x == 0 or pow(x, (P - 1) // 2, P) == 1
        assert int.from_bytes(isq[32 * k:32 * k + 32], "big") == (1 if is_sq else 0)
        r = int.from_bytes(sq[32 * k:32 * k + 32], "big")
        assert (r * r % P == x) if is_sq else r == 0
    # Fp2: elements are c0 || c1
    a2 = a + b; n2 = n                      # pair (av[k], bv[k])?  build explicitly
    xs = [(rnd.randrange(P), rnd.randrange(P)) for _ in range(n2)]
    ys = [(rnd.randrange(P), rnd.randrange(P)) for _ in range(n2)]
    xs[0] = (0, 0); xs[1] = (1, 0); xs[2] = (0, 1); ys[3] = (P - 1, P - 1)
    f2b = lambda vs: b"".join(c0.to_bytes(32, "big") + c1.to_bytes(32, "big") for c0, c1 in vs)
    assert oracle.field_op_batch(16, f2b(xs), f2b(ys), n2) == f2b([B.f2_mul(x, y) for x, y in zip(xs, ys)])
    assert oracle.field_op_batch(17, f2b(xs), None, n2) == f2b([B.f2_sqr(x) for x in xs])
    assert oracle.field_op_batch(18, f2b(xs[1:]), None, n2 - 1) == f2b([B.f2_inv(x) for x in xs[1:]])
    assert oracle.field_op_batch(19, f2b(xs), None, n2) == f2b([B.f2_mul(x, B.XI) for x in xs])
    assert oracle.field_op_batch(20, f2b(xs), None, n2) == f2b([B.f2_conj(x) for x in xs])
    rt = oracle.field_op_batch(21, f2b(xs), None, n2)
    for k, x in enumerate(xs):
        r = (int.from_bytes(rt[64 * k:64 * k + 32], "big"), int.from_bytes(rt[64 * k + 32:64 * k + 64], "big"))
        assert B.f2_sqr(r) == x if B.f2_is_square(x) else r == (0, 0)
    # Fp12 through the Gt byte layout
    m = 10
    fa = [[(rnd.randrange(P), rnd.randrange(P)) for _ in range(6)] for _ in range(m)]
    fb = [[(rnd.randrange(P), rnd.randrange(P)) for _ in range(6)] for _ in range(m)]
    fa[0] = [(1, 0)] + [(0, 0)] * 5
    A12 = b"".join(B.f12_to_bytes(x) for x in fa); B12 = b"".join(B.f12_to_bytes(x) for x in fb)
    j = lambda xs_: b"".join(B.f12_to_bytes(x) for x in xs_)
    assert oracle.field_op_batch(48, A12, B12, m) == j([B.f12_mul(x, y) for x, y in zip(fa, fb)])
    assert oracle.field_op_batch(49, A12, None, m) == j([B.f12_sqr(x) for x in fa])
    assert oracle.field_op_batch(50, A12, None, m) == j([B.f12_inv(x) for x in fa])
    assert oracle.field_op_batch(51, A12, None, m) == j([B.f12_conj(x) for x in fa])
    for k in (1, 2, 3):
        assert oracle.field_op_batch(51 + k, A12, None, m) == j([B.f12_frob(x, k) for x in fa])
    # sparse product: only the 0 / 3 / 4 slots (w^0, w^1, w^3 in the model's w-power indexing) of b are used
    sparse = []
    for y in fb:
        full = B.f12_from_bytes(B.f12_to_bytes(y))
        keep = B.f12_from_bytes(B.f12_to_bytes(y)[:64] + bytes(128) + B.f12_to_bytes(y)[192:320] + bytes(64))
        sparse.append(keep)
    assert oracle.field_op_batch(56, A12, B12, m) == j([B.f12_mul(x, y) for x, y in zip(fa, sparse)])
    # Fp6 via Fp12 with c1 = 0: (a0, 0) * (b0, 0) = (a0 b0, 0)
    six = lambda x: B.f12_to_bytes(x)[:192]
    a6 = b"".join(six(x) for x in fa); b6 = b"".join(six(x) for x in fb)
    emb = lambda s6: b"".join(s6[192 * k:192 * k + 192] + bytes(192) for k in range(m))
    prod = oracle.field_op_batch(32, a6, b6, m)
    assert emb(prod) == oracle.field_op_batch(48, emb(a6), emb(b6), m)
    assert oracle.field_op_batch(33, a6, None, m) == oracle.field_op_batch(32, a6, a6, m)
    inv6 = oracle.field_op_batch(34, a6, None, m)
    one6 = (1).to_bytes(32, "big") + bytes(160)
    assert oracle.field_op_batch(32, a6, inv6, m) == one6 * m
    # cyclotomic squaring == squaring on cyclotomic elements (easy part of the final exponentiation output)
    g = oracle.pairing_batch(oracle.g1_generator(), oracle.g2_generator(), 1)
    assert oracle.field_op_batch(55, g, None, 1) == oracle.field_op_batch(49, g, None, 1)
