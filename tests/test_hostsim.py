"""CPU-only: the DEVICE arithmetic headers compiled for the host with -DBN_CHECK.

Two purposes: (1) the interval discipline of the lazy radix-2^29 limbs (fp29.h) is asserted at every
multiply on every code path the kernels run -- the tracked bounds are data independent, so one pass
proves them for all inputs; (2) the exact code the GPU runs is compared with the oracle without a GPU.
This is a test tool; the product has no CPU path."""
import ctypes
import json
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tests", "hostsim")


@pytest.fixture(scope="module")
def hs():
    so = os.path.join(SIM, "libhostsim.so")
    src = [os.path.join(SIM, "hostsim.cpp")] + [os.path.join(ROOT, "bls-bn254_amd", "csrc", f)
                                               for f in os.listdir(os.path.join(ROOT, "bls-bn254_amd", "csrc")) if f.endswith(".h")]
    if not os.path.exists(so) or any(os.path.getmtime(p) > os.path.getmtime(so) for p in src):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-DBN_CHECK", "-DBN_VERIFY_PARK_T", "-fPIC", "-shared", "-pthread", "-o", so, os.path.join(SIM, "hostsim.cpp")])
    return ctypes.CDLL(so)


def b32(x):
    return x.to_bytes(32, "big")


def test_fp_arithmetic(hs, pyref):
    P = pyref.P
    rnd = random.Random(1)
    out = ctypes.create_string_buffer(32)
    vals = [0, 1, 2, P - 1, P - 2, 2**253, (1 << 29) - 1, 1 << 232] + [rnd.randrange(P) for _ in range(60)]
    for x in vals:
        for y in vals[:12] + [rnd.randrange(P) for _ in range(3)]:
            hs.hs_fp_mul(b32(x), b32(y), out); assert int.from_bytes(out.raw, "big") == x * y % P
            hs.hs_fp_mix(b32(x), b32(y), out); assert int.from_bytes(out.raw, "big") == ((x + y) * (x - y) + 9 * x - y) % P
        hs.hs_fp_sqr(b32(x), out); assert int.from_bytes(out.raw, "big") == x * x % P
    # fp_inv = the divstep recurrence (21 batches of 29 steps), checked against Fermat's power (the reference's Fp::invert) and
    # against the host-side copy of the power ladder: edge values, powers of two, long carry runs, random
    cases = [0, 1, 2, 3, P - 1, P - 2, (P + 1) // 2, (P - 1) // 2, P // 3, 2 ** 253, 2 ** 253 + 1, 2 ** 200, 2 ** 29, 2 ** 29 - 1, 2 ** 58,
             2 ** 232, 2 ** 232 - 1, (1 << 253) - 1] + [rnd.randrange(P) for _ in range(300)] + [rnd.randrange(1 << 64) for _ in range(20)]
    for x in cases:
        hs.hs_fp_inv(b32(x), out); assert int.from_bytes(out.raw, "big") == pow(x, P - 2, P), hex(x)      # inv0(0) = 0
    for x in cases[:24]:
        hs.hs_fp_inv_pow(b32(x), out); assert int.from_bytes(out.raw, "big") == pow(x, P - 2, P)
        y = rnd.randrange(P)
        hs.hs_fp_inv_lazy(b32(x), b32(y), out); assert int.from_bytes(out.raw, "big") == pow((x - y) % P, P - 2, P)
    assert hs.hs_fp_decode_ok(b32(P - 1)) == 1 and hs.hs_fp_decode_ok(b32(P)) == 0 and hs.hs_fp_decode_ok(b"\xff" * 32) == 0
    for okm in [os.urandom(48) for _ in range(10)] + [b"\xff" * 48, bytes(48)]:
        hs.hs_fp_from_okm(okm, out); assert int.from_bytes(out.raw, "big") == int.from_bytes(okm, "big") % P


def test_pairing_path_bit_exact_and_bounded(hs, oracle, pyref, kats):
    rnd = random.Random(2)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    ml = ctypes.create_string_buffer(384); gt = ctypes.create_string_buffer(384); st = ctypes.c_int(0)
    hs.hs_stats_reset()
    hs.hs_miller1(G1, G2, ml, ctypes.byref(st))
    assert st.value == 3 and ml.raw == oracle.miller_loop_batch(G1, G2, 1)
    assert hs.hs_final_exp(ml.raw, gt) == 0
    assert gt.raw.hex() == kats["constants"]["gt_generator_bytes_hex"]
    for _ in range(2):
        p, q = oracle.g1_mul(G1, rnd.randrange(1, pyref.R)), oracle.g2_mul(G2, rnd.randrange(1, pyref.R))
        hs.hs_miller1(p, q, ml, ctypes.byref(st)); assert ml.raw == oracle.miller_loop_batch(p, q, 1)
        hs.hs_pairing(p, q, gt); assert gt.raw == oracle.pairing_batch(p, q, 1)
    assert hs.hs_expx_chain_matches(oracle.miller_loop_batch(G1, G2, 1)) == 1       # t^x: addition chain == binary ladder
    assert hs.hs_fe_h3_loop_matches(oracle.miller_loop_batch(G1, G2, 1)) == 1       # interpreter h3 (kernel) == register h3
    assert hs.hs_fe_tails_match(oracle.miller_loop_batch(G1, G2, 1)) == 1           # fe_h1 / fe_h2 as tails of the t^x kernels
    ident = bytes(32) + (1).to_bytes(32, "big")
    hs.hs_miller1(ident, G2, ml, ctypes.byref(st))
    assert st.value == 7 and ml.raw == (1).to_bytes(32, "big") + bytes(352)
    stats = (ctypes.c_double * 8)()
    hs.hs_stats(stats)
    assert stats[0] < 2.97 and stats[1] < 2.97 and stats[2] < 167          # proven worst-case budgets (fp29.h)


def test_hash_checks_and_verify(hs, oracle, pyref, kats):
    rnd = random.Random(3)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    o64 = ctypes.create_string_buffer(64); o128 = ctypes.create_string_buffer(128)
    g1k = kats["g1"]
    for v in g1k["hash"]:
        m = v["msg"].encode(); d = g1k["hash_dst"].encode()
        hs.hs_hash_to_g1(m, len(m), d, len(d), 1, o64); assert o64.raw.hex() == v["p_x"] + v["p_y"]
    for v in g1k["encode"][:3]:
        m = v["msg"].encode(); d = g1k["encode_dst"].encode()
        hs.hs_hash_to_g1(m, len(m), d, len(d), 0, o64); assert o64.raw.hex() == v["p_x"] + v["p_y"]
    g2k = kats["g2"]
    for name, ro in (("hash", 1), ("encode", 0)):
        for v in g2k[name][:2]:
            m = v["msg"].encode(); d = g2k[name + "_dst"].encode()
            hs.hs_hash_to_g2(m, len(m), d, len(d), ro, o128)
            assert o128.raw.hex() == v["x_c1"] + v["x_c0"] + v["y_c1"] + v["y_c0"]
    bp = g2k["bad_point"]
    bad = bytes.fromhex(bp["x_c1"] + bp["x_c0"] + bp["y_c1"] + bp["y_c0"])
    from tests import synth
    assert hs.hs_g2_check(G2) == 1 and hs.hs_g2_check(bad) == 0 and hs.hs_g2_check(synth.NON_SUBGROUP_PK) == 0
    ident = bytes(32) + (1).to_bytes(32, "big")
    assert hs.hs_g1_check(G1) == 1 and hs.hs_g1_check(ident) == 1 and hs.hs_g1_check(G1[:32] + (3).to_bytes(32, "big")) == 0
    p = oracle.g1_mul(G1, rnd.randrange(1, pyref.R))
    hs.hs_g1_add(p, G1, o64); assert o64.raw == oracle.g1_add(p, G1)
    hs.hs_g1_add(p, p, o64); assert o64.raw == oracle.g1_add(p, p)
    hs.hs_g1_add(p, ident, o64); assert o64.raw == p
    k = rnd.randrange(1, pyref.R)
    hs.hs_g1_mul(p, k.to_bytes(32, "big"), o64); assert o64.raw == oracle.g1_mul(p, k)
    # Mul<Scalar> by 4-bit windows (blsbn254_g1_mul_batch / g2_mul_batch), incl. 0, 1, r - 1, a scalar with zero digits and the identity
    q2 = oracle.g2_mul(G2, rnd.randrange(1, pyref.R))
    ident2 = bytes(64) + bytes(32) + (1).to_bytes(32, "big")
    for kk in (0, 1, pyref.R - 1, 0x1000_0000_0000_0000_0000_0000_0000_0000_f, k):
        kb = kk.to_bytes(32, "big")
        hs.hs_g1_mul_win4(p, kb, o64); assert o64.raw == oracle.g1_mul(p, kk)
        hs.hs_g2_mul_win4(q2, kb, o128); assert o128.raw == oracle.g2_mul(q2, kk)
    hs.hs_g1_mul_win4(ident, k.to_bytes(32, "big"), o64); assert o64.raw == ident
    hs.hs_g2_mul_win4(ident2, k.to_bytes(32, "big"), o128); assert o128.raw == ident2
    dst = pyref.DEFAULT_DST
    sk = rnd.randrange(1, pyref.R)
    pk = oracle.sk_to_pk(sk); msg = b"hello"; sig = oracle.sign(sk, msg, dst)
    ml = ctypes.create_string_buffer(384)
    assert hs.hs_verify(pk, msg, len(msg), sig, dst, len(dst), ml) == 1
    negG2 = pyref.g2_to_bytes(pyref.g2_neg(pyref.G2_GEN))
    H = oracle.hash_to_g1_batch([msg], dst)
    assert ml.raw == oracle.multi_miller_loop(sig + H, negG2 + pk, 2)        # fixed-Q line table == on-the-fly lines
    ml2 = ctypes.create_string_buffer(384); fl = ctypes.c_int(0)
    hs.hs_miller_verify_ws(pk, sig, H, ml2, ctypes.byref(fl))                # workspace-reload loop (the kernel's) == register loop
    assert ml2.raw == ml.raw and fl.value == 3
    assert hs.hs_verify(pk, b"hellp", 5, sig, dst, len(dst), None) == 0
    assert hs.hs_verify(synth.NON_SUBGROUP_PK, msg, len(msg), sig, dst, len(dst), None) == 0


def test_compressed_codecs(hs, oracle, pyref):
    rnd = random.Random(8)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    c32 = ctypes.create_string_buffer(32); b64 = ctypes.create_string_buffer(64)
    c64 = ctypes.create_string_buffer(64); b128 = ctypes.create_string_buffer(128)
    for _ in range(6):
        p = oracle.g1_mul(G1, rnd.randrange(1, pyref.R)); q = oracle.g2_mul(G2, rnd.randrange(1, pyref.R))
        assert hs.hs_g1_codec_roundtrip(p, c32, b64) == 1 and b64.raw == p and c32.raw == oracle.g1_compress(p)
        assert hs.hs_g2_codec_roundtrip(q, c64, b128) == 1 and b128.raw == q and c64.raw == oracle.g2_compress(q)


def test_keygen_and_hash_to_scalar(hs, pyref):
    """HMAC / HKDF / XMD-to-scalar device code against the stdlib hmac + big-int restatement."""
    rnd = random.Random(7)
    out = ctypes.create_string_buffer(32)
    for okm in [bytes(48), b"\xff" * 48, (pyref.R).to_bytes(48, "big"), (pyref.R - 1).to_bytes(48, "big"),
                (2 * pyref.R + 5).to_bytes(48, "big")] + [rnd.randbytes(48) for _ in range(20)]:
        hs.hs_fr_from_okm(okm, out)
        assert int.from_bytes(out.raw, "big") == int.from_bytes(okm, "big") % pyref.R
    for ikm_len, info in ((32, b""), (32, b"key info"), (33, b"x" * 70), (64, b""), (100, b"abc")):
        ikm = rnd.randbytes(ikm_len)
        assert hs.hs_keygen(ikm, ctypes.c_size_t(ikm_len), info, ctypes.c_size_t(len(info)), out) == 1
        assert int.from_bytes(out.raw, "big") == pyref.keygen(ikm, info)
    dst = b"QUUX-V01-CS02-with-BN254FR_XMD:SHA-256"
    for msg in (b"", b"abc", rnd.randbytes(200)):
        hs.hs_hash_to_scalar(msg, ctypes.c_size_t(len(msg)), dst, len(dst), out)
        assert int.from_bytes(out.raw, "big") == pyref.hash_to_scalar(msg, dst)


def test_svdw_shared_inversion_edge_cases(hs, pyref):
    """hash_to_g1_from_fields shares one inversion between its two SVDW maps; u = +-1/2 makes a map's denominator
    zero (inv0 path), which the shared inversion must not leak into the other map."""
    P = pyref.P
    half = (P + 1) // 2
    rnd = random.Random(11)
    out = ctypes.create_string_buffer(64)
    us = [0, 1, half, P - half, rnd.randrange(P), rnd.randrange(P)]
    for u0 in us:
        for u1 in us:
            hs.hs_g1_from_fields(b32(u0), b32(u1), out)
            want = pyref.g1_add(pyref.svdw_g1(u0), pyref.svdw_g1(u1))
            assert out.raw == pyref.g1_to_bytes(want), (u0, u1)


def test_is_square_by_jacobi_symbol(hs, pyref):
    """fp_is_square (binary Jacobi algorithm, no exponentiation) == Euler's criterion, zero counted as a square."""
    P = pyref.P
    rnd = random.Random(12)
    vals = [0, 1, 2, 3, 4, P - 1, P - 2, (P + 1) // 2, 2**253, 2**253 + 1] + [rnd.randrange(P) for _ in range(300)]
    vals += [v * v % P for v in vals[:40]]
    for v in vals:
        assert hs.hs_fp_is_square(b32(v)) == (1 if pow(v, (P - 1) // 2, P) in (0, 1) else 0), v


def test_glv_split_and_endomorphism(hs, oracle, pyref):
    """glv.h on the host: k = k1 + k2 lambda (mod r) with both halves below 2^127, and phi(P) = (beta x, y) = [lambda] P
    (the decomposition the threshold MSM runs on the device)."""
    R = pyref.R
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    assert (lam * lam + lam + 1) % R == 0
    rnd = random.Random(9)
    out = ctypes.create_string_buffer(32)
    for k in [0, 1, 2, R - 1, R - 2, lam, R - lam, (R - 1) // 2, 1 << 253] + [rnd.randrange(R) for _ in range(3000)]:
        sg = hs.hs_glv_split(b32(k), out)
        k1 = int.from_bytes(out.raw[:16], "big") * (-1 if sg & 1 else 1)
        k2 = int.from_bytes(out.raw[16:], "big") * (-1 if sg & 2 else 1)
        assert (k1 + k2 * lam - k) % R == 0 and abs(k1) < (1 << 127) and abs(k2) < (1 << 127), hex(k)
    G1 = oracle.g1_generator()
    o64 = ctypes.create_string_buffer(64)
    for _ in range(4):
        p = oracle.g1_mul(G1, rnd.randrange(1, R))
        hs.hs_glv_phi(p, o64)
        assert o64.raw == oracle.g1_mul(p, lam)


def test_parked_and_prepared_miller_loops(hs, oracle, pyref):
    """The loops behind k_miller_1 (invariants and T parked), k_miller_hpk2 (two variable pairs sharing f^2), k_miller_hpk2p
    (two prepared pairs) and k_miller_prepared (pair tables: key line x -G2gen line) on the host with the interval checker:
    every one of them produces the oracle's Miller value bit for bit."""
    rnd = random.Random(21)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    neg_g2 = pyref.g2_to_bytes(pyref.g2_neg(pyref.G2_GEN))
    o1 = ctypes.create_string_buffer(384); o2 = ctypes.create_string_buffer(384)
    for _ in range(2):
        ha, hb, sig = (oracle.g1_mul(G1, rnd.randrange(1, pyref.R)) for _ in range(3))
        qa, qb = (oracle.g2_mul(G2, rnd.randrange(1, pyref.R)) for _ in range(2))
        hs.hs_miller1_ws(ha, qa, o1)
        assert o1.raw == oracle.miller_loop_batch(ha, qa, 1)
        hs.hs_miller2(ha, qa, hb, qb, 1, o1, o2)
        want = oracle.multi_miller_loop(ha + hb, qa + qb, 2)
        assert o1.raw == want and o2.raw == want
        hs.hs_miller2(ha, qa, hb, qb, 0, o1, o2)                 # second pair is padding: contributes exactly 1
        want = oracle.multi_miller_loop(ha, qa, 1)
        assert o1.raw == want and o2.raw == want
        want = oracle.multi_miller_loop(sig + ha, neg_g2 + qa, 2)
        hs.hs_miller_prepared(sig, ha, qa, 1, o1, o2)            # H = (x : y : 1): the textbook Miller value itself
        assert o1.raw == want and o2.raw == oracle.final_exponentiation(want, 1)
        hs.hs_miller_prepared(sig, ha, qa, 5, o1, o2)            # H = (5x : 5y : 5): differs by 5^88 in Fp, gone after the final exponentiation
        assert o1.raw != want and o2.raw == oracle.final_exponentiation(want, 1)


def test_wide_final_exponentiation_one_wave_per_tuple(hs, oracle, pyref):
    """wide.h: the hard part of the final exponentiation with the 64 lanes of a wave sharing ONE tuple (Fp12 product = 36
    Fp2 products + 6 sums, cyclotomic squaring = 9 Fp2 squarings + 6 combinations), run here phase by phase over the 64 lane
    ids under the interval checker: every primitive equals its serial counterpart of tower.h byte for byte, and
    fe_easy + wide hard part == final_exponentiation == the oracle."""
    rnd = random.Random(77)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()

    def ml():
        return oracle.miller_loop_batch(oracle.g1_mul(G1, rnd.randrange(1, pyref.R)), oracle.g2_mul(G2, rnd.randrange(1, pyref.R)), 1)
    a = ctypes.create_string_buffer(384); b = ctypes.create_string_buffer(384)
    x, y = ml(), ml()
    for op in (0, 2, 4, 5, 6):                                   # product, conjugate, Frobenius^1..3 on arbitrary Fp12 values
        assert hs.hs_wide_op(op, x, y, a, b) == 0 and a.raw == b.raw, op
    one = (1).to_bytes(32, "big") + bytes(352)
    assert hs.hs_wide_op(0, x, one, a, b) == 0 and a.raw == b.raw == x
    gt = oracle.pairing_batch(oracle.g1_mul(G1, 5), oracle.g2_mul(G2, 7), 1)
    assert hs.hs_wide_op(1, gt, None, a, b) == 0 and a.raw == b.raw                                 # cyclotomic squaring
    assert a.raw == oracle.gt_mul(gt, gt) if hasattr(oracle, "gt_mul") else True
    for m in (x, y, ml()):
        assert hs.hs_final_exp_wide(m, a) == 0 and hs.hs_final_exp(m, b) == 0
        assert a.raw == b.raw == oracle.final_exponentiation(m, 1)


def test_easy_part_split_at_its_inversion(hs, oracle, pyref):
    """fe_easy_head / fp_inv4 / fe_easy_tail (the three launches of the easy part from a round of waves up) == fe_easy, for
    every position of the tuple among the four that share one inversion, with a zero and a one as neighbours; under the
    interval checker."""
    rnd = random.Random(78)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    for slot in range(4):
        m = oracle.miller_loop_batch(oracle.g1_mul(G1, rnd.randrange(1, pyref.R)), oracle.g2_mul(G2, rnd.randrange(1, pyref.R)), 1)
        assert hs.hs_fe_easy_split_matches(m, slot) == 1, slot
    one = (1).to_bytes(32, "big") + bytes(352)
    assert hs.hs_fe_easy_split_matches(one, 0) == 1


def test_wide_miller_loops_over_prepared_keys(hs, oracle, pyref):
    """wide.h Miller loops (R <- R^2, R <- R * L with L assembled by five lanes from the key's table): the one-pair loop equals
    miller_loop_1prepared and the oracle's Miller value, the verify loop over the pair table equals miller_loop_prepared, byte
    for byte, under the interval checker."""
    rnd = random.Random(78)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    out = ctypes.create_string_buffer(4 * 384)
    for _ in range(2):
        sig, h = (oracle.g1_mul(G1, rnd.randrange(1, pyref.R)) for _ in range(2))
        pk = oracle.g2_mul(G2, rnd.randrange(1, pyref.R))
        assert hs.hs_miller_wide(sig, h, pk, out) == 0
        w1, s1, w2, s2 = (out.raw[384 * k:384 * k + 384] for k in range(4))
        assert w1 == s1 == oracle.miller_loop_batch(h, pk, 1)
        assert w2 == s2
        # variable G2 point: lane 0 computes the lines, the wave multiplies them in
        assert hs.hs_miller_wide_var(h, pk, out) == 0
        assert out.raw[:384] == out.raw[384:768] == oracle.miller_loop_batch(h, pk, 1)


def test_tri_three_lanes_per_tuple(hs, oracle, pyref):
    """tri.h: the Fp12 arithmetic with a QUAD of lanes per tuple (lane 0: c0, lane 1: c1, lane 2: the Karatsuba cross product),
    run here as four threads per quad with a rendezvous for every DPP fetch, under the interval checker: every primitive equals
    its serial counterpart of tower.h byte for byte; the table-only verify Miller loop equals miller_loop_prepared; the hard
    part after the serial easy part equals final_exponentiation and the oracle."""
    rnd = random.Random(79)
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()

    def ml():
        return oracle.miller_loop_batch(oracle.g1_mul(G1, rnd.randrange(1, pyref.R)), oracle.g2_mul(G2, rnd.randrange(1, pyref.R)), 1)
    a = ctypes.create_string_buffer(384); b = ctypes.create_string_buffer(384)
    x, y = ml(), ml()
    for op in (0, 1, 3, 4, 5, 6):                              # product, square, conjugate, Frobenius^1..3 on arbitrary Fp12 values
        assert hs.hs_tri_op(op, x, y, a, b) == 0 and a.raw == b.raw, op
    one = (1).to_bytes(32, "big") + bytes(352)
    assert hs.hs_tri_op(0, x, one, a, b) == 0 and a.raw == b.raw == x
    gt = oracle.pairing_batch(oracle.g1_mul(G1, 5), oracle.g2_mul(G2, 7), 1)
    assert hs.hs_tri_op(2, gt, None, a, b) == 0 and a.raw == b.raw == oracle.gt_mul(gt, gt)          # cyclotomic squaring
    dst = pyref.DEFAULT_DST
    sk = rnd.randrange(1, pyref.R)
    pk = oracle.sk_to_pk(sk); msg = b"tri"; sig = oracle.sign(sk, msg, dst)
    H = oracle.hash_to_g1_batch([msg], dst)
    negG2 = pyref.g2_to_bytes(pyref.g2_neg(pyref.G2_GEN))
    assert hs.hs_tri_miller(sig, H, pk, 1, a, b) == 0 and a.raw == b.raw == oracle.multi_miller_loop(sig + H, negG2 + pk, 2)
    assert hs.hs_tri_miller(sig, H, pk, 3, a, b) == 0 and a.raw == b.raw                              # homogeneous H (Z = 3)
    flag = ctypes.c_int(-1)
    assert hs.hs_tri_final_exp(a.raw, b, ctypes.byref(flag)) == 0 and flag.value == 1 and b.raw == one   # a valid tuple: e(sig, -G2) e(H, pk) = 1
    for m in (x, y):
        assert hs.hs_tri_final_exp(m, a, ctypes.byref(flag)) == 0 and flag.value == 0
        assert a.raw == oracle.final_exponentiation(m, 1)


def test_quad_four_lanes_per_key(hs, oracle, pyref):
    """quad.h: the per-key preparation (88 line triples, psi subgroup test) with four lanes per G2 point, run as four threads per
    quad under the interval checker: the line table equals g2_prepare_lines limb for limb, the subgroup bit equals
    g2_torsion_free -- for keys in the subgroup, the reference's bad point and a point of the twist outside the r-torsion."""
    from tests import synth
    rnd = random.Random(80)
    G2 = oracle.g2_generator()
    bits = ctypes.c_int(0)
    for pk in (oracle.sk_to_pk(rnd.randrange(1, pyref.R)), G2):
        assert hs.hs_quad_prepare(pk, ctypes.byref(bits)) == 1 and bits.value == 7
    assert hs.hs_quad_prepare(synth.NON_SUBGROUP_PK, ctypes.byref(bits)) == 1 and bits.value == 4      # same table, both say "outside", all lanes agree
    # the one-pair Miller loop with a variable G2 point on a quad (quad line steps + three-lane accumulator) == miller_loop_1 == the oracle
    a = ctypes.create_string_buffer(384); b = ctypes.create_string_buffer(384)
    g1 = oracle.g1_mul(oracle.g1_generator(), rnd.randrange(1, pyref.R)); g2 = oracle.g2_mul(G2, rnd.randrange(1, pyref.R))
    assert hs.hs_tri_miller_1(g1, g2, a, b) == 0 and a.raw == b.raw == oracle.miller_loop_batch(g1, g2, 1)
    assert hs.hs_tri_miller_1prepared(g1, g2, a, b) == 0 and a.raw == b.raw == oracle.miller_loop_batch(g1, g2, 1)     # lines from the key's table
