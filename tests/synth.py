"""Deterministic synthetic BLS batches: seeded key pool, 32-byte messages, signatures
made with the CPU oracle, and a known pattern of invalid tuples.  Test/bench data generation only."""
import hashlib

import numpy as np

SEED = b"BLSBN254"
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47


def sk_of(k):
    h = hashlib.sha256(SEED + b"sk" + k.to_bytes(4, "little")).digest() + hashlib.sha256(SEED + b"sk" + k.to_bytes(4, "little") + b"\x01").digest()
    return int.from_bytes(h[:48], "big") % R or 1


def msg_of(i):
    return hashlib.sha256(SEED + b"msg" + i.to_bytes(8, "little")).digest()


def make_batch(oracle, n, dst, pool=8, invalid_every=0, uniq=None):
    """Returns (pks bytes, msgs list, sigs bytes, expected bool list).  `uniq` distinct signed tuples are
    generated with the oracle and tiled to n (signing is CPU-bound); every `invalid_every`-th tuple is
    corrupted in a rotating way."""
    uniq = min(n, uniq or n)
    sks = [sk_of(k) for k in range(pool)]
    pk_pool = [oracle.sk_to_pk(s) for s in sks]
    base = []
    for i in range(uniq):
        m = msg_of(i)
        base.append((pk_pool[i % pool], m, oracle.sign(sks[i % pool], m, dst)))
    g1 = oracle.g1_generator()
    pks, msgs, sigs, exp = [], [], [], []
    for i in range(n):
        pk, m, s = base[i % uniq]
        ok = True
        if invalid_every and i % invalid_every == invalid_every - 1:
            kind = (i // invalid_every) % 5
            ok = False
            if kind == 0:
                m = bytes([m[0] ^ 1]) + m[1:]                         # flipped message bit
            elif kind == 1:
                s = oracle.g1_add(s, g1)                              # wrong signature (still on curve)
            elif kind == 2:
                pk = pk_pool[(i % uniq + 1) % pool]                   # wrong key
            elif kind == 3:
                s = s[:32] + (int.from_bytes(s[32:], "big") ^ 1).to_bytes(32, "big")   # off-curve signature
            else:
                pk = NON_SUBGROUP_PK                                  # on the twist, not in the r-torsion
        pks.append(pk); msgs.append(m); sigs.append(s); exp.append(ok)
    return b"".join(pks), msgs, b"".join(sigs), exp


def make_batch_gpu(engine, oracle, n, dst, pool=1024, invalid_every=0, spot=1000, seed=1, base=0):
    """n UNIQUE signed tuples: key pool of `pool` keys, 32-byte messages msg_g, signatures
    sig_g = [sk_(g mod pool)] H(msg_g) made by the engine's own GPU signing kernels and spot-checked at `spot`
    random indices against the CPU oracle.  g = base + i is the tuple's GLOBAL index (rank r of a sharded run
    passes base = r * n, so every rank holds different messages); corruption pattern as in make_batch, keyed on g."""
    import random
    pool = min(pool, max(n, 1))
    sks = [sk_of(k) for k in range(pool)]
    skb = b"".join(s.to_bytes(32, "big") for s in sks)
    pk_pool = engine.sk_to_pk_batch(skb, pool)
    msgs = [msg_of(base + i) for i in range(n)]
    rot = base % pool                                   # tuple i uses key (base + i) mod pool
    skr = skb[32 * rot:] + skb[:32 * rot]
    pkr = pk_pool[128 * rot:] + pk_pool[:128 * rot]
    sk_all = skr * (n // pool) + skr[:32 * (n % pool)]
    sigs = bytearray(engine.sign_batch(sk_all, msgs, dst))
    pks = bytearray(pkr * (n // pool) + pkr[:128 * (n % pool)])
    rnd = random.Random(seed + base)
    for i in ([0, n - 1] + [rnd.randrange(n) for _ in range(spot)])[:spot + 2]:
        k = (base + i) % pool
        assert bytes(sigs[64 * i:64 * i + 64]) == oracle.sign(sks[k], msgs[i], dst), "GPU signature differs from the oracle at %d" % i
        if i < pool or i % 97 == 0:
            assert bytes(pks[128 * i:128 * i + 128]) == oracle.sk_to_pk(sks[k])
    exp = [True] * n
    if invalid_every:
        g1 = oracle.g1_generator()
        first = (invalid_every - 1 - base) % invalid_every
        for i in range(first, n, invalid_every):
            g = base + i
            kind = (g // invalid_every) % 5
            exp[i] = False
            if kind == 0:
                msgs[i] = bytes([msgs[i][0] ^ 1]) + msgs[i][1:]
            elif kind == 1:
                sigs[64 * i:64 * i + 64] = oracle.g1_add(bytes(sigs[64 * i:64 * i + 64]), g1)
            elif kind == 2:
                j = (g + 1) % pool
                pks[128 * i:128 * i + 128] = pk_pool[128 * j:128 * j + 128]
            elif kind == 3:
                sigs[64 * i + 63] ^= 1
            else:
                pks[128 * i:128 * i + 128] = NON_SUBGROUP_PK
    return bytes(pks), msgs, bytes(sigs), exp


def expected_bits(n_total, invalid_every):
    """closed form of the validity pattern over global indices 0..n_total-1"""
    a = np.ones(n_total, dtype=np.uint8)
    if invalid_every:
        a[invalid_every - 1::invalid_every] = 0
    return a


def bitmap_of(bools):
    a = np.packbits(np.array(bools, dtype=np.uint8), bitorder="little")
    return a.tobytes()


# a point of E'(Fp2) outside the r-torsion (found with the pure-Python model, seed 11); x.c1|x.c0|y.c1|y.c0
NON_SUBGROUP_PK = None


def _init_non_subgroup():
    global NON_SUBGROUP_PK
    from oracle.pyref import bn254 as B
    import random
    rnd = random.Random(11)
    while True:
        x = (rnd.randrange(P), rnd.randrange(P))
        y = B.f2_sqrt(B.f2_add(B.f2_mul(B.f2_sqr(x), x), B.B2))
        if y is not None:
            NON_SUBGROUP_PK = B.g2_to_bytes((x, y))
            return


_init_non_subgroup()
