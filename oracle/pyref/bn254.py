"""Pure-Python big-int model of the BLS-BN254 verification path.  TEST INFRASTRUCTURE ONLY.

This is the *slow, independent* second restatement SURVEY.md §8(c) asks for: affine
group law, generic (dense) Fp12 arithmetic, affine line functions.  It exists to
  (1) pin the convention against the reference's golden vectors
      (hash-to-curve KATs g1.rs:992-1121 / g2.rs:1046-1305, Gt::generator()
      pairings.rs:387-479, gt^r == 1 pairings.rs:977-979),
  (2) cross-check the C oracle (oracle/bn254_oracle.c), which uses different
      (projective / sparse / Montgomery) formulas, and
  (3) derive constants (Frobenius coefficients, NAFs, Montgomery constants).

Nothing under bls-bn254_amd/ may import it.  Pure-Python loops: small cases only.

Reference files followed (read as a specification, /root/reference/src/inner_types):
  fp.rs:115-122 (from_okm), :164-168 (sgn0), :284-371 (SVDW map to G1), :433-458 (hash/encode)
  fp2.rs:95-99 (sgn0), :221-287 (SVDW map to G2), :441-452 (is_square), :454-487 (hash/encode)
  g1.rs:297-302,339-360 (codec), :910-928 (hash/encode)
  g2.rs:292-300,350-388 (codec, order c1||c0), :685-693 (clear_cofactor), :938-954 (psi)
  pairings.rs:499-514 (Gt byte layout)
Errata NOT reproduced: E1-E7 (tower/pairing built on xi = 9+u, optimal ate,
BN final exponent), E9 (y==1 heuristic), E15 (inv0(0) = 0).
"""
import hashlib

P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
X = 0x44e992b44a6909f1
ATE_LOOP = 6 * X + 2
assert P == 36 * X**4 + 36 * X**3 + 24 * X**2 + 6 * X + 1
assert R == 36 * X**4 + 36 * X**3 + 18 * X**2 + 6 * X + 1
# final exponent pinned by Gt::generator(): (p^12-1)/r * 2x(6x^2+3x+1)
FINAL_EXP = ((P**12 - 1) // R) * (2 * X * (6 * X * X + 3 * X + 1))
LAMBDA = (12 * X**3 + 12 * X**2 + 6 * X + 1, 12 * X**3 + 6 * X**2 + 4 * X,
          12 * X**3 + 6 * X**2 + 6 * X, 12 * X**3 + 6 * X**2 + 4 * X - 1)
assert FINAL_EXP == (P**6 - 1) * (P**2 + 1) * (LAMBDA[0] + LAMBDA[1] * P + LAMBDA[2] * P**2 + LAMBDA[3] * P**3)


def naf(n):
    """Non-adjacent form, least-significant digit first."""
    out = []
    while n:
        if n & 1:
            d = 2 - (n & 3)
            n -= d
        else:
            d = 0
        out.append(d)
        n >>= 1
    return out


# ---------------------------------------------------------------- Fp
def fp_inv(a):
    return pow(a, P - 2, P)          # inv0: 0 -> 0  (E15 fixed)


def fp_is_square(a):
    return pow(a, (P - 1) // 2, P) in (0, 1)     # fp.rs:428-431


def fp_sqrt(a):
    y = pow(a, (P + 1) // 4, P)      # p = 3 mod 4 (fp.rs:212-243 with v = 1)
    return y if y * y % P == a % P else None


def fp_sgn0(a):
    return a & 1


# ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (9, 1)


def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return (-a[0] % P, -a[1] % P)
def f2_conj(a): return (a[0], -a[1] % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_sqr(a): return f2_mul(a, a)
def f2_muls(a, s): return (a[0] * s % P, a[1] * s % P)


def f2_inv(a):
    t = fp_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * t % P, -a[1] * t % P)


def f2_pow(a, e):
    r = F2_ONE
    for bit in bin(e)[2:]:
        r = f2_sqr(r)
        if bit == '1':
            r = f2_mul(r, a)
    return r


def f2_is_square(a):     # fp2.rs:441-452 (norm is a square in Fp)
    return fp_is_square((a[0] * a[0] + a[1] * a[1]) % P)


def f2_sgn0(a):          # fp2.rs:95-99
    return (a[0] & 1) | ((a[0] == 0) & (a[1] & 1))


def f2_sqrt(a):
    """Any square root, or None.  (Sign is fixed by the caller through sgn0.)"""
    if a == F2_ZERO:
        return F2_ZERO
    a1 = f2_pow(a, (P - 3) // 4)
    alpha = f2_mul(f2_sqr(a1), a)
    x0 = f2_mul(a1, a)
    if alpha == (P - 1, 0):
        r = (-x0[1] % P, x0[0])
    else:
        r = f2_mul(f2_pow(f2_add(alpha, F2_ONE), (P - 1) // 2), x0)
    return r if f2_sqr(r) == a else None


# ---------------------------------------------------------------- dense Fp12 = Fp2[w]/(w^6 - xi)
# element = list of 6 Fp2 coefficients of w^0..w^5.  Tower <-> w-basis:
#   w^0=c0.c0  w^1=c1.c0  w^2=c0.c1  w^3=c1.c1  w^4=c0.c2  w^5=c1.c2
F12_ONE = [F2_ONE] + [F2_ZERO] * 5
TOWER_ORDER = (0, 2, 4, 1, 3, 5)     # w-index of c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2


def f12_mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        if a[i] == F2_ZERO:
            continue
        for j in range(6):
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    return [f2_add(t[k], f2_mul(t[k + 6], XI)) if k < 5 else t[k] for k in range(6)]


def f12_sqr(a): return f12_mul(a, a)


def f12_pow(a, e):
    r = list(F12_ONE)
    for bit in bin(e)[2:]:
        r = f12_sqr(r)
        if bit == '1':
            r = f12_mul(r, a)
    return r


def f12_conj(a):     # the p^6 Frobenius: w -> -w
    return [a[i] if i % 2 == 0 else f2_neg(a[i]) for i in range(6)]


# gamma[k][i] = xi^(i*(p^k-1)/6): Frobenius^k acts on coefficient i as conj^k(.) * gamma[k][i]
GAMMA = {k: [f2_pow(XI, i * (P**k - 1) // 6) for i in range(6)] for k in (1, 2, 3)}


def f12_frob(a, k=1):
    out = []
    for i in range(6):
        c = a[i]
        if k % 2 == 1:
            c = f2_conj(c)
        out.append(f2_mul(c, GAMMA[k][i]))
    return out


def f12_inv(a):
    # a^-1 = a^(p^12 - 2); slow but independent of any tower formula
    # use norm trick instead: a * conj-products ... keep it simple via Fermat in Fp12
    return f12_pow(a, P**12 - 2)


def f12_to_bytes(a):     # Gt::to_repr, pairings.rs:499-514
    return b"".join(a[i][0].to_bytes(32, "big") + a[i][1].to_bytes(32, "big") for i in TOWER_ORDER)


def f12_from_bytes(b):
    assert len(b) == 384
    a = [None] * 6
    for slot, i in enumerate(TOWER_ORDER):
        c0 = int.from_bytes(b[64 * slot:64 * slot + 32], "big")
        c1 = int.from_bytes(b[64 * slot + 32:64 * slot + 64], "big")
        if c0 >= P or c1 >= P:
            return None
        a[i] = (c0, c1)
    return a


# ---------------------------------------------------------------- curves (affine, None = identity)
B1 = 3
B2 = f2_mul((3, 0), f2_inv(XI))                     # 3/(9+u), fp2.rs:335-348
G1_GEN = (1, 2)
G2_GEN = ((0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed,
           0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2),
          (0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa,
           0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b))


def g1_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B1) % P == 0


def g1_neg(pt): return None if pt is None else (pt[0], -pt[1] % P)


def g1_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * fp_inv(2 * a[1]) % P
    else:
        lam = (b[1] - a[1]) * fp_inv(b[0] - a[0]) % P
    x3 = (lam * lam - a[0] - b[0]) % P
    return (x3, (lam * (a[0] - x3) - a[1]) % P)


def g1_mul(pt, k):
    acc = None
    for bit in bin(k)[2:] if k else '':
        acc = g1_add(acc, acc)
        if bit == '1':
            acc = g1_add(acc, pt)
    return acc


def g2_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B2)) == F2_ZERO


def g2_neg(pt): return None if pt is None else (pt[0], f2_neg(pt[1]))


def g2_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if f2_add(a[1], b[1]) == F2_ZERO:
            return None
        lam = f2_mul(f2_muls(f2_sqr(a[0]), 3), f2_inv(f2_muls(a[1], 2)))
    else:
        lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), a[0]), b[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(a[0], x3)), a[1]))


def g2_mul(pt, k):
    acc = None
    for bit in bin(k)[2:] if k else '':
        acc = g2_add(acc, acc)
        if bit == '1':
            acc = g2_add(acc, pt)
    return acc


PSI_X = f2_pow(XI, (P - 1) // 3)     # g2.rs:939-942
PSI_Y = f2_pow(XI, (P - 1) // 2)     # g2.rs:944-947


def g2_psi(pt):                      # g2.rs:938-954
    if pt is None:
        return None
    return (f2_mul(f2_conj(pt[0]), PSI_X), f2_mul(f2_conj(pt[1]), PSI_Y))


def g2_clear_cofactor(pt):           # g2.rs:685-693
    p0 = g2_mul(pt, X)
    p1 = g2_psi(g2_mul(p0, 3))
    p2 = g2_psi(g2_psi(p0))
    p3 = g2_psi(g2_psi(g2_psi(pt)))
    return g2_add(g2_add(p0, p1), g2_add(p2, p3))


def g2_in_subgroup_slow(pt):         # g2.rs:733-736  ([r]P == O)
    return g2_mul(pt, R) is None


def g2_in_subgroup_fast(pt):
    """[x+1]P + psi([x]P) + psi^2([x]P) == psi^3([2x]P)   (the test the HIP kernel uses)."""
    if pt is None:
        return True
    xp = g2_mul(pt, X)
    lhs = g2_add(g2_add(xp, pt), g2_add(g2_psi(xp), g2_psi(g2_psi(xp))))
    rhs = g2_psi(g2_psi(g2_psi(g2_add(xp, xp))))
    return lhs == rhs


# ---------------------------------------------------------------- codecs
def g1_to_bytes(pt):                 # g1.rs:297-302 ; identity = (0, 1) (g1.rs:265-271)
    if pt is None:
        return (0).to_bytes(32, "big") + (1).to_bytes(32, "big")
    return pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def g1_from_bytes(b):
    """Returns (ok, point).  Strict: coordinates must be canonical (< p); x == 0 -> identity
    (g1.rs:352-353).  No on-curve check here (the caller validates, E10)."""
    x = int.from_bytes(b[:32], "big"); y = int.from_bytes(b[32:64], "big")
    if x >= P:
        return False, None
    if x == 0:
        return True, None
    if y >= P:
        return False, None
    return True, (x, y)


def g2_to_bytes(pt):                 # g2.rs:292-300: x.c1 || x.c0 || y.c1 || y.c0
    if pt is None:
        pt = (F2_ZERO, F2_ONE)
    (x, y) = pt
    return b"".join(v.to_bytes(32, "big") for v in (x[1], x[0], y[1], y[0]))


def g2_from_bytes(b):
    v = [int.from_bytes(b[32 * i:32 * i + 32], "big") for i in range(4)]
    if v[0] >= P or v[1] >= P:
        return False, None
    if v[0] == 0 and v[1] == 0:
        return True, None
    if v[2] >= P or v[3] >= P:
        return False, None
    return True, ((v[1], v[0]), (v[3], v[2]))


# ---------------------------------------------------------------- hash to curve
def expand_message_xmd(msg, dst, n):         # RFC 9380 5.3.1 with SHA-256
    if len(dst) > 255:
        dst = hashlib.sha256(b"H2C-OVERSIZE-DST-" + dst).digest()
    ell = (n + 31) // 32
    assert ell <= 255
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(b"\x00" * 64 + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
    bi = hashlib.sha256(b0 + b"\x01" + dst_prime).digest()
    out = bi
    for i in range(2, ell + 1):
        bi = hashlib.sha256(bytes(a ^ b for a, b in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
        out += bi
    return out[:n]


def hash_to_fp(msg, dst, count):             # fp.rs:433-458
    okm = expand_message_xmd(msg, dst, 48 * count)
    return [int.from_bytes(okm[48 * i:48 * i + 48], "big") % P for i in range(count)]


def hash_to_fp2(msg, dst, count):            # fp2.rs:454-487
    okm = expand_message_xmd(msg, dst, 96 * count)
    return [(int.from_bytes(okm[96 * i:96 * i + 48], "big") % P,
             int.from_bytes(okm[96 * i + 48:96 * i + 96], "big") % P) for i in range(count)]


SVDW1_C1 = 4
SVDW1_C2 = (P - 1) // 2
SVDW1_C3 = 0x16789af3a83522eb353c98fc6b36d713d5d8d1cc5dffffffa
SVDW1_C4 = 0x10216f7ba065e00de81ac1e7808072c9dd2b2385cd7b438469602eb24829a9bd
assert SVDW1_C3 * SVDW1_C3 % P == -12 % P and SVDW1_C3 & 1 == 0
assert SVDW1_C4 * 3 % P == -16 % P


def svdw_g1(u):                              # fp.rs:292-370
    tv1 = u * u % P * SVDW1_C1 % P
    tv2 = (1 + tv1) % P
    tv1 = (1 - tv1) % P
    tv3 = fp_inv(tv1 * tv2 % P)
    tv4 = u * tv1 % P * tv3 % P * SVDW1_C3 % P
    x1 = (SVDW1_C2 - tv4) % P
    gx1 = (x1 * x1 % P * x1 + B1) % P
    x2 = (SVDW1_C2 + tv4) % P
    gx2 = (x2 * x2 % P * x2 + B1) % P
    x3 = tv2 * tv2 % P * tv3 % P
    x3 = (x3 * x3 % P * SVDW1_C4 + 1) % P
    e1 = fp_is_square(gx1)
    x = x1 if e1 else x3
    if fp_is_square(gx2) and not e1:
        x = x2
    gx = (x * x % P * x + B1) % P
    y = fp_sqrt(gx)
    assert y is not None
    if fp_sgn0(u) != fp_sgn0(y):
        y = -y % P
    return (x, y)


SVDW2_C1 = f2_add(F2_ONE, B2)                # g(Z), Z = 1
SVDW2_C2 = ((P - 1) // 2, 0)
SVDW2_C3 = (0x29fd332ab7260112b801fa95b21af64e2e6da55f90a3e510fcbe57377b5ca1ec,
            0x303d1eff1426764bf8408aee24ba0b865e76f77b1267a846b1e9154d01565034)
SVDW2_C4 = (0x17365bbe63b1d2078632fe0eb2ac5a41b4e6a9c08b98676721010b008d4eaf99,
            0x0f57ffe5fc79e19cd689d7aa4209cad8fe164d7f4694786b388732a995d03755)
assert f2_sqr(SVDW2_C3) == f2_neg(f2_muls(SVDW2_C1, 3)) and f2_sgn0(SVDW2_C3) == 0
assert f2_muls(SVDW2_C4, 3) == f2_neg(f2_muls(SVDW2_C1, 4))


def svdw_g2(u):                              # fp2.rs:224-286
    tv1 = f2_mul(f2_sqr(u), SVDW2_C1)
    tv2 = f2_add(F2_ONE, tv1)
    tv1 = f2_sub(F2_ONE, tv1)
    tv3 = f2_inv(f2_mul(tv1, tv2))
    tv4 = f2_mul(f2_mul(f2_mul(u, tv1), tv3), SVDW2_C3)
    x1 = f2_sub(SVDW2_C2, tv4)
    gx1 = f2_add(f2_mul(f2_sqr(x1), x1), B2)
    x2 = f2_add(SVDW2_C2, tv4)
    gx2 = f2_add(f2_mul(f2_sqr(x2), x2), B2)
    x3 = f2_mul(f2_sqr(tv2), tv3)
    x3 = f2_add(f2_mul(f2_sqr(x3), SVDW2_C4), F2_ONE)
    e1 = f2_is_square(gx1)
    x = x1 if e1 else x3
    if f2_is_square(gx2) and not e1:
        x = x2
    gx = f2_add(f2_mul(f2_sqr(x), x), B2)
    y = f2_sqrt(gx)
    assert y is not None
    if f2_sgn0(u) != f2_sgn0(y):
        y = f2_neg(y)
    return (x, y)


def hash_to_g1(msg, dst):                    # g1.rs:910-919
    u0, u1 = hash_to_fp(msg, dst, 2)
    return g1_add(svdw_g1(u0), svdw_g1(u1))


def encode_to_g1(msg, dst):                  # g1.rs:922-928
    return svdw_g1(hash_to_fp(msg, dst, 1)[0])


def hash_to_g2(msg, dst):                    # g2.rs:919-927
    u0, u1 = hash_to_fp2(msg, dst, 2)
    return g2_clear_cofactor(g2_add(svdw_g2(u0), svdw_g2(u1)))


def encode_to_g2(msg, dst):                  # g2.rs:930-936
    return g2_clear_cofactor(svdw_g2(hash_to_fp2(msg, dst, 1)[0]))


# ---------------------------------------------------------------- pairing (affine lines, dense Fp12)
def _line(T, Q2, Pt):
    """Line through T and Q2 (tangent if equal) on the twist, evaluated at the G1 point Pt, as
    a dense Fp12:  yP - lam*xP*w + (lam*xT - yT)*w^3   (SURVEY.md App. A, D-type untwist)."""
    if T[0] == Q2[0] and T[1] == Q2[1]:
        lam = f2_mul(f2_muls(f2_sqr(T[0]), 3), f2_inv(f2_muls(T[1], 2)))
    else:
        lam = f2_mul(f2_sub(Q2[1], T[1]), f2_inv(f2_sub(Q2[0], T[0])))
    l = [F2_ZERO] * 6
    l[0] = (Pt[1], 0)
    l[1] = f2_neg(f2_muls(lam, Pt[0]))
    l[3] = f2_sub(f2_mul(lam, T[0]), T[1])
    return l


def miller_loop(Pt, Q):
    """Optimal ate f_{6x+2,Q}(P) * l_{T,pi(Q)} * l_{T+pi(Q),-pi^2(Q)}, plain binary loop.
    (Its value differs from the C oracle's by Fp6-subfield factors only; compare after final exp.)"""
    if Pt is None or Q is None:
        return list(F12_ONE)
    f = list(F12_ONE)
    T = Q
    for bit in bin(ATE_LOOP)[3:]:
        f = f12_mul(f12_sqr(f), _line(T, T, Pt))
        T = g2_add(T, T)
        if bit == '1':
            f = f12_mul(f, _line(T, Q, Pt))
            T = g2_add(T, Q)
    Q1 = g2_psi(Q)
    Q2 = g2_neg(g2_psi(Q1))
    f = f12_mul(f, _line(T, Q1, Pt))
    T = g2_add(T, Q1)
    f = f12_mul(f, _line(T, Q2, Pt))
    return f


def final_exponentiation_slow(f):
    return f12_pow(f, FINAL_EXP)


def final_exponentiation(f):
    """Structured: easy part by Frobenius, hard part = lambda0 + lambda1 p + lambda2 p^2 + lambda3 p^3."""
    t = f12_mul(f12_conj(f), f12_inv(f))                 # f^(p^6-1)
    t = f12_mul(f12_frob(t, 2), t)                       # ^(p^2+1)
    out = f12_pow(t, LAMBDA[0])
    out = f12_mul(out, f12_frob(f12_pow(t, LAMBDA[1]), 1))
    out = f12_mul(out, f12_frob(f12_pow(t, LAMBDA[2]), 2))
    out = f12_mul(out, f12_frob(f12_pow(t, LAMBDA[3]), 3))
    return out


def pairing(Pt, Q):
    return final_exponentiation(miller_loop(Pt, Q))


def multi_pairing(pairs):
    f = list(F12_ONE)
    for Pt, Q in pairs:
        f = f12_mul(f, miller_loop(Pt, Q))
    return final_exponentiation(f)


# ---------------------------------------------------------------- BLS (min-sig: sig in G1, pk in G2)
DEFAULT_DST = b"BLS_SIG_BN254G1_XMD:SHA-256_SVDW_RO_NUL_"


def sk_to_pk(sk): return g2_mul(G2_GEN, sk % R)
def sign(sk, msg, dst=DEFAULT_DST): return g1_mul(hash_to_g1(msg, dst), sk % R)


POP_DST = b"BLS_POP_BN254G1_XMD:SHA-256_SVDW_RO_POP_"
KEYGEN_SALT = b"BLS-SIG-KEYGEN-SALT-"          # helpers.rs:3


def hash_to_scalar(msg, dst):        # Scalar::hash, scalar.rs:554-563, with the reduction from_okm is meant to do
    return int.from_bytes(expand_message_xmd(msg, dst, 48), "big") % R


def keygen(ikm, key_info=b""):       # draft-irtf-cfrg-bls-signature-05 section 2.3, HKDF from the stdlib hmac
    import hmac
    assert len(ikm) >= 32
    salt, L = KEYGEN_SALT, 48
    while True:
        salt = hashlib.sha256(salt).digest()
        prk = hmac.new(salt, ikm + b"\x00", hashlib.sha256).digest()
        info = key_info + L.to_bytes(2, "big")
        t1 = hmac.new(prk, info + b"\x01", hashlib.sha256).digest()
        t2 = hmac.new(prk, t1 + info + b"\x02", hashlib.sha256).digest()
        sk = int.from_bytes((t1 + t2)[:L], "big") % R
        if sk:
            return sk


def pop_prove(sk, dst=POP_DST): return sign(sk, g2_to_bytes(sk_to_pk(sk)), dst)
def pop_verify_bytes(pk_b, proof_b, dst=POP_DST): return verify_bytes(pk_b, pk_b, proof_b, dst)


def verify_bytes(pk_b, msg, sig_b, dst=DEFAULT_DST):
    ok, sig = g1_from_bytes(sig_b)
    if not ok or sig is None or not g1_on_curve(sig):
        return False
    ok, pk = g2_from_bytes(pk_b)
    if not ok or pk is None or not g2_on_curve(pk) or not g2_in_subgroup_slow(pk):
        return False
    h = hash_to_g1(msg, dst)
    return multi_pairing([(sig, g2_neg(G2_GEN)), (h, pk)]) == F12_ONE


def aggregate_verify_bytes(pk_bs, msgs, sig_b, dst=DEFAULT_DST):
    ok, sig = g1_from_bytes(sig_b)
    if not ok or sig is None or not g1_on_curve(sig) or len(pk_bs) == 0:
        return False
    pairs = [(sig, g2_neg(G2_GEN))]
    for pk_b, m in zip(pk_bs, msgs):
        ok, pk = g2_from_bytes(pk_b)
        if not ok or pk is None or not g2_on_curve(pk) or not g2_in_subgroup_slow(pk):
            return False
        pairs.append((hash_to_g1(m, dst), pk))
    return multi_pairing(pairs) == F12_ONE


def lagrange_at_zero(ids):
    """lambda_i = prod_{j != i} x_j / (x_j - x_i)  in Fr."""
    out = []
    for i, xi in enumerate(ids):
        num, den = 1, 1
        for j, xj in enumerate(ids):
            if i != j:
                num = num * xj % R
                den = den * (xj - xi) % R
        out.append(num * pow(den, R - 2, R) % R)
    return out


def threshold_combine(ids, sigs):
    acc = None
    for lam, s in zip(lagrange_at_zero(ids), sigs):
        acc = g1_add(acc, g1_mul(s, lam))
    return acc
