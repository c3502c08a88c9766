#!/usr/bin/env python3
"""Generates bls-bn254_amd/csrc/bn254_consts.h: every curve/field constant the HIP kernels use,
in the device representation (radix-2^29, 9 signed limbs, Montgomery form with R = 2^261).

Self-contained integer arithmetic (it must not import anything under oracle/).  Constants follow the
reference's definitions (read as a specification):
  p, r, x            fp.rs:29-33, scalar.rs:29-33, inner_types.rs:28
  b = 3, b' = 3/(9+u)  g1.rs (y^2 = x^3 + 3), fp2.rs:335-348
  G2 generator       fp2.rs:305-333
  psi coefficients   g2.rs:939-947  (= xi^((p-1)/3), xi^((p-1)/2))
  SVDW constants     fp.rs:298-317 (G1), fp2.rs:230-248 (G2)
  line formulas      pairings.rs:901-962 (Jacobian doubling / mixed addition step)
The tower non-residue is xi = 9+u (SURVEY.md errata E1); Frobenius coefficients are
xi^(i(p^k-1)/6) (E3); the Miller loop runs over NAF(6x+2) (E4).

Usage: python gen_constants.py [out.h]      (deterministic; the output is committed)
"""
import os
import sys

P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
RR = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
X = 0x44e992b44a6909f1
RB = 29
NL = 9
MASK = (1 << RB) - 1
MONT_R = 1 << (RB * NL)


def limbs(v):
    assert 0 <= v < MONT_R
    return [(v >> (RB * i)) & MASK for i in range(NL)]


def mont(v):
    return limbs(v % P * MONT_R % P)


def inv(a):
    return pow(a, P - 2, P)


# --- Fp2 helpers (tuples), xi = 9+u
def f2mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2sqr(a): return f2mul(a, a)
def f2add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2neg(a): return (-a[0] % P, -a[1] % P)
def f2conj(a): return (a[0], -a[1] % P)
def f2dbl(a): return f2add(a, a)


def f2inv(a):
    t = inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * t % P, -a[1] * t % P)


def f2pow(a, e):
    r = (1, 0)
    for bit in bin(e)[2:]:
        r = f2sqr(r)
        if bit == '1':
            r = f2mul(r, a)
    return r


def f2sgn0(a):
    return (a[0] & 1) | ((a[0] == 0) & (a[1] & 1))


def f2sqrt(a):
    a1 = f2pow(a, (P - 3) // 4)
    alpha = f2mul(f2sqr(a1), a)
    x0 = f2mul(a1, a)
    if alpha == (P - 1, 0):
        r = (-x0[1] % P, x0[0])
    else:
        r = f2mul(f2pow(f2add(alpha, (1, 0)), (P - 1) // 2), x0)
    assert f2sqr(r) == a
    return r


XI = (9, 1)
B2 = f2mul((3, 0), f2inv(XI))
G2X = (0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed,
       0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2)
G2Y = (0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa,
       0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b)
assert f2sub(f2sqr(G2Y), f2add(f2mul(f2sqr(G2X), G2X), B2)) == (0, 0)


def naf(n):
    out = []
    while n:
        d = 0
        if n & 1:
            d = 2 - (n & 3)
            n -= d
        out.append(d)
        n >>= 1
    return out


# --- Miller-loop line steps on a Jacobian T (pairings.rs:901-962)
def doubling_step(T):
    x, y, z = T
    tmp0 = f2sqr(x); tmp1 = f2sqr(y); tmp2 = f2sqr(tmp1)
    tmp3 = f2dbl(f2sub(f2sub(f2sqr(f2add(tmp1, x)), tmp0), tmp2))
    tmp4 = f2add(f2dbl(tmp0), tmp0)
    tmp6 = f2add(x, tmp4)
    tmp5 = f2sqr(tmp4)
    zsq = f2sqr(z)
    nx = f2sub(f2sub(tmp5, tmp3), tmp3)
    nz = f2sub(f2sub(f2sqr(f2add(z, y)), tmp1), zsq)
    ny = f2sub(f2mul(f2sub(tmp3, nx), tmp4), f2dbl(f2dbl(f2dbl(tmp2))))
    c1 = f2neg(f2dbl(f2mul(tmp4, zsq)))
    c2 = f2sub(f2sub(f2sub(f2sqr(tmp6), tmp0), tmp5), f2dbl(f2dbl(tmp1)))
    c0 = f2dbl(f2mul(nz, zsq))
    return (nx, ny, nz), (c0, c1, c2)


def addition_step(T, Q):
    x, y, z = T
    qx, qy = Q
    zsq = f2sqr(z); ysq = f2sqr(qy)
    t0 = f2mul(zsq, qx)
    t1 = f2mul(f2sub(f2sub(f2sqr(f2add(qy, z)), ysq), zsq), zsq)
    t2 = f2sub(t0, x)
    t3 = f2sqr(t2)
    t4 = f2dbl(f2dbl(t3))
    t5 = f2mul(t4, t2)
    t6 = f2sub(f2sub(t1, y), y)
    t9 = f2mul(t6, qx)
    t7 = f2mul(t4, x)
    nx = f2sub(f2sub(f2sub(f2sqr(t6), t5), t7), t7)
    nz = f2sub(f2sub(f2sqr(f2add(z, t2)), zsq), t3)
    t10 = f2add(qy, nz)
    t8 = f2mul(f2sub(t7, nx), t6)
    ny = f2sub(t8, f2dbl(f2mul(y, t5)))
    t10 = f2sub(f2sub(f2sqr(t10), ysq), f2sqr(nz))
    c2 = f2sub(f2dbl(t9), t10)
    c0 = f2dbl(nz)
    c1 = f2dbl(f2neg(t6))
    return (nx, ny, nz), (c0, c1, c2)


def line_table(Q, gamma1):
    """All line coefficient triples of the optimal ate loop for a fixed Q, in evaluation order."""
    digits = naf(6 * X + 2)
    T = (Q[0], Q[1], (1, 0))
    out = []
    for j in range(len(digits) - 2, -1, -1):
        T, c = doubling_step(T); out.append(c)
        if digits[j]:
            q = Q if digits[j] > 0 else (Q[0], f2neg(Q[1]))
            T, c = addition_step(T, q); out.append(c)
    q1 = (f2mul(f2conj(Q[0]), gamma1[2]), f2mul(f2conj(Q[1]), gamma1[3]))
    q2 = (f2mul(f2conj(q1[0]), gamma1[2]), f2neg(f2mul(f2conj(q1[1]), gamma1[3])))
    T, c = addition_step(T, q1); out.append(c)
    T, c = addition_step(T, q2); out.append(c)
    return out


def fmt_limbs(ls):
    return "{" + ", ".join("%d" % v for v in ls) + "}"


def fp_c(name, v):
    return "BN_CONST int32_t %s[9] = %s;\n" % (name, fmt_limbs(mont(v)))


def fp2_c(name, v):
    return "BN_CONST int32_t %s[18] = %s;\n" % (name, fmt_limbs(mont(v[0]) + mont(v[1])))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "..", "csrc", "bn254_consts.h")
    s = []
    s.append("// GENERATED by bls-bn254_amd/tools/gen_constants.py -- do not edit.\n")
    s.append("// Device representation: radix 2^29, 9 limbs, Montgomery form with R = 2^261 (see fp29.h).\n")
    s.append("#pragma once\n#include <stdint.h>\n\n")
    s.append("#ifndef BN_CONST\n#define BN_CONST static constexpr\n#endif\n\n")
    s.append("namespace bnc {\n")
    s.append("BN_CONST int32_t P[9] = %s;      // modulus, plain limbs\n" % fmt_limbs(limbs(P)))
    pinv = (-pow(P, -1, 1 << RB)) % (1 << RB)
    s.append("BN_CONST int32_t PINV = %d;      // -p^-1 mod 2^29\n" % pinv)
    s.append(fp_c("ONE", 1))
    s.append("BN_CONST int32_t R2[9] = %s;     // R^2 mod p (plain limbs): to-Montgomery multiplier\n" % fmt_limbs(limbs(MONT_R * MONT_R % P)))
    s.append("BN_CONST int32_t R3[9] = %s;     // R^3 mod p: multiplier for the 2^261 part of a 48-byte okm\n" % fmt_limbs(limbs(pow(MONT_R, 3, P))))
    s.append("BN_CONST int32_t LC_QINV = %d;  // floor(2^284 / p): q = (top * LC_QINV) >> 52 ~ floor(value / p) in fp_lc\n" % ((1 << 284) // P))
    s.append(fp_c("THREE", 3))
    s.append(fp_c("INV2", inv(2)))
    s.append(fp2_c("B2", B2))
    b3 = (B2[0] * 3 % P, B2[1] * 3 % P)
    s.append(fp2_c("B2_3", b3))
    # Frobenius coefficients gamma_k[i] = xi^(i (p^k-1)/6), i = 1..5
    g1 = [f2pow(XI, i * (P - 1) // 6) for i in range(6)]
    g2 = [f2mul(g, f2conj(g)) for g in g1]
    g3 = [f2mul(a, b) for a, b in zip(g1, g2)]
    for i in range(6):
        assert g2[i][1] == 0
        assert g3[i] == f2pow(XI, i * (P**3 - 1) // 6) and g2[i] == f2pow(XI, i * (P**2 - 1) // 6)
    s.append("BN_CONST int32_t GAMMA1[5][18] = {%s};\n" % ", ".join(fmt_limbs(mont(g[0]) + mont(g[1])) for g in g1[1:]))
    s.append("BN_CONST int32_t GAMMA2[5][9] = {%s};   // in Fp\n" % ", ".join(fmt_limbs(mont(g[0])) for g in g2[1:]))
    s.append("BN_CONST int32_t GAMMA3[5][18] = {%s};\n" % ", ".join(fmt_limbs(mont(g[0]) + mont(g[1])) for g in g3[1:]))
    # SVDW G1 (fp.rs:298-317)
    c3 = pow(-12 % P, (P + 1) // 4, P)
    assert c3 * c3 % P == -12 % P
    if c3 & 1:
        c3 = P - c3
    s.append(fp_c("SVDW1_C2", (P - 1) // 2))
    s.append(fp_c("SVDW1_C3", c3))
    s.append(fp_c("SVDW1_C4", (-16 * inv(3)) % P))
    # SVDW G2 (fp2.rs:230-248)
    c1 = f2add((1, 0), B2)
    c3_2 = f2sqrt(f2neg((c1[0] * 3 % P, c1[1] * 3 % P)))
    if f2sgn0(c3_2):
        c3_2 = f2neg(c3_2)
    c4_2 = f2neg((c1[0] * 4 * inv(3) % P, c1[1] * 4 * inv(3) % P))
    s.append(fp2_c("SVDW2_C1", c1))
    s.append(fp2_c("SVDW2_C3", c3_2))
    s.append(fp2_c("SVDW2_C4", c4_2))
    # generators
    s.append(fp2_c("G2_GEN_X", G2X))
    s.append(fp2_c("G2_GEN_Y", G2Y))
    # loop digits
    dg = naf(6 * X + 2)
    s.append("BN_CONST int ATE_NAF_LEN = %d;\n" % len(dg))
    s.append("BN_CONST int8_t ATE_NAF[%d] = {%s};   // NAF(6x+2), least significant first\n" % (len(dg), ", ".join(str(d) for d in dg)))
    s.append("BN_CONST uint64_t BN_X = 0x%xULL;\n" % X)
    # exponents as 64-bit words (little endian)
    def words(v):
        return "{" + ", ".join("0x%016xULL" % ((v >> (64 * i)) & (2**64 - 1)) for i in range(4)) + "}"
    s.append("BN_CONST uint64_t EXP_PM2[4] = %s;       // p-2\n" % words(P - 2))
    s.append("BN_CONST uint64_t EXP_PM1_2[4] = %s;     // (p-1)/2\n" % words((P - 1) // 2))
    s.append("BN_CONST uint64_t EXP_PP1_4[4] = %s;     // (p+1)/4\n" % words((P + 1) // 4))
    s.append("BN_CONST uint64_t EXP_PM3_4[4] = %s;     // (p-3)/4\n" % words((P - 3) // 4))
    # scalar field Fr (threshold path: Lagrange coefficients), same radix / Montgomery radius
    s.append("BN_CONST int32_t FR_MOD[9] = %s;   // r, plain limbs\n" % fmt_limbs(limbs(RR)))
    s.append("BN_CONST int32_t FR_PINV = %d;     // -r^-1 mod 2^29\n" % ((-pow(RR, -1, 1 << RB)) % (1 << RB)))
    s.append("BN_CONST int32_t FR_ONE[9] = %s;   // R mod r\n" % fmt_limbs(limbs(MONT_R % RR)))
    s.append("BN_CONST int32_t FR_R2[9] = %s;    // R^2 mod r\n" % fmt_limbs(limbs(MONT_R * MONT_R % RR)))
    s.append("BN_CONST int32_t FR_R3[9] = %s;    // R^3 mod r\n" % fmt_limbs(limbs(pow(MONT_R, 3, RR))))
    s.append("BN_CONST uint64_t EXP_RM2[4] = %s;       // r-2\n" % words(RR - 2))
    # GLV endomorphism of G1 (threshold MSM): phi(x, y) = (beta x, y) = [lambda](x, y); short lattice basis of
    # {(a, b): a + b lambda = 0 mod r} by the extended Euclidean algorithm; rounding constants scaled by 2^256
    glv = glv_constants()
    s.append(fp_c("GLV_BETA", glv["beta"]))
    w8 = lambda v: "{" + ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xffffffff) for i in range(8)) + "}"
    for name in ("A1", "B1", "A2", "B2"):
        s.append("BN_CONST uint32_t GLV_%s[8] = %s;   // basis entry %s as a 256-bit two's-complement number (%d)\n" % (name, w8(glv[name.lower()] % (1 << 256)), name.lower(), glv[name.lower()]))
    for name in ("G1", "G2"):
        s.append("BN_CONST uint32_t GLV_%s[8] = %s;   // |%s| = floor(2^256 |%s| / r): c = (k * G) >> 256\n" % (name, w8(glv[name.lower()]), "c1" if name == "G1" else "c2", "b2" if name == "G1" else "b1"))
    s.append("BN_CONST bool GLV_C1_NEG = %s, GLV_C2_NEG = %s;   // signs of c1 = round(b2 k / r), c2 = round(-b1 k / r)\n" % ("true" if glv["c1_neg"] else "false", "true" if glv["c2_neg"] else "false"))
    s.append("BN_CONST uint64_t GLV_LAMBDA[4] = %s;   // the eigenvalue (documentation / tests)\n" % words(glv["lam"]))
    s.append("}  // namespace bnc\n\n")
    # fixed-Q line table for -G2gen (the verify equation pairs the signature with -G2gen)
    negG2 = (G2X, f2neg(G2Y))
    tab = line_table(negG2, g1)
    s.append("#define BN_ATE_NAF_INIT {%s}\n" % ", ".join(str(d) for d in dg))
    s.append("// Line coefficients (c0, c1, c2) of the optimal ate loop for the fixed point Q = -G2gen, in\n")
    s.append("// evaluation order (doubling step, then the addition step if the NAF digit is non-zero, ...,\n")
    s.append("// then the two Frobenius additions).  Each entry: 3 Fp2 = 54 limbs.\n")
    s.append("#define BN_NEG_G2_LINES %d\n" % len(tab))
    s.append("#ifdef BN_WANT_LINE_TABLE\n")
    s.append("BN_LINE_TABLE_QUAL int32_t BN_NEG_G2_LINE_TABLE[%d][54] = {\n" % len(tab))
    for c in tab:
        ls = []
        for v in c:
            ls += mont(v[0]) + mont(v[1])
        s.append(" " + fmt_limbs(ls) + ",\n")
    s.append("};\n#endif\n")
    with open(out, "w") as f:
        f.write("".join(s))
    print("wrote", os.path.normpath(out), "(%d line entries)" % len(tab))


def g1_mul_affine(pt, k):
    """affine double-and-add on y^2 = x^3 + 3 (generator-time check of the endomorphism only)"""
    def add(a, b):
        if a is None: return b
        if b is None: return a
        if a[0] == b[0]:
            if (a[1] + b[1]) % P == 0: return None
            l = 3 * a[0] * a[0] * inv(2 * a[1]) % P
        else:
            l = (b[1] - a[1]) * inv((b[0] - a[0]) % P) % P
        x = (l * l - a[0] - b[0]) % P
        return (x, (l * (a[0] - x) - a[1]) % P)
    acc = None
    for bit in bin(k)[2:]:
        acc = add(acc, acc)
        if bit == '1':
            acc = add(acc, pt)
    return acc


def glv_split_words(k, c):
    """The device algorithm (glv.h) restated on Python integers, word-exact: returns (k1, k2) as signed integers."""
    M = 1 << 256
    c1 = (k * c["g1"]) >> 256
    c2 = (k * c["g2"]) >> 256
    C1 = (-c1 if c["c1_neg"] else c1) % M
    C2 = (-c2 if c["c2_neg"] else c2) % M
    k1 = (k - C1 * (c["a1"] % M) - C2 * (c["a2"] % M)) % M
    k2 = (-(C1 * (c["b1"] % M) + C2 * (c["b2"] % M))) % M
    sgn = lambda v: v - M if v >> 255 else v
    return sgn(k1), sgn(k2)


def glv_constants():
    import math
    import random
    # a primitive cube root of unity mod r and mod p; pick the pair with phi = [lambda]
    def cube_root_of_unity(m):
        g = 2
        while True:
            c = pow(g, (m - 1) // 3, m)
            if c != 1:
                return c
            g += 1
    lam = cube_root_of_unity(RR)
    beta = cube_root_of_unity(P)
    G = (1, 2)
    Q = g1_mul_affine(G, lam)
    if Q != (beta * G[0] % P, G[1]):
        beta = beta * beta % P
    assert Q == (beta * G[0] % P, G[1])
    # extended Euclid on (r, lambda): rows (rem, t) with rem = s r + t lambda
    rows = [(RR, 0), (lam, 1)]
    sq = math.isqrt(RR)
    while rows[-1][0] >= sq:
        q = rows[-2][0] // rows[-1][0]
        rows.append((rows[-2][0] - q * rows[-1][0], rows[-2][1] - q * rows[-1][1]))
    q = rows[-2][0] // rows[-1][0]
    nxt = (rows[-2][0] - q * rows[-1][0], rows[-2][1] - q * rows[-1][1])
    v1 = (rows[-1][0], -rows[-1][1])
    cands = [(rows[-2][0], -rows[-2][1]), (nxt[0], -nxt[1])]
    v2 = min(cands, key=lambda v: v[0] * v[0] + v[1] * v[1])
    a1, b1 = v1
    a2, b2 = v2
    det = a1 * b2 - a2 * b1
    assert abs(det) == RR and (a1 + b1 * lam) % RR == 0 and (a2 + b2 * lam) % RR == 0
    # c1 = round(b2 k / det), c2 = round(-b1 k / det); on the device: magnitude (k * g) >> 256, sign separate
    n1, n2 = b2 * (1 if det > 0 else -1), -b1 * (1 if det > 0 else -1)
    c = {"lam": lam, "beta": beta, "a1": a1, "b1": b1, "a2": a2, "b2": b2,
         "g1": (abs(n1) << 256) // RR, "g2": (abs(n2) << 256) // RR, "c1_neg": n1 < 0, "c2_neg": n2 < 0}
    # Analytic bound on the halves (the device code hard-codes 128-bit magnitudes: glv.h, k_msm_window): with exact rounding
    # |k1| <= (|a1| + |a2|) / 2, |k2| <= (|b1| + |b2|) / 2; each of the two truncated quotients is off by less than 2 (one
    # for the floor of g = 2^256 n / r, one for the floor of the product), i.e. less than 1.5 further basis vectors each.
    assert 2 * (abs(a1) + abs(a2)) < 1 << 128 and 2 * (abs(b1) + abs(b2)) < 1 << 128, "GLV halves may exceed 128 bits"
    # the truncated quotients are off by at most one: the halves must still fit 128 bits with room to spare
    rnd = random.Random(7)
    worst = 0
    for k in [0, 1, 2, RR - 1, RR - 2, lam, RR - lam, (RR - 1) // 2] + [rnd.randrange(RR) for _ in range(20000)]:
        k1, k2 = glv_split_words(k, c)
        assert (k1 + k2 * lam - k) % RR == 0
        worst = max(worst, abs(k1).bit_length(), abs(k2).bit_length())
    assert worst <= 127, worst
    return c


if __name__ == "__main__":
    main()
