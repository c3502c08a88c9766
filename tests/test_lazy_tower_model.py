"""CPU-only: the lazily reduced 10 x 26-bit Fp6 product measured by bench_micro/lazy_tower.hip is a
correct Fp6 product -- checked against Python big integers -- so that its GPU timing compares like with like.  The shipped
arithmetic (9 x 29-bit limbs, tower.h) is not touched by this experiment."""
import ctypes
import os
import random
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 1 << 260


def _lib(tmp_path):
    so = str(tmp_path / "liblazy_tower.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "bench_micro", "lazy_tower_host.cpp")])
    return ctypes.CDLL(so)


def limbs(x):
    return [(x >> (26 * k)) & ((1 << 26) - 1) for k in range(10)]


def val(l):
    return sum(int(v) << (26 * k) for k, v in enumerate(l))


def f2mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def xi(a): return ((9 * a[0] - a[1]) % P, (a[0] + 9 * a[1]) % P)


def f6mul(a, b):
    c0 = f2add(f2mul(a[0], b[0]), xi(f2add(f2mul(a[1], b[2]), f2mul(a[2], b[1]))))
    c1 = f2add(f2add(f2mul(a[0], b[1]), f2mul(a[1], b[0])), xi(f2mul(a[2], b[2])))
    c2 = f2add(f2add(f2mul(a[0], b[2]), f2mul(a[2], b[0])), f2mul(a[1], b[1]))
    return (c0, c1, c2)


def test_lazy_fp6_product_equals_big_integer_model(tmp_path):
    lib = _lib(tmp_path)
    src = open(os.path.join(ROOT, "bench_micro", "lazy_tower.h")).read()
    assert ", ".join("0x%07x" % x for x in limbs(P)) in src and "PINV26 = 0x%07x" % ((-pow(P, -1, 1 << 26)) % (1 << 26)) in src
    rnd = random.Random(5)
    rinv = pow(R, -1, P)
    for it in range(300):
        top = (1 << 256) if it % 2 else P               # lazily reduced operands (up to ~8 p) and canonical ones
        a = [[rnd.randrange(top) for _ in range(2)] for _ in range(3)]
        b = [[rnd.randrange(top) for _ in range(2)] for _ in range(3)]
        if it == 0:
            a = [[(1 << 256) - 1] * 2] * 3; b = [[(1 << 256) - 1] * 2] * 3      # every limb at its maximum: the column budget
        if it == 2:
            a = [[0, 0], [P - 1, 1], [0, 0]]
        A = (ctypes.c_int32 * 60)(*[v for c in a for x in c for v in limbs(x)])
        B = (ctypes.c_int32 * 60)(*[v for c in b for x in c for v in limbs(x)])
        out = (ctypes.c_int32 * 60)()
        lib.lz_fp6_mul(A, B, out)
        got = [[val(out[10 * (2 * c + j):10 * (2 * c + j) + 10]) % P for j in range(2)] for c in range(3)]
        want = f6mul([tuple(x % P for x in c) for c in a], [tuple(x % P for x in c) for c in b])
        assert got == [[w * rinv % P for w in c] for c in want], it
        for c in range(6):
            l = list(out[10 * c:10 * c + 10])
            assert all(0 <= v < (1 << 26) for v in l[:9]) and abs(l[9]) < (1 << 24), l      # limbs fit to be the next product's operand (the top limb carries the sign)
