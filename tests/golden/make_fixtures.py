#!/usr/bin/env python3
"""Transcribe the reference's own known-answer vectors into JSON fixtures (DATA only).

Run in the build container (needs /root/reference, which does not travel to the GPU box):
    python tests/golden/make_fixtures.py
Writes tests/golden/reference_kats.json.  The script reads the reference's Rust test modules
as *text* and extracts inputs/expected outputs; it copies no code.  Sources:
  g1.rs:981-1049   G1 encode (NU) vectors: msg, u, mapped point
  g1.rs:1052-1141  G1 hash   (RO) vectors: msg, u0, u1, Q0, Q1, P
  g2.rs:994-1013   G2 "bad" point (not on curve, not torsion free)
  g2.rs:1039-1225  G2 encode (NU) vectors (little-endian u64 words)
  g2.rs:1228-1313  G2 hash   (RO) vectors (big-endian hex)
  fp2.rs:305-348   G2 generator and twist coefficient b' (Montgomery words, R = 2^256)
  pairings.rs:387-479  Gt::generator() = the pairing golden vector (Montgomery words)
Montgomery-form constants are decoded here as words * 2^-256 mod p (fp.rs:464-466).
"""
import json
import os
import re
import sys

REF = "/root/reference/src/inner_types"
P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
RINV = pow(1 << 256, P - 2, P)


def words_le(ws):
    return sum(int(w, 16) << (64 * i) for i, w in enumerate(ws))


def from_mont(ws):
    return words_le(ws) * RINV % P


def hx(v):
    return "%064x" % v


def read(name):
    with open(os.path.join(REF, name)) as f:
        return f.read()


def rust_bytes_literal(s):
    # the reference's messages are plain ASCII b"..." literals without escapes
    assert "\\" not in s
    return s


def section(text, start_pat, end_pat):
    a = re.search(start_pat, text).start()
    b = re.search(end_pat, text[a:]).start() + a
    return text[a:b]


def g1_vectors():
    t = read("g1.rs")
    out = {}
    enc = section(t, r"fn encode\(\)", r"fn hash\(\)")
    out["encode_dst"] = re.search(r'DST_ENCODE[^=]*=\s*b"([^"]*)"', enc).group(1)
    vecs = []
    for m in re.finditer(r'TestVector\s*\{(.*?)\}', enc, re.S):
        body = m.group(1)
        if "msg: b\"" not in body:
            continue
        d = {"msg": rust_bytes_literal(re.search(r'msg:\s*b"([^"]*)"', body).group(1))}
        for k in ("p_x", "p_y", "q_x", "q_y", "u"):
            d[k] = re.search(k + r'\s*:\s*"([0-9a-f]+)"', body).group(1)
        vecs.append(d)
    assert len(vecs) == 5
    out["encode"] = vecs
    hs = section(t, r"fn hash\(\)", r"fn arithmetic\(\)")
    out["hash_dst"] = re.search(r'DST_HASH[^=]*=\s*b"([^"]*)"', hs).group(1)
    vecs = []
    for m in re.finditer(r'TestVector\s*\{(.*?)\}', hs, re.S):
        body = m.group(1)
        if "msg: b\"" not in body:
            continue
        d = {"msg": rust_bytes_literal(re.search(r'msg:\s*b"([^"]*)"', body).group(1))}
        for k in ("p_x", "p_y", "q0_x", "q0_y", "q1_x", "q1_y", "u0", "u1"):
            d[k] = re.search(k + r'\s*:\s*"([0-9a-f]+)"', body).group(1)
        vecs.append(d)
    assert len(vecs) == 5
    out["hash"] = vecs
    return out


def parse_fp(expr):
    """Fp::from_words([..]) (canonical LE words) | Fp::from_be_hex("..") | Fp::from_montgomery([..])"""
    expr = expr.strip()
    m = re.match(r'Fp::from_words\(\[(.*?)\]\)', expr, re.S)
    if m:
        return words_le(re.findall(r'0x[0-9a-fA-F]+', m.group(1)))
    m = re.match(r'Fp::from_montgomery\(\[(.*?)\]\)', expr, re.S)
    if m:
        return from_mont(re.findall(r'0x[0-9a-fA-F]+', m.group(1)))
    m = re.match(r'Fp::from_be_hex\(\s*"([0-9a-f]+)"\s*,?\s*\)', expr, re.S)
    if m:
        return int(m.group(1), 16)
    raise ValueError(expr[:80])


FP_EXPR = r'(Fp::from_(?:words|montgomery)\(\[.*?\]\)|Fp::from_be_hex\(\s*"[0-9a-f]+"\s*,?\s*\))'


def parse_fp2_list(text):
    """All `c0: <Fp>, c1: <Fp>` pairs in order of appearance."""
    out = []
    for m in re.finditer(r'c0:\s*' + FP_EXPR + r'\s*,\s*c1:\s*' + FP_EXPR, text, re.S):
        out.append((parse_fp(m.group(1)), parse_fp(m.group(2))))
    return out


def g2_vectors():
    t = read("g2.rs")
    out = {}
    bad = section(t, r"let bad = G2Projective", r"z: Fp2::ONE")
    (bx, by) = parse_fp2_list(bad)
    out["bad_point"] = {"x_c0": hx(bx[0]), "x_c1": hx(bx[1]), "y_c0": hx(by[0]), "y_c1": hx(by[1])}
    for name, start, end, dstname in (("encode", r"fn encode\(\)", r"fn hash\(\)", "DST_ENCODE"),
                                      ("hash", r"fn hash\(\)", r"\Z", "DST_HASH")):
        sec = section(t, start, end) if end != r"\Z" else t[re.search(start, t).start():]
        out[name + "_dst"] = re.search(dstname + r'[^=]*=\s*b"([^"]*)"', sec).group(1)
        vecs = []
        parts = re.split(r'TestVector\s*\{', sec)[2:]      # [0]=prefix, [1]=struct decl
        for part in parts:
            mm = re.search(r'msg:\s*b"([^"]*)"', part)
            if not mm:
                continue
            f2 = parse_fp2_list(part)
            assert len(f2) >= 2
            (x, y) = f2[0], f2[1]
            vecs.append({"msg": rust_bytes_literal(mm.group(1)),
                         "x_c0": hx(x[0]), "x_c1": hx(x[1]), "y_c0": hx(y[0]), "y_c1": hx(y[1])})
        assert len(vecs) == 5, (name, len(vecs))
        out[name] = vecs
    return out


def constants():
    t = read("fp2.rs")
    out = {}
    for nm in ("GEN_X", "GEN_Y", "B"):
        sec = section(t, r"pub const %s: Self = Self \{" % nm, r"\};")
        (c,) = parse_fp2_list(sec)
        out["fp2_" + nm.lower()] = {"c0": hx(c[0]), "c1": hx(c[1])}
    t = read("g2.rs")
    sec = section(t, r"const ENDO_U", r"Self \{\s*x: self")
    (eu, ev) = parse_fp2_list(sec)
    out["psi_endo_u"] = {"c0": hx(eu[0]), "c1": hx(eu[1])}
    out["psi_endo_v"] = {"c0": hx(ev[0]), "c1": hx(ev[1])}
    t = read("pairings.rs")
    sec = section(t, r"fn generator\(\) -> Self \{\s*// pairing", r"fn is_identity")
    f2 = parse_fp2_list(sec)
    assert len(f2) == 6
    names = ("c0.c0", "c0.c1", "c0.c2", "c1.c0", "c1.c1", "c1.c2")
    out["gt_generator"] = {n: {"c0": hx(v[0]), "c1": hx(v[1])} for n, v in zip(names, f2)}
    out["gt_generator_bytes_hex"] = "".join(hx(v[0]) + hx(v[1]) for v in f2)   # pairings.rs:499-514 order
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures are committed, nothing to do")
    data = {"_source": "transcribed from /root/reference/src/inner_types/{g1,g2,fp2,pairings}.rs test modules "
                       "and constants by tests/golden/make_fixtures.py",
            "g1": g1_vectors(), "g2": g2_vectors(), "constants": constants()}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
    with open(dst, "w") as f:
        json.dump(data, f, indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
