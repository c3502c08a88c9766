"""Executed MADs per tuple of the dominant loops, counted by running the DEVICE headers on the host (tests/hostsim, -DBN_CHECK):
every fp_mul / fp_sqr / fp_dot2 / fp_lc pass of a loop is counted while it runs; the counts are data independent (uniform
control flow), so one run counts for all inputs.  MADs per operation are those of the code as compiled for gfx950
(fp29.h: product 81 + reduction 81 = 162; squaring 45 + 81 = 126; double product 162 + 81 = 243; an fp_lc pass 9 per term with
-DBN_LC_MAD, which is how the Miller / t^x / h3 units are built).  Writes profiles/r03_executed_mads.json (read by bench.py).
Usage: python scripts/executed_mads.py"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIM = os.path.join(ROOT, "tests", "hostsim")


def main():
    so = os.path.join(SIM, "libhostsim.so")
    src = [os.path.join(SIM, "hostsim.cpp")] + [os.path.join(ROOT, "bls-bn254_amd", "csrc", f)
                                               for f in os.listdir(os.path.join(ROOT, "bls-bn254_amd", "csrc")) if f.endswith(".h")]
    if not os.path.exists(so) or any(os.path.getmtime(p) > os.path.getmtime(so) for p in src):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-DBN_CHECK", "-DBN_VERIFY_PARK_T", "-fPIC", "-shared", "-pthread", "-o", so, os.path.join(SIM, "hostsim.cpp")])
    hs = ctypes.CDLL(so)
    from oracle import oracle as O
    from tests import synth
    dst = b"D"
    sk = synth.sk_of(3)
    pk = O.sk_to_pk(sk); msg = b"count"; sig = O.sign(sk, msg, dst)
    h = O.hash_to_g1_batch([msg], dst)
    out = (ctypes.c_double * 30)()
    hs.hs_executed_ops(sig, h, pk, out)
    names = ["miller_loop_prepared (k_miller_prepared)", "miller_loop_verify_ws2 (k_miller_verify)", "cyclotomic_exp_x_chain (one t^x launch)",
             "fe_easy", "fe_h1 + fe_h2 + fe_h3"]
    core = O.verify_core_counts()
    alg = {0: core[4] + 2 * core[1], 1: core[0] + core[1]}
    res = {"how": "tests/hostsim hs_executed_ops: the device headers run on the host with operation counters (data independent)",
           "mads_per_op": {"fp_mul": 162, "fp_sqr": 126, "fp_dot2": 243, "fp_lc_term": 9}, "phases": {}}
    for k, nm in enumerate(names):
        mul, sqr, dot, norm, lcs, terms = [int(out[6 * k + j]) for j in range(6)]
        mads = 162 * mul + 126 * sqr + 243 * dot + 9 * terms
        e = {"fp_mul": mul, "fp_sqr": sqr, "fp_dot2": dot, "fp_norm": norm, "fp_lc_passes": lcs, "fp_lc_terms": terms,
             "executed_mads": mads, "executed_mads_products_only": 162 * mul + 126 * sqr + 243 * dot}
        if k in alg:
            e["algorithmic_fp_mul"] = alg[k]
            e["algorithmic_mads_at_136"] = alg[k] * 136
            e["executed_over_algorithmic"] = round(mads / (alg[k] * 136.0), 4)
        res["phases"][nm] = e
    path = os.path.join(ROOT, "profiles", "r03_executed_mads.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
