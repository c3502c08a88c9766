"""CPU-only: every kernel in the built library has a real body.  A kernel whose source hits undefined behaviour (an accidental
self-recursion turned k_miller_tri_prepared into a 15-instruction endless loop during round 3) still compiles and links; only
its size gives it away before a GPU run hangs on it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# kernels that are legitimately tiny (index fills, flag reductions, scans) -- everything else does field arithmetic
SMALL_OK = ("k_iota", "k_pack_bitmap", "k_prep_unsort", "k_status_reduce", "k_and_reduce", "k_kd_", "k_scan_excl", "k_rlc2_", "k_fp12_mask_one", "k_valu_peak", "k_place_bitmap")


def test_no_degenerate_kernels():
    import blsbn254_loader
    M = blsbn254_loader.load()
    path = M.library_path()
    if not os.path.exists(path):
        __import__("bls_bn254_amd.build", fromlist=["x"]).build()
    from kernel_resources import code_objects
    tmp = tempfile.mkdtemp()
    sizes = {}
    for idx, (data, off, size) in enumerate(code_objects(path)):
        co = os.path.join(tmp, "co_%d.o" % idx)
        open(co, "wb").write(data[off:off + size])
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1); sizes.setdefault(cur, 0)
            elif cur and re.match(r"^\s+[a-z_0-9]+", line):
                sizes[cur] += 1
    kernels = {k: v for k, v in sizes.items() if re.search(r"\dk_[a-z]", k) or k.startswith("k_")}
    assert len(kernels) >= 60, sorted(kernels)
    bad = [(k, v) for k, v in kernels.items() if v < 200 and not any(s in k for s in SMALL_OK)]
    assert not bad, "suspiciously small kernels (undefined behaviour in the source?): %s" % bad


# The register-resident loops of the throughput path: built to fit the 512 registers of one wave per SIMD with NO spills to
# memory (spills into the accumulation registers are part of the plan and do not count).  A header change that has nothing to
# do with them can push them over (adding one bool to the workspace reference made k_fe_expx_h1 / _h2 spill 203 / 217 registers
# to scratch: 4.65 -> 6.5 ms each, a 10 % slower step, with every test still green).  Limit = bytes of scratch per lane.
SCRATCH_LIMIT = {"k_miller_prepared": 0, "k_fe_expx": 0, "k_fe_expx_h1": 0, "k_fe_expx_h2": 0, "k_fe_h3": 0, "k_fe_easy_tail": 0, "k_miller_tri_prepared": 0,
                 "k_fe_tri_hard": 0, "k_miller_tri_1": 0, "k_miller_tri_1p": 0,
                 "k_miller_wide_prepared": 64, "k_fe_hard_wide": 64}      # the wave-per-tuple kernels call real functions: a few stack words


def test_throughput_kernels_do_not_spill_to_memory():
    import blsbn254_loader
    M = blsbn254_loader.load()
    path = M.library_path()
    if not os.path.exists(path):
        __import__("bls_bn254_amd.build", fromlist=["x"]).build()
    from kernel_resources import code_objects
    tmp = tempfile.mkdtemp()
    scratch = {}
    for idx, (data, off, size) in enumerate(code_objects(path)):
        co = os.path.join(tmp, "co_%d.o" % idx)
        open(co, "wb").write(data[off:off + size])
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s*(\S+)", blk).group(1)
            if name.startswith("_Z"):
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0].strip()
            scratch[name] = int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk).group(1))
    missing = [k for k in SCRATCH_LIMIT if k not in scratch]
    assert not missing, missing
    bad = {k: scratch[k] for k, lim in SCRATCH_LIMIT.items() if scratch[k] > lim}
    assert not bad, "scratch (spilled registers) in register-resident kernels, bytes per lane: %s" % bad


# Serialised accesses: a buffer descriptor or scalar offset that reaches a load through VECTOR registers is wrapped in a
# read-first-lane loop per access (k_fe_tri_hard had 1378 of them, the wave-per-tuple Miller loop paid 8.5 us per line value for
# a lane-dependent base), and an LDS address that lost its address space becomes a flat access.  Limits per kernel:
# (read-first-lane instructions, flat accesses); scripts/isa_lint.py prints the table for every kernel.
SERIALISED_LIMITS = {"k_miller_prepared": (8, 0), "k_fe_expx": (8, 0), "k_fe_expx_h1": (8, 0), "k_fe_expx_h2": (8, 0), "k_fe_h3": (16, 0),
                     "k_miller_tri_prepared": (16, 0), "k_fe_tri_hard": (100, 0), "k_miller_tri_1": (16, 0), "k_miller_tri_1p": (16, 0),
                     "k_miller_wide_prepared": (100, 0), "k_fe_hard_wide": (300, 0), "k_hash_to_g1": (8, 0)}


def test_hot_kernels_have_no_serialised_accesses():
    import blsbn254_loader
    M = blsbn254_loader.load()
    path = M.library_path()
    if not os.path.exists(path):
        __import__("bls_bn254_amd.build", fromlist=["x"]).build()
    from isa_lint import lint
    found = {}
    for sym, c in lint(path).items():
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.split("(")[0].strip() if sym.startswith("_Z") else sym
        if name in SERIALISED_LIMITS:
            found[name] = (c["readfirstlane"], c["flat"])
    missing = [k for k in SERIALISED_LIMITS if k not in found]
    assert not missing, missing
    bad = {k: found[k] for k, (rfl, flat) in SERIALISED_LIMITS.items() if found[k][0] > rfl or found[k][1] > flat}
    assert not bad, "read-first-lane instructions / flat accesses over the limit (kernel: (rfl, flat)): %s" % bad
