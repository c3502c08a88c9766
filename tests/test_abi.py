"""CPU-only: the C-ABI library loads, exports every symbol include/blsbn254.h declares, and refuses to
run without a gfx950 device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def M():
    import blsbn254_loader
    return blsbn254_loader.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "blsbn254.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(blsbn254_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_surveyed_entry_points():
    syms = declared_symbols()
    for want in ("blsbn254_ctx_create", "blsbn254_ctx_destroy", "blsbn254_pairing_batch", "blsbn254_multi_miller_loop",
                 "blsbn254_final_exponentiation", "blsbn254_hash_to_g1_batch", "blsbn254_hash_to_g2_batch",
                 "blsbn254_g1_check_batch", "blsbn254_g2_check_batch", "blsbn254_verify_batch", "blsbn254_aggregate_verify",
                 "blsbn254_aggregate_sigs", "blsbn254_threshold_combine"):
        assert want in syms


def test_rust_shim_extern_block_matches_the_header():
    """integration/rust/ffi.rs is generated from include/blsbn254.h; the committed copy must be current and must
    declare every entry point exactly once (the reference-side binding)."""
    import subprocess
    import sys
    gen = os.path.join(ROOT, "integration", "rust", "gen_ffi.py")
    assert subprocess.call([sys.executable, gen, "--check"]) == 0, "run python integration/rust/gen_ffi.py"
    ffi = open(os.path.join(ROOT, "integration", "rust", "ffi.rs")).read()
    assert sorted(re.findall(r"pub fn (blsbn254_[a-z0-9_]+)\(", ffi)) == declared_symbols()
    shim = open(os.path.join(ROOT, "integration", "rust", "gpu.rs")).read()
    for used in set(re.findall(r"ffi::(blsbn254_[a-z0-9_]+)", shim)):
        assert used in declared_symbols()


def test_library_exports_every_declared_symbol(M):
    path = M.library_path()
    if not os.path.exists(path):
        bld = __import__("bls_bn254_amd.build", fromlist=["x"])
        bld.build()
    lib = ctypes.CDLL(path)
    for s in declared_symbols():
        assert hasattr(lib, s), "missing export " + s


def test_no_cpu_fallback(M):
    """Without a GPU the engine must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(M.Bn254Error) as e:
        M.Engine(0)
    assert e.value.code == -4
    lib = M.load_library()
    assert b"no CPU fallback" in lib.blsbn254_strerror(-4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bls-bn254_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "bn254_oracle" not in src, f


def test_error_class_mapping(M):
    assert M.InvalidScalarBytes.__mro__[1] is M.Bn254Error            # error.rs:4-10 order
    lib = M.load_library()
    assert [lib.blsbn254_strerror(i) for i in (1, 2, 3, 4)] == [b"invalid scalar bytes", b"invalid G1 bytes", b"invalid G2 bytes", b"invalid Gt bytes"]


def test_pack_messages(M):
    data, off = M.engine.pack_messages([b"", b"ab", b"cde"])
    assert data == b"abcde" and list(off) == [0, 0, 2, 5]


def test_header_is_valid_c_and_library_loads_from_a_c_program(tmp_path):
    """include/blsbn254.h compiled as strict C99 by a plain-C consumer that opens the shared library the way a cgo / Rust-FFI
    caller would (no HIP headers, no C++): every symbol it needs resolves, error strings match the reference's error names,
    and without a gfx950 device ctx_create fails loudly with BLSBN254_E_NO_DEVICE instead of falling back to a CPU path."""
    import subprocess
    exe = str(tmp_path / "consumer")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_c", "consumer.c"), "-o", exe, "-ldl"])
    lib = os.path.join(ROOT, "bls-bn254_amd", "libblsbn254_hip.so")
    out = subprocess.run([exe, lib], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.startswith("no device") or out.stdout.startswith("ctx ok")
