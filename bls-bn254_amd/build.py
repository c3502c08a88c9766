"""Builds the HIP shared library in-tree: bls-bn254_amd/libblsbn254_hip.so (gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "kernels.hip")
OUT = os.path.join(HERE, "libblsbn254_hip.so")
DEPS = ["kernels.hip", "lane_ops.h", "pairing.h", "curve.h", "tower.h", "fp29.h", "fr29.h", "sha256.h", "bn254_consts.h"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    paths = [os.path.join(HERE, "csrc", d) for d in DEPS] + [os.path.join(HERE, "..", "include", "blsbn254.h")]
    return any(os.path.getmtime(p) > t for p in paths if os.path.exists(p))


def build(force=False, waves_per_simd=None, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-o", OUT, SRC]
    wps = waves_per_simd or os.environ.get("BN_WAVES_PER_SIMD")
    if wps:
        cmd.append("-DBN_WAVES_PER_SIMD=%s" % wps)
    if verbose:
        print("[bls-bn254_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
