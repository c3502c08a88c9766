"""Builds the HIP shared library in-tree: bls-bn254_amd/libblsbn254_hip.so (gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "libblsbn254_hip.so")


def build(force=False, jobs=None, verbose=True):
    if force:
        subprocess.check_call(["make", "-C", HERE, "clean"])
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    jobs = jobs or max(1, min(os.cpu_count() or 4, 12))      # the inlined kernel units take ~2-4 min each
    cmd = ["make", "-C", HERE, "-j%d" % jobs]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd, env=env)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
