"""Per-kernel resource usage of the built library (registers, spills, scratch, LDS), read from the code objects'
AMDGPU metadata notes -- no GPU and no recompilation needed.  Usage: python scripts/kernel_resources.py [lib.so]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "bls-bn254_amd", "libblsbn254_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    """(offset, size) of every gfx950 code object inside the clang offload bundles of a fat binary"""
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    pos = data.find(magic)
    while pos >= 0:
        n = int.from_bytes(data[pos + 24:pos + 32], "little")
        q = pos + 32
        for _ in range(n):
            off, size, tl = (int.from_bytes(data[q + 8 * k:q + 8 * k + 8], "little") for k in range(3))
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append((data, pos + off, size))
        pos = data.find(magic, pos + 24)
    return out


def main():
    tmp = tempfile.mkdtemp()
    rows = []
    for idx, (data, off, size) in enumerate(code_objects(LIB)):
        co = os.path.join(tmp, "co_%d.o" % idx)
        open(co, "wb").write(data[off:off + size])
        notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
            name = g("name")
            if name.startswith("_Z"):
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0].strip()
            rows.append((name, g("vgpr_count"), blk.split("\n")[0].strip(), g("vgpr_spill_count"), g("sgpr_spill_count"),
                         g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    print("%-28s %5s %5s %7s %7s %9s %8s" % ("kernel", "vgpr", "agpr", "v-spill", "s-spill", "scratch B", "LDS B"))
    for r in sorted(set(rows)):
        print("%-28s %5s %5s %7s %7s %9s %8s" % r)


if __name__ == "__main__":
    main()
