"""Static checks of the built kernels for the code-generation accidents that cost this project time (each once, each invisible to
the tests): per kernel the number of read-first-lane loops (a buffer descriptor or scalar offset that reached a load through
vector registers: every access is serialised), flat accesses (an LDS or scratch address that lost its address space), the
scratch bytes per lane (spilled registers) and the instruction count.  Usage: python scripts/isa_lint.py [lib.so | obj.o ...]
-> one line per kernel; tests/test_kernel_sanity.py asserts limits on the kernels of the hot path."""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import code_objects

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lint(path):
    """{symbol: Counter(total, readfirstlane, waterfall_loops, flat)} for every function symbol of the gfx950 code objects"""
    out = {}
    tmp = tempfile.mkdtemp()
    for idx, (data, off, size) in enumerate(code_objects(path)):
        co = os.path.join(tmp, "co_%d.o" % idx)
        open(co, "wb").write(data[off:off + size])
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        cur = None
        prev = ""
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = out.setdefault(m.group(1), collections.Counter())
                continue
            t = line.strip().split()
            if not t or cur is None:
                continue
            op = t[0]
            cur["total"] += 1
            if op == "v_readfirstlane_b32":
                cur["readfirstlane"] += 1
            if op == "s_and_saveexec_b64" and prev.startswith(("v_cmp_eq", "s_and_b64", "s_nop")):
                cur["waterfall_loops"] += 1
            if op.startswith(("flat_load", "flat_store")):
                cur["flat"] += 1
            prev = op
    return out


def main():
    paths = sys.argv[1:] or [os.path.join(ROOT, "bls-bn254_amd", "libblsbn254_hip.so")]
    print("%-44s %8s %8s %8s %6s" % ("symbol", "instrs", "rfl", "wf-loops", "flat"))
    for p in paths:
        for name, c in sorted(lint(p).items()):
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0].strip() if name.startswith("_Z") else name
            print("%-44s %8d %8d %8d %6d" % (short[:44], c["total"], c["readfirstlane"], c["waterfall_loops"], c["flat"]))


if __name__ == "__main__":
    main()
