"""Instruction mix of the kernels in a built object / library: extracts the gfx950 code objects, disassembles them with
llvm-objdump and counts instruction classes per kernel (whole kernel, static counts).  Usage: python scripts/isa_mix.py file [kernel-substring]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import code_objects

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def classify(op):
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")): return "mad64"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_", "v_mul_")): return "mul32"
    if op.startswith("v_accvgpr") or op.startswith("v_mov") : return "mov/acc"
    if op.startswith(("v_and", "v_or", "v_xor", "v_bfe", "v_lshr", "v_lshl", "v_ashr", "v_alignbit", "v_bfi", "v_not")): return "bit/shift"
    if op.startswith(("v_add", "v_sub", "v_addc", "v_subb", "v_lshl_add", "v_add3")): return "add/sub"
    if op.startswith("v_cndmask") or op.startswith("v_cmp"): return "select/cmp"
    if op.startswith(("scratch_", "buffer_", "global_", "flat_")): return "vmem:" + ("scratch" if op.startswith("scratch") else "global")
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "salu/other"
    if op.startswith("v_"): return "valu-other"
    return "other"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    tmp = tempfile.mkdtemp()
    for idx, (data, off, size) in enumerate(code_objects(path)):
        co = os.path.join(tmp, "co_%d.o" % idx)
        open(co, "wb").write(data[off:off + size])
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        cur, counts = None, None
        def flush():
            if cur and want in cur and counts:
                tot = sum(counts.values())
                print("%s: %d instructions" % (cur, tot))
                for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
                    print("   %-14s %8d  %5.1f%%" % (k, v, 100.0 * v / tot))
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                flush()
                cur, counts = m.group(1), collections.Counter()
                continue
            m = re.match(r"^\s+(\w+)", line)
            if m and cur:
                counts[classify(m.group(1))] += 1
        flush()


if __name__ == "__main__":
    main()
