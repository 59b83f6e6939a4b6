#!/usr/bin/env python3
"""Instruction-class timeline of a kernel's longest loop: tools/isa_timeline.py <disassembly.s> <mangled-name-substring>
M = v_mfma 16x16x32 bf16, m = v_mfma 16x16x16 bf16, F = fp32 mfma, v = VALU, t = transcendental, L = ds_read, l = ds_write,
G = global/buffer memory, w = s_waitcnt, n = s_nop, s = other SALU, b = branch.  Disassemble with
llvm-objdump -d --no-show-raw-insn on the gfx950 code object (tools/kernel_regs.py shows how to extract it)."""
import re
import sys


def cls(l):
    op = l.split()[0]
    if op.startswith("v_mfma"):
        return "M" if "16x16x32" in op else ("m" if "16x16x16" in op else "F")
    if op.startswith(("v_exp", "v_rcp", "v_sin", "v_cos", "v_log", "v_sqrt", "v_rsq")):
        return "t"
    if op.startswith("v_"):
        return "v"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "L"
    if op.startswith("ds_"):
        return "l"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "G"
    if op == "s_waitcnt":
        return "w"
    if op == "s_nop":
        return "n"
    if op.startswith(("s_cbranch", "s_branch")):
        return "b"
    return "s"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <", l) and sys.argv[2] in l)
    end = next(i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i]))
    body = [l.strip() for l in lines[start + 1:end] if l.strip()]
    tl = "".join(cls(l) for l in body)
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(tl))
    seg = tl[lo:hi]
    for i in range(0, len(seg), 120):
        print(f"{lo + i:5d} {seg[i:i + 120]}")
    from collections import Counter

    print(Counter(seg))


if __name__ == "__main__":
    main()
