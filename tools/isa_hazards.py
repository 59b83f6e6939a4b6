#!/usr/bin/env python3
"""Static check of the inline-asm MFMAs of a compiled source: tools/isa_hazards.py [umhs_field]   (no GPU needed).

hipcc's hazard recognizer does not look inside inline asm.  The dW products of the transpose-free field backward are inline-asm MFMAs
(accumulators pinned to AGPRs); gfx950 needs two wait states between a VALU write of a VGPR (v_perm, v_mov, v_accvgpr_read, a
conversion ...) and an MFMA that reads it as its A or B operand, and for an inline-asm MFMA nobody inserts them.  This walks the
disassembly and reports every MFMA with AGPR destination whose A / B operand registers -- VGPRs, or AGPRs (the carried dW products
take their operand tiles from AGPRs, `dwm_a`: a register-allocator copy `v_accvgpr_write` / `v_accvgpr_mov` into such an operand
right in front of the MFMA is the same hazard) -- were written by one of the two preceding instructions (s_nop k counts as k + 1).
Exit status 1 if any is found.  `check(text, stats)` also counts the MFMAs it looked at by operand file."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def regs(tok):
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, "k.o")
        shutil.copy(obj, tmp)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], capture_output=True, cwd=d)
        co = next(os.path.join(d, f) for f in os.listdir(d) if "hipv4" in f)
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout


def check(text, stats=None):
    bad, kernel, hist = [], "?", []
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel, hist = m.group(1), []
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        toks = [t.strip(",") for t in ins.split()]
        op = toks[0]
        if op.startswith("v_mfma") and len(toks) >= 4 and toks[1].startswith("a"):
            srcs = {r for t in toks[2:4] for r in regs(t)}
            if stats is not None:
                for f in {r[0] for r in srcs}:
                    stats[f] = stats.get(f, 0) + 1
            states = 0
            for prev in reversed(hist[-4:]):
                if states >= 2:
                    break
                pt = prev.split()
                if pt[0] == "s_waitcnt":  # (memory results are ordered by the counters, not by wait states)
                    continue
                if pt[0] == "s_nop":
                    states += int(pt[1]) + 1
                    continue
                if pt[0].startswith("v_") and len(pt) > 1 and regs(pt[1].strip(",")) & srcs:
                    bad.append((kernel, prev, ins))
                    break
                states += 1
        hist.append(ins)
    return bad


def main():
    names = sys.argv[1:] or ["umhs_field", "umhs_field_p0", "umhs_field_p1", "umhs_field_p0f"]  # default: the field file's four translation units
    bad = []
    for name in names:
        obj = name if name.endswith(".o") else os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd", "csrc", name + ".o")
        bad += check(disassemble(obj))
    for kernel, prev, ins in bad[:20]:
        print(f"{kernel[:60]}: '{prev}' right in front of '{ins}'")
    print(f"{len(bad)} inline-asm MFMA(s) read a VGPR written fewer than two wait states earlier")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
