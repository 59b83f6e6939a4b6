#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of a compiled source: tools/kernel_regs.py umhs_field [substring]
(reads the gfx950 code object out of csrc/<name>.o; no GPU needed)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def notes(name):
    obj = name if name.endswith(".o") else os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd", "csrc", name + ".o")
    with tempfile.TemporaryDirectory() as d:
        import shutil

        tmp = os.path.join(d, "k.o")
        shutil.copy(obj, tmp)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], capture_output=True, cwd=d)  # writes k.o.0.hipv4-...gfx950
        co = next(os.path.join(d, f) for f in os.listdir(d) if "hipv4" in f)
        return subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "umhs_field"
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    # umhs_field = the file's four translation units (umhsnerf/build.py)
    txt = "".join(notes(n) for n in (["umhs_field", "umhs_field_p0", "umhs_field_p1", "umhs_field_p0f"] if name == "umhs_field" else [name]))
    for e in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
        e = ".agpr_count:" + e
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", e) or [None, "?"])[1]
        n = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if want in n:
            print(f"{n[:80]:80s} vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>4} sgpr {g('sgpr_count'):>4} spill {g('vgpr_spill_count'):>4} "
                  f"scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}")


if __name__ == "__main__":
    main()
