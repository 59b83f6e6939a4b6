#!/usr/bin/env python3
"""A/B builds that differ in ONE source of csrc (default umhs_kernels.hip; `--file umhs_sampler.hip` for another single-unit source):
`python tools/alt_kernels.py NAME [--file F.hip] [--src FILE] [-DFOO=1 ...]` compiles that file (or FILE, e.g. an older revision written
to /tmp) with the extra flags and links it with the in-tree objects of the other sources into tools/_alt/libumhs_NAME.so.
UMHS_LIB_PATH=tools/_alt/libumhs_NAME.so selects it (tools/ab_lib.sh)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")]
from umhsnerf import build as B
name, args = sys.argv[1], sys.argv[2:]
which = "umhs_kernels.hip"
if "--file" in args:
    i = args.index("--file")
    which = args[i + 1]
    del args[i:i + 2]
src = os.path.join(B.CSRC, which)
if "--src" in args:
    i = args.index("--src")
    src = args[i + 1]
    del args[i:i + 2]
ALT = os.path.join(ROOT, "tools", "_alt")
os.makedirs(ALT, exist_ok=True)
B.build_lib(verbose=False)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
obj = os.path.join(ALT, f"{which[:-4]}.{name}.o")
subprocess.check_call([hipcc, *B.FLAGS, *B.EXTRA_FLAGS.get(which, []), *args, f"-I{B.INCLUDE}", f"-I{B.CSRC}", "-c", src, "-o", obj])
objs = [obj]
for s in B.SOURCES:
    if s != which:
        objs += [os.path.join(B.CSRC, s.replace(".hip", suffix + ".o")) for suffix, _ in B.UNITS.get(s, (("", []),))]
out = os.path.join(ALT, f"libumhs_{name}.so")
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
os.remove(obj)
print(out)
