#!/usr/bin/env python3
"""A/B builds of the HIP library: `python tools/alt_build.py NAME -DFOO=1 ...` compiles csrc with the extra flags into
tools/_alt/libumhs_NAME.so (travels to the GPU box, git-ignored); tools that honour UMHS_ALT_LIB=NAME load it instead of the product
library (this process only; the product never reads that variable)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")]
ALT = os.path.join(ROOT, "tools", "_alt")


def alt_path(name):
    return os.path.join(ALT, f"libumhs_{name}.so")


def use_alt_from_env():
    name = os.environ.get("UMHS_ALT_LIB")
    if name:
        from umhsnerf import _hip
        assert os.path.exists(alt_path(name)), alt_path(name)
        _hip.LIB_PATH = alt_path(name)
        print(f"[alt lib] {alt_path(name)}", flush=True)


if __name__ == "__main__":
    from umhsnerf import build as B
    name, extra = sys.argv[1], sys.argv[2:]
    os.makedirs(ALT, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in B.SOURCES:
        obj = os.path.join(ALT, src.replace(".hip", f".{name}.o"))
        procs.append(subprocess.Popen([hipcc, *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), *extra, f"-I{B.INCLUDE}", f"-I{B.CSRC}", "-c",
                                       os.path.join(B.CSRC, src), "-o", obj]))
        objs.append(obj)
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", alt_path(name)])
    for o in objs:
        os.remove(o)
    print(alt_path(name))
