"""Build libumhs_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
LIB = os.path.join(HERE, "libumhs_hip.so")
SOURCES = ("umhs_kernels.hip", "umhs_field.hip", "umhs_sampler.hip", "umhs_data.hip", "umhs_metrics.hip", "umhs_rgb.hip")
# umhs_field.hip compiles as four translation units side by side (its header comment): object suffix -> extra defines
UNITS = {"umhs_field.hip": (("_p0", ["-DUMHS_FIELD_TU=1", "-Wno-unused-function"]), ("_p1", ["-DUMHS_FIELD_TU=2", "-Wno-unused-function"]),
                            ("_p0f", ["-DUMHS_FIELD_TU=3", "-Wno-unused-function"]), ("", ["-DUMHS_FIELD_TU=0"]))}
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-munsafe-fp-atomics", "-std=c++17"]
# umhs_field.hip: MFMAs written as builtins get the VGPR C/D form even in the kernels whose register budget exceeds 256 (the
# transpose-free backward); their long-lived dW accumulators are inline-asm MFMAs on AGPRs (see dw_row in that file)
# -fno-slp-vectorize: hipcc packs the two residual subtractions of a bf16 split into one v_pk_add_f32, which costs more beside MFMAs
# than the two v_sub_f32 it replaces and needs an s_nop in front of the conversion that reads it (field backward: -8 us at C2)
EXTRA_FLAGS = {"umhs_field.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"]}


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "umhs_hip.h")]


def _cmd_stamp(cmd) -> str:
    import hashlib

    # (paths reduced to their last component: the same tree under another root -- a gpurun snapshot -- is the same build)
    return hashlib.sha256("\0".join(("-I" + os.path.basename(c[2:])) if c.startswith("-I") else os.path.basename(c) if os.sep in c else c
                                     for c in cmd).encode()).hexdigest()


def _obj_stale(src: str, obj: str, cmd) -> bool:
    """An object is stale when a source / header is newer, or when it was compiled with another command line (flags, defines, unit
    split: `<obj>.cmd` holds the hash of the command that produced it -- an A/B of compiler flags must never link a mixed build)."""
    if not os.path.exists(obj):
        return True
    try:
        with open(obj + ".cmd") as f:
            if f.read().strip() != _cmd_stamp(cmd):
                return True
    except OSError:
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src, *_headers()])


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for src in SOURCES:  # an object compiled with another command line (or never stamped) makes the library stale too
        for suffix, defines in UNITS.get(src, (("", []),)):
            obj = os.path.join(CSRC, src.replace(".hip", suffix + ".o"))
            cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), *defines, f"-I{INCLUDE}", f"-I{CSRC}", "-c", os.path.join(CSRC, src), "-o", obj]
            if os.path.exists(obj) and not os.path.exists(obj + ".cmd"):
                continue  # (objects that travelled without their stamp -- a GPU box's copy -- are judged by their mtimes alone)
            if _obj_stale(os.path.join(CSRC, src), obj, cmd):
                return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip")] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    """Compiles the sources that changed (all of them when a header did, or with ``force``), the files side by side, and links."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, running, t0 = [], [], time.time()
    for src in SOURCES:
        for suffix, defines in UNITS.get(src, (("", []),)):
            obj = os.path.join(CSRC, src.replace(".hip", suffix + ".o"))
            objs.append(obj)
            cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), *defines, f"-I{INCLUDE}", f"-I{CSRC}", "-c", os.path.join(CSRC, src), "-o", obj]
            if not force and not _obj_stale(os.path.join(CSRC, src), obj, cmd):
                continue
            if os.path.exists(obj + ".cmd"):
                os.remove(obj + ".cmd")
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
    for cmd, proc in running:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        with open(cmd[-1] + ".cmd", "w") as f:
            f.write(_cmd_stamp(cmd))
        if verbose:
            print(f"[build] {os.path.basename(cmd[-1])}: done {time.time() - t0:.0f} s after the start", flush=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
