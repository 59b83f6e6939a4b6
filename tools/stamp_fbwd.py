#!/usr/bin/env python3
"""Phase breakdown of the transpose-free field backward: builds csrc with -DUMHS_TF_STAMP into tools/_alt/libumhs_stamp.so (CPU box:
`python tools/stamp_fbwd.py build`), then on the GPU box runs one backward per case and prints the cycles wave 0 of workgroup 0 spent
between consecutive stamps (s_memtime, pinned by scheduling barriers -- the stamped build is a little slower than the product)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")
ALT = os.path.join(ROOT, "tools", "_alt")
LIB = os.path.join(ALT, "libumhs_stamp.so")
sys.path[:0] = [ROOT, PKG]
if sys.argv[1:2] == ["build"]:
    from umhsnerf import build as B
    os.makedirs(ALT, exist_ok=True)
    objs = []
    for src in B.SOURCES:
        obj = os.path.join(ALT, src.replace(".hip", ".stamp.o"))
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), "-DUMHS_TF_STAMP",
                               f"-I{B.INCLUDE}", f"-I{B.CSRC}", "-c", os.path.join(B.CSRC, src), "-o", obj])
        objs.append(obj)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    sys.exit(0)
import torch
from umhsnerf import _hip, ops
_hip.LIB_PATH = LIB  # the stamped build instead of the product library (this process only)
lib = _hip.lib()
lib.umhs_debug_tf_stamps.argtypes = [ctypes.c_void_p]
dev = "cuda:0"
CASES = {"C2": (6, 31, True, 262144), "C3": (9, 128, True, 524288), "C5": (4, 141, False, 524288)}
NAMES = {0: {1: "head fwd H0 H1 H2", 2: "epilogue + dir fwd + swaps", 3: "band tiles", 7: "head outputs (+ dir hidden dW)", 8: "swap a2, dW2",
             9: "T2 chain + mask", 10: "swap dz1, a1", 11: "dW1 (4x4)", 12: "T1 chain + mask", 13: "swap dz0, dW0 (4x2)", 14: "T0 chain", 18: "store"},
         1: {1: "base fwd B0 B1", 2: "feature fwd F0 F1", 7: "swap x27", 8: "swap a2, dW2", 9: "T2 chain + mask", 10: "swap dz1, a1", 11: "dW1 (4x4)",
             12: "T1 chain + mask", 13: "swap dz0, dW0 (4x2)", 14: "T0 chain", 15: "base: swap z1, h, dW B1", 16: "T_B1 + swaps + dW B0",
             17: "T_B0 chain", 18: "store d_enc"}}
if os.environ.get("UMHS_BWD_TF", "3") != "1":  # the zipped kernels (csrc/umhs_field_zip.h): part 1 VALU blocks with their slots, gemms bare; part 0 by stretch
    NAMES[0] = {1: "head MLP forward (zipped)", 2: "epilogue + dir hidden fwd", 3: "swaps m, dir, pe, hdir", 4: "band tiles", 5: "per-ray sums (folded)",
                6: "head outputs", 7: "dir hidden dW", 8: "head MLP backward (zipped)", 9: "store + end slots"}
    NAMES[1] = {1: "V: pe, split enc", 2: "gemm B0", 3: "V: relu + split h", 4: "gemm B1", 5: "V: split in27", 6: "gemm F0", 7: "V: relu + split a1",
                8: "gemm F1", 9: "V: relu + split a2, d_fl", 10: "fp32 gemm T_F2", 11: "V: mask + split dz1", 12: "gemm T_F1", 13: "V: mask + split dz0",
                14: "gemm T_F0", 15: "V: dzb1 + fp32 gemm T_B1", 16: "V: mask + split dzb0", 17: "gemm T_B0", 18: "store + end slots"}
for name in os.environ.get("CASES", "C2").split(","):
    C, B, spec, N = CASES[name]
    layout = ops.FieldLayout(C, B, spec, 19)
    g = torch.Generator().manual_seed(0)
    flat = ((torch.rand(layout.total, generator=g) - 0.5) * 0.5).to(dev)
    fs = ops.FieldSpec(layout, 0.4, True, scalings=ops.hash_scalings().to(dev))
    enc = (torch.rand(16, N, 2, device=dev) - 0.5)
    wpos = torch.rand(N, 3, device=dev) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    sel = torch.ones(N, device=dev)
    out = ops.field_fwd(fs, flat, enc, True, wpos, dirs, sel, want_emb=True, want_logits=True)
    dsig, dspec = torch.rand(N, device=dev), torch.rand(N, B, device=dev)
    dflat = torch.zeros_like(flat)
    fn = lambda: ops.field_bwd(fs, flat, enc, True, wpos, dirs, sel, out["sigma_raw"], out["emb"], dsig, dspec, None, dflat, feat_logits=out["feat_logits"])
    if os.environ.get("FUSED", "0") == "1":  # the folded compositing backward (64-sample rays)
        R = N // 64
        pinfo = torch.stack([torch.arange(R, device=dev) * 64, torch.full((R,), 64, device=dev)], 1).contiguous()
        ray = torch.arange(R, device=dev).repeat_interleave(64).contiguous()
        t0 = torch.rand(N, device=dev)
        comp = dict(sigma=out["sigma"], t0=t0, t1=t0 + 0.01, packed_info=pinfo, ray_indices=ray, weights=torch.rand(N, device=dev) * 0.05,
                    d_comp=torch.randn(R, B, device=dev), d_acc=torch.randn(R, device=dev), grad_scaling=True)
        b16 = torch.cat([out["sigma_raw"][:, None], out["emb"]], 1).contiguous()
        fn = lambda: ops.field_bwd(fs, flat, enc, True, wpos, dirs, sel, out["sigma_raw"], b16, None, None, None, dflat, feat_logits=out["feat_logits"],
                                   comp=dict(comp))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.umhs_debug_tf_stamps_clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 48)()
    lib.umhs_debug_tf_stamps(buf)
    print(f"{name}: stamped backward {e0.elapsed_time(e1) * 1e3:.1f} us")
    for part in (0, 1):
        v = list(buf[24 * part: 24 * part + 24])
        tiles = max(v[0], 1)
        tot = sum(v[1:19])
        print(f"  part {part}: {tiles} tiles, {tot / tiles:.0f} cycles per tile")
        for k in range(1, 19):
            if v[k]:
                print(f"    {NAMES[part].get(k, str(k)):34s} {v[k] / tiles:8.0f}  {100.0 * v[k] / tot:5.1f} %")
        print(f"    whole kernel (workgroup 0): LDS image {v[20]} + tile loop {v[21]} + accumulator reduce / slab {v[22]} cycles")
        print(f"    prologue: fp32 packs {v[16]}, transposed packs {v[17]}, bf16x3 packs {v[19]}, barrier {v[23]} cycles (last launch)")
