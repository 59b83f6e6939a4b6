#!/usr/bin/env python3
"""Field backward alone (all kernels of umhs_field_bwd) at the bench shapes; UMHS_BWD_TF=1 selects the fp32-chain kernels.  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd"), os.path.join(ROOT, "tools")]
import torch
from alt_build import use_alt_from_env
use_alt_from_env()
from umhsnerf import ops

dev = "cuda:0"
CASES = {"C2": (6, 31, True, 262144), "C3": (9, 128, True, 524288), "C5": (4, 141, False, 524288)}
for name in os.environ.get("CASES", "C2,C3,C5").split(","):
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name}: N={N} B={B} C={C} spec={spec} TF={os.environ.get('UMHS_BWD_TF', '3')} FUSED={os.environ.get('FUSED', '0')}: field_bwd {us:8.1f} us = {us * 1e3 / N:.2f} ns/sample", flush=True)
