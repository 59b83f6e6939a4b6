#!/usr/bin/env python3
"""Time the field kernels alone at several N (fixed cost vs per-sample cost).  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")]
import numpy as np, torch
from umhsnerf import ops

dev = "cuda:0"
C, B, spec = int(os.environ.get("C", 6)), int(os.environ.get("B", 31)), os.environ.get("SPEC", "1") == "1"
layout = ops.FieldLayout(C, B, spec, 19)
g = torch.Generator().manual_seed(0)
flat = ((torch.rand(layout.total, generator=g) - 0.5) * 0.5).to(dev)
fs = ops.FieldSpec(layout, 0.4, True, scalings=ops.hash_scalings().to(dev))

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for N in ([int(os.environ.get('N', 262144))] if os.environ.get('ONLY') else [16384, 65536, 262144, 1048576]):
    enc = (torch.rand(16, N, 2, device=dev) - 0.5)
    wpos = torch.rand(N, 3, device=dev) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    sel = torch.ones(N, device=dev)
    out = ops.field_fwd(fs, flat, enc, True, wpos, dirs, sel, want_emb=True, want_logits=True)
    dsig, dspec = torch.rand(N, device=dev), torch.rand(N, B, device=dev)
    dflat = torch.zeros_like(flat)
    t_f = timeit(lambda: ops.field_fwd(fs, flat, enc, True, wpos, dirs, sel, want_emb=True))
    t_d = timeit(lambda: ops.field_fwd(fs, flat, enc, True, None, None, sel, density_only=True))
    t_b = timeit(lambda: ops.field_bwd(fs, flat, enc, True, wpos, dirs, sel, out["sigma_raw"], out["emb"], dsig, dspec, None, dflat,
                                       feat_logits=out["feat_logits"]))
    pos01 = torch.rand(N, 3, device=dev)
    table = layout.view(flat, "mlp_base.encoder.hash_table")
    t_h = timeit(lambda: ops.hashgrid_fwd(pos01, table, fs.scalings, 19, True))
    d_tab = torch.zeros_like(table)
    t_hb = timeit(lambda: ops.hashgrid_bwd(pos01, enc, fs.scalings, 19, d_tab, True))
    print(f"N={N:8d}  field_fwd {t_f:8.1f} us  density_only {t_d:7.1f} us  field_bwd(all) {t_b:8.1f} us  hash_fwd {t_h:7.1f} us  hash_bwd {t_hb:7.1f} us", flush=True)
