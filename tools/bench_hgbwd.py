#!/usr/bin/env python3
"""Hash-grid backward alone on the bench batch's positions: prepare (histogram + scans, unthrottled vs as the step issues it) and
apply (scatter + reduce with the dense levels' Adam step), C2 / C5 sample counts.  GPU box.  UMHS_LIB_PATH selects another build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")]
import torch
import bench
from umhsnerf import ops
dev = torch.device("cuda", 0)
log2_T = 19
layout = ops.FieldLayout(6, 31, True, log2_T)
fs = ops.FieldSpec(layout, 0.4, True, scalings=ops.hash_scalings().to(dev))
def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, R in (("C2", 4096), ("C5", 8192)):
    N = R * 64
    b = bench.synthetic_batch(R, 64, 31, seed=42, device=dev)
    _, pos, _ = ops.positions_fwd(b["origins"], b["directions"], b["starts"].view(-1), b["ends"].view(-1), fs)
    d_enc = torch.randn(16, N, 2, device=dev) * 1e-3
    table, p, m, v = (torch.zeros(16 << log2_T, 2, device=dev) for _ in range(4))
    adam = dict(table=p, exp_avg=m, exp_avg_sq=v, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, step=1, level_begin=5)
    prep = lambda: ops.hashgrid_bwd_prepare(pos, fs.scalings, log2_T)
    def both(adam_=adam):
        prep()
        ops.hashgrid_bwd_apply(pos, d_enc, fs.scalings, log2_T, table, True, overwrite=True, adam=adam_)
    tp = timeit(prep)
    print(f"{name}: prepare {tp:7.1f} us   apply+adam {timeit(both) - tp:7.1f} us   apply {timeit(lambda: both(None)) - tp:7.1f} us", flush=True)
