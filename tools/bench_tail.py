"""Time the small per-ray kernels of the training step (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")]
import torch
from umhsnerf import ops
dev = "cuda:0"
R, B, C, S = (int(x) for x in os.environ.get("SHAPE", "4096,31,6,64").split(","))  # SHAPE=8192,128,9,64: C3
N = R * S
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.rand(*s, generator=g).to(dev)
spec, gt, M, E, acc, depth, colors, gt_rgb, bg = rnd(R, B), rnd(R, B), rnd(B, 3) / B, rnd(C, B), rnd(R), rnd(R), rnd(C, 3), rnd(R, 3), rnd(R, 3)
t0 = rnd(N); t1 = t0 + 0.01
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
mm = ops.tmid_minmax(t0, t1)
print(f"tmid_minmax     {timeit(lambda: ops.tmid_minmax(t0, t1)):7.1f} us")
print(f"ray_train_tail  {timeit(lambda: ops.ray_train_tail(spec, M, E, acc, depth, mm, colors, gt, gt_rgb, bg, 0.2, 5.0, 1.0, True)):7.1f} us")
print(f"ray_epilogue    {timeit(lambda: ops.ray_epilogue_fwd(spec, M, E, acc, depth, mm, colors, 0.2)):7.1f} us")
sigma = rnd(N) * 20
pinfo = torch.stack([torch.arange(R, device=dev) * S, torch.full((R,), S, device=dev)], 1).contiguous()
vals = [rnd(N, B), rnd(N, B), rnd(N, B), rnd(N, C)]
w, a_, d_, outs = ops.composite_fwd(sigma, t0, t1, pinfo, vals)
print(f"composite_fwd   {timeit(lambda: ops.composite_fwd(sigma, t0, t1, pinfo, vals)):7.1f} us")
dsp, dacc = rnd(R, B), rnd(R)
print(f"composite_bwd   {timeit(lambda: ops.composite_bwd(sigma, t0, t1, pinfo, w, vals[:1], [dsp], [True], dacc, True)):7.1f} us")
o, d = rnd(N, 3), rnd(N, 3)
fs = ops.FieldSpec(ops.FieldLayout(C, B, True, 19), 0.4, True, scalings=ops.hash_scalings().to(dev))
print(f"positions_fwd   {timeit(lambda: ops.positions_fwd(o, d, t0, t1, fs)):7.1f} us")
