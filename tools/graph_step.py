#!/usr/bin/env python3
"""What would a hipGraph of the synthetic training step buy?  Captures ONE `UMHSPipeline.train_iteration` (C2 shape by default) with
torch.cuda.graph (= hipGraph on ROCm) and replays it next to the eager loop.  TIMING ONLY: Adam's step-dependent scalars (bias
corrections, decayed learning rate) are launch arguments and stay frozen at the captured step, so the replayed updates are not a valid
training run -- a product version would first have to move them into a device buffer.  GPU box:
    python tools/graph_step.py [C2|C3|C5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")]
import numpy as np, torch
from bench import CONFIGS, synthetic_batch, trained_like_init
from umhsnerf import ops
from umhsnerf._ns_compat import packed_ray_samples
from umhsnerf.umhs_model import UMHSConfig
from umhsnerf.umhs_pipeline import UMHSPipeline

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = CONFIGS[name]
dev = torch.device("cuda", 0)
R, S, B, Cn = cfg["R"], cfg["S"], cfg["B"], cfg["C"]
mc = UMHSConfig(method=cfg["method"], pred_specular=cfg["pred_specular"], temperature=cfg["temperature"], per_band_outputs=True)
pipe = UMHSPipeline.from_packed_samples(mc, dev, metadata={"wavelengths": list(np.linspace(400, 700, B)), "num_classes": Cn}, seed=42)
trained_like_init(pipe.model.field, seed=42)
b = synthetic_batch(R, S, B, seed=42, device=dev)
rs = packed_ray_samples(b["origins"], b["directions"], b["starts"], b["ends"])
pinfo = ops.pack_info(b["ray_indices"], R)
with torch.no_grad():
    batch = {"image": pipe.model.converter(b["gt_spectral"]), "hs_image": b["gt_spectral"]}
step = lambda: pipe.train_iteration(rs, b["ray_indices"], R, batch, packed_info=pinfo)


def timed(fn, n=100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(20):
    step()
eager = [timed(step) for _ in range(3)]
print(f"{name} eager : {min(eager):.4f} ms per step (3 x 100 steps: {', '.join(f'{x:.4f}' for x in eager)})", flush=True)
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    replay = [timed(g.replay) for _ in range(3)]
    print(f"{name} graph : {min(replay):.4f} ms per replay ({', '.join(f'{x:.4f}' for x in replay)})  -> {100 * (1 - min(replay) / min(eager)):.1f} % of the eager step", flush=True)
except Exception as e:  # a launch the capture cannot hold (a sync, an un-joined side stream, ...)
    print(f"{name} graph : capture failed: {type(e).__name__}: {str(e)[:400]}", flush=True)
