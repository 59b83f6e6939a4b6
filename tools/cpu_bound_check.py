"""How much of the step is host-side launch time?  Enqueue-only time vs synchronised time of the C2 train step."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd"))
from bench import C2, synthetic_batch, trained_like_init
from umhsnerf import ops
from umhsnerf._ns_compat import packed_ray_samples
from umhsnerf.umhs_model import UMHSConfig
from umhsnerf.umhs_pipeline import UMHSPipeline

dev = torch.device("cuda", 0)
cfg = C2; R, S, B, Cn = cfg["R"], cfg["S"], cfg["B"], cfg["C"]
mc = UMHSConfig(method=cfg["method"], pred_specular=cfg["pred_specular"], temperature=cfg["temperature"], per_band_outputs=True)
pipe = UMHSPipeline.from_packed_samples(mc, dev, metadata={"wavelengths": list(np.linspace(400, 700, B)), "num_classes": Cn}, seed=42)
trained_like_init(pipe.model.field, seed=42)
b = synthetic_batch(R, S, B, seed=42, device=dev)
rs = packed_ray_samples(b["origins"], b["directions"], b["starts"], b["ends"])
pinfo = ops.pack_info(b["ray_indices"], R)
with torch.no_grad():
    batch = {"image": pipe.model.converter(b["gt_spectral"]), "hs_image": b["gt_spectral"]}
step = lambda: pipe.train_iteration(rs, b["ray_indices"], R, batch, packed_info=pinfo)
for _ in range(10): step()
torch.cuda.synchronize()
for K in (20, 100):
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K}: enqueue {1e3*(t1-t0)/K:.3f} ms/step, total {1e3*(t2-t0)/K:.3f} ms/step", flush=True)
