"""Training-step time of the other BASELINE configs (parity-test cases, not bench lines): C3 (128 bands, 9 endmembers, 8192 rays) and
C5 (141 bands, 4 endmembers, no specular), same step definition as bench.py.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")]
import numpy as np, torch
import bench
from umhsnerf import ops
from umhsnerf._ns_compat import packed_ray_samples
from umhsnerf.umhs_model import UMHSConfig
from umhsnerf.umhs_pipeline import UMHSPipeline
dev = torch.device("cuda", 0)
CFGS = {"C2": bench.C2,
        "C3": dict(R=8192, S=64, B=128, C=9, temperature=0.3, pred_specular=True, method="rgb+spectral"),
        "C5": dict(R=8192, S=64, B=141, C=4, temperature=0.7, pred_specular=False, method="rgb+spectral")}
for name, c in CFGS.items():
    mc = UMHSConfig(method=c["method"], pred_specular=c["pred_specular"], temperature=c["temperature"], per_band_outputs=True)
    pipe = UMHSPipeline.from_packed_samples(mc, dev, metadata={"wavelengths": list(np.linspace(400, 700, c["B"])), "num_classes": c["C"]}, seed=42)
    bench.trained_like_init(pipe.model.field, seed=42)
    b = bench.synthetic_batch(c["R"], c["S"], c["B"], seed=42, device=dev)
    rs = packed_ray_samples(b["origins"], b["directions"], b["starts"], b["ends"])
    pinfo = ops.pack_info(b["ray_indices"], c["R"])
    with torch.no_grad():
        batch = {"image": pipe.model.converter(b["gt_spectral"]), "hs_image": b["gt_spectral"]}
    for _ in range(10):
        pipe.train_iteration(rs, b["ray_indices"], c["R"], batch, packed_info=pinfo)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 100
    for _ in range(K):
        pipe.train_iteration(rs, b["ray_indices"], c["R"], batch, packed_info=pinfo)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"{name}: R={c['R']} B={c['B']} C={c['C']} spec={c['pred_specular']}: {dt * 1e3:.3f} ms/step = {c['R'] / dt / 1e6:.2f} M rays/s", flush=True)
    del pipe
    torch.cuda.empty_cache()
