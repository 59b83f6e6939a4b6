"""Longer runs of both training loops (fixed packed batch; data manager + occupancy sampler): finite, decreasing losses."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd"), os.path.join(ROOT, "tests")]
from bench import C2, synthetic_batch, trained_like_init
from umhsnerf import ops
from umhsnerf._ns_compat import packed_ray_samples
from umhsnerf.umhs_model import UMHSConfig
from umhsnerf.umhs_pipeline import UMHSPipeline

dev = torch.device("cuda", 0)
R, S, B, Cn = C2["R"], C2["S"], C2["B"], C2["C"]
mc = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, per_band_outputs=True)
pipe = UMHSPipeline.from_packed_samples(mc, dev, metadata={"wavelengths": list(np.linspace(400, 700, B)), "num_classes": Cn}, seed=42)
trained_like_init(pipe.model.field, seed=42)
b = synthetic_batch(R, S, B, seed=42, device=dev)
rs = packed_ray_samples(b["origins"], b["directions"], b["starts"], b["ends"])
pinfo = ops.pack_info(b["ray_indices"], R)
with torch.no_grad():
    batch = {"image": pipe.model.converter(b["gt_spectral"]), "hs_image": b["gt_spectral"]}
t0 = time.perf_counter()
for i in range(3000):
    out, loss = pipe.train_iteration(rs, b["ray_indices"], R, batch, packed_info=pinfo)
    if i % 500 == 0 or i == 2999:
        l = {k: round(float(v), 5) for k, v in loss.items()}
        print(f"fixed batch step {i}: {l} psnr {float(pipe.model.psnr(out['spectral'], b['gt_spectral'])):.3f}", flush=True)
torch.cuda.synchronize()
print(f"3000 steps in {time.perf_counter() - t0:.2f} s; params finite: {bool(torch.isfinite(pipe.model.field.flat).all())}", flush=True)

from test_hip_data import _split
from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig
torch.manual_seed(0)
Bd = 8
split, _, _, _ = _split(n=6, B=Bd, const=0.6)
dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=4096), device="cuda:0", seed=1, train=split)
cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, background_color="black")
p2 = UMHSPipeline.from_packed_samples(cfg, "cuda:0", metadata={"wavelengths": list(np.linspace(420, 680, Bd)), "num_classes": 3}, seed=2, datamanager=dm)
with torch.no_grad():
    split.image = p2.model.converter(split.hs_image.view(-1, Bd)).view(*split.hs_image.shape[:3], 3).contiguous()
t0 = time.perf_counter()
for step in range(2000):
    _, loss_dict, metrics = p2.get_train_loss_dict(step)
    if step % 400 == 0 or step == 1999:
        print(f"datamanager step {step}: loss {float(sum(v.detach() for v in loss_dict.values())):.5f} psnr_spectral {float(metrics['psnr_spectral']):.2f} "
              f"samples {int(metrics['num_samples_per_batch'])}", flush=True)
torch.cuda.synchronize()
print(f"2000 sampler-driven steps in {time.perf_counter() - t0:.2f} s; params finite: {bool(torch.isfinite(p2.model.field.flat).all())}")
md, _ = p2.get_eval_image_metrics_and_images(0)
print("eval image metrics:", {k: round(v, 4) for k, v in md.items()})
