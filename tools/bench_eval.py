"""Eval-image rendering rate (get_outputs_for_camera_ray_bundle) after a short training run on a synthetic scene.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from test_hip_data import _split
from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig
from umhsnerf.umhs_model import UMHSConfig
from umhsnerf.umhs_pipeline import UMHSPipeline
torch.manual_seed(0)
Bd, H = 31, int(os.environ.get("EVAL_HW", "256"))
split, _, _, _ = _split(n=4, H=H, W=H, B=Bd, const=0.6)
dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=4096), device="cuda:0", seed=1, train=split)
cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, background_color="random")
p = UMHSPipeline.from_packed_samples(cfg, "cuda:0", metadata={"wavelengths": list(np.linspace(400, 700, Bd)), "num_classes": 6}, seed=2, datamanager=dm)
with torch.no_grad():
    split.image = p.model.converter(split.hs_image.view(-1, Bd)).view(*split.hs_image.shape[:3], 3).contiguous()
for step in range(300):
    p.get_train_loss_dict(step)
m = p.model.eval()
rb = split.image_rays(0)
with torch.no_grad():
    for _ in range(2):
        out = m.get_outputs_for_camera_ray_bundle(rb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        out = m.get_outputs_for_camera_ray_bundle(rb)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"eval image {H}x{H}: {dt * 1e3:.2f} ms = {H * H / dt / 1e6:.2f} M rays/s, samples/ray {float(out['num_samples_per_ray'].float().mean()):.0f}")
