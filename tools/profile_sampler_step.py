"""Per-phase timing of the sampler-driven training step (occupancy grid update, marching, visibility, hot path)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd"), os.path.join(ROOT, "tests")]
from test_hip_data import _split
from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig
from umhsnerf.umhs_model import UMHSConfig
from umhsnerf.umhs_pipeline import UMHSPipeline
torch.manual_seed(0)
Bd = 31
split, _, _, _ = _split(n=6, H=64, W=64, B=Bd, const=0.6)
dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=4096), device="cuda:0", seed=1, train=split)
cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, background_color="random")
p = UMHSPipeline.from_packed_samples(cfg, "cuda:0", metadata={"wavelengths": list(np.linspace(400, 700, Bd)), "num_classes": 6}, seed=2, datamanager=dm)
with torch.no_grad():
    split.image = p.model.converter(split.hs_image.view(-1, Bd)).view(*split.hs_image.shape[:3], 3).contiguous()
for step in range(300):
    p.get_train_loss_dict(step)
torch.cuda.synchronize()
def t(fn, reps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
m = p.model
ms, _ = t(lambda: m.update_occupancy_grid(300)); print(f"update_occupancy_grid {ms:.3f} ms")
ms, (rb, batch) = t(lambda: dm.next_train(0)); print(f"next_train           {ms:.3f} ms")
ms, (rs, ri) = t(lambda: m.sample(rb)); print(f"sample (march+vis)   {ms:.3f} ms   samples {ri.numel()}")
def fb():
    p.optimizer.zero_grad(set_to_none=True)
    return m.forward_backward_from_samples(rs, ri, len(rb), batch)
ms, _ = t(fb); print(f"forward_backward     {ms:.3f} ms")
ms, _ = t(lambda: p.optimizer.step()); print(f"optimizer            {ms:.3f} ms")
step = [300]
def full():
    step[0] += 1
    return p.get_train_loss_dict(step[0])
ms, _ = t(full, 50); print(f"full step            {ms:.3f} ms")
