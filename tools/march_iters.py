"""Marcher diagnostics on the sampler-driven scene: per-ray sample counts and the time of one walk (empty / full grid).  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, ctypes as C
from test_hip_data import _split
from umhsnerf import _hip
from umhsnerf._hip import ptr
from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig
from umhsnerf.umhs_model import UMHSConfig
from umhsnerf.umhs_pipeline import UMHSPipeline
torch.manual_seed(0)
Bd = 31
split, _, _, _ = _split(n=6, H=64, W=64, B=Bd, const=0.6)
dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=4096), device="cuda:0", seed=1, train=split)
cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, background_color="random")
p = UMHSPipeline.from_packed_samples(cfg, "cuda:0", metadata={"wavelengths": list(np.linspace(400, 700, Bd)), "num_classes": 6}, seed=2, datamanager=dm)
m = p.model
m.occupancy_grid.mark_all_occupied() if os.environ.get("ALLOCC") else None
rb, _ = dm.next_train(0)
g = m.occupancy_grid
o, d = rb.origins.contiguous(), rb.directions.contiguous()
R = o.shape[0]
counts = torch.empty(R, dtype=torch.int64, device=o.device)
roi = (C.c_float * 6)(*g._roi)
c = m.config
print("levels", g.levels, "res", g.res, "roi", g._roi, "near", c.near_plane, "far", c.far_plane, "step", c.render_step_size, "cone", c.cone_angle, "occupied frac", float(g.binaries.float().mean()))
_hip.check(_hip.lib().umhs_march_count(ptr(o), ptr(d), R, ptr(g.binaries.view(torch.uint8)), roi, g.levels, g.res, c.near_plane, c.far_plane if c.far_plane else 1e10, c.render_step_size, c.cone_angle, None, None, None, 0.0, ptr(counts), None, 0, _hip.stream()), "x")
cc = counts.float().cpu()
print("per-ray value: mean %.1f median %.1f p99 %.1f max %.0f" % (cc.mean(), cc.median(), cc.quantile(0.99), cc.max()))
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
lib = _hip.lib()
call = lambda: lib.umhs_march_count(ptr(o), ptr(d), R, ptr(g.binaries.view(torch.uint8)), roi, g.levels, g.res, c.near_plane, 1000.0, c.render_step_size, c.cone_angle, None, None, None, 0.0, ptr(counts), None, 0, _hip.stream())
print("march_count, empty grid: %.1f us" % timeit(call))
g.mark_all_occupied()
print("march_count, full grid : %.1f us  (samples/ray mean %.0f)" % (timeit(call), float(counts.float().mean())))
