"""cProfile of the host side of the sampler-driven training step (where does the Python time between the syncs go?).  GPU box."""
import cProfile, pstats, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
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
pr = cProfile.Profile()
pr.enable()
for step in range(300, 500):
    p.get_train_loss_dict(step)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
