"""Cost of running the hash-grid backward's apply half (scatter + reduce) in 1 / 2 / 4 / 8 level groups (what the multi-GPU path does to
start the all-reduce of finished levels early).  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")]
import torch
from umhsnerf import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
N, log2_T = 262144, 19
# ray-coherent positions like the bench batch: 4096 rays x 64 samples
o = torch.rand(4096, 1, 3, device=dev) * 0.5 + 0.25
d = torch.nn.functional.normalize(torch.randn(4096, 1, 3, device=dev), dim=-1)
pos = (o + d * torch.linspace(0, 0.25, 64, device=dev).view(1, 64, 1)).clamp(1e-4, 1 - 1e-4).reshape(-1, 3).contiguous()
sc = ops.hash_scalings(16, 16, 2048).to(dev)
d_enc = torch.randn(16, N, 2, device=dev)
table = torch.zeros(16 << log2_T, 2, device=dev)
def run(groups):
    ops.hashgrid_bwd_prepare(pos, sc, log2_T)
    g = 16 // groups
    for l0 in range(0, 16, g):
        ops.hashgrid_bwd_apply(pos, d_enc, sc, log2_T, table, True, overwrite=True, level_begin=l0, level_count=g)
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
base = timeit(lambda: ops.hashgrid_bwd_prepare(pos, sc, log2_T))
for groups in (1, 2, 4, 8):
    print(f"groups {groups}: apply {timeit(lambda: run(groups)) - base:7.1f} us")
