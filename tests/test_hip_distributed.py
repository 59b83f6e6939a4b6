"""N>1 flow on the GPU box: 2 ranks sharing the one GPU with gloo as the transport, and -- whenever the box has two or more
GPUs -- 2 ranks with one device each over RCCL (backend "nccl").  Both check the early per-segment gradient reduction
(parallel.FlatGradSink) against the plain post-backward all-reduce, bit for bit, and an accumulation window."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_async_segment_reduce_is_bit_identical_to_plain_allreduce():
    from umhsnerf import ops

    # group-wise hash-grid backward == one call over all levels (single process)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    n = 5000
    pos = torch.rand(n, 3, generator=g).to(dev)
    d_enc = torch.randn(n, 32, generator=g).to(dev)
    sc = ops.hash_scalings(ops.NUM_LEVELS, 16, 2048).to(dev)
    full = torch.empty(ops.NUM_LEVELS << 19, 2, device=dev)
    ops.hashgrid_bwd(pos, d_enc, sc, 19, full, True, overwrite=True)
    parts = torch.full_like(full, float("nan"))
    for l0 in range(0, 16, 4):
        ops.hashgrid_bwd(pos, d_enc, sc, 19, parts, True, overwrite=True, level_begin=l0, level_count=4)
    assert torch.equal(full, parts)
    parts.fill_(1.0)  # accumulate mode on a sub-range leaves the other levels alone
    ops.hashgrid_bwd(pos, d_enc, sc, 19, parts, True, overwrite=False, level_begin=12, level_count=4)
    assert torch.equal(parts[: 12 << 19], torch.ones_like(parts[: 12 << 19]))
    torch.testing.assert_close(parts[12 << 19:], full[12 << 19:] + 1.0, rtol=1e-6, atol=1e-7)

    # coarse levels: every row outside the enumerated live set keeps an exactly zero gradient (what the compact all-reduce relies on)
    from umhsnerf import parallel

    ns, rows = parallel.live_hash_rows(sc.cpu(), 19)
    assert ns == 5 and rows.numel() == 288066
    big = torch.rand(200000, 3, generator=g).to(dev)
    big[:50] = 0.0  # unselected samples sit at the origin corner
    big[50:60] = 0.999999
    gt = torch.empty(ops.NUM_LEVELS << 19, 2, device=dev)
    ops.hashgrid_bwd(big, torch.randn(16, 200000, 2, generator=g).to(dev), sc, 19, gt, True, overwrite=True)
    dead = torch.ones(ns << 19, dtype=torch.bool, device=dev)
    dead[rows.to(dev)] = False
    assert float(gt[: ns << 19][dead].abs().max()) == 0.0 and float(gt[: ns << 19][~dead].abs().max()) > 0
    touched = (gt[: ns << 19].abs().sum(-1) > 0)
    assert float(touched.sum()) > 0.95 * rows.numel()  # ...and the set is tight: random positions reach nearly all of it

    _two_ranks("gloo", 29533)


def _two_ranks(backend: str, port: int):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", UMHS_CHECK_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(HERE, "dist_async_reduce_check.py")], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and f"ASYNC_REDUCE_CHECK OK {backend}" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: runs only on a box with >= 2 GPUs")
def test_async_segment_reduce_over_rccl_one_device_per_rank():
    """The same check with backend "nccl" (= RCCL on ROCm), one GPU per rank: the first place RCCL itself executes the sink's
    per-segment async all-reduces (the 1-GPU boxes of the build pool can only run the gloo variant above)."""
    _two_ranks("nccl", 29534)
