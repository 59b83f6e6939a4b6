"""N>1 path on CPU: two gloo ranks exercise the ray sharding and the flat-gradient all-reduce that bench.py / UMHSAdam
use over RCCL.  Equivalence checked: averaging per-rank gradients of per-rank ray shards == gradient of the global batch
(equal shard sizes), and both ranks end with identical parameters after the same Adam step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import torch_ref as T


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from umhsnerf import parallel

        torch.set_num_threads(2)
        R, S, B, C = 16, 8, 5, 3
        p = T.FieldParams(C, B, True, log2_hashmap_size=10, table_scale=0.5, seed=3)
        flat = torch.cat([v.detach().reshape(-1) for v in p.parameters()])
        parallel.broadcast_params(flat, 0)
        batch = T.synthetic_batch(R, S, B, seed=5)
        M = T.colour_matrix(np.linspace(400, 700, B))
        b0, b1 = parallel.shard_rays(R, rank, world)
        sel = (batch["ray_indices"] >= b0) & (batch["ray_indices"] < b1)
        out_ = T.model_outputs(p, batch["origins"][sel], batch["directions"][sel], batch["starts"][sel], batch["ends"][sel],
                               batch["ray_indices"][sel] - b0, b1 - b0, 0.4, M)
        loss = 5 * torch.nn.functional.mse_loss(out_["spectral"], batch["gt_spectral"][b0:b1])
        params = list(p.parameters())
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        g = torch.cat([(gi if gi is not None else torch.zeros_like(v)).reshape(-1) for gi, v in zip(grads, params)])
        scale = parallel.allreduce_flat_grad(g)
        assert scale == 1.0 / world
        g = g * scale
        m, v = torch.zeros_like(flat), torch.zeros_like(flat)
        T.adam_step([flat], [g], [m], [v], 1, 2e-2)
        if rank == 0:
            # single-process gradient of the GLOBAL batch (mean over all rays) must equal the averaged shard gradients
            og = T.model_outputs(p, batch["origins"], batch["directions"], batch["starts"], batch["ends"], batch["ray_indices"], R, 0.4, M)
            lg = 5 * torch.nn.functional.mse_loss(og["spectral"], batch["gt_spectral"])
            gg = torch.autograd.grad(lg, params, allow_unused=True)
            gg = torch.cat([(gi if gi is not None else torch.zeros_like(v_)).reshape(-1) for gi, v_ in zip(gg, params)])
            np.testing.assert_allclose(g.numpy(), gg.numpy(), rtol=1e-4, atol=1e-7)
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert torch.equal(gathered[0], gathered[1])  # replicas stay bit-identical
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce_matches_global_batch():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert dict(out) == {0: 1, 1: 1}


def test_shard_rays_partitions_exactly():
    from umhsnerf.parallel import rank_seed, shard_rays

    for R, W in [(4096, 8), (65536, 8), (10, 3), (7, 8)]:
        sl = [shard_rays(R, r, W) for r in range(W)]
        assert sl[0][0] == 0 and sl[-1][1] == R and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        assert max(e - b for b, e in sl) - min(e - b for b, e in sl) <= 1
    assert rank_seed(42, 3) == 45


def _sink_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from umhsnerf import parallel
        from umhsnerf.optim import UMHSAdam  # noqa: F401  (import only: the fused step itself needs the GPU)

        p = torch.nn.Parameter(torch.zeros(64))
        sink = parallel.FlatGradSink(p, level_groups=4)
        assert sink.groups(16) == [(0, 4), (4, 4), (8, 4), (12, 4)]
        sink.sparse_levels = 5  # the compact message of the sparse coarse levels must fit in the first group
        assert sink.groups(16) == [(0, 8), (8, 4), (12, 4)]
        sink.sparse_levels = 0
        for it in range(2):  # the buffer is reused from one backward to the next
            assert sink.owns_next_backward()
            buf = sink.begin()
            buf.copy_(torch.arange(64.0) * (rank + 1) + it)
            sink.segment_done(buf[48:])
            for l0, cnt in sink.groups(12):
                sink.segment_done(buf[l0 * 4:(l0 + cnt) * 4])
            sink.commit()
            assert p.grad is buf and sink.owns_next_backward()  # a further backward would ADD to the buffer (gradient accumulation) ...
            with pytest.raises(RuntimeError, match="already started its all-reduce"):
                sink.begin()  # ... which more than one rank may only do when the exchange was deferred to the last micro-step
            assert sink.finish(p.grad)
            torch.testing.assert_close(p.grad, torch.arange(64.0) * 3 + 2 * it)
            assert not sink.finish(p.grad)  # nothing pending any more
            p.grad = None
        # an accumulation window of three micro-steps: no exchange (one level group) until the last, which reduces the SUM
        for micro in range(3):
            sink.defer_reduce = micro < 2
            assert sink.owns_next_backward()
            buf = sink.begin()
            assert sink.accumulating == (micro > 0)
            assert sink.groups(12) == ([(0, 12)] if micro < 2 else [(0, 3), (3, 3), (6, 3), (9, 3)])
            if micro == 0:
                buf.copy_(torch.full((64,), float(rank + 1)))
            else:
                buf.add_(float(rank + 1))  # the kernels' += mode
            sink.segment_done(buf[48:])
            for l0, cnt in sink.groups(12):
                sink.segment_done(buf[l0 * 4:(l0 + cnt) * 4])
            sink.commit()
            assert bool(sink.works) == (micro == 2)
        assert sink.finish(p.grad)
        torch.testing.assert_close(p.grad, torch.full((64,), 9.0))  # 3 micro-steps x (1 + 2)
        p.grad = torch.zeros(64)  # a gradient somebody else put there: autograd's job, not the sink's
        assert not sink.owns_next_backward()
        p.grad = None
        sink.async_reduce = False
        assert sink.groups(16) == [(0, 16)]
        sink.begin().fill_(1.0)
        sink.segment_done(sink.buffer)
        sink.commit()
        assert not sink.finish(p.grad) and parallel.allreduce_flat_grad(p.grad) == 0.5 and float(p.grad[0]) == 2.0
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_flat_grad_sink_reduces_segments_early_and_falls_back():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sink_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert dict(out) == {0: 1, 1: 1}
