"""Helper launched by test_hip_distributed.py under torch.distributed.run: 2 ranks over gloo sharing one GPU, or -- with
UMHS_CHECK_BACKEND=nccl on a box with >= 2 GPUs -- one device per rank over RCCL.

Each rank draws its own rays; the gradient that reaches Adam must be bit-identical whether the flat gradient is all-reduced
per finished segment during the backward (FlatGradSink, the default) or in one piece afterwards, and a short training
run must end in bit-identical parameters on both ranks and in both modes."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd"))


def main():
    from bench import synthetic_batch, trained_like_init
    from umhsnerf import ops
    from umhsnerf._ns_compat import packed_ray_samples
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("UMHS_CHECK_BACKEND", "gloo")
    dev_index = rank if backend == "nccl" else 0  # RCCL needs one device per rank; gloo ranks share the GPU
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group("gloo")
    flag_dev = device if backend == "nccl" else "cpu"
    R, S, B, Cn = 256, 32, 31, 6
    bands = list(np.linspace(400, 700, B))
    b = synthetic_batch(R, S, B, seed=7 + rank, device=device)
    rs = packed_ray_samples(b["origins"], b["directions"], b["starts"], b["ends"])
    pinfo = ops.pack_info(b["ray_indices"], R)

    def run(async_reduce, steps):
        torch.manual_seed(11)  # the training background colour is random (umhs_model.py:466-470 in the reference)
        torch.cuda.manual_seed(11)
        mc = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, per_band_outputs=True)
        pipe = UMHSPipeline.from_packed_samples(mc, device, metadata={"wavelengths": bands, "num_classes": Cn}, world_size=world, local_rank=dev_index,
                                                seed=3)
        trained_like_init(pipe.model.field, seed=3)
        dist.broadcast(pipe.model.field.flat.data, src=0)
        with torch.no_grad():
            batch = {"image": pipe.model.converter(b["gt_spectral"]), "hs_image": b["gt_spectral"]}
        field = pipe.model.field
        field._spec()  # creates the sink
        field._grad_sink.async_reduce = async_reduce
        # one backward, gradient as Adam will see it
        pipe.optimizer.zero_grad(set_to_none=True)
        out = pipe.model.get_outputs_from_samples(rs, b["ray_indices"], R, pinfo)
        sum(pipe.model.get_loss_dict(out, batch).values()).backward()
        g = field.flat.grad
        if not field._grad_sink.finish(g):
            assert not async_reduce
            dist.all_reduce(g)
        else:
            assert async_reduce
        g = g.clone()
        for _ in range(steps):
            pipe.train_iteration(rs, b["ray_indices"], R, batch, packed_info=pinfo)
        return g, field.flat.detach().clone()

    g1, p1 = run(True, 5)
    g0, p0 = run(False, 5)
    if os.environ.get("UMHS_CHECK_VERBOSE"):
        g2, p2 = run(True, 5)
        g3, p3 = run(False, 5)
        L = ops.FieldLayout(Cn, B, True, 19)
        tail = L.offset("mlp_base.mlp.layers.0.weight")
        for nm, (a, c) in {"async-sync": (g1, g0), "async-async": (g1, g2), "sync-sync": (g0, g3)}.items():
            d = (a - c).abs()
            print(rank, nm, "table", float(d[:tail].max()), [float(d[i << 20:(i + 4) << 20].max()) for i in range(0, 16, 4)], "tail", float(d[tail:].max()),
                  "ref", float(c[:tail].abs().max()), float(c[tail:].abs().max()), flush=True)
    ok = torch.equal(g1, g0) and torch.equal(p1, p0) and bool(g1.abs().sum() > 0)
    other = p1.clone()
    dist.broadcast(other, src=0)
    ok = ok and torch.equal(other, p1)  # ranks stay in lock step
    # an accumulation window (2 micro-steps, exchange deferred to the second) leaves both ranks with the same parameters as well
    torch.manual_seed(11)
    torch.cuda.manual_seed(11)
    mc = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, per_band_outputs=True)
    acc = UMHSPipeline.from_packed_samples(mc, device, metadata={"wavelengths": bands, "num_classes": Cn}, world_size=world, local_rank=dev_index, seed=3,
                                           gradient_accumulation_steps=2)
    trained_like_init(acc.model.field, seed=3)
    dist.broadcast(acc.model.field.flat.data, src=0)
    with torch.no_grad():
        batch = {"image": acc.model.converter(b["gt_spectral"]), "hs_image": b["gt_spectral"]}
    before = acc.model.field.flat.detach().clone()
    for _ in range(4):
        acc.train_iteration(rs, b["ray_indices"], R, batch, packed_info=pinfo)
    pa = acc.model.field.flat.detach().clone()
    other = pa.clone()
    dist.broadcast(other, src=0)
    ok = ok and torch.equal(other, pa) and not torch.equal(pa, before)
    # the C5 shape of the reference's scripts/rgb+spectral.sh (141 bands, 4 endmembers, no specular head, gradient accumulation 3): the
    # folded compositing backward deposits into the sink three times per window, the exchange runs on the third micro-step only
    B5, C5 = 141, 4
    bands5 = list(np.linspace(400, 1000, B5))
    b5 = synthetic_batch(R, S, B5, seed=17 + rank, device=device)
    rs5 = packed_ray_samples(b5["origins"], b5["directions"], b5["starts"], b5["ends"])
    pinfo5 = ops.pack_info(b5["ray_indices"], R)

    def run_c5(async_reduce):
        torch.manual_seed(13)
        torch.cuda.manual_seed(13)
        mc5 = UMHSConfig(method="rgb+spectral", pred_specular=False, temperature=0.4)
        pipe = UMHSPipeline.from_packed_samples(mc5, device, metadata={"wavelengths": bands5, "num_classes": C5}, world_size=world, local_rank=dev_index,
                                                seed=5, gradient_accumulation_steps=3)
        trained_like_init(pipe.model.field, seed=5)
        dist.broadcast(pipe.model.field.flat.data, src=0)
        pipe.model.field._spec()
        pipe.model.field._grad_sink.async_reduce = async_reduce
        with torch.no_grad():
            batch5 = {"image": pipe.model.converter(b5["gt_spectral"]), "hs_image": b5["gt_spectral"]}
        start = pipe.model.field.flat.detach().clone()
        after = []
        for it in range(6):  # two windows = two optimizer steps
            pipe.train_iteration(rs5, b5["ray_indices"], R, batch5, packed_info=pinfo5)
            after.append(pipe.model.field.flat.detach().clone())
        # parameters move on the third and sixth micro-step only
        moved = [not torch.equal(x, y) for x, y in zip([start] + after[:-1], after)]
        return after[-1], moved

    p5a, moved_a = run_c5(True)
    p5s, moved_s = run_c5(False)
    other = p5a.clone()
    dist.broadcast(other, src=0)
    ok5 = torch.equal(other, p5a) and torch.equal(p5a, p5s) and moved_a == [False, False, True, False, False, True] == moved_s
    if not ok5:
        print(rank, "C5 accumulation window mismatch", moved_a, moved_s, float((p5a - p5s).abs().max()), flush=True)
    ok = ok and ok5
    flag = torch.tensor([1 if ok else 0], device=flag_dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("ASYNC_REDUCE_CHECK", "OK" if int(flag) == 1 else "MISMATCH", backend, dist.get_backend(),
              float((g1 - g0).abs().max()), float((p1 - p0).abs().max()), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
