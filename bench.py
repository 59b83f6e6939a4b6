#!/usr/bin/env python3
"""bench.py -- train rays/sec of the UMHS hot path on N MI355X GPUs (one process per GPU over RCCL).

A "step" is one full training iteration of the reference on a synthetic packed batch of the hotdog-shaped
config C2 (B=31 bands, C=6 endmembers, temperature 0.4, specular on, rgb+spectral, 4096 rays x 64
samples per GPU): field forward -> compositing of every stream -> spec->sRGB -> losses -> backward -> (RCCL
all-reduce of the flat gradient when N>1) -> fused Adam with the endmember clamp.  Inputs are resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed live with HIP events on the
launch stream; `cpu_baseline` is the oracle (CPU restatement of the reference's torch path) timed on this
box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver does dmabuf IPC only (RCCL between the ranks of a node)
ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist

C2 = dict(R=4096, S=64, B=31, C=6, temperature=0.4, pred_specular=True, method="rgb+spectral")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
ATOMIC_PEAK_GBS = 1300.0  # ibid. "Global float atomics": chip-wide ~1.3 TB/s of added bytes


def synthetic_batch(R, S, B, seed, device):
    """Same recipe as oracle.torch_ref.synthetic_batch, built with torch on the host then moved."""
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(R, 3, generator=g)
    u = u / u.norm(dim=-1, keepdim=True)
    o = (torch.rand(R, 3, generator=g) * 2 - 1) * 0.5 + 2.5 * u
    d = -o + torch.randn(R, 3, generator=g) * 0.1
    d = d / d.norm(dim=-1, keepdim=True)
    ray_indices = torch.arange(R).repeat_interleave(S)
    step = float(np.sqrt(12.0) / 1000.0)
    t_near = (o.norm(dim=-1) - 1.3).clamp(min=0.05) + torch.rand(R, generator=g) * step
    k = torch.arange(R * S) % S
    t0 = t_near[ray_indices] + k * (2.6 / S)
    t1 = t0 + step * (1 + 0.004 * t0)
    b = dict(origins=o[ray_indices].contiguous(), directions=d[ray_indices].contiguous(), starts=t0[:, None].contiguous(),
             ends=t1[:, None].contiguous(), ray_indices=ray_indices, gt_spectral=torch.rand(R, B, generator=g))
    return {k_: v.to(device) for k_, v in b.items()}


def trained_like_init(field, seed):
    """Untrained tables give constant outputs; use a 'trained-like' state: table U(+-0.5), endmembers in
    [0,1], density bias raised so ~30 % of the samples have alpha > 0.01."""
    g = torch.Generator().manual_seed(seed)
    L = field.layout
    with torch.no_grad():
        flat = field.flat.data.cpu()
        tab = L.view(flat, "mlp_base.encoder.hash_table")
        tab.copy_((torch.rand(tab.shape, generator=g) * 2 - 1) * 0.5)
        L.view(flat, "endmembers").copy_(torch.rand(L.entries["endmembers"][1], generator=g))
        L.view(flat, "mlp_base.mlp.layers.1.bias")[0] += 1.5
        field.flat.data.copy_(flat.to(field.flat.device))


class KernelTimer:
    """HIP events around each C-ABI call on the launch stream (torch's current stream is the stream every op uses)."""

    def __init__(self, ops):
        self.ops, self.records, self.orig = ops, {}, {}
        self.enabled, self.only = False, None

    def wrap(self, name):
        fn = getattr(self.ops, name)
        self.orig[name] = fn

        def timed(*a, **k):
            if not self.enabled or (self.only is not None and name not in self.only):
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.records.setdefault(name, []).append((e0, e1))
            return out

        setattr(self.ops, name, timed)

    def summary(self):
        return {k: (float(np.mean([a.elapsed_time(b) for a, b in v])), len(v)) for k, v in self.records.items()}


def cpu_baseline(cfg, seed, budget_s=15.0):
    """Oracle train step (fwd + loss + bwd + Adam) on the host cores: the 'reference CPU path' stand-in."""
    from oracle import torch_ref as T

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))  # a 1-GPU box owns a 16-core share of the host; more threads only oversubscribe
    torch.set_num_threads(cores)
    R = 1024  # bounded sample: 1024 of the 4096 rays (x64 samples), same distributions
    p = T.FieldParams(cfg["C"], cfg["B"], cfg["pred_specular"], cfg["method"], table_scale=0.5, seed=seed)
    with torch.no_grad():
        p.base_b[1][0] += 1.5
    b = T.synthetic_batch(R, cfg["S"], cfg["B"], seed=seed)
    M = T.colour_matrix(np.linspace(400, 700, cfg["B"]))
    gt_rgb = T.colour_system(b["gt_spectral"], M)
    params = [v for _, v in p.named_parameters()]
    ms, vs = [torch.zeros_like(v) for v in params], [torch.zeros_like(v) for v in params]

    def step(i):
        out = T.model_outputs(p, b["origins"], b["directions"], b["starts"], b["ends"], b["ray_indices"], R, cfg["temperature"], M)
        loss = sum(T.model_loss(out, b["gt_spectral"], gt_rgb, b["bg_random"], cfg["method"]).values())
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(v) for g, v in zip(grads, params)]
        with torch.no_grad():
            T.adam_step(params, grads, ms, vs, i, 2e-2)
            p.endmembers.clamp_(0, 1)

    print(f"[bench] cpu_baseline: {cores} threads, {R} rays x {cfg['S']} samples ...", file=sys.stderr, flush=True)
    step(1)  # warm-up
    t0, n = time.perf_counter(), 0
    while n < 2 or (time.perf_counter() - t0 < budget_s and n < 8):
        n += 1
        step(n + 1)
        print(f"[bench] cpu_baseline step {n}: {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
    dt = (time.perf_counter() - t0) / n
    return dict(value=R / dt, unit="rays/s", cores=cores, kind="port",
                sample=f"oracle/torch_ref.py fp32 train step (fwd+loss+bwd+Adam), {R} of {cfg['R']} rays x {cfg['S']} samples, {n} steps, {dt:.2f} s/step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hot path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)  # (rehearsals with more ranks than GPUs share a device; the driver runs one rank per GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("UMHS_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only for 1-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from umhsnerf import ops
    from umhsnerf._ns_compat import packed_ray_samples
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    cfg = C2
    R, S, B, Cn = cfg["R"], cfg["S"], cfg["B"], cfg["C"]
    N = R * S
    bands = list(np.linspace(400, 700, B))
    mc = UMHSConfig(method=cfg["method"], pred_specular=cfg["pred_specular"], temperature=cfg["temperature"], per_band_outputs=True)
    pipe = UMHSPipeline.from_packed_samples(mc, device, metadata={"wavelengths": bands, "num_classes": Cn}, world_size=world, local_rank=local, seed=42)
    trained_like_init(pipe.model.field, seed=42)
    if world > 1:
        dist.broadcast(pipe.model.field.flat.data, src=0)
    b = synthetic_batch(R, S, B, seed=42 + rank, device=device)  # every rank draws its own rays (weak scaling)
    rs = packed_ray_samples(b["origins"], b["directions"], b["starts"], b["ends"])
    pinfo = ops.pack_info(b["ray_indices"], R)
    with torch.no_grad():
        gt_rgb = pipe.model.converter(b["gt_spectral"])
    batch = {"image": gt_rgb, "hs_image": b["gt_spectral"]}

    timer = KernelTimer(ops)
    OPS = ("positions_fwd", "hashgrid_fwd", "field_fwd", "composite_fwd", "tmid_minmax", "ray_train_tail", "composite_bwd", "field_bwd",
           "hashgrid_bwd", "hashgrid_bwd_prepare", "hashgrid_bwd_apply", "field_fwd_prepare", "field_bwd_prepare", "adam_step",
           "adam_step_rows", "adam_step_rows_range")  # (one GPU: the dense hash levels' Adam step rides in hashgrid_bwd_apply)
    for name in OPS:
        timer.wrap(name)
    # Inside the timed region only the dominant operator carries HIP events (2 per step): an event is a barrier packet on the
    # queue, and a pair around every one of the ~12 operators costs ~12 % of the step.  The full per-operator table comes from
    # a separate, untimed pass over the same step.
    timer.only = {"field_bwd"}

    def step():
        return pipe.train_iteration(rs, b["ray_indices"], R, batch, packed_info=pinfo)

    for _ in range(args.warmup):
        step()
    timer.enabled = True
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outputs, loss_dict = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.enabled = False
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    dom_live = timer.summary()
    timer.records.clear()
    timer.only, timer.enabled = None, True  # untimed breakdown pass (all ranks: the step contains the collectives)
    for _ in range(min(args.steps, 10)):
        step()
    torch.cuda.synchronize()
    timer.enabled = False

    if rank == 0:
        ksum = timer.summary()
        ksum.update(dom_live)  # the dominant operator's figure is the one measured inside the timed region
        ms_step = dt / args.steps * 1e3
        psnr = float(pipe.model.psnr(outputs["spectral"].detach(), b["gt_spectral"]))
        loss_dict = {k: v.detach() for k, v in loss_dict.items()}
        print(f"[bench] gpu: {ms_step:.3f} ms/step, {R * world * args.steps / dt:.0f} rays/s", file=sys.stderr, flush=True)
        # algorithmic bytes per launch (1024 B/sample of hash-grid gather resp. gradient scatter,
        # + the level-major feature rows (128 B) and positions (12 B) each kernel streams)
        alg = {
            "hashgrid_bwd": N * (1024 + 128 + 12),
            "hashgrid_bwd_apply": N * (1024 + 128 + 12),
            "hashgrid_fwd": N * (1024 + 128 + 12),
            "adam_step": pipe.model.field.flat.numel() * 28,
        }
        kern = {k: round(v[0], 4) for k, v in sorted(ksum.items(), key=lambda kv: -kv[1][0])}
        dom = next(iter(kern))
        # HBM traffic / MFMA-busy of the dominant operator from the committed rocprofv3 --pmc passes of this same command
        # (profiles/r01/pmc_summary.json; counters cannot be read from inside the process)
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r01", "pmc_summary.json")) as f:
                pmc = json.load(f)["kernels"]
        except Exception:
            pass
        op_kernels = {"field_bwd": ("field_bwd_part_kernel", "field_bwd_heads_kernel", "field_bwd_base_kernel", "field_reduce_kernel"),
                      "field_fwd": ("field_fwd_kernel", "field_pack_fwd"), "hashgrid_fwd": ("hashgrid_fwd_kernel",),
                      "hashgrid_bwd": ("hg_partition_kernel", "hg_reduce_kernel", "hg_scan_kernel"),
                      "hashgrid_bwd_apply": ("hg_partition_kernel<true>", "hg_reduce_kernel"), "adam_step": ("adam_kernel",)}
        traffic = sum(v["hbm_traffic_bytes"] for k, v in pmc.items() if any(k.startswith(p_) for p_ in op_kernels.get(dom, ()))) or None
        busy = [v["mfma_util"] for k, v in pmc.items() if "mfma_util" in v and
                k.startswith(("field_bwd_part_kernel<0", "field_bwd_heads") if dom == "field_bwd" else "field_fwd_kernel")]
        roof = None
        if dom in alg:
            ach = alg[dom] / (ksum[dom][0] * 1e-3) / 1e9
            roof = dict(bound="hbm", kernel=dom, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                        traffic=traffic, avg_ms=round(ksum[dom][0], 4), launches=ksum[dom][1])
        else:  # fp32-input MFMA (v_mfma_f32_16x16x4_f32), dense peak 157.3 TFLOP/s; algorithmic FLOPs per sample: SURVEY 8d
            fwd_flop = 33.9e3
            flops = {"field_fwd": fwd_flop * N, "field_bwd": 2 * fwd_flop * N}.get(dom)  # backward = dX + dW = 2x forward
            if flops:
                ach = flops / (ksum[dom][0] * 1e-3) / 1e12
                roof = dict(bound="mfma", kernel=dom + " (two heads kernels + base kernel + slab reduce)" if dom == "field_bwd" else dom,
                            achieved=round(ach, 2), peak=157.3, unit="TFLOP/s", frac=round(ach / 157.3, 4), traffic=traffic,
                            avg_ms=round(ksum[dom][0], 4), launches=ksum[dom][1],
                            mfma_busy_frac_pmc=(busy[0] if busy else None),
                            note="frac = algorithmic FLOPs / time / peak; mfma_busy_frac_pmc = SQ_VALU_MFMA_BUSY_CYCLES share incl. recompute and tile padding")
        line = {
            "metric": "train rays/sec (hotdog-shaped 31-band, C2)", "value": round(R * world * args.steps / dt, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2 hotdog 31-band rgb+spectral train step (fwd+loss+bwd+Adam)", "rays_per_gpu": R, "samples_per_ray": S,
                       "bands": B, "endmembers": Cn, "global_rays": R * world, "hash_table": "16x2^19x2 f32", "parallelism": f"dp{world}"},
            "spectral_psnr_db": round(psnr, 3), "loss": {k: round(float(v), 6) for k, v in loss_dict.items()},
            "kernels_ms": kern, "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, seed=42)
            line["gpu_over_cpu"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
