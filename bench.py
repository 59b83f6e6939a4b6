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
import gc
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
# the other single-GPU shapes of BASELINE.json (scripts/cbox_dragon.sh:3-9, pinecone.sh:5-12 per GPU, rgb+spectral.sh:6-14); C2 is the
# configuration the headline metric is quoted on and the default, the others are selected with --config
CONFIGS = {
    "C2": dict(C2, workload="C2 hotdog 31-band rgb+spectral train step (fwd+loss+bwd+Adam)"),
    "C3": dict(R=8192, S=64, B=128, C=9, temperature=0.3, pred_specular=True, method="rgb+spectral",
               workload="C3 cbox_dragon 128-band rgb+spectral train step (fwd+loss+bwd+Adam)"),
    "C4": dict(R=8192, S=64, B=31, C=4, temperature=0.5, pred_specular=True, method="rgb+spectral",
               workload="C4 pinecone 31-band, 8192 rays per GPU (65536 over 8 GPUs) rgb+spectral train step"),
    "C5": dict(R=8192, S=64, B=141, C=4, temperature=0.7, pred_specular=False, method="rgb+spectral",
               workload="C5 141-band joint rgb+spectral train step, 4 endmembers, no specular head"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3  # ibid.: dense fp32-input MFMA (v_mfma_f32_16x16x4_f32)


def csrc_hash() -> str:
    """Content hash of the kernel sources: profiles/*/pmc_summary.json records the hash it was measured on, and counters from a
    different build are not reported."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


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


# operators that have a roofline entry (the dominant one of them is timed live inside the timed region)
ROOFLINE_OPS = ("field_bwd", "hashgrid_bwd_apply", "hashgrid_bwd", "field_fwd", "field_heads_fwd", "hashgrid_fwd", "hashgrid_fwd_count", "composite_fwd",
                "composite_bwd")


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
        """{operator: (median ms, launches, mean ms, max ms)} -- the MEDIAN is what every table quotes: one host-side stall between an
        event pair (BENCH_r03: 131 ms inside a 0.02 ms operator) must not poison a row; the max shows that it happened."""
        out = {}
        for k, v in self.records.items():
            t = [a.elapsed_time(b) for a, b in v]
            out[k] = (float(np.median(t)), len(t), float(np.mean(t)), float(np.max(t)))
        return out


def cpu_baseline(cfg, seed, budget_s=15.0):
    """Oracle train step (fwd + loss + bwd + Adam) on the host cores: the 'reference CPU path' stand-in."""
    from oracle import torch_ref as T

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))  # a 1-GPU box owns a 16-core share of the host; more threads only oversubscribe
    torch.set_num_threads(cores)
    R = cfg["R"] if cfg["B"] <= 32 else 1024  # the full batch where a step fits the budget (~1.6 s at C2), else a 1024-ray sample
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
    while n < 2 or (time.perf_counter() - t0 < budget_s and n < 10):
        n += 1
        step(n + 1)
        print(f"[bench] cpu_baseline step {n}: {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
    dt = (time.perf_counter() - t0) / n
    return dict(value=R / dt, unit="rays/s", cores=cores, kind="port",
                sample=f"oracle/torch_ref.py fp32 train step (fwd+loss+bwd+Adam), {R} of {cfg['R']} rays x {cfg['S']} samples, {n} steps, {dt:.2f} s/step")


def sampler_scene(cfg, device, warm=300):
    """The resident synthetic scene of `sampler_step` / `eval_image`: 6 cameras of 64 x 64 pixels on a sphere around a constant-spectrum
    target, 4096 rays per batch, 4-level 128^3 occupancy grid, `warm` training steps so that the grid has settled on the target."""
    from umhsnerf.data.umhs_datamanager import ResidentSplit, UMHSDataManager, UMHSDataManagerConfig
    from umhsnerf.data.umhs_dataparser import Cameras
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    B, Cn, R, n, H, W = cfg["B"], cfg["C"], 4096, 6, 64, 64
    # the random background colours come from the device generator: seeded, the scene -- how far the grid has pruned, samples per ray --
    # is the same in every run (every kernel of the step is bitwise reproducible), unseeded it came out at 270 .. 540 samples per ray
    torch.manual_seed(20240611)
    torch.cuda.manual_seed_all(20240611)
    g = torch.Generator().manual_seed(3)
    pos = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 0.9
    z = torch.nn.functional.normalize(pos, dim=-1)
    x = torch.nn.functional.normalize(torch.linalg.cross(torch.tensor([[0.0, 0, 1]]).expand(n, 3), z), dim=-1)
    c2w = torch.stack([x, torch.linalg.cross(z, x), z, pos], -1).contiguous()
    cams = Cameras(c2w, torch.full((n,), 30.0), torch.full((n,), 30.0), torch.full((n,), W / 2), torch.full((n,), H / 2), H, W)
    hs = torch.full((n, H, W, B), 0.6)
    split = ResidentSplit(cams, torch.rand(n, H, W, 3, generator=g), hs, device)
    dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=R), device=device, seed=1, train=split)
    mc = UMHSConfig(method=cfg["method"], pred_specular=cfg["pred_specular"], temperature=cfg["temperature"], background_color="random")
    pipe = UMHSPipeline.from_packed_samples(mc, device, metadata={"wavelengths": list(np.linspace(400, 700, B)), "num_classes": Cn}, seed=2,
                                            datamanager=dm)
    with torch.no_grad():
        split.image = pipe.model.converter(split.hs_image.view(-1, B)).view(n, H, W, 3).contiguous()
    for step in range(warm):  # the grid settles on the target (update every 16 steps) and the model starts to fit it
        pipe.get_train_loss_dict(step)
    torch.cuda.synchronize()
    return pipe, c2w


def eval_image(cfg, device, pipe, c2w, H=256, reps=5):
    """Gradient-free rendering of one H x H camera of the sampler scene through `get_outputs_for_camera_ray_bundle` (reference
    umhs_model.py:593-620): march + density query of every candidate + pruning + the per-ray heads path, whole image in
    chunks of 32,768 rays.  Roofline: SURVEY 8d's forward-only figure, 32 + 1024 B per hashed sample (= candidate: survivors reuse the
    query's features) + (B + C + 5) x 4 B per ray, against HBM."""
    from umhsnerf import ops
    from umhsnerf import sampler as smp
    from umhsnerf.data.umhs_datamanager import ResidentSplit
    from umhsnerf.data.umhs_dataparser import Cameras

    B, Cn = cfg["B"], cfg["C"]
    f = 30.0 * H / 64.0  # the training cameras' field of view
    cams = Cameras(c2w[:1].clone(), torch.full((1,), f), torch.full((1,), f), torch.full((1,), H / 2), torch.full((1,), H / 2), H, H)
    split = ResidentSplit(cams, torch.zeros(1, H, H, 3), None, device)
    rb = split.image_rays(0)
    m = pipe.model.eval()
    try:
        with torch.no_grad():
            for _ in range(2):
                out = m.get_outputs_for_camera_ray_bundle(rb)
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            t0 = time.perf_counter()
            e[0].record()
            for i in range(reps):
                out = m.get_outputs_for_camera_ray_bundle(rb)
                e[i + 1].record()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            cand, finish = 0, smp.march_finish

            def counting_finish(h):  # candidates of the image = the samples every chunk's march hands to the density query
                nonlocal cand
                r = finish(h)
                cand += int(r[1].numel())
                return r

            smp.march_finish = counting_finish
            try:
                m.get_outputs_for_camera_ray_bundle(rb)
            finally:
                smp.march_finish = finish
            ms_med = float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)]))
            t_ops, t_smp = KernelTimer(ops), KernelTimer(smp)
            for name in ("positions_fwd", "hashgrid_fwd", "field_fwd", "field_base_fwd", "field_heads_fwd", "enc_gather", "composite_fwd",
                         "ray_epilogue_fwd", "tmid_minmax", "field_fwd_prepare"):
                t_ops.wrap(name)
            for name in ("march_begin", "march_finish", "visibility_mask", "compact_samples", "sample_midpoints"):
                t_smp.wrap(name)
            t_ops.enabled = t_smp.enabled = True
            m.get_outputs_for_camera_ray_bundle(rb)
            torch.cuda.synchronize()
            t_ops.enabled = t_smp.enabled = False
            ks = {**t_ops.summary(), **t_smp.summary()}
            for tm in (t_ops, t_smp):
                for name, fn in tm.orig.items():
                    setattr(tm.ops, name, fn)
    finally:
        m.train()
    R = H * H
    surv = float(out["num_samples_per_ray"].float().sum())
    n_hashed = cand if cand else surv
    nbytes = n_hashed * (32 + 1024) + R * (B + Cn + 5) * 4
    ach = nbytes / (ms_med * 1e-3) / 1e9
    per_image = {k: round(v[0] * v[1], 4) for k, v in ks.items()}  # ms per image (an operator runs once per chunk, or twice: query + render)
    return dict(rays=R, image=f"{H}x{H}", ms=round(ms_med, 4), ms_wall=round(dt * 1e3, 4), rays_per_s=round(R / (ms_med * 1e-3), 1),
                samples_per_ray=round(surv / R, 1), candidates_per_ray=round(cand / R, 1) if cand else None, chunks=-(-R // 32768),
                roofline=dict(kernel="eval image (march + candidate density query + heads per ray)", bound="hbm", achieved=round(ach, 1),
                              peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), algorithmic_bytes=int(nbytes),
                              note="32 + 1024 B per hashed sample + (B + C + 5) x 4 B per ray (SURVEY 8d, inference)"),
                kernels_ms=dict(sorted(per_image.items(), key=lambda kv: -kv[1])),
                note="get_outputs_for_camera_ray_bundle on a camera of the sampler scene after its training steps; ms = median of "
                     f"{reps} images between HIP events, kernels_ms from one more image with an event pair around every operator")


def sampler_step(cfg, device, pipe, steps=40, warm=300):
    """The iteration `ns-train umhsnerf` runs (reference umhs_model.py:229-237,549-554): occupancy-grid update, pixel batch + ray
    generation, the march through the grid, the density query of every candidate + visibility pruning, then the hot path on the
    survivors, backward, Adam -- `UMHSPipeline.get_train_loss_dict` on a resident synthetic scene (6 cameras of 64 x 64 pixels on a
    sphere around a constant-spectrum target, 4096 rays per batch, 4-level 128^3 grid; the scene of tools/profile_sampler_step.py).
    Not the headline metric (its sample count is the scene's, not BASELINE's): reported beside it so that the share of the marcher and
    the candidate query is on the driver's line.  -> dict(ms, rays_per_s, samples, candidates, kernels_ms, rooflines)."""
    from umhsnerf import ops
    from umhsnerf import sampler as smp

    R = 4096
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(warm, warm + steps):
        out, _, _ = pipe.get_train_loss_dict(step)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # per-operator pass (an event pair around every operator: untimed, like the main line's breakdown)
    t_ops, t_smp = KernelTimer(ops), KernelTimer(smp)
    for name in ("positions_fwd", "hashgrid_fwd", "field_fwd", "enc_gather", "composite_fwd", "ray_train_tail", "composite_bwd",
                 "field_bwd", "hashgrid_bwd_prepare", "hashgrid_bwd_apply", "adam_step_rows_range", "adam_step", "pixel_indices", "pixel_gather",
                 "raygen"):
        t_ops.wrap(name)
    for name in ("march_begin", "march_finish", "visibility_mask", "compact_samples", "sample_midpoints"):
        t_smp.wrap(name)
    t_ops.enabled = t_smp.enabled = True
    cand = surv = 0
    grid = pipe.model.sampler.occupancy_grid
    for step in range(warm + steps, warm + steps + 10):
        out, _, _ = pipe.get_train_loss_dict(step)
        surv += int(out["num_samples_per_ray"].sum())
        cand += int(getattr(grid, "last_candidates", 0))
    torch.cuda.synchronize()
    t_ops.enabled = t_smp.enabled = False
    ks = {**{k: v for k, v in t_ops.summary().items()}, **{k: v for k, v in t_smp.summary().items()}}
    per_step = {k: round(v[0] * v[1] / 10.0, 4) for k, v in ks.items()}  # ms per step (an operator may run more than once per step)
    for tm in (t_ops, t_smp):  # un-wrap: the module objects are shared with the caller
        for name, fn in tm.orig.items():
            setattr(tm.ops, name, fn)
    n_surv = surv / 10.0
    n_cand = (cand / 10.0) if cand else None
    roofs = []

    def roof(name, ops_, nbytes):
        t = sum(per_step.get(o, 0.0) for o in ops_)
        if t > 0 and nbytes:
            ach = nbytes / (t * 1e-3) / 1e9
            roofs.append(dict(kernel=name, operators=list(ops_), bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                              frac=round(ach / HBM_PEAK_GBS, 4), ms_per_step=round(t, 4)))

    if n_cand:
        # march: the walk is one wave per ray (lanes = segments of the ray, 0.03 ms), the sample emission one dependent chain per ray
        # (t += max(t * cone, step) in float): bound by the LATENCY of the longest ray's chain, not by bytes; it runs one
        # step ahead on its own stream.  Reported as what it is: candidates per second, no peak to divide by.
        t_m = sum(per_step.get(o, 0.0) for o in ("march_begin", "march_finish"))
        if t_m > 0:
            roofs.append(dict(kernel="march (walk + compaction)", operators=["march_begin", "march_finish"], bound="latency (longest ray's chain)",
                              achieved=round(n_cand / (t_m * 1e-3) / 1e6, 1), unit="M candidates/s", peak=None, frac=None, ms_per_step=round(t_m, 4),
                              note="hidden: issued one step ahead on its own stream"))
        # density query of every candidate: 1024 B of table rows + 128 B of features + 12 B position + 4 B sigma
        roof("density query of the candidates", ("sample_midpoints", "positions_fwd", "hashgrid_fwd", "field_fwd"), n_cand * (1024 + 128 + 12 + 4))
        roof("visibility + compaction", ("visibility_mask", "compact_samples", "enc_gather"), n_cand * 13 + n_surv * (48 + 128))
    return dict(ms=round(ms, 4), rays_per_s=round(R / ms * 1e3, 1), rays=R, survivors_per_step=round(n_surv, 1), candidates_per_step=n_cand,
                kernels_ms=dict(sorted(per_step.items(), key=lambda kv: -kv[1])), rooflines=roofs,
                scene="6 cameras 64x64 around a constant-spectrum target, 4-level 128^3 occupancy grid, after 300 warm-up steps",
                note="get_train_loss_dict: grid update + pixel batch + march + candidate density query + pruning + hot path + Adam; "
                     "kernels_ms from an untimed pass with an event pair around every operator (positions_fwd / hashgrid_fwd / field_fwd run "
                     "for the candidates' density query AND for the survivors' forward)")


OPS = ("positions_fwd", "hashgrid_fwd", "hashgrid_fwd_count", "hashgrid_bwd_prepare_counted", "field_fwd", "field_base_fwd", "field_heads_fwd", "accumulate_fwd", "composite_fwd", "tmid_minmax",
       "ray_train_tail", "composite_bwd", "field_bwd",
       "hashgrid_bwd", "hashgrid_bwd_prepare", "hashgrid_bwd_apply", "field_fwd_prepare", "field_bwd_prepare", "adam_step",
       "adam_step_rows", "adam_step_rows_range")  # (one GPU: the dense hash levels' Adam step rides in hashgrid_bwd_apply)


class Case:
    """One BASELINE configuration on this rank: the plugin-surface pipeline, its resident synthetic batch and the step closure."""

    def __init__(self, name, device, world, local, rank):
        from umhsnerf import ops
        from umhsnerf._ns_compat import packed_ray_samples
        from umhsnerf.umhs_model import UMHSConfig
        from umhsnerf.umhs_pipeline import UMHSPipeline

        self.name, self.cfg = name, CONFIGS[name]
        cfg = self.cfg
        self.R, self.S, self.B, self.Cn = cfg["R"], cfg["S"], cfg["B"], cfg["C"]
        self.N = self.R * self.S
        bands = list(np.linspace(400, 700, self.B))
        mc = UMHSConfig(method=cfg["method"], pred_specular=cfg["pred_specular"], temperature=cfg["temperature"], per_band_outputs=True)
        self.pipe = UMHSPipeline.from_packed_samples(mc, device, metadata={"wavelengths": bands, "num_classes": self.Cn}, world_size=world,
                                                     local_rank=local, seed=42)
        trained_like_init(self.pipe.model.field, seed=42)
        if world > 1:
            dist.broadcast(self.pipe.model.field.flat.data, src=0)
        self.b = synthetic_batch(self.R, self.S, self.B, seed=42 + rank, device=device)  # every rank draws its own rays (weak scaling)
        self.rs = packed_ray_samples(self.b["origins"], self.b["directions"], self.b["starts"], self.b["ends"])
        self.pinfo = ops.pack_info(self.b["ray_indices"], self.R)
        with torch.no_grad():
            gt_rgb = self.pipe.model.converter(self.b["gt_spectral"])
        self.batch = {"image": gt_rgb, "hs_image": self.b["gt_spectral"]}

    def step(self):
        return self.pipe.train_iteration(self.rs, self.b["ray_indices"], self.R, self.batch, packed_info=self.pinfo)


def measure(case, timer, steps, warmup, world, device):
    """warm-up -> pick the dominant operator from three fully timed (still warm-up) steps -> the timed region: EXACTLY `steps` steps
    between a barrier + device sync on both sides, one HIP event per step boundary (median) and an event pair around the dominant
    operator only -> (wall seconds [max over ranks], median ms, the dominant operator's live summary, last outputs, last losses)."""
    # Host hygiene first: a generation-2 pass of Python's cycle collector takes ~0.1 s in this process (torch + numpy + the oracle's
    # modules are tens of thousands of tracked objects); BENCH_r03's breakdown pass caught one between an event pair.
    # Collect now and freeze the survivors, so the passes below only ever see young-generation collections.  It happens HERE, in front
    # of the warm-up: 0.1 s of idle GPU right before the timed region cost its first step 0.9 ms and the next ten a lower clock.
    gc.collect()
    gc.freeze()
    for _ in range(warmup):
        case.step()
    # Which operator is the dominant one is measured, not assumed: three steps with every operator timed (still warm-up), and the
    # largest carries the events of the timed region.  field_bwd and hashgrid_bwd_apply are within a few percent of each other at
    # C2 -- a ranking taken AFTER the timed region (from the untimed pass) would quote an operator that was not timed live.
    timer.records.clear()
    timer.only, timer.enabled = None, True
    for _ in range(3):
        case.step()
    torch.cuda.synchronize()
    timer.enabled = False
    pre = timer.summary()
    pick = torch.tensor([float(pre.get(k, (0.0, 0))[0]) for k in ROOFLINE_OPS], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(pick, op=dist.ReduceOp.MAX)  # every rank times the same operator
    timer.only = {ROOFLINE_OPS[int(torch.argmax(pick).item())]}
    timer.records.clear()
    timer.enabled = True
    # untimed: the launch queue refills and the clocks settle after the fully timed steps above (their host syncs drained it); the
    # step times of a timed region that starts cold fall from 0.73 to 0.69 ms over its first thirty steps
    for _ in range(max(20, warmup)):
        case.step()
    timer.records.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    outputs = loss_dict = None
    for i in range(steps):
        outputs, loss_dict = case.step()
        marks[i + 1].record()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.enabled = False
    per_step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]  # GPU-side step boundaries on the launch stream
    tmax = torch.tensor([dt, float(np.median(per_step_ms))], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dom_live = timer.summary()
    timer.records.clear()
    if np.max(per_step_ms) > 3 * np.median(per_step_ms):  # a stall inside the timed region: say where (stderr; the line carries max + median)
        print(f"[bench] {case.name}: slow step(s) in the timed region, ms per step = {[round(t, 3) for t in per_step_ms]}", file=sys.stderr, flush=True)
    return float(tmax[0].item()), float(tmax[1].item()), dom_live, outputs, loss_dict, float(np.max(per_step_ms))


def breakdown(case, timer, steps):
    """Untimed pass with an event pair around every operator -> {operator: (median ms, launches, mean, max)}."""
    timer.records.clear()
    timer.only, timer.enabled = None, True
    for _ in range(steps):
        case.step()
    torch.cuda.synchronize()
    timer.enabled = False
    out = timer.summary()
    timer.records.clear()
    return out


def roofline_tables(case, ksum, dom, world, config_name):
    """Per-operator rooflines from the measured operator times: algorithmic work per launch -- hash gather / scatter 1024 B
    per sample + the level-major feature rows (128 B) and positions (12 B); field MLPs 2 x MAC per sample (forward), backward = dX +
    dW = 2 x forward; compositing and the optimizer by the bytes they must move once."""
    cfg, pipe, N, R, B, Cn = case.cfg, case.pipe, case.N, case.R, case.B, case.Cn
    kern = {k: round(v[0], 4) for k, v in sorted(ksum.items(), key=lambda kv: -kv[1][0])}
    spec = cfg["pred_specular"]
    mac = 3072 + 2 * (1728 + 4096) + 64 * Cn + 64 * (Cn + (1 if spec else 0)) + ((448 + 16 * B) if spec else 0) + Cn * B + 256
    nstream = (3 * B if spec else B) + Cn
    # (one GPU: the Adam step of the dense hash levels rides in the bucket reduce of hashgrid_bwd_apply -- 28 B per parameter of
    # levels 5..15, the same bytes the separate adam_step would move -- so that operator's algorithmic bytes include them)
    sink = getattr(pipe.model.field, "_grad_sink", None)
    fused_adam_bytes = 28 * (16 - int(getattr(sink, "sparse_levels", 0) or 0)) * (1 << 19) * 2 if world == 1 else 0
    # (hashgrid_fwd_count = the gather with the backward's bucket histogram in the same launch: the gather's bytes, nothing added)
    alg_bytes = {"hashgrid_fwd": N * (1024 + 128 + 12), "hashgrid_fwd_count": N * (1024 + 128 + 12), "hashgrid_bwd": N * (1024 + 128 + 12),
                 "hashgrid_bwd_apply": N * (1024 + 128 + 12) + fused_adam_bytes,
                 # (above 32 bands the compositing pass carries no value stream -- the band sums are formed inside the heads kernel and
                 # the value half of its backward inside field_bwd, DESIGN 4.3: sigma, t0, t1 in, weights out + two floats per ray)
                 "composite_fwd": (N * 16 + R * 8) if "field_heads_fwd" in ksum else N * (nstream + 4) * 4, "composite_bwd": N * (2 * B + 5) * 4,
                 "adam_step": pipe.model.field.flat.numel() * 28}
    # (wide-band models run the forward as mlp_base + heads with the per-ray band sums inside the heads kernel, and the backward
    # with the compositing backward's value half inside field_bwd: ops.field_base_fwd / field_heads_fwd; field_bwd then includes it)
    alg_flops = {"field_fwd": 2.0 * mac * N, "field_bwd": 4.0 * mac * N, "field_base_fwd": 2.0 * 3072 * N,
                 "field_heads_fwd": 2.0 * (mac - 3072) * N}
    # HBM traffic / MFMA-busy from the committed rocprofv3 --pmc passes of this same command -- only when they were taken on
    # THIS build of the kernels (the summary records the source hash); counters cannot be read from inside the process
    pmc, pmc_src = {}, None
    for rel in (f"profiles/r04/pmc_summary_{config_name}.json", f"profiles/r03/pmc_summary_{config_name}.json"):
        try:
            with open(os.path.join(ROOT, rel)) as f:
                js = json.load(f)
            if js.get("csrc_sha") == csrc_hash() and js.get("config", "C2") == config_name:
                pmc, pmc_src = js["kernels"], rel
                break
        except Exception:
            pass
    op_kernels = {"field_bwd": ("field_bwd_tf_kernel", "field_bwd_tfz0_kernel", "field_bwd_tfz1_kernel", "field_slab_fold", "field_reduce_tf", "field_mix_"),
                  "field_fwd": ("field_fwd_kernel", "field_pack_all"), "field_base_fwd": ("field_fwd_kernel<false, true",),
                  "field_heads_fwd": ("field_fwd_kernel", "field_heads_finish"), "hashgrid_fwd": ("hashgrid_fwd_kernel",),
                  "hashgrid_fwd_count": ("hashgrid_fwd_count_kernel",),
                  "hashgrid_bwd": ("hg_partition_kernel", "hg_reduce_kernel", "hg_scan_kernel", "hg_pairs_kernel", "hg_level_absmax"),
                  "hashgrid_bwd_apply": ("hg_partition_kernel<true>", "hg_reduce_kernel", "hg_pairs_kernel<true>", "hg_level_absmax"),
                  "adam_step": ("adam_kernel",),
                  "composite_fwd": ("composite_fwd_kernel",), "composite_bwd": ("composite_bwd_kernel",)}

    def roof_of(op):
        ms, launches = ksum[op][0], ksum[op][1]
        extra = {"avg_ms": round(ms, 4), "launches": launches}
        if len(ksum[op]) > 3:
            extra.update(mean_ms=round(ksum[op][2], 4), max_ms=round(ksum[op][3], 4))
        traffic = sum(v["hbm_traffic_bytes"] for k, v in pmc.items() if "hbm_traffic_bytes" in v and any(k.startswith(p_) for p_ in op_kernels.get(op, ()))) or None
        if op in alg_bytes:
            ach = alg_bytes[op] / (ms * 1e-3) / 1e9
            return dict(kernel=op, bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                        traffic=traffic, **extra)
        if op in alg_flops:
            ach = alg_flops[op] / (ms * 1e-3) / 1e12
            # MFMA-busy share of the operator = of all its kernels (the reductions included), weighted by their cycles
            mk = [v for k, v in pmc.items() if v.get("kernel_cycles_per_xcd") and any(k.startswith(p_) for p_ in op_kernels[op])]
            busy = round(sum(v.get("mfma_util", 0.0) * v["kernel_cycles_per_xcd"] for v in mk) / sum(v["kernel_cycles_per_xcd"] for v in mk), 4) if mk else None
            return dict(kernel=op, bound="mfma", achieved=round(ach, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(ach / MFMA_F32_PEAK_TF, 4),
                        traffic=traffic, mfma_busy_frac_pmc=busy, **extra)
        return None

    rooflines = [r for r in (roof_of(op) for op in kern) if r is not None]
    roof = next((r for r in rooflines if r["kernel"] == dom), rooflines[0] if rooflines else None)
    if roof is not None:
        roof = dict(roof, pmc_source=pmc_src,
                    note="dominant operator, timed with HIP events inside the timed region (avg_ms = median of those launches); achieved = "
                         "algorithmic bytes or FLOPs (SURVEY 8d) / time; traffic / mfma_busy_frac_pmc only when profiles/*/pmc_summary.json was "
                         "taken on this build")
    return kern, roof, rooflines


def other_config(name, device, timer, steps=20, warmup=10):
    """The other single-GPU BASELINE shapes through the same measurement, on the driver's line: C3 = configs[2]
    (scripts/cbox_dragon.sh:3-9, 128 bands, 8192 rays, 1 GPU), C4 = one GPU's 8192-ray shard of configs[3], C5 = configs[4]'s shard."""
    case = Case(name, device, 1, 0, 0)
    dt, median_ms, dom_live, outputs, loss_dict, worst_ms = measure(case, timer, steps, warmup, 1, device)
    ksum = breakdown(case, timer, 5)
    ksum.update(dom_live)
    dom = next(iter(dom_live)) if dom_live else None
    kern, roof, _ = roofline_tables(case, ksum, dom, 1, name)
    out = dict(workload=case.cfg["workload"], rays=case.R, samples_per_ray=case.S, bands=case.B, endmembers=case.Cn,
               ms_per_step=round(dt / steps * 1e3, 4), ms_per_step_median=round(median_ms, 4), ms_per_step_max=round(worst_ms, 4),
               rays_per_s=round(case.R * steps / dt, 1),
               steps=steps, warmup=warmup, roofline=roof, kernels_ms=kern,
               spectral_psnr_db_vs_uniform_random_gt=round(float(case.pipe.model.psnr(outputs["spectral"].detach(), case.b["gt_spectral"])), 3))
    del case
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampler-step", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-eval-image", action="store_true")
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hot path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)  # (rehearsals with more ranks than GPUs share a device; the driver runs one rank per GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    backend = None
    if world > 1:
        backend = os.environ.get("UMHS_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only for 1-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from umhsnerf import ops, parallel

    case = Case(args.config, device, world, local, rank)
    cfg, pipe, b = case.cfg, case.pipe, case.b
    R, S, B, Cn = case.R, case.S, case.B, case.Cn

    timer = KernelTimer(ops)
    for name in OPS:
        timer.wrap(name)
    # Inside the timed region only the dominant operator carries HIP events (2 per step) plus one event per step boundary (for the
    # median): an event is a barrier packet on the queue, and a pair around every one of the ~12 operators costs ~12 % of the
    # step.  The full per-operator table comes from a separate, untimed pass over the same step.
    parallel.STATS.update(bytes=0, messages=0)
    pauses = []

    def gc_cb(phase, info, _t=[0.0]):  # every pass of Python's cycle collector during the measurement, with its length
        if phase == "start":
            _t[0] = time.perf_counter()
        else:  # (frozen = after measure()'s own deliberate collect + freeze, i.e. inside the timed region or the breakdown pass)
            pauses.append({"generation": info.get("generation"), "ms": round((time.perf_counter() - _t[0]) * 1e3, 3), "frozen": gc.get_freeze_count() > 0})

    gc.callbacks.append(gc_cb)
    dt, median_ms, dom_live, outputs, loss_dict, worst_ms = measure(case, timer, args.steps, args.warmup, world, device)
    exchanged = {k: v for k, v in parallel.STATS.items()}
    ksum_all = breakdown(case, timer, min(args.steps, 10))  # (all ranks: the step contains the collectives)
    gc.callbacks.remove(gc_cb)
    dist_info = None
    if world > 1:  # how much of the step is exchange that did NOT hide under the backward: the same step with the collectives off
        parallel.EXCHANGE_DISABLED = True
        for _ in range(3):
            case.step()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(min(args.steps, 20)):
            case.step()
        torch.cuda.synchronize()
        t_noex = torch.tensor([(time.perf_counter() - t1) / min(args.steps, 20)], device=device, dtype=torch.float64)
        parallel.EXCHANGE_DISABLED = False
        dist.all_reduce(t_noex, op=dist.ReduceOp.MAX)
        sink = pipe.model.field._grad_sink
        # (the counters saw every step of measure(): warm-up, the three dominant-operator steps, the re-warm steps, the timed ones)
        n_exchanging_steps = args.warmup + 3 + max(20, args.warmup) + args.steps
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "devices_visible": ndev,
                     "mb_exchanged_per_step": round(exchanged["bytes"] / n_exchanging_steps / 1e6, 2),
                     "messages_per_step": round(exchanged["messages"] / n_exchanging_steps, 2),
                     "level_groups": getattr(sink, "level_groups", None), "async_reduce": getattr(sink, "async_reduce", None),
                     "ms_per_step_without_exchange": round(float(t_noex.item()) * 1e3, 4),
                     "exposed_exchange_ms": round(dt / args.steps * 1e3 - float(t_noex.item()) * 1e3, 4)}

    if rank == 0:
        ksum = dict(ksum_all)
        ksum.update(dom_live)  # the dominant operator's figure is the one measured inside the timed region
        ms_step = dt / args.steps * 1e3
        psnr = float(pipe.model.psnr(outputs["spectral"].detach(), b["gt_spectral"]))
        loss_dict = {k: v.detach() for k, v in loss_dict.items()}
        print(f"[bench] gpu: {ms_step:.3f} ms/step (median {median_ms:.3f}), {R * world * args.steps / dt:.0f} rays/s", file=sys.stderr, flush=True)
        dom = next(iter(dom_live)) if dom_live else None  # the operator that carried the events of the timed region
        kern, roof, rooflines = roofline_tables(case, ksum, dom, world, args.config)
        line = {
            "metric": f"train rays/sec (hotdog-shaped 31-band, C2)" if args.config == "C2" else f"train rays/sec ({args.config})",
            "value": round(R * world * args.steps / dt, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "ms_per_step_median": round(median_ms, 4), "ms_per_step_max": round(worst_ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "arithmetic": "fp32 storage and accumulation; MLP chains (forward, recompute, dX) as three-piece bf16 products on the bf16 MFMA "
                          "(24 significant bits, fp32 accumulate); dW operands as two bf16 pieces x three products (2^-16 per product, "
                          "unbiased: tests/test_hip_fullsize.py against a float64 oracle at N = 262,144); band tiles and 16-wide layers on the fp32 MFMA",
            "data": "synthetic",
            "config": {"workload": cfg["workload"], "rays_per_gpu": R, "samples_per_ray": S,
                       "bands": B, "endmembers": Cn, "global_rays": R * world, "hash_table": "16x2^19x2 f32", "parallelism": f"dp{world}"},
            "sanity": {"loss": {k: round(float(v), 6) for k, v in loss_dict.items()},
                       "spectral_psnr_db_vs_uniform_random_gt": round(psnr, 3),
                       "note": "the synthetic ground truth is uniform noise: these only show the step is numerically alive, not image quality"},
            "kernels_ms": kern,
            "kernels_ms_note": "median over the launches of an untimed pass with an event pair around every operator (the dominant one: inside "
                               "the timed region); kernels_ms_max = the slowest launch of each",
            "kernels_ms_max": {k: round(v[3], 4) for k, v in ksum.items()},
            # (collector passes INSIDE the timed region / breakdown pass, i.e. after measure()'s own deliberate collect + freeze; none expected)
            "host_gc_pauses": [p for p in pauses if p["ms"] >= 1.0 and p["frozen"]],
            "host_gc_collect_before_timed_region_ms": max([p["ms"] for p in pauses if not p["frozen"]] or [0.0]),
            "roofline": roof, "rooflines": rooflines, "csrc_sha": csrc_hash(),
        }
        if dist_info is not None:
            line["dist"] = dist_info
        if world == 1:
            del pipe, b, outputs, case  # (the hot-path pipeline's buffers: the passes below build their own models)
            torch.cuda.empty_cache()
        if world == 1 and not args.no_other_configs:
            line["other_configs"] = {}
            for name in ("C3", "C4", "C5", "C2"):
                if name != args.config:
                    line["other_configs"][name] = other_config(name, device, timer)
                    print(f"[bench] {name}: {line['other_configs'][name]['ms_per_step']:.3f} ms/step", file=sys.stderr, flush=True)
        if world == 1 and not (args.no_eval_image and args.no_sampler_step):
            scene_pipe, scene_c2w = sampler_scene(cfg, device)
            if not args.no_sampler_step:
                line["sampler_step"] = sampler_step(cfg, device, scene_pipe)
            if not args.no_eval_image:
                line["eval_image"] = eval_image(cfg, device, scene_pipe, scene_c2w)
            del scene_pipe
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, seed=42)
            line["gpu_over_cpu"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
