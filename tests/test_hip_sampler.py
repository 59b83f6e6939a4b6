"""GPU tests of the occupancy-grid marcher / visibility pruning / grid update against the oracle restatement
(oracle/torch_ref.py march_ray_ref; nerfacc itself is not available offline -> parity unpinned), plus a short end-to-end
training run through UMHSModel.get_outputs with the occupancy sampler."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rays(R, seed, inside_frac=0.3):
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(R, 3, generator=g)
    u = u / u.norm(dim=-1, keepdim=True)
    o = 2.5 * u + (torch.rand(R, 3, generator=g) - 0.5)
    k = int(R * inside_frac)
    o[:k] = (torch.rand(k, 3, generator=g) * 2 - 1) * 0.9  # cameras inside the roi
    d = -o + torch.randn(R, 3, generator=g) * 0.4
    d[:k] = torch.randn(k, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    d[-1] = torch.tensor([1.0, 0.0, 0.0])  # axis-aligned: two zero components
    o[-1] = torch.tensor([-3.0, 0.13, -0.27])
    d[-2] = torch.tensor([0.0, -1.0, 0.0])
    o[-2] = torch.tensor([0.31, 5.0, 0.2])
    return o, d


@pytest.mark.parametrize("levels,res,cone,step,far", [(1, 8, 0.0, 0.05, 1e3), (3, 16, 0.004, 0.02, 1e3), (4, 16, 0.01, 0.01, 6.0), (2, 32, 0.0, 0.013, 1e3)])
def test_marcher_matches_oracle(levels, res, cone, step, far):
    from umhsnerf.sampler import march_rays

    rng = np.random.default_rng(levels * 100 + res)
    binaries = rng.random((levels, res, res, res)) < 0.25
    o, d = _rays(48, seed=res)
    roi = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
    ri_ref, s_ref, e_ref = T.march_rays_ref(o, d, binaries, roi, 0.05, far, step, cone)
    bin_u8 = torch.from_numpy(binaries.astype(np.uint8)).to(DEV)
    ri, s, e, pinfo = march_rays(o.to(DEV), d.to(DEV), bin_u8, roi, levels, res, 0.05, far, step, cone)
    assert ri.shape[0] == ri_ref.shape[0] and ri_ref.shape[0] > 50, (ri.shape, ri_ref.shape)
    assert torch.equal(ri.cpu(), ri_ref)  # same number of samples on every ray
    assert torch.equal(s.cpu(), s_ref) and torch.equal(e.cpu(), e_ref)  # same float32 arithmetic, bit for bit
    assert torch.equal(pinfo.cpu(), T.pack_info(ri_ref, 48))
    # geometric invariants: mid-points lie in occupied voxels of the finest level containing them; rays are sorted
    mid = (o[ri_ref] + d[ri_ref] * ((s_ref + e_ref) / 2)[:, None]).numpy()
    m = np.abs(mid).max(-1)
    lvl = np.where(m < 1, 0, np.ceil(np.log2(np.maximum(m, 1e-9)) + 1e-7)).astype(int).clip(0, levels - 1)
    ok = 0
    for i in range(len(lvl)):
        half = 2.0 ** lvl[i]
        idx = np.clip(np.floor((mid[i] + half) / (2 * half / res)).astype(int), 0, res - 1)
        ok += bool(binaries[lvl[i], idx[0], idx[1], idx[2]])
    assert ok >= 0.995 * len(lvl)  # voxel-boundary ties aside
    assert bool((s_ref[1:] >= s_ref[:-1])[ri_ref[1:] == ri_ref[:-1]].all())
    # the three host paths -- single pass with scratch rows, overflowing rows (falls back to the second walk), two passes -- agree
    import os

    for cap in ("8", "0"):
        os.environ["UMHS_MARCH_CAP"] = cap
        try:
            ri2, s2, e2, p2 = march_rays(o.to(DEV), d.to(DEV), bin_u8, roi, levels, res, 0.05, far, step, cone)
        finally:
            del os.environ["UMHS_MARCH_CAP"]
        assert torch.equal(ri2, ri) and torch.equal(s2, s) and torch.equal(e2, e) and torch.equal(p2, pinfo), cap


def _env(**kv):
    import contextlib
    import os

    @contextlib.contextmanager
    def cm():
        old = {k: os.environ.get(k) for k in kv}
        os.environ.update({k: str(v) for k, v in kv.items()})
        try:
            yield
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
    return cm()


@pytest.mark.parametrize("levels,res,cone,per_ray", [(4, 128, 0.004, False), (4, 128, 0.0, True), (1, 64, 0.004, True), (6, 32, 0.01, False)])
def test_walk_split_over_the_lanes_of_a_wave_is_the_serial_walk_bit_for_bit(levels, res, cone, per_ray):
    """The default march takes the voxel walk out of the one-thread-per-ray kernel: umhs_march_walk gives every ray a wave, every lane a
    start somewhere in the ray's range; a lane falls onto the sequential walk at its first voxel face and must land EXACTLY on the next
    lane's start, else the window ends there; the emission kernel replays the lists.  UMHS_MARCH_SERIAL=1 is the form the oracle is
    written in (test_marcher_matches_oracle pins both to it on small grids).  Here: the bench scene's grid shape and a few thousand
    rays, a blob + noise occupancy, optional per-ray near / far planes and jitter -- identical samples, also with the lanes' voxel
    budget cut to 1 / 2 / 5 (many windows, lanes that run out in front of the next lane's start), with lists too short for the ray
    (the emission kernel walks on by itself from where the list ends) and in the two-pass form."""
    from umhsnerf.sampler import march_rays

    R = 3000
    g = torch.Generator().manual_seed(levels * 7 + res)
    ax = torch.linspace(-1, 1, res)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    blob = ((X * X + Y * Y * 1.7 + Z * Z) < 0.35)
    binaries = torch.stack([(blob if l == 0 else torch.zeros_like(blob)) | (torch.rand(res, res, res, generator=g) < (0.02 if l else 0.05)) for l in range(levels)])
    o, d = _rays(R, seed=res + 1)
    o[:64] = o[:64] * 40.0  # far outside: most of them miss the grid or cross it far from the centre
    d[64:96, 0] *= 0.004  # almost parallel to a grid axis: the walk creeps ulp by ulp at the faces of that axis
    d[64:96] = d[64:96] / d[64:96].norm(dim=-1, keepdim=True)
    roi = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
    kw = {}
    if per_ray:
        kw = dict(nears=torch.rand(R, generator=g) * 0.5, fars=2.0 + torch.rand(R, generator=g) * 8.0, jitter=torch.rand(R, generator=g), jitter_step=0.005)
        kw = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    bin_u8 = binaries.to(torch.uint8).to(DEV)
    args = (o.to(DEV), d.to(DEV), bin_u8, roi, levels, res, 0.05, 1e3, 0.005, cone)
    with _env(UMHS_MARCH_SERIAL=1):
        ref = march_rays(*args, **kw)
    assert ref[0].shape[0] > 50 * R // 8
    for knobs in ({}, dict(UMHS_MARCH_VSEG=1), dict(UMHS_MARCH_VSEG=2), dict(UMHS_MARCH_VSEG=5), dict(UMHS_MARCH_VCAP=16), dict(UMHS_MARCH_VCAP=128, UMHS_MARCH_VSEG=3),
                  dict(UMHS_MARCH_VCAP=64), dict(UMHS_MARCH_CAP=0), dict(UMHS_MARCH_CAP=8, UMHS_MARCH_VCAP=96)):
        with _env(**knobs):
            got = march_rays(*args, **kw)
        for a, b in zip(got, ref):
            assert torch.equal(a, b), knobs


def test_visibility_matches_oracle():
    from umhsnerf import ops
    from umhsnerf.sampler import visibility_mask

    b = T.synthetic_batch(40, 90, 3, seed=5, ragged=True)
    N = b["origins"].shape[0]
    g = torch.Generator().manual_seed(1)
    sigma = torch.exp(torch.randn(N, generator=g) * 2.0 + 2.0)
    t0, t1 = b["starts"][:, 0], b["ends"][:, 0]
    pinfo = T.pack_info(b["ray_indices"], 40)
    for eps, thre in [(1e-4, 0.01), (1e-4, 0.0), (0.0, 0.02), (0.3, 0.0)]:
        ref = T.render_visibility_from_density(t0, t1, sigma, pinfo, eps, thre)
        got = visibility_mask(sigma.to(DEV), t0.to(DEV), t1.to(DEV), pinfo.to(DEV), eps, thre).cpu()
        assert int((got != ref).sum()) <= max(2, N // 5000), (eps, thre, int((got != ref).sum()))  # ties at the thresholds
        assert 0 < int(ref.sum()) < N


def test_grid_update_and_sampling_follow_the_density():
    from umhsnerf.sampler import OccGridEstimator, VolumetricSampler
    from umhsnerf._ns_compat import RayBundle

    grid = OccGridEstimator([-1, -1, -1, 1, 1, 1], resolution=32, levels=2).to(DEV).train()
    ball = lambda x: (x.norm(dim=-1, keepdim=True) < 0.5).float() * 5.0  # density 5 inside a radius-0.5 ball
    assert not bool(grid.binaries.any())
    for step in range(0, 48, 16):
        grid.update_every_n_steps(step, occ_eval_fn=ball, occ_thre=0.01, warmup_steps=256)
    occ0 = grid.binaries[0].float().mean().item()
    assert abs(occ0 - (4 / 3 * np.pi * 0.5**3) / 8) < 0.02  # the ball fills ~6.5 % of the level-0 cube
    grid.update_every_n_steps(512 * 16, occ_eval_fn=ball)  # post-warm-up branch: uniform + occupied cells
    o, d = _rays(256, seed=9, inside_frac=0.0)
    sampler = VolumetricSampler(grid, density_fn=ball).to(DEV).train()
    rs, ri = sampler(RayBundle(origins=o.to(DEV), directions=d.to(DEV)), render_step_size=0.01, near_plane=0.05, far_plane=1e3,
                     alpha_thre=0.01, cone_angle=0.0)
    mid = rs.frustums.get_positions()
    assert mid.shape[0] > 1000 and float(mid.norm(dim=-1).max()) < 0.5 + 2 * (2 / 32) * np.sqrt(3)  # only near the ball
    assert bool((ri[1:] >= ri[:-1]).all())
    sampler.eval()
    rs2, _ = sampler(RayBundle(origins=o.to(DEV), directions=d.to(DEV)), render_step_size=0.01, near_plane=0.05, far_plane=1e3)
    assert rs2.frustums.starts.shape[0] >= rs.frustums.starts.shape[0]  # eval: no sigma_fn pruning


def test_short_training_run_with_occupancy_sampler():
    """ns-train-shaped loop: grid update -> occupancy sampling -> HIP hot path -> loss -> backward -> fused Adam."""
    from umhsnerf._ns_compat import RayBundle
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    B, Cn, R = 21, 4, 1024
    bands = list(range(450, 651, 10))
    cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.5, log2_hashmap_size=15, grid_resolution=32,
                     grid_levels=2, background_color="black", per_band_outputs=False)
    pipe = UMHSPipeline.from_packed_samples(cfg, torch.device(DEV), metadata={"wavelengths": bands, "num_classes": Cn}, seed=11)
    model = pipe.model.train()
    o, d = _rays(R, seed=21, inside_frac=0.0)
    bundle = RayBundle(origins=o.to(DEV), directions=d.to(DEV))
    g = torch.Generator().manual_seed(2)
    target = torch.rand(1, B, generator=g).expand(R, B).contiguous().to(DEV) * 0.8
    batch = {"hs_image": target, "image": model.converter(target)}
    losses = []
    for step in range(60):
        model.update_occupancy_grid(step)
        pipe.optimizer.zero_grad(set_to_none=True)
        out = model.get_outputs(bundle)
        loss = sum(model.get_loss_dict(out, batch).values())
        loss.backward()
        pipe.optimizer.step()
        losses.append(float(loss))
    assert out["spectral"].shape == (R, B) and out["num_samples_per_ray"].shape == (R,)
    assert np.isfinite(losses).all() and losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    E = model.field.endmembers
    assert float(E.min()) >= 0.0 and float(E.max()) <= 1.0  # clamp_endmembers fused in the optimizer step


def test_training_forward_reuses_the_samplers_hash_features(monkeypatch):
    """The sampler's density query already hash-encodes every marched candidate; the training forward gathers the survivors' rows
    instead of encoding them again.  Same positions, same table -> bit-identical features, losses and gradients."""
    import sys, os

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_hip_data import _split
    from umhsnerf import ops
    from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    res = {}
    for reuse in ("1", "0"):
        monkeypatch.setenv("UMHS_REUSE_ENC", reuse)
        torch.manual_seed(3)
        B = 8
        split, _, _, _ = _split(n=4, B=B, const=0.5)
        dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=2048), device=DEV, seed=4, train=split)
        cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, background_color="black")
        pipe = UMHSPipeline.from_packed_samples(cfg, DEV, metadata={"wavelengths": list(np.linspace(420, 680, B)), "num_classes": 3}, seed=5, datamanager=dm)
        for step in range(20):
            pipe.get_train_loss_dict(step)
        rb, batch = dm.next_train(20)
        rs, ri = pipe.model.sample(rb)
        cached = (rs.metadata or {}).get("umhs_enc") if reuse == "1" else None
        assert (cached is not None) == (reuse == "1")
        if cached is not None:  # the gathered rows ARE the encoding of the surviving samples
            fr = rs.frustums
            n = ri.numel()
            _, pos01, _ = ops.positions_fwd(fr.origins.view(n, 3), fr.directions.view(n, 3), fr.starts.view(-1), fr.ends.view(-1), pipe.model.field._spec())
            L = pipe.model.field.layout
            fresh = ops.hashgrid_fwd(pos01, L.view(pipe.model.field.flat.detach(), "mlp_base.encoder.hash_table"), pipe.model.field.scalings, 19, True)
            assert cached[0].shape[1] >= n and torch.equal(ops.enc_gather(*cached), fresh)
        pipe.optimizer.zero_grad(set_to_none=True)
        out, loss = pipe.model.forward_backward_from_samples(rs, ri, len(rb), batch)
        res[reuse] = ({k: float(v) for k, v in loss.items()}, pipe.model.field.flat.grad.clone(), ri.numel())
    (l1, g1, n1), (l0, g0, n0) = res["1"], res["0"]
    assert n1 == n0 and l1 == l0 and torch.equal(g1, g0)


def test_march_prefetched_one_step_ahead_is_the_same_training_run(monkeypatch):
    """UMHSPipeline issues the next batch's occupancy march on a side stream under the current step's forward/backward.  Same rays,
    same grid (steps that rewrite the grid are not prefetched), same order of generator draws -> the very same training run."""
    import sys, os

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_hip_data import _split
    from umhsnerf._ns_compat import RayBundle
    from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    res = {}
    for pre in ("1", "0"):
        monkeypatch.setenv("UMHS_PREFETCH_MARCH", pre)
        torch.manual_seed(7)
        B = 8
        split, _, _, _ = _split(n=4, B=B, const=0.5)
        dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=1024), device=DEV, seed=4, train=split)
        cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, background_color="random")
        pipe = UMHSPipeline.from_packed_samples(cfg, DEV, metadata={"wavelengths": list(np.linspace(420, 680, B)), "num_classes": 3}, seed=5, datamanager=dm)
        used, losses = 0, []
        for step in range(40):  # crosses the grid updates at steps 16 and 32
            ahead = getattr(pipe, "_ahead", None)
            used += int(ahead is not None and ahead[0] == step)
            _, loss, _ = pipe.get_train_loss_dict(step)
            losses.append(tuple(float(v) for v in loss.values()))
            if step == 20:  # an eval-time march in between must not disturb the march that is in flight
                pipe.model.eval()
                full = dm.train_split.image_rays(0)
                with torch.no_grad():
                    pipe.model.sample(RayBundle(origins=full.origins.view(-1, 3), directions=full.directions.view(-1, 3)))
                pipe.model.train()
        torch.cuda.synchronize()
        res[pre] = (losses, pipe.model.field.flat.detach().clone(), used)
    assert res["1"][2] >= 35 and res["0"][2] == 0  # every step but the first and the grid-update steps ran on a prefetched march
    assert res["1"][0] == res["0"][0]
    assert torch.equal(res["1"][1], res["0"][1])


@pytest.mark.parametrize("knob", ["UMHS_MARCH_SERIAL=1", "UMHS_FUSED_COUNT=0", "UMHS_REUSE_ENC=0"])
def test_round4_shortcuts_do_not_change_the_training_run(monkeypatch, knob):
    """The walk split over the lanes of a wave (vs one thread per ray) and the backward's bucket histogram taken inside the forward
    gather's launch (vs its own kernel; exercised with UMHS_REUSE_ENC=0, where the step hashes the survivors itself) are exact
    rearrangements: 40 sampler-driven steps -- grid updates, random backgrounds, Adam -- end in the same losses and the same
    parameters, bit for bit, with and without each."""
    import sys, os

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_hip_data import _split
    from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    name, value = knob.split("=")
    fixed = {"UMHS_REUSE_ENC": "0"} if name == "UMHS_FUSED_COUNT" else {}
    res = []
    for on in (False, True):
        for k, v in fixed.items():
            monkeypatch.setenv(k, v)
        if on:
            monkeypatch.setenv(name, value)
        else:
            monkeypatch.delenv(name, raising=False)
        torch.manual_seed(11)
        B = 8
        split, _, _, _ = _split(n=4, B=B, const=0.5)
        dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=1024), device=DEV, seed=4, train=split)
        cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, background_color="random")
        pipe = UMHSPipeline.from_packed_samples(cfg, DEV, metadata={"wavelengths": list(np.linspace(420, 680, B)), "num_classes": 3}, seed=5, datamanager=dm)
        losses = []
        for step in range(40):
            _, loss, _ = pipe.get_train_loss_dict(step)
            losses.append(tuple(float(v) for v in loss.values()))
        torch.cuda.synchronize()
        res.append((losses, pipe.model.field.flat.detach().clone()))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])
    assert res[0][0][-1] != res[0][0][0]  # (the run trains: the comparison is not between two constant sequences)


def test_fused_sampler_steps_match_the_torch_ops_they_replace():
    """ray_prefix / in-walk jitter / sample_midpoints / visibility_count + compact_samples vs cumsum, nears + rand * step, the
    sigma_fn expression, nonzero + index_select + gathers + pack_info: every one bit for bit."""
    from umhsnerf.sampler import compact_samples, march_rays, ray_prefix, sample_midpoints, visibility_mask

    g = torch.Generator().manual_seed(5)
    for R in (1, 7, 1024, 2500):
        counts = torch.randint(0, 300, (R,), generator=g)
        counts[torch.rand(R, generator=g) < 0.2] = 0
        pinfo, stats = ray_prefix(counts.to(DEV))
        ends = torch.cumsum(counts, 0)
        assert torch.equal(pinfo.cpu(), torch.stack([ends - counts, counts], -1))
        assert stats.tolist() == [int(ends[-1]), int(counts.max())]
    # stratified start inside the walk == per-ray near planes prepared by torch
    levels, res, step = 3, 16, 0.02
    binaries = np.random.default_rng(3).random((levels, res, res, res)) < 0.3
    bin_u8 = torch.from_numpy(binaries.astype(np.uint8)).to(DEV)
    R = 300
    o, d = (t.to(DEV) for t in _rays(R, seed=9))
    roi = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
    jit = torch.rand(R, generator=g).to(DEV)
    nears = torch.full((R,), 0.05, device=DEV) + jit * step
    a = march_rays(o, d, bin_u8, roi, levels, res, 0.05, 1e3, step, 0.004, jitter=jit, jitter_step=step)
    b = march_rays(o, d, bin_u8, roi, levels, res, 0.05, 1e3, step, 0.004, nears=nears, fars=torch.full((R,), 1e3, device=DEV))
    assert a[0].numel() > 2000 and all(torch.equal(x, y) for x, y in zip(a, b))
    ri, t0, t1, pinfo = a
    # midpoints
    pos = sample_midpoints(o, d, ri, t0, t1)
    assert torch.equal(pos, o[ri] + d[ri] * (t0 + t1)[:, None] / 2.0)
    # pruning + compaction
    sigma = (torch.rand(ri.numel(), generator=g) * 40).to(DEV)
    sigma[torch.rand(ri.numel(), generator=g).to(DEV) < 0.3] = 0.0
    cam = torch.randint(0, 5, (R, 1), generator=g).to(DEV)
    keep = visibility_mask(sigma, t0, t1, pinfo, 1e-4, 0.01)
    mask, kept = visibility_mask(sigma, t0, t1, pinfo, 1e-4, 0.01, with_counts=True)
    assert torch.equal(mask.bool(), keep)
    sel = torch.nonzero(keep).view(-1)
    assert 0 < sel.numel() < ri.numel()
    pinfo2, stats = ray_prefix(kept)
    assert int(stats[0]) == sel.numel() and torch.equal(pinfo2.cpu(), T.pack_info(ri[sel].cpu(), R))
    for c in (cam, None):
        out = compact_samples(mask, pinfo, pinfo2, sel.numel(), t0, t1, o, d, c)
        assert torch.equal(out["sel"], sel) and torch.equal(out["ray_indices"], ri[sel])
        assert torch.equal(out["t_starts"], t0[sel]) and torch.equal(out["t_ends"], t1[sel])
        assert torch.equal(out["origins"], o[ri[sel]]) and torch.equal(out["directions"], d[ri[sel]])
        assert (out["camera_indices"] is None) if c is None else torch.equal(out["camera_indices"], cam[ri[sel]])
    # nothing survives / nothing marched
    none = compact_samples(torch.zeros_like(mask), pinfo, torch.zeros_like(pinfo2), 0, t0, t1, o, d, cam)
    assert none["ray_indices"].numel() == 0 and none["origins"].shape == (0, 3)
    p0, s0 = ray_prefix(torch.zeros(0, dtype=torch.int64, device=DEV))
    assert p0.shape == (0, 2) and s0.tolist() == [0, 0]
