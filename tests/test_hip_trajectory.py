"""GPU parity over a TRAINING RUN and on the branches single-step tests do not reach ("next round" 1a-1e):
  * 50 Adam steps (exp-decay LR, endmember clamp) of the HIP plugin path vs the CPU oracle, same init and batches;
  * gradient accumulation (3 micro-batches, one optimizer step -- scripts/rgb+spectral.sh:5) through the gradient sink;
  * trunc_exp's clamped backward (|sigma_raw| > 15, umhs_field.py:17,327);
  * the B > 32 heads backward at N >= 4096 (several workgroups + a tail tile) for B in {128, 141};
  * the positional encoding at world |x| in [4, 8] (v_sin on revolutions vs the reference's sin(fl(2 pi) x f)).
Tolerances are written at each assertion; radiance 1e-4 relative and PSNR 0.05 dB are north_star's."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as T
from test_hip_parity import DEV, assert_close, dev, make_case, relerr

pytestmark = pytest.mark.gpu


def _ops():
    from umhsnerf import ops

    return ops


# ------------------------------------------------------------------------------------------------------------------- #
# shared: a small learnable problem (smooth target spectrum as a function of the viewing ray), oracle trainer, HIP trainer
# ------------------------------------------------------------------------------------------------------------------- #
def learnable_batches(nb, R, S, B, seed=100):
    out = []
    for i in range(nb):
        b = T.synthetic_batch(R, S, B, seed=seed + i)
        o = b["origins"].view(R, S, 3)[:, 0].double()
        u = o / o.norm(dim=-1, keepdim=True)
        lam = torch.linspace(0, 1, B, dtype=torch.float64)
        b["gt_spectral"] = (0.5 + 0.35 * torch.sin(3.0 * u[:, :1] + 4.0 * lam[None, :]) * torch.cos(2.0 * u[:, 1:2])).float()
        out.append(b)
    return out


def oracle_params(C, B, spec, log2_T, seed=9):
    p = T.FieldParams(C, B, spec, log2_hashmap_size=log2_T, table_scale=0.5, seed=seed)
    with torch.no_grad():
        p.base_b[1][0] += 1.0
    return p


def oracle_grads(p, b, R, temp, M, method="rgb+spectral"):
    out = T.model_outputs(p, b["origins"], b["directions"], b["starts"], b["ends"], b["ray_indices"], R, temp, M)
    gt_rgb = T.colour_system(b["gt_spectral"], M)
    loss = T.model_loss(out, b["gt_spectral"], gt_rgb, torch.zeros(R, 3), method)
    params = [v for _, v in p.named_parameters()]
    grads = torch.autograd.grad(sum(loss.values()), params, allow_unused=True)
    return out, loss, [g if g is not None else torch.zeros_like(v) for g, v in zip(grads, params)]


def hip_pipeline(p, C, B, spec, temp, log2_T, bands, **kw):
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    cfg = UMHSConfig(method="rgb+spectral", pred_specular=spec, temperature=temp, log2_hashmap_size=log2_T, background_color="black")
    pipe = UMHSPipeline.from_packed_samples(cfg, torch.device(DEV), metadata={"wavelengths": bands, "num_classes": C}, seed=3, **kw)
    pipe.model.field.load_state_dict(p.reference_state_dict())
    return pipe


def hip_batch(b, M):
    from umhsnerf._ns_compat import packed_ray_samples

    d = dev(b)
    rs = packed_ray_samples(d["origins"], d["directions"], d["starts"], d["ends"])
    return rs, d["ray_indices"], {"image": T.colour_system(b["gt_spectral"], M).to(DEV), "hs_image": d["gt_spectral"]}


# ------------------------------------------------------------------------------------------------------------------- #
@pytest.mark.parametrize("path", ["per-sample [N,B] arrays", "band sums inside the field kernels"])
def test_fifty_step_training_trajectory_matches_the_oracle(path, monkeypatch):
    """(Both forms of the training step: the field forward + compositing kernels over per-sample [N,B] arrays -- the default up to 32
    bands -- and the two-launch forward / folded compositing backward that never materialises them -- the default above.)
    Same init, same four batches cycled, 50 steps of Adam(lr 2e-2 exp-decayed, eps 1e-15) + clamp_endmembers
    (umhs_config.py:59-64, umhs_model.py:358-370,568-572).  Bounds: loss curve 1e-3 relative at every step, final spectral PSNR
    0.05 dB, endmembers 1e-3 absolute.  (The fp32 oracle drifts from its own fp64 run by 4e-5 / 2e-4 dB / 4e-5 over these 50
    steps, so the bounds are ~25x the arithmetic noise floor, not slack for a wrong update rule.)"""
    fused = path.startswith("band sums")
    monkeypatch.setenv("UMHS_FUSED_BWD", "1" if fused else "0")
    R, S, B, C, temp, log2_T, steps = 64, 24, 31, 6, 0.4, 14, 50
    bands = list(np.linspace(400, 700, B))
    M = T.colour_matrix(bands)
    bs = learnable_batches(4, R, S, B)
    p = oracle_params(C, B, True, log2_T)
    pipe = hip_pipeline(p, C, B, True, temp, log2_T, bands)
    hb = [hip_batch(b, M) for b in bs]
    params = [v for _, v in p.named_parameters()]
    ms, vs = [torch.zeros_like(v) for v in params], [torch.zeros_like(v) for v in params]
    ref_loss, got_loss, ref_psnr, got_psnr = [], [], [], []
    for step in range(steps):
        b = bs[step % 4]
        out, loss, grads = oracle_grads(p, b, R, temp, M)
        with torch.no_grad():
            T.adam_step(params, grads, ms, vs, step + 1, T.exp_decay_lr(step))
            p.endmembers.clamp_(0, 1)
        ref_loss.append(float(sum(loss.values()).detach()))
        ref_psnr.append(float(T.psnr(out["spectral"].detach(), b["gt_spectral"])))
        rs, ri, batch = hb[step % 4]
        outputs, loss_dict = pipe.train_iteration(rs, ri, R, batch)
        got_loss.append(float(sum(loss_dict.values())))
        got_psnr.append(float(T.psnr(outputs["spectral"].cpu(), b["gt_spectral"])))
    ref_loss, got_loss = np.array(ref_loss), np.array(got_loss)
    assert ref_loss[-1] < 0.25 * ref_loss[0], "the problem must actually train"
    rel = np.abs(got_loss - ref_loss) / np.abs(ref_loss)
    assert rel.max() <= 1e-3, f"loss curve: max relative deviation {rel.max():.2e} at step {int(rel.argmax())}"
    assert abs(got_psnr[-1] - ref_psnr[-1]) <= 0.05 and np.abs(np.array(got_psnr) - np.array(ref_psnr)).max() <= 0.05
    sd = pipe.model.field.state_dict()
    e_err = float((sd["endmembers"].cpu() - p.endmembers.detach()).abs().max())
    assert e_err <= 1e-3, f"endmembers after {steps} steps differ by {e_err:.2e}"
    for k, v in p.reference_state_dict().items():
        if k != "mlp_base.encoder.hash_table":  # sign-like first Adam steps make single table entries chaotic; MLPs are the check
            assert_close(f"after {steps} steps: {k}", sd[k], v, 5e-3)
    assert_close("hash table after training", sd["mlp_base.encoder.hash_table"], p.hash_table.detach(), 5e-2)


@pytest.mark.parametrize("bias", [20.0, -20.0])
def test_trunc_exp_clamped_backward(bias):
    """sigma = exp(raw) is not clamped forward, its gradient is g * exp(clamp(raw, -15, 15)) (nerfstudio trunc_exp, umhs_field.py:17,327).
    Every sample here has |raw| > 15, so an unclamped backward would be off by e^5 = 148x."""
    ops = _ops()
    C, B, temp = 6, 31, 0.4
    p, b, layout, flat, fs = make_case(C, B, True, 9, 23, log2_T=12, temperature=temp, seed=4)
    with torch.no_grad():
        p.base_b[1][0] += bias - 1.0
        p.base_w[1][0] *= 0.05
        for k, v in p.reference_state_dict().items():
            layout.view(flat, k).copy_(v.to(DEV))
    pos = T.frustum_positions(b["origins"], b["directions"], b["starts"], b["ends"])
    q = (T.scene_contraction_linf(pos) + 2.0) / 4.0
    sel = ((q > 0) & (q < 1)).all(-1)
    enc = T.hash_encode(q * sel[:, None], p.hash_table, p.scalings, p.log2_T).detach().requires_grad_()
    N = enc.shape[0]
    h = T.mlp_forward(enc, list(p.base_w), list(p.base_b))
    sraw, emb = torch.split(h, [1, 15], dim=-1)
    assert float(sraw.abs().min()) > 15.5
    density = T.trunc_exp(sraw) * sel[:, None]
    g = torch.Generator().manual_seed(2)
    cot_d = torch.rand(N, 1, generator=g) - 0.3
    names = [k for k, _ in p.named_parameters() if k.startswith("base_")]
    params = [v for k, v in p.named_parameters() if k.startswith("base_")]
    grads = torch.autograd.grad((density * cot_d).sum(), [enc] + params)
    e = enc.detach().view(N, 16, 2).permute(1, 0, 2).contiguous().to(DEV)
    args = (fs, flat, e, True, pos.to(DEV), b["directions"].to(DEV), sel.float().to(DEV))
    fwd = ops.field_fwd(*args, want_emb=True, want_logits=True)
    assert_close("sigma (forward is NOT clamped)", fwd["sigma"], density.view(-1), 2e-5)
    for logits in (fwd["feat_logits"],):
        d_flat = torch.zeros_like(flat)
        d_enc = ops.field_bwd(*args, fwd["sigma_raw"], fwd["emb"], cot_d.view(-1).to(DEV), torch.zeros(N, B, device=DEV), None, d_flat,
                              feat_logits=logits)
        assert_close("d_enc", d_enc.permute(1, 0, 2).reshape(N, 32), grads[0], 5e-5)
        for k, gref in zip(names, grads[1:]):
            pre, idx = k.rsplit(".", 1)
            got = layout.view(d_flat, f"mlp_base.mlp.layers.{idx}.{'weight' if pre.endswith('_w') else 'bias'}")
            assert_close(f"grad {k}", got, gref, 5e-5)
    unclamped = float(torch.exp(sraw.detach()).abs().max() / torch.exp(sraw.detach().clamp(-15, 15)).abs().max())
    assert unclamped > 100 or unclamped < 0.01  # the clamp is what this test sees


@pytest.mark.parametrize("C,B,spec,temp", [(9, 128, True, 0.3), (4, 141, False, 0.7), (15, 160, True, 0.5)])
def test_field_bwd_many_bands_many_workgroups(C, B, spec, temp):
    """The B > 32 heads backward against the oracle at N = 4,288 (>= 33 workgroup tiles and a partial last tile), d_enc and
    every parameter gradient <= 5e-5; shapes of scripts/cbox_dragon.sh (128 bands) and rgb+spectral.sh (141), and the kernels' limit."""
    ops = _ops()
    R, S = 64, 67
    p, b, layout, flat, fs = make_case(C, B, spec, R, S, log2_T=12, temperature=temp, seed=6)
    pos = T.frustum_positions(b["origins"], b["directions"], b["starts"], b["ends"])
    q = (T.scene_contraction_linf(pos) + 2.0) / 4.0
    sel = ((q > 0) & (q < 1)).all(-1)
    enc = T.hash_encode(q * sel[:, None], p.hash_table, p.scalings, p.log2_T).detach().requires_grad_()
    N = enc.shape[0]
    assert N == 4288 and N % 128 != 0
    h = T.mlp_forward(enc, list(p.base_w), list(p.base_b))
    sraw, emb = torch.split(h, [1, 15], dim=-1)
    density = T.trunc_exp(sraw) * sel[:, None]
    outs = T.field_outputs(p, b["origins"], b["directions"], b["starts"], b["ends"], emb, temp)
    g = torch.Generator().manual_seed(9)
    cot_s, cot_d = torch.rand(N, B, generator=g) - 0.3, torch.rand(N, 1, generator=g) - 0.3
    loss = (outs["spectral"].view(N, B) * cot_s).sum() + (density * cot_d).sum()
    names = [k for k, _ in p.named_parameters() if k != "hash_table"]
    params = [v for k, v in p.named_parameters() if k != "hash_table"]
    grads = torch.autograd.grad(loss, [enc] + params, allow_unused=True)
    e = enc.detach().view(N, 16, 2).permute(1, 0, 2).contiguous().to(DEV)
    args = (fs, flat, e, True, pos.to(DEV), b["directions"].to(DEV), sel.float().to(DEV))
    fwd = ops.field_fwd(*args, want_emb=True, want_logits=True)
    assert_close("spectral", fwd["spectral"], outs["spectral"].view(N, B), 2e-5)
    key = {"base": "mlp_base.mlp", "head": "mlp_head", "feat": "feature_mlp", "dir": "mlp_directional"}
    for logits in (fwd["feat_logits"],):
        d_flat = torch.full_like(flat, float("nan"))
        d_flat[: layout.offset("mlp_base.mlp.layers.0.weight")] = 0
        d_enc = ops.field_bwd(*args, fwd["sigma_raw"], fwd["emb"], cot_d.view(-1).to(DEV), cot_s.to(DEV), None, d_flat, feat_logits=logits)
        assert_close("d_enc", d_enc.permute(1, 0, 2).reshape(N, 32), grads[0], 5e-5)
        for k, gref in zip(names, grads[1:]):
            if k == "endmembers":
                got = layout.view(d_flat, "endmembers")
            else:
                pre, idx = k.rsplit(".", 1)
                got = layout.view(d_flat, f"{key[pre[:-2]]}.layers.{idx}.{'weight' if pre.endswith('_w') else 'bias'}")
            if gref is None:  # mlp_directional without the specular head: unused by the reference (no gradient), left alone here
                assert not spec and k.startswith("dir_"), k
            else:
                assert_close(f"grad {k}", got, gref, 5e-5)


def test_positional_encoding_far_from_the_origin():
    """NeRFEncoding on RAW world positions (umhs_field.py:183-184) with |x| up to the outer occupancy level's extent 8: the kernel
    takes v_sin of fract(x f) in revolutions, the reference sin(fl32(2 pi) x f) -- at |x f| = 16 those differ by ~7e-6 absolute,
    which must stay inside the radiance budget after three MLPs (<= 1e-4; the near-origin cases hold 2e-5)."""
    ops = _ops()
    C, B, temp = 6, 31, 0.4
    p, b, layout, flat, fs = make_case(C, B, True, 40, 32, log2_T=12, temperature=temp, seed=8)
    g = torch.Generator().manual_seed(5)
    N = b["origins"].shape[0]
    # positions uniformly in the shell 4 <= |x|_inf <= 8 (contracted into the grid like any far sample)
    pos = (torch.rand(N, 3, generator=g) * 2 - 1) * 8
    far = pos.abs().max(-1).values < 4
    pos[far] = pos[far] / pos[far].abs().max(-1, keepdim=True).values * (4 + 4 * torch.rand(int(far.sum()), 1, generator=g))
    assert float(pos.abs().max(-1).values.min()) >= 4 - 1e-4
    b["origins"], b["starts"], b["ends"] = pos.clone(), torch.zeros(N, 1), torch.zeros(N, 1)  # o + d * 0 = pos exactly
    density, emb, sraw, sel = T.field_density(p, b["origins"], b["directions"], b["starts"], b["ends"])
    outs = T.field_outputs(p, b["origins"], b["directions"], b["starts"], b["ends"], emb, temp)
    d = dev(b)
    o = ops.FieldFn.apply(flat, d["origins"], d["directions"], d["starts"], d["ends"], fs)
    assert int(sel.sum()) > N // 2
    assert_close("sigma", o[0].view(-1), density.view(-1), 2e-5)
    assert_close("spectral", o[2], outs["spectral"].view(N, B), 1e-4)
    assert_close("specular", o[4], outs["specular"].view(N, B), 1e-4)
    assert_close("abundances", o[5], outs["abundances"].view(N, C), 1e-4)
    pe_ref = T.nerf_encoding(pos)
    pe_exact = torch.sin(2 * np.pi * torch.cat([pos.double()[:, :, None] * torch.tensor([1.0, 2.0], dtype=torch.float64)], -1).reshape(N, 6))
    assert float((pe_ref[:, :6].double() - pe_exact).abs().max()) < 2e-5  # the reference's own distance from exact arithmetic here


# ------------------------------------------------------------------------------------------------------------------- #
@pytest.mark.parametrize("direct", ["1", "0"])
def test_gradient_accumulation_three_micro_batches(direct, monkeypatch):
    """scripts/rgb+spectral.sh:5 trains with --gradient-accumulation_steps 3: gradients of three batches are SUMMED, then one
    optimizer step.  Here: three micro-steps through the gradient sink (the hash-grid scatter in its += mode, the MLP tail
    through a small temporary) -- as the launch-sequence step and as the autograd path -- against the oracle's summed gradient
    (2e-4, the single-step bound) and its Adam step on that sum (1e-4)."""
    monkeypatch.setenv("UMHS_DIRECT_STEP", direct)
    R, S, B, C, temp, log2_T = 48, 24, 31, 6, 0.4, 15
    bands = list(np.linspace(400, 700, B))
    M = T.colour_matrix(bands)
    bs = learnable_batches(3, R, S, B, seed=40)
    p = oracle_params(C, B, True, log2_T, seed=21)
    pipe = hip_pipeline(p, C, B, True, temp, log2_T, bands, gradient_accumulation_steps=3)
    params = [v for _, v in p.named_parameters()]
    total = [torch.zeros_like(v) for v in params]
    field = pipe.model.field
    before = field.flat.detach().clone()
    for i, b in enumerate(bs):
        _, loss, grads = oracle_grads(p, b, R, temp, M)
        total = [t + g for t, g in zip(total, grads)]
        rs, ri, batch = hip_batch(b, M)
        _, loss_dict = pipe.train_iteration(rs, ri, R, batch)
        for k in loss:
            assert abs(float(loss_dict[k]) - float(loss[k])) <= 1e-4 * abs(float(loss[k])), (i, k)
        if i < 2:
            assert torch.equal(field.flat.detach(), before), "no optimizer step inside the accumulation window"
            assert field.flat.grad is not None and field._grad_sink.accumulating == (i > 0)
    assert not torch.equal(field.flat.detach(), before)
    g = field.flat.grad
    L = field.layout
    names = dict(zip([k for k, _ in p.named_parameters()], total))
    assert_close("sum grad hash_table", L.view(g, "mlp_base.encoder.hash_table"), names["hash_table"], 2e-4)
    assert_close("sum grad endmembers", L.view(g, "endmembers"), names["endmembers"], 2e-4)
    assert_close("sum grad base_w0", L.view(g, "mlp_base.mlp.layers.0.weight"), names["base_w.0"], 2e-4)
    assert_close("sum grad head_w1", L.view(g, "mlp_head.layers.1.weight"), names["head_w.1"], 2e-4)
    assert_close("sum grad feat_b2", L.view(g, "feature_mlp.layers.2.bias"), names["feat_b.2"], 2e-4)
    assert_close("sum grad dir_w1", L.view(g, "mlp_directional.layers.1.weight"), names["dir_w.1"], 2e-4)
    with torch.no_grad():
        T.adam_step(params, total, [torch.zeros_like(v) for v in params], [torch.zeros_like(v) for v in params], 1, T.exp_decay_lr(0))
        p.endmembers.clamp_(0, 1)
    sd = field.state_dict()
    for k, v in p.reference_state_dict().items():
        assert_close(f"param after the accumulated step: {k}", sd[k], v, 1e-4)
    # the next window starts from a zeroed gradient: one more micro-step must overwrite, not add
    rs, ri, batch = hip_batch(bs[0], M)
    pipe.train_iteration(rs, ri, R, batch)
    assert field._grad_sink.accumulating is False


def test_trainer_driven_loop_equals_the_standalone_pipeline():
    """The pipeline as nerfstudio's Trainer builds and drives it -- ``config.setup(device=, test_mode=, world_size=, local_rank=,
    grad_scaler=)``, then per step: BEFORE callbacks, zero_grad, get_train_loss_dict, ``loss.backward()``, optimizer.step,
    scheduler.step, AFTER callbacks (Trainer.train_iteration, nerfstudio 1.1.5 [upstream-recalled]) -- against the stand-alone
    pipeline that does all of that itself.  Same data seed, same initial state: 12 steps, parameters within 2e-5."""
    import functools

    from umhsnerf._ns_compat import TrainingCallbackLocation as Loc
    from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig
    from umhsnerf.optim import UMHSAdam, exp_decay_lr
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline, UMHSPipelineConfig
    from test_hip_data import _split

    Bn = 8
    bands = list(np.linspace(420, 680, Bn))
    split, _, _, _ = _split(n=6, B=Bn, const=0.6)
    meta = {"wavelengths": bands, "num_classes": 3}
    mk_dm = lambda cfg, **kw: UMHSDataManager(cfg, device=kw.get("device", DEV), seed=1, train=split, metadata=dict(meta),
                                              **{k: v for k, v in kw.items() if k in ("test_mode", "world_size", "local_rank", "num_classes")})
    mcfg = lambda: UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, background_color="random", log2_hashmap_size=15)
    alone = UMHSPipeline.from_packed_samples(mcfg(), DEV, metadata=meta, seed=2, datamanager=mk_dm(UMHSDataManagerConfig(train_num_rays_per_batch=1024)))
    pcfg = UMHSPipelineConfig(datamanager=UMHSDataManagerConfig(_target=mk_dm, train_num_rays_per_batch=1024), model=mcfg(), num_classes=3)
    driven = pcfg.setup(device=DEV, test_mode="val", world_size=1, local_rank=0, grad_scaler=None)
    assert driven.trainer_driven and driven.optimizer is None and isinstance(driven, UMHSPipeline)
    driven.model.load_state_dict(alone.model.state_dict())
    with pytest.raises(RuntimeError, match="own optimizer"):
        driven.train_iteration(None, None, 0, {})
    steps = 12
    torch.manual_seed(0)
    ref_losses = [float(sum(alone.get_train_loss_dict(s)[1].values())) for s in range(steps)]
    # --- what the Trainer does around the pipeline -----------------------------------------------------------------
    groups = driven.get_param_groups()
    assert list(groups) == ["fields"]
    opt = UMHSAdam(groups["fields"], lr=2e-2, eps=1e-15, weight_decay=0)  # AdamOptimizerConfig(_target=UMHSAdam).setup(params)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: exp_decay_lr(s) / 2e-2)
    cbs = driven.get_training_callbacks(None)
    torch.manual_seed(0)
    got_losses = []
    for step in range(steps):
        for cb in cbs:
            cb.run_callback_at_location(step, location=Loc.BEFORE_TRAIN_ITERATION)
        opt.zero_grad()
        _, loss_dict, metrics = driven.get_train_loss_dict(step=step)
        loss = functools.reduce(torch.add, loss_dict.values())
        assert loss.requires_grad
        loss.backward()
        opt.step()
        sched.step()
        for cb in cbs:
            cb.run_callback_at_location(step, location=Loc.AFTER_TRAIN_ITERATION)
        got_losses.append(float(loss.detach()))
    np.testing.assert_allclose(got_losses, ref_losses, rtol=2e-5)
    tail = alone.model.field.layout.tail_offset()
    assert_close("MLP / endmember parameters", driven.model.field.flat[tail:], alone.model.field.flat[tail:], 2e-5)
    assert_close("hash table", driven.model.field.flat[:tail], alone.model.field.flat[:tail], 1e-3)
    assert "psnr_spectral" in metrics


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_hashgrid_bwd_propagates_non_finite_gradients(bad):
    """the fixed-point bucket reduce must not turn a NaN / Inf upstream gradient into an arbitrary finite table gradient
    (the reference's index_add would hand the optimizer a NaN).  The level that saw it comes back NaN, the others stay exact."""
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    n, log2_T = 3000, 19
    pos = torch.rand(n, 3, generator=g)
    d_enc = torch.randn(16, n, 2, generator=g)
    sc = ops.hash_scalings().to(DEV)
    clean = torch.empty(16 << log2_T, 2, device=DEV)
    ops.hashgrid_bwd(pos.to(DEV), d_enc.to(DEV), sc, log2_T, clean, True, method="partition", overwrite=True)
    assert bool(torch.isfinite(clean).all())
    d_bad = d_enc.clone()
    d_bad[7, 1234, 1] = bad
    got = torch.empty_like(clean)
    ops.hashgrid_bwd(pos.to(DEV), d_bad.to(DEV), sc, log2_T, got, True, method="partition", overwrite=True)
    T = 1 << log2_T
    lvl = got[7 * T:8 * T]
    assert bool(torch.isnan(lvl).any()) and not bool(torch.isfinite(lvl[lvl != 0]).any())
    keep = torch.ones(16 * T, dtype=torch.bool, device=DEV)
    keep[7 * T:8 * T] = False
    assert torch.equal(got[keep], clean[keep])
