"""GPU tests at BASELINE.json's full sizes (where the CPU oracle is too slow) through size-independent properties, plus
edge cases and the stand-alone renderer API against the reference's own golden outputs."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from umhsnerf import ops

    return ops


def _c2(seed=0, R=4096, S=64, B=31, C=6):
    ops = _ops()
    layout = ops.FieldLayout(C, B, True, 19)
    g = torch.Generator().manual_seed(seed)
    flat = ((torch.rand(layout.total, generator=g) - 0.5) * 0.6)
    layout.view(flat, "endmembers").copy_(torch.rand(C, B, generator=g))
    layout.view(flat, "mlp_base.mlp.layers.1.bias")[0] += 1.5
    fs = ops.FieldSpec(layout, 0.4, True, scalings=ops.hash_scalings().to(DEV))
    b = T.synthetic_batch(R, S, B, seed=seed + 1)
    return layout, flat.to(DEV), fs, {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}


def test_full_size_c2_properties():
    """N = 262,144 (config C2): determinism, layout independence, atomic vs partitioned scatter, compositing invariants."""
    ops = _ops()
    layout, flat, fs, b = _c2()
    R, N = 4096, 4096 * 64
    o, d, s, e = b["origins"], b["directions"], b["starts"].view(-1), b["ends"].view(-1)
    wpos, pos01, sel = ops.positions_fwd(o, d, s, e, fs)
    table = layout.view(flat, "mlp_base.encoder.hash_table")
    enc_lm = ops.hashgrid_fwd(pos01, table, fs.scalings, 19, True)
    enc_rm = ops.hashgrid_fwd(pos01, table, fs.scalings, 19, False)
    assert torch.equal(enc_lm.permute(1, 0, 2).reshape(N, 32), enc_rm)  # the two layouts hold identical bits
    out1 = ops.field_fwd(fs, flat, enc_lm, True, wpos, d, sel, want_emb=True)
    out2 = ops.field_fwd(fs, flat, enc_rm, False, wpos, d, sel, want_emb=True)
    for k in ("sigma", "spectral", "spectral2", "specular", "abundances", "emb"):
        assert torch.equal(out1[k], out2[k]), k  # deterministic, layout-independent
    ab = out1["abundances"]
    assert float((ab.sum(-1) - 1).abs().max()) < 1e-5 and float(ab.min()) >= 0  # softmax rows
    assert torch.allclose(out1["spectral"], out1["spectral2"] + out1["specular"], rtol=0, atol=1e-6)
    assert bool(torch.isfinite(out1["spectral"]).all()) and float(out1["sigma"].min()) >= 0
    # compositing invariants
    pinfo = ops.pack_info(b["ray_indices"], R)
    assert torch.equal(pinfo[:, 1], torch.full((R,), 64, device=DEV)) and int(pinfo[-1, 0]) == N - 64
    w, acc, depth, (c1,) = ops.composite_fwd(out1["sigma"], s, e, pinfo, [out1["spectral"]])
    assert float(acc.max()) <= 1 + 1e-5 and float(w.min()) >= 0
    assert torch.allclose(w.view(R, 64).sum(1), acc, atol=1e-5)
    _, _, _, (c2,) = ops.composite_fwd(out1["sigma"], s, e, pinfo, [out1["spectral"] * 3.0])
    assert torch.allclose(c2, 3.0 * c1, rtol=1e-6, atol=1e-7)  # linearity in the composited values
    ones = torch.ones(N, 4, device=DEV)
    _, _, _, (c3,) = ops.composite_fwd(out1["sigma"], s, e, pinfo, [ones])
    assert torch.allclose(c3, acc[:, None].expand(R, 4), atol=1e-5)  # compositing 1 gives the accumulation
    # hash-grid backward: memory-side float atomics vs the partitioned integer accumulation, full size
    d_enc = torch.randn(16, N, 2, device=DEV) * torch.rand(1, N, 1, device=DEV)
    ga, gp = torch.zeros_like(table), torch.zeros_like(table)
    ops.hashgrid_bwd(pos01, d_enc, fs.scalings, 19, ga, True, method="atomic")
    ops.hashgrid_bwd(pos01, d_enc, fs.scalings, 19, gp, True, method="partition")
    assert float((ga - gp).abs().max() / ga.abs().max()) < 2e-5
    assert float(gp.sum() / ga.sum() - 1) < 1e-4
    gp2 = torch.zeros_like(table)
    ops.hashgrid_bwd(pos01, d_enc, fs.scalings, 19, gp2, True, method="partition")
    assert torch.equal(gp, gp2)  # bitwise reproducible at full size


def test_full_size_field_bwd_linearity_and_reduction():
    """dL/dparams is linear in the upstream gradients and sums over disjoint sample sets (N = 262,144 vs two halves)."""
    ops = _ops()
    layout, flat, fs, b = _c2(seed=3)
    N, B = 4096 * 64, 31
    o, d, s, e = b["origins"], b["directions"], b["starts"].view(-1), b["ends"].view(-1)
    wpos, pos01, sel = ops.positions_fwd(o, d, s, e, fs)
    enc = ops.hashgrid_fwd(pos01, layout.view(flat, "mlp_base.encoder.hash_table"), fs.scalings, 19, True)
    out = ops.field_fwd(fs, flat, enc, True, wpos, d, sel, want_emb=True, want_logits=True)
    g = torch.Generator(device=DEV).manual_seed(1)
    ds, dsp = torch.rand(N, device=DEV, generator=g) - 0.5, torch.rand(N, B, device=DEV, generator=g) - 0.5

    def bwd(dsig, dspec, idx=None):
        df = torch.zeros_like(flat)
        if idx is None:
            de = ops.field_bwd(fs, flat, enc, True, wpos, d, sel, out["sigma_raw"], out["emb"], dsig, dspec, None, df,
                               feat_logits=out["feat_logits"])
        else:
            sl = lambda t: t[idx].contiguous()
            ee = enc[:, idx].contiguous()
            de = ops.field_bwd(fs, flat, ee, True, sl(wpos), sl(d), sl(sel), sl(out["sigma_raw"]), sl(out["emb"]), sl(dsig),
                               sl(dspec), None, df, feat_logits=sl(out["feat_logits"]))
        return df[layout.offset("mlp_base.mlp.layers.0.weight"):], de

    g_full, de_full = bwd(ds, dsp)
    g_2x, _ = bwd(2 * ds, 2 * dsp)
    assert float((g_2x - 2 * g_full).abs().max() / g_full.abs().max()) < 1e-5
    half = N // 2
    g_a, de_a = bwd(ds, dsp, slice(0, half))
    g_b, _ = bwd(ds, dsp, slice(half, N))
    assert float((g_a + g_b - g_full).abs().max() / g_full.abs().max()) < 2e-5
    assert torch.equal(de_a, de_full[:, :half])  # per-sample d_enc does not depend on the batch it is part of


def test_edge_cases_empty_and_tiny():
    ops = _ops()
    layout, flat, fs, _ = _c2(R=4, S=4)
    # one sample, one ray
    b = T.synthetic_batch(1, 1, 31, seed=2)
    bd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}
    fl = flat.clone().requires_grad_()
    dens, emb, spec, spec2, spl, ab = ops.FieldFn.apply(fl, bd["origins"], bd["directions"], bd["starts"], bd["ends"], fs)
    assert spec.shape == (1, 31) and dens.shape == (1, 1)
    pinfo = ops.pack_info(bd["ray_indices"], 1)
    w, acc, depth, comp = ops.CompositeFn.apply(dens, bd["starts"], bd["ends"], pinfo, True, spec)
    comp.sum().backward()
    assert bool(torch.isfinite(fl.grad).all()) and float(fl.grad.abs().sum()) > 0
    # rays without samples: zero outputs, zero-length pack_info rows, no NaN from 0/(0+eps)
    ri = torch.tensor([1, 1, 1, 3, 3], device=DEV)
    pinfo = ops.pack_info(ri, 5)
    assert pinfo.tolist() == [[0, 0], [0, 3], [3, 0], [3, 2], [5, 0]]
    sig = torch.rand(5, device=DEV) * 50
    t0 = torch.tensor([0.1, 0.2, 0.3, 0.1, 0.2], device=DEV)
    v = torch.rand(5, 7, device=DEV)
    w, acc, depth, (o,) = ops.composite_fwd(sig, t0, t0 + 0.05, pinfo, [v])
    assert o[0].abs().sum() == 0 and o[2].abs().sum() == 0 and o[4].abs().sum() == 0 and acc[0] == 0 and depth[4] == 0
    # unsupported shapes are refused with an error code, not a fault
    big = ops.FieldLayout(16, 31, True, 12)
    with pytest.raises(RuntimeError, match="not supported"):
        ops.field_fwd(ops.FieldSpec(big, 0.4, True, scalings=fs.scalings), torch.zeros(big.total, device=DEV),
                      torch.zeros(16, 8, 2, device=DEV), True, torch.zeros(8, 3, device=DEV), torch.zeros(8, 3, device=DEV),
                      torch.ones(8, device=DEV))


def test_renderer_api_against_reference_goldens(golden_dir):
    """SpectralRenderer.forward / get_weights_spectral / blend_background_for_loss_computation vs G3, G5 and the oracle."""
    from umhsnerf.umhs_renderer import SpectralRenderer, get_weights_spectral

    ops = _ops()
    g3 = np.load(os.path.join(golden_dir, "g3_weights.npz"))
    w = get_weights_spectral(torch.from_numpy(g3["deltas"]).to(DEV), torch.from_numpy(g3["densities"]).to(DEV))
    np.testing.assert_allclose(w.cpu().numpy(), g3["weights"], rtol=2e-5, atol=2e-7)  # 1-exp(-x): one ulp of 1.0 absolute
    g5 = np.load(os.path.join(golden_dir, "g5_blend.npz"))
    r = SpectralRenderer()
    torch.manual_seed(1234)
    pred = torch.from_numpy(g5["pred"]).to(DEV)
    p2, gt2 = r.blend_background_for_loss_computation(pred, torch.from_numpy(g5["acc"]).to(DEV), torch.from_numpy(g5["gt"]).to(DEV),
                                                       torch.from_numpy(g5["gt"]).to(DEV))
    assert p2.shape == pred.shape and torch.equal(gt2.cpu(), torch.from_numpy(g5["gt_out"]))  # bg RNG differs per device
    # packed accumulate with caller-provided weights, incl. the leading-1 squeeze and gradients to both inputs
    b = T.synthetic_batch(23, 17, 5, seed=4, ragged=True)
    N = b["origins"].shape[0]
    g = torch.Generator().manual_seed(0)
    vals, wts = torch.rand(1, N, 5, generator=g, requires_grad=True), torch.rand(N, 1, generator=g, requires_grad=True)
    ref = T.spectral_renderer(vals, wts, b["ray_indices"], 23)
    cot = torch.rand(23, 5, generator=g)
    gv, gw = torch.autograd.grad((ref * cot).sum(), [vals, wts])
    vd, wd = vals.detach().to(DEV).requires_grad_(), wts.detach().to(DEV).requires_grad_()
    got = r(vd, wd, b["ray_indices"].to(DEV), 23)
    assert float((got.cpu() - ref).abs().max()) < 1e-5
    (got * cot.to(DEV)).sum().backward()
    assert float((vd.grad.cpu() - gv).abs().max()) < 1e-5 and float((wd.grad.cpu() - gw).abs().max()) < 1e-5
    dense = r(vd.detach()[:, :9], wd.detach()[:9])  # no ray_indices: one ray, sum over dim -2
    assert float((dense.cpu() - (vals[0, :9] * wts[:9]).sum(0).detach()).abs().max()) < 1e-5


@pytest.mark.parametrize("method,specular", [("rgb+spectral", True), ("spectral", False)])
def test_direct_training_step_equals_the_autograd_path(method, specular, monkeypatch):
    """UMHSPipeline.train_iteration runs the launch sequence without an autograd graph; the autograd Functions stay the general
    path.  Same kernels, same order: losses, outputs, gradient and parameters after 3 steps must agree."""
    from umhsnerf import ops
    from umhsnerf._ns_compat import packed_ray_samples
    from umhsnerf.umhs_model import BandOutputs, UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    R, S, B, Cn = 512, 24, 31, 5
    b = T.synthetic_batch(R, S, B, seed=9)
    b = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}
    rs = packed_ray_samples(b["origins"], b["directions"], b["starts"], b["ends"])
    pinfo = ops.pack_info(b["ray_indices"], R)
    res = {}
    for direct in ("1", "0"):
        monkeypatch.setenv("UMHS_DIRECT_STEP", direct)
        torch.manual_seed(4)
        cfg = UMHSConfig(method=method, pred_specular=specular, temperature=0.5, per_band_outputs=True)
        pipe = UMHSPipeline.from_packed_samples(cfg, DEV, metadata={"wavelengths": list(np.linspace(400, 700, B)), "num_classes": Cn}, seed=8)
        with torch.no_grad():
            tab = pipe.model.field.layout.view(pipe.model.field.flat.data, "mlp_base.encoder.hash_table")
            tab.mul_(300.0)
            batch = {"image": pipe.model.converter(b["gt_spectral"]), "hs_image": b["gt_spectral"]}
        assert pipe.model.direct_step_supported(batch) == (direct == "1")
        out, loss = pipe.train_iteration(rs, b["ray_indices"], R, batch, packed_info=pinfo)
        g1 = pipe.model.field.flat.grad.clone()
        for _ in range(2):
            out, loss = pipe.train_iteration(rs, b["ray_indices"], R, batch, packed_info=pinfo)
        res[direct] = (out, {k: float(v) for k, v in loss.items()}, g1, pipe.model.field.flat.detach().clone())
    (o1, l1, g1, p1), (o0, l0, g0, p0) = res["1"], res["0"]
    assert isinstance(o1, BandOutputs) and l1.keys() == l0.keys()
    for k in l0:
        assert abs(l1[k] - l0[k]) <= 1e-6 * abs(l0[k])
    assert float(g0.abs().max()) > 0
    torch.testing.assert_close(g1, g0, rtol=1e-5, atol=1e-9 + 1e-6 * float(g0.abs().max()))
    torch.testing.assert_close(p1, p0, rtol=0, atol=2e-3)  # Adam turns last-bit gradient differences into +-lr steps on ~zero entries
    assert float((p1 - p0).abs().mean()) < 1e-6
    for k in o0:  # every key of the eager dict is reachable (per-band ones lazily), with the same values
        assert k in o1
        torch.testing.assert_close(o1[k].float(), o0[k].detach().float(), rtol=1e-5, atol=1e-6)
    assert "wv_31" not in o1 and "foo" not in o1
    with pytest.raises(KeyError):
        o1["residual_99"]
    assert set(o1.materialize().keys()) == set(o0.keys())


@pytest.mark.parametrize("name,R,S,B,C,spec,temp", [
    ("C3_cbox_dragon_128band", 8192, 64, 128, 9, True, 0.3),
    ("C4_pinecone_per_gpu_share", 8192, 64, 31, 4, True, 0.5),
    ("C5_joint_141band_nospec", 8192, 64, 141, 4, False, 0.7),
])
def test_full_size_training_step_other_configs(name, R, S, B, C, spec, temp, monkeypatch):
    """BASELINE configs C3 / C4 (one GPU's share) / C5 at full size through the pipeline: the straight launch sequence and the
    autograd path agree on losses and gradient, outputs satisfy the model's invariants, and three steps reduce the loss."""
    from umhsnerf import ops
    from umhsnerf._ns_compat import packed_ray_samples
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    b = T.synthetic_batch(R, S, B, seed=21)
    b = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}
    rs = packed_ray_samples(b["origins"], b["directions"], b["starts"], b["ends"])
    pinfo = ops.pack_info(b["ray_indices"], R)
    res = {}
    for direct in ("1", "0"):
        monkeypatch.setenv("UMHS_DIRECT_STEP", direct)
        torch.manual_seed(5)
        cfg = UMHSConfig(method="rgb+spectral", pred_specular=spec, temperature=temp, per_band_outputs=False)
        pipe = UMHSPipeline.from_packed_samples(cfg, DEV, metadata={"wavelengths": list(np.linspace(400, 700, B)), "num_classes": C}, seed=6)
        with torch.no_grad():
            pipe.model.field.layout.view(pipe.model.field.flat.data, "mlp_base.encoder.hash_table").mul_(300.0)
            batch = {"image": pipe.model.converter(b["gt_spectral"]), "hs_image": b["gt_spectral"]}
        out, loss = pipe.train_iteration(rs, b["ray_indices"], R, batch, packed_info=pinfo)
        g = pipe.model.field.flat.grad.clone()
        first = {k: float(v.detach()) for k, v in loss.items()}
        if direct == "1":
            assert out["spectral"].shape == (R, B) and out["abundances"].shape == (R, C) and out["rgb"].shape == (R, 3)
            assert bool(torch.isfinite(out["spectral"]).all()) and float(out["accumulation"].max()) <= 1 + 1e-5
            assert float((out["abundances"].sum(-1) - out["accumulation"][:, 0]).abs().max()) < 1e-4  # composited softmax rows
            if spec:
                torch.testing.assert_close(out["spectral"], out["spectral2"] + out["specular"], rtol=1e-5, atol=1e-6)
            for _ in range(2):
                out, loss = pipe.train_iteration(rs, b["ray_indices"], R, batch, packed_info=pinfo)
            assert sum(float(v.detach()) for v in loss.values()) < sum(first.values())
        res[direct] = (first, g)
    (l1, g1), (l0, g0) = res["1"], res["0"]
    for k in l0:
        assert abs(l1[k] - l0[k]) <= 2e-6 * abs(l0[k]), (name, k, l1[k], l0[k])
    torch.testing.assert_close(g1, g0, rtol=1e-5, atol=1e-9 + 2e-6 * float(g0.abs().max()))


def test_adam_step_riding_in_the_reduce_pass_is_the_same_update(monkeypatch):
    """One GPU: UMHSPipeline arms the optimizer before the backward, and the bucket reduce of the hash-grid gradient applies Adam to
    the dense table levels in its epilogue (umhs_hashgrid_bwd_apply_adam); optimizer.step() does the rest.  Same arithmetic ->
    the very same parameters, moments and gradients as the separate Adam launch, step after step (lr decays in between)."""
    from umhsnerf import ops
    from umhsnerf._ns_compat import packed_ray_samples
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    R, S, B, Cn = 512, 24, 31, 5
    b = T.synthetic_batch(R, S, B, seed=9)
    b = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}
    rs = packed_ray_samples(b["origins"], b["directions"], b["starts"], b["ends"])
    pinfo = ops.pack_info(b["ray_indices"], R)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("UMHS_FUSED_ADAM", fused)
        torch.manual_seed(4)
        cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.5, per_band_outputs=False)
        pipe = UMHSPipeline.from_packed_samples(cfg, DEV, metadata={"wavelengths": list(np.linspace(400, 700, B)), "num_classes": Cn}, seed=8)
        with torch.no_grad():
            pipe.model.field.layout.view(pipe.model.field.flat.data, "mlp_base.encoder.hash_table").mul_(300.0)
            batch = {"image": pipe.model.converter(b["gt_spectral"]), "hs_image": b["gt_spectral"]}
        sink, rode = pipe.model.field._spec().grad_sink, 0
        losses = []
        for _ in range(6):
            pipe.optimizer.zero_grad(set_to_none=True)
            armed = pipe.optimizer.arm_fused()
            assert armed == (fused == "1")
            _, loss = pipe.model.forward_backward_from_samples(rs, b["ray_indices"], R, batch, pinfo)
            rode += int(sink.adam_done is not None)
            pipe.optimizer.step()
            assert sink.adam_done is None and sink.fused_adam is None
            losses.append([float(v) for v in loss.values()])
        st = pipe.optimizer.state[pipe.model.field.flat]
        res[fused] = (losses, pipe.model.field.flat.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                      pipe.model.field.flat.grad.clone(), st["step"], rode)
    assert res["1"][6] == 6 and res["0"][6] == 0 and res["1"][5] == res["0"][5] == 6
    assert res["1"][0] == res["0"][0]
    for i in (1, 2, 3, 4):
        assert torch.equal(res["1"][i], res["0"][i]), i
    # an armed step that the next optimizer.step() does not match is refused rather than applied twice
    pipe.optimizer.zero_grad(set_to_none=True)
