"""``method="rgb"`` (the reference's default method and BASELINE configs[0]: scripts/rgb.sh, 256 rays x 64 samples, forward render) on
the GPU path: per-sample colour and density against the oracle's rgb branch, the rendered image per element, gradients, a few
training steps, checkpoints under the reference's key names."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")]
from oracle import torch_ref as T  # noqa: E402
from test_hip_parity import DEV, assert_close, assert_elementwise, dev  # noqa: E402

pytestmark = pytest.mark.gpu
BANDS = [400.0 + 10 * i for i in range(31)]


def _pair(log2_T=14, seed=3, R=256, S=64):
    """The oracle's rgb-method parameters and a model of this package holding the same values (C1 shape by default)."""
    from umhsnerf.umhs_model import UMHSConfig

    p = T.FieldParams(6, 31, False, method="rgb", log2_hashmap_size=log2_T, table_scale=0.5, seed=seed)
    with torch.no_grad():
        p.base_b[1][0] += 1.0
    cfg = UMHSConfig(log2_hashmap_size=log2_T)  # every default of the reference: method="rgb", implementation="torch", random background
    assert cfg.method == "rgb"
    m = cfg.setup(scene_box=None, num_train_data=1, metadata={"wavelengths": BANDS, "num_classes": 6}, num_classes=6, seed=1).to(DEV)
    m.field.load_state_dict({k: v for k, v in p.reference_state_dict().items()}, strict=False)
    b = T.synthetic_batch(R, S, 31, seed=seed + 1)
    return p, m, b


def _samples(b):
    from umhsnerf._ns_compat import packed_ray_samples

    d = dev(b)
    return packed_ray_samples(d["origins"], d["directions"], d["starts"], d["ends"]), d


def test_forward_render_matches_the_oracle_at_the_c1_shape():
    p, m, b = _pair()
    rs, d = _samples(b)
    m.eval()
    with torch.no_grad():
        fo = m.field(rs)
        out = m.get_outputs_from_samples(rs, d["ray_indices"], 256)
    density, emb, _, _ = T.field_density(p, b["origins"], b["directions"], b["starts"], b["ends"])
    ref = T.field_outputs(p, b["origins"], b["directions"], b["starts"], b["ends"], emb, 0.4)
    from umhsnerf._ns_compat import FieldHeadNames

    assert_close("density", fo[FieldHeadNames.DENSITY], density, 2e-5)
    assert_elementwise("per-sample rgb", fo[FieldHeadNames.RGB], ref["rgb"])
    M = T.colour_matrix(np.asarray(BANDS))
    want = T.model_outputs(p, b["origins"], b["directions"], b["starts"], b["ends"], b["ray_indices"], 256, 0.4, M)
    assert set(out) >= {"rgb", "accumulation", "depth", "num_samples_per_ray"}
    assert_elementwise("rgb", out["rgb"], want["rgb"].clamp(0, 1))
    assert_elementwise("accumulation", out["accumulation"], want["accumulation"])
    assert_elementwise("depth", out["depth"], want["depth"])


def test_training_step_gradients_and_loss_match_the_oracle():
    p, m, b = _pair(R=64, S=32)
    rs, d = _samples(b)
    m.train()
    gt = torch.rand(64, 3, generator=torch.Generator().manual_seed(9))
    bg = torch.rand(64, 3, generator=torch.Generator().manual_seed(10))
    out = m.get_outputs_from_samples(rs, d["ray_indices"], 64)
    loss = m.get_loss_dict(out, {"image": gt.to(DEV)}, background=bg.to(DEV))  # (the oracle's draw of the random background)
    assert set(loss) == {"rgb_loss"}
    l = loss["rgb_loss"]
    l.backward()
    M = T.colour_matrix(np.asarray(BANDS))
    want = T.model_outputs(p, b["origins"], b["directions"], b["starts"], b["ends"], b["ray_indices"], 64, 0.4, M)
    lw = T.model_loss(want, None, gt, bg, "rgb")["rgb_loss"]
    assert abs(float(l) - float(lw)) <= 1e-4 * abs(float(lw))
    params = list(p.parameters())
    grads = torch.autograd.grad(lw, params, allow_unused=True)
    g = m.field.flat.grad
    for k, v in p.reference_state_dict().items():
        ref_g = next(gi for gi, pv in zip(grads, params) if pv.data_ptr() == v.data_ptr())
        assert_close("d " + k, m.field.layout.view(g, k), ref_g if ref_g is not None else torch.zeros_like(v), 2e-4)


def test_a_few_training_steps_reduce_the_loss_and_checkpoints_round_trip():
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    _, _, b = _pair(R=128, S=32)
    rs, d = _samples(b)
    pipe = UMHSPipeline.from_packed_samples(UMHSConfig(log2_hashmap_size=14, background_color="black"), DEV,
                                            metadata={"wavelengths": BANDS, "num_classes": 6}, seed=2)
    assert type(pipe.model.field).__name__ == "UMHSRGBField"
    batch = {"image": torch.full((128, 3), 0.25, device=DEV)}
    losses = []
    for _ in range(30):
        _, ld = pipe.train_iteration(rs, d["ray_indices"], 128, batch)
        losses.append(float(ld["rgb_loss"]))
    assert losses[-1] < 0.5 * losses[0], losses[::6]
    sd = pipe.model.state_dict()
    assert {"field.mlp_base.encoder.hash_table", "field.mlp_base.mlp.layers.1.bias", "field.mlp_head.layers.2.weight", "field.aabb"} <= set(sd)
    assert sd["field.mlp_head.layers.0.weight"].shape == (64, 31) and sd["field.mlp_head.layers.2.weight"].shape == (3, 64)
    assert not any(k.startswith(("field.feature_mlp", "field.mlp_directional", "field.endmembers")) for k in sd)
    other = UMHSConfig(log2_hashmap_size=14).setup(scene_box=None, num_train_data=1, metadata={"wavelengths": BANDS, "num_classes": 6},
                                                   num_classes=6, seed=7).to(DEV)
    other.load_state_dict(sd)
    assert torch.equal(other.field.flat, pipe.model.field.flat)


def test_sampler_driven_render_runs():
    """The occupancy grid + marcher in front of the rgb field (density_fn), one eval image chunk."""
    from umhsnerf._ns_compat import RayBundle
    from umhsnerf.umhs_model import UMHSConfig

    m = UMHSConfig(log2_hashmap_size=14).setup(scene_box=None, num_train_data=1, metadata={"wavelengths": BANDS, "num_classes": 6},
                                               num_classes=6, seed=5).to(DEV)
    g = torch.Generator().manual_seed(0)
    o = torch.tensor([0.0, 0.0, -3.0]).repeat(64, 1) + 0.01 * torch.randn(64, 3, generator=g)
    dd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]).repeat(64, 1) + 0.1 * torch.randn(64, 3, generator=g), dim=-1)
    m.train()
    m.update_occupancy_grid(0)
    m.eval()
    with torch.no_grad():
        out = m(RayBundle(origins=o.to(DEV), directions=dd.to(DEV)))
    assert out["rgb"].shape == (64, 3) and bool(torch.isfinite(out["rgb"]).all()) and float(out["rgb"].min()) >= 0.0


def test_a_batch_without_surviving_samples_renders_the_background():
    """with no sample left after the march the rgb method must give an all-background image and a legal (zero) backward,
    not an argument error from the hash-grid entry point (empty tensors carry NULL data pointers)."""
    from umhsnerf import ops
    from umhsnerf._ns_compat import packed_ray_samples

    _, m, _ = _pair(R=8, S=4)
    z3, z1 = torch.zeros(0, 3, device=DEV), torch.zeros(0, 1, device=DEV)
    rs = packed_ray_samples(z3, z3, z1, z1)
    ri = torch.zeros(0, dtype=torch.int64, device=DEV)
    m.train()
    out = m.get_outputs_from_samples(rs, ri, 8)
    assert out["rgb"].shape == (8, 3) and out["accumulation"].shape == (8, 1)
    assert float(out["accumulation"].abs().max()) == 0.0 and float(out["rgb"].abs().max()) == 0.0
    assert int(out["num_samples_per_ray"].sum()) == 0
    out["rgb"].sum().backward()
    assert m.field.flat.grad is None or float(m.field.flat.grad.abs().max()) == 0.0
    # the C entry points themselves: an empty batch is a no-op, not an argument error
    enc = ops.hashgrid_fwd(torch.zeros(0, 3, device=DEV), m.field.layout.view(m.field.flat.detach(), "mlp_base.encoder.hash_table"),
                           m.field.scalings, m.field.layout.log2_hashmap_size, level_major=False)
    assert enc.shape[0] == 0


def test_the_rgb_mlp_kernels_alone_against_the_oracle_with_clamped_trunc_exp():
    """csrc/umhs_rgb.hip by itself (no library GEMM in the rgb field): mlp_base with trunc_exp (incl. pre-activations
    beyond +-15, where the backward clamps) and the SH + mlp_head + sigmoid kernel, outputs and every gradient against the oracle's
    autograd on the same numbers; tail tiles (N not a multiple of 16 or 64), bitwise run-to-run reproducibility of the gradients."""
    from umhsnerf import ops

    g = torch.Generator().manual_seed(5)
    N = 1000 + 13
    enc = (torch.rand(N, 32, generator=g) - 0.5)
    sel = (torch.rand(N, generator=g) > 0.1).float()
    w0, b0 = torch.randn(64, 32, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1
    w1, b1 = torch.randn(16, 64, generator=g) * 0.3, torch.randn(16, generator=g) * 0.1
    b1[0] = 2.0
    w1[0] *= 14.0  # sigma_raw spreads beyond +-15 on part of the samples (|z| up to ~45)
    d_density, d_emb = torch.randn(N, generator=g), torch.randn(N, 15, generator=g)

    def ref(enc_, w0_, b0_, w1_, b1_):
        h = T.mlp_forward(enc_, [w0_, w1_], [b0_, b1_])
        return T.trunc_exp(h[:, 0]) * sel, h[:, 1:]

    leaves = [t.clone().requires_grad_() for t in (enc, w0, b0, w1, b1)]
    dens_r, emb_r = ref(*leaves)
    assert float((T.mlp_forward(enc, [w0, w1], [b0, b1])[:, 0].abs() > 15).float().mean()) > 0.02
    grads_r = torch.autograd.grad([dens_r, emb_r], leaves, [d_density, d_emb])
    dl = [t.to(DEV).requires_grad_() for t in (enc, w0, b0, w1, b1)]
    dens, emb = ops.RgbBaseFn.apply(dl[0], sel.to(DEV), *dl[1:])
    assert_close("rgb base density", dens, dens_r.detach(), 1e-4)  # (exp of pre-activations up to ~40: 1e-6 of the argument is 4e-5 of the value)
    assert_close("rgb base emb", emb, emb_r.detach(), 2e-5)
    grads = torch.autograd.grad([dens, emb], dl, [d_density.to(DEV), d_emb.to(DEV)])
    for name, a, b in zip(("d_enc", "d_w0", "d_b0", "d_w1", "d_b1"), grads, grads_r):
        assert_close(f"rgb base {name}", a, b, 5e-5)
    again = torch.autograd.grad(ops.RgbBaseFn.apply(dl[0], sel.to(DEV), *dl[1:]), dl, [d_density.to(DEV), d_emb.to(DEV)])
    assert all(torch.equal(a, b) for a, b in zip(grads, again)), "the slab reduce sums in a fixed order"
    # trunc_exp's backward below -15 on its own (an unclamped exp(x) would give gradients e^-25 instead of e^-15 there)
    raw = T.mlp_forward(enc, [w0, w1], [b0, b1])[:, 0]
    low = ((raw < -16) & (sel > 0)).float()
    assert float(low.sum()) >= 5
    g_low_r = torch.autograd.grad(ref(*leaves)[0], leaves[0], d_density * low)[0]
    g_low = torch.autograd.grad(ops.RgbBaseFn.apply(dl[0], sel.to(DEV), *dl[1:])[0], dl[0], (d_density * low).to(DEV))[0]
    assert_close("rgb base d_enc through the clamp", g_low, g_low_r, 5e-5)
    assert float(g_low_r.abs().max()) > 1e-9

    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    embs = torch.randn(N, 15, generator=g) * 0.5
    hw = [torch.randn(64, 31, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1, torch.randn(64, 64, generator=g) * 0.2,
          torch.randn(64, generator=g) * 0.1, torch.randn(3, 64, generator=g) * 0.3, torch.randn(3, generator=g) * 0.1]
    d_rgb = torch.randn(N, 3, generator=g)
    hl = [t.clone().requires_grad_() for t in [embs] + hw]
    x = torch.cat([T.sh_encoding_deg4((dirs + 1.0) / 2.0), hl[0]], dim=-1)
    rgb_r = torch.sigmoid(T.mlp_forward(x, [hl[1], hl[3], hl[5]], [hl[2], hl[4], hl[6]]))
    gr = torch.autograd.grad(rgb_r, hl, d_rgb)
    hd = [t.to(DEV).requires_grad_() for t in [embs] + hw]
    rgb = ops.RgbHeadFn.apply(dirs.to(DEV), *hd)
    assert_elementwise("rgb head", rgb, rgb_r.detach(), rtol=2e-5, atol=1e-6)
    gh = torch.autograd.grad(rgb, hd, d_rgb.to(DEV))
    for name, a, b in zip(("d_emb", "d_w0", "d_b0", "d_w1", "d_b1", "d_w2", "d_b2"), gh, gr):
        assert_close(f"rgb head {name}", a, b, 5e-5)
