"""GPU parity tests: every C-ABI operator of libumhs_hip.so against the CPU oracle (oracle/torch_ref.py)
on the same seeded inputs.  Tolerances: north_star asks 1e-4 relative on rendered radiance; per-operator
checks are tighter where the arithmetic allows.  Integer/index outputs (pack_info) are bit-exact."""
import math
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


def relerr(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def assert_close(name, got, want, rtol):
    e = relerr(got, want)
    assert e <= rtol, f"{name}: max|diff|/max|ref| = {e:.3e} > {rtol:.1e} (shape {tuple(want.shape)})"


def assert_elementwise(name, got, want, rtol=1e-4, atol=1e-6) -> float:
    """|got - want| <= rtol |want| + atol for EVERY element (north_star: rendered radiance within 1e-4 relative -- a max-norm
    bound would let a dark band be off by far more than that of its own value).  Returns the worst |diff| / (rtol |want| + atol)."""
    a, b = got.detach().double().cpu().reshape(-1), want.detach().double().cpu().reshape(-1)
    assert a.shape == b.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    ratio = (a - b).abs() / (rtol * b.abs() + atol)
    worst = int(ratio.argmax()) if ratio.numel() else 0
    r = float(ratio[worst]) if ratio.numel() else 0.0
    assert r <= 1.0, (f"{name}: element {worst}: got {float(a[worst]):.8g}, want {float(b[worst]):.8g} "
                      f"(|diff| = {float((a - b).abs()[worst]):.3e} = {r:.2f} x the bound rtol {rtol:.0e} / atol {atol:.0e}; "
                      f"{int((ratio > 1).sum())} of {ratio.numel()} elements over)")
    return r


def make_case(C, B, spec, R, S, ragged=False, log2_T=19, seed=0, temperature=0.4):
    ops = _ops()
    p = T.FieldParams(C, B, spec, log2_hashmap_size=log2_T, table_scale=0.5, seed=seed)
    with torch.no_grad():
        p.base_b[1][0] += 1.0  # raise sigma so a good share of samples has alpha > 0.01
    batch = T.synthetic_batch(R, S, B, seed=seed + 1, ragged=ragged)
    layout = ops.FieldLayout(C, B, spec, log2_T)
    flat = torch.zeros(layout.total)
    for k, v in p.reference_state_dict().items():
        layout.view(flat, k).copy_(v)
    fs = ops.FieldSpec(layout, temperature, True, scalings=ops.hash_scalings().to(DEV))
    return p, batch, layout, flat.to(DEV), fs


def dev(batch):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}


# --------------------------------------------------------------------------------------------- #
def test_positions_and_selector():
    ops = _ops()
    _, b, _, _, fs = make_case(4, 8, False, 64, 16, log2_T=12)
    d = dev(b)
    wpos, pos01, sel = ops.positions_fwd(d["origins"], d["directions"], d["starts"].view(-1), d["ends"].view(-1), fs)
    pos = T.frustum_positions(b["origins"], b["directions"], b["starts"], b["ends"])
    q = (T.scene_contraction_linf(pos) + 2.0) / 4.0
    s = ((q > 0) & (q < 1)).all(-1)
    assert_close("world_pos", wpos, pos, 1e-6)
    assert_close("pos01", pos01, q * s[:, None], 1e-6)
    assert torch.equal(sel.cpu() > 0.5, s)
    # density_fn path (positions given) and the aabb-normalised branch, incl. points outside the box
    g = torch.Generator().manual_seed(3)
    P = (torch.rand(500, 3, generator=g) * 6 - 3)
    fs2 = ops.FieldSpec(fs.layout, 0.4, False, aabb=(-1, -1, -1, 1, 1, 1), scalings=fs.scalings)
    _, p2, s2 = ops.positions_fwd(None, None, None, None, fs2, world_pos_in=P.to(DEV))
    q2 = (P + 1) / 2
    m2 = ((q2 > 0) & (q2 < 1)).all(-1)
    assert torch.equal(s2.cpu() > 0.5, m2) and 0 < int(m2.sum()) < 500
    assert_close("pos01_aabb", p2, q2 * m2[:, None], 1e-6)


@pytest.mark.parametrize("method", ["partition", "atomic"])
@pytest.mark.parametrize("level_major", [True, False])
@pytest.mark.parametrize("log2_T", [19, 12, 20])  # 20: 128 buckets per level (two per lane of the partition kernel's scan wave)
def test_hashgrid_fwd_bwd(level_major, log2_T, method):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    if log2_T == 20 and (method == "atomic" or not level_major):
        pytest.skip("the 2^20 table is only there for the 128-bucket partition path")
    N = 3000
    x = torch.rand(N, 3, generator=g)
    x[:7] = 0.0  # masked samples sit exactly on a vertex
    x[7] = torch.tensor([1.0 / 16, 0.5, 0.25])  # exact integer coordinates at level 0 (ceil == floor)
    table = ((torch.rand(16 << log2_T, 2, generator=g) * 2 - 1) * 0.5).requires_grad_()
    sc = T.hash_scalings()
    ref = T.hash_encode(x, table, sc, log2_T)
    enc = ops.hashgrid_fwd(x.to(DEV), table.detach().to(DEV), sc.to(DEV), log2_T, level_major)
    got = enc.permute(1, 0, 2).reshape(N, 32) if level_major else enc
    assert_close("enc", got, ref, 2e-6)
    cot = torch.rand(N, 32, generator=g)
    cot[100:200] = 0.0
    (gref,) = torch.autograd.grad((ref * cot).sum(), table)
    d_enc = cot.view(N, 16, 2).permute(1, 0, 2).contiguous() if level_major else cot
    d_table = torch.zeros_like(table.detach()).to(DEV)
    ops.hashgrid_bwd(x.to(DEV), d_enc.to(DEV), sc.to(DEV), log2_T, d_table, level_major, method=method)
    assert_close("d_table", d_table, gref, 2e-5)
    # accumulate semantics: a second call doubles the gradient
    ops.hashgrid_bwd(x.to(DEV), d_enc.to(DEV), sc.to(DEV), log2_T, d_table, level_major, method=method)
    assert_close("d_table x2", d_table, 2 * gref, 2e-5)
    # overwrite semantics: garbage in, gradient out
    d_table.fill_(123.0)
    ops.hashgrid_bwd(x.to(DEV), d_enc.to(DEV), sc.to(DEV), log2_T, d_table, level_major, method=method, overwrite=True)
    assert_close("d_table overwrite", d_table, gref, 2e-5)


def test_hashgrid_bwd_partition_reproducible_and_keeps_small_gradients():
    """The int64 fixed-point LDS accumulation is exact: bitwise identical run to run, and slots whose gradient is
    many orders of magnitude below the level maximum keep float32-level RELATIVE accuracy (Adam normalises per slot)."""
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    N, log2_T = 20000, 19
    x = torch.rand(N, 3, generator=g)
    x[:4096] = torch.rand(64, 1, 3, generator=g).expand(64, 64, 3).reshape(-1, 3) + torch.linspace(0, 2e-3, 64).view(1, 64, 1).expand(64, 64, 3).reshape(-1, 3) * 0.1
    x = x.clamp(1e-4, 1 - 1e-4)
    sc = T.hash_scalings()
    cot = (torch.rand(N, 32, generator=g) - 0.5) * (10.0 ** (-7 * torch.rand(N, 1, generator=g)))
    # Reference: every contribution w_corner*g computed in fp32 exactly as the reference's torch ops would (offsets
    # from the fp32-rounded x*scale), but ACCUMULATED in float64 -- so it has no cancellation error of its own.
    L, Tn = 16, 1 << log2_T
    scaled = x[:, None, :] * sc.view(-1, 1)
    cc, ff = torch.ceil(scaled).to(torch.int32), torch.floor(scaled).to(torch.int32)
    off = scaled - ff
    lo = torch.arange(L, dtype=torch.int64) * Tn
    pick = lambda a, b_, c: torch.cat([a[..., 0:1], b_[..., 1:2], c[..., 2:3]], dim=-1)
    corners = [pick(cc, cc, cc), pick(cc, ff, cc), pick(ff, ff, cc), pick(ff, cc, cc), pick(cc, cc, ff), pick(cc, ff, ff), pick(ff, ff, ff), pick(ff, cc, ff)]
    ox, oy, oz = off[..., 0], off[..., 1], off[..., 2]
    rx, ry, rz = 1 - ox, 1 - oy, 1 - oz
    ws = [ox * oy * oz, ox * ry * oz, rx * ry * oz, rx * oy * oz, ox * oy * rz, ox * ry * rz, rx * ry * rz, rx * oy * rz]
    gg = cot.view(N, L, 2)
    ref = torch.zeros(L * Tn, 2, dtype=torch.float64)
    mag = torch.zeros(L * Tn, 2, dtype=torch.float64)
    for cn, wc in zip(corners, ws):
        idx = T.hash_fn(cn, Tn, lo).reshape(-1)
        contrib = (wc[..., None] * gg).reshape(-1, 2)  # fp32 products, like the kernel
        ref.index_add_(0, idx, contrib.double())
        mag.index_add_(0, idx, contrib.double().abs())
    d_enc = cot.view(N, 16, 2).permute(1, 0, 2).contiguous().to(DEV)
    outs = []
    for _ in range(2):
        d_table = torch.zeros(16 << log2_T, 2, device=DEV)
        ops.hashgrid_bwd(x.to(DEV), d_enc, sc.to(DEV), log2_T, d_table, True, method="partition")
        outs.append(d_table.cpu())
    assert torch.equal(outs[0], outs[1])  # order-independent integer sums: bitwise reproducible
    got = outs[0].double()
    touched = mag > 0
    assert float(mag[touched].min() / mag.max()) < 1e-7  # slots spanning > 7 decades of gradient magnitude
    err = (got - ref).abs()
    assert bool((err[touched] <= 1e-5 * mag[touched] + 1e-13 * mag.max()).all()), float((err / mag.clamp_min(1e-300))[touched].max())
    assert float(got[~touched].abs().max()) == 0.0


def test_hashgrid_bwd_pairs_that_straddle_two_buckets_and_the_overflow_guard():
    """Round 4's record format pairs the two x-neighbour corners of a sample in ONE record, which needs both slots in one bucket of
    2^13 slots: always so below resolution 8192.  Above it (any scale is legal at the C ABI) a floor coordinate = 8191 mod 8192 puts
    the pair in two buckets: it is emitted as two singles, the workgroup may exceed its LDS staging (then it writes straight to
    memory), and a level whose records exceed its region (5 per sample) comes back as NaN -- loudly -- never silently wrong."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    N, log2_T = 4096, 19
    sc = T.hash_scalings().clone()
    sc[12:] = torch.tensor([9000.0, 17000.0, 33000.0, 40000.0])  # four levels above 8192
    x = torch.rand(N, 3, generator=g) * 0.98 + 0.01
    # every fourth sample: x chosen so that floor(x * scale_13) = 8191 (mod 8192) at level 13 -> its four x-pairs straddle two buckets there
    pick = torch.arange(0, N, 4)
    x[pick, 0] = (8191.0 + 0.25 + 0.5 * torch.rand(pick.numel(), generator=g)) / float(sc[13])
    table = torch.zeros(16 << log2_T, 2, requires_grad=True)
    cot = torch.rand(N, 32, generator=g)
    (gref,) = torch.autograd.grad((T.hash_encode(x, table, sc, log2_T) * cot).sum(), table)
    fl = torch.floor(x[:, 0] * sc[13]).long()
    assert int(((fl % 8192) == 8191).sum()) >= N // 4 - 8
    d_enc = cot.view(N, 16, 2).permute(1, 0, 2).contiguous()
    outs = []
    for _ in range(2):
        d_table = torch.zeros(16 << log2_T, 2, device=DEV)
        ops.hashgrid_bwd(x.to(DEV), d_enc.to(DEV), sc.to(DEV), log2_T, d_table, True, method="partition")
        outs.append(d_table.cpu())
    assert torch.equal(outs[0], outs[1])
    assert_close("d_table with straddling pairs", outs[0], gref, 2e-5)
    # ALL samples straddling at level 13: 8 records per sample > the region's 5 -> that level (only) is NaN, every other level is right
    x2 = x.clone()
    x2[:, 0] = (8191.0 + 0.25 + 0.5 * torch.rand(N, generator=g)) / float(sc[13])
    (gref2,) = torch.autograd.grad((T.hash_encode(x2, table, sc, log2_T) * cot).sum(), table)
    d_table = torch.zeros(16 << log2_T, 2, device=DEV)
    ops.hashgrid_bwd(x2.to(DEV), d_enc.to(DEV), sc.to(DEV), log2_T, d_table, True, method="partition", overwrite=True)
    got = d_table.cpu()
    Tn = 1 << log2_T
    assert bool(torch.isnan(got[13 * Tn:14 * Tn]).all()), "an overflowing level must be NaN"
    keep = torch.ones(16 * Tn, dtype=torch.bool)
    keep[13 * Tn:14 * Tn] = False
    assert_close("the other levels", got[keep], gref2[keep], 2e-5)


@pytest.mark.parametrize("kind", ["ray_samples", "one_cell", "scattered", "straddling"])
def test_hashgrid_fwd_count_equals_forward_plus_prepare(kind):
    """The training step takes the backward's bucket histogram inside the forward gather's launch (umhs_hashgrid_fwd_count +
    umhs_hashgrid_bwd_prepare_counted).  It must be umhs_hashgrid_fwd + umhs_hashgrid_bwd_prepare bit for bit: the features, every
    byte the prepare leaves in the workspace (per-workgroup counts, prefixes, bucket offsets), and so the gradient.  Sample sets:
    consecutive samples along rays (runs merge on the coarse levels), all in one cell, independent positions, and pairs that
    straddle two buckets; N not a multiple of the 512-sample run."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    N, log2_T = 3 * 512 * 37 + 129, 19
    sc = T.hash_scalings().clone()
    if kind == "ray_samples":
        o = torch.rand(N // 48 + 1, 1, 3, generator=g) * 0.5 + 0.1
        d = torch.nn.functional.normalize(torch.randn(N // 48 + 1, 1, 3, generator=g), dim=-1)
        x = (o + d * torch.linspace(0, 0.3, 48).view(1, 48, 1)).reshape(-1, 3)[:N].clamp(0.0, 1.0).contiguous()
    elif kind == "one_cell":
        x = 0.4 + torch.rand(N, 3, generator=g) * 1e-3
    else:
        x = torch.rand(N, 3, generator=g)
        if kind == "straddling":
            sc[12:] = torch.tensor([9000.0, 17000.0, 33000.0, 40000.0])
            pick = torch.arange(0, N, 4)
            x[pick, 0] = (8191.0 + 0.25 + 0.5 * torch.rand(pick.numel(), generator=g)) / float(sc[13])
    x, sc = x.to(DEV), sc.to(DEV)
    table = ((torch.rand(16 << log2_T, 2, generator=g) - 0.5) * 0.2).to(DEV)
    d_enc = torch.rand(16, N, 2, generator=g).to(DEV)
    nbytes = ops._hip.lib().umhs_hashgrid_bwd_workspace_bytes(N, 16, log2_T)
    assert nbytes > 0
    ws = ops._workspace(nbytes, x.device, slot=1)

    def run(fused):
        ws.zero_()
        if fused:
            enc = ops.hashgrid_fwd_count(x, table, sc, log2_T)
            assert ops.hashgrid_bwd_prepare_counted(x, sc, log2_T)
        else:
            enc = ops.hashgrid_fwd(x, table, sc, log2_T, True)
            assert ops.hashgrid_bwd_prepare(x, sc, log2_T)
        meta = ws[:nbytes].clone()  # (zeroed above; the record regions behind the counts are written by the scatter pass only)
        d_table = torch.empty(16 << log2_T, 2, device=DEV)
        ops.hashgrid_bwd_apply(x, d_enc, sc, log2_T, d_table, True, overwrite=True)
        return enc, meta, d_table

    a, b = run(False), run(True)
    assert bool(a[1].any())
    assert torch.equal(a[0], b[0]), "features"
    assert torch.equal(a[1], b[1]), "histogram / scans in the workspace"
    assert torch.equal(a[2].view(torch.int32), b[2].view(torch.int32)), "gradient"


def test_hashgrid_bwd_skewed_all_samples_in_one_cell():
    """Worst case for the bucket partition: every contribution of the coarse levels lands in 8 slots."""
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    N, log2_T = 5000, 19
    x = 0.4 + torch.rand(N, 3, generator=g) * 1e-3
    table = torch.zeros(16 << log2_T, 2, requires_grad=True)
    sc = T.hash_scalings()
    cot = torch.rand(N, 32, generator=g)
    (gref,) = torch.autograd.grad((T.hash_encode(x, table, sc, log2_T) * cot).sum(), table)
    d_enc = cot.view(N, 16, 2).permute(1, 0, 2).contiguous()
    d_table = torch.zeros(16 << log2_T, 2, device=DEV)
    ops.hashgrid_bwd(x.to(DEV), d_enc.to(DEV), sc.to(DEV), log2_T, d_table, True, method="partition")
    assert_close("d_table skew", d_table, gref, 2e-5)


CASES = [
    pytest.param(6, 31, True, 0.4, id="C6_B31_spec"),
    pytest.param(9, 128, True, 0.3, id="C9_B128_spec"),
    pytest.param(4, 141, False, 0.7, id="C4_B141_nospec"),
    pytest.param(4, 21, True, 0.5, id="C4_B21_spec"),
    pytest.param(15, 3, True, 0.2, id="C15_B3_spec"),
    pytest.param(5, 21, False, 0.6, id="C5_B21_nospec"),  # B <= 32 without the specular head: split backward, no-spec variants
    pytest.param(7, 32, True, 0.4, id="C7_B32_spec"),     # the largest band count the split backward takes
]


def _oracle_field(p, b, temperature):
    density, emb, sraw, sel = T.field_density(p, b["origins"], b["directions"], b["starts"], b["ends"])
    outs = T.field_outputs(p, b["origins"], b["directions"], b["starts"], b["ends"], emb, temperature)
    return density, emb, sraw, sel, outs


@pytest.mark.parametrize("C,B,spec,temp", CASES)
def test_field_fwd(C, B, spec, temp):
    """R3-R9 with the hash features taken from the ORACLE (isolates the MFMA kernels from the gather kernel)."""
    ops = _ops()
    p, b, layout, flat, fs = make_case(C, B, spec, 7, 29, log2_T=12, temperature=temp)  # N = 203: ragged tail tile
    density, emb, sraw, sel, outs = _oracle_field(p, b, temp)
    pos = T.frustum_positions(b["origins"], b["directions"], b["starts"], b["ends"])
    q = (T.scene_contraction_linf(pos) + 2.0) / 4.0
    enc = T.hash_encode(q * sel[:, None], p.hash_table, p.scalings, p.log2_T).detach()
    N = enc.shape[0]
    for level_major in (True, False):
        e = enc.view(N, 16, 2).permute(1, 0, 2).contiguous() if level_major else enc
        out = ops.field_fwd(fs, flat, e.to(DEV), level_major, pos.to(DEV), b["directions"].to(DEV), sel.float().to(DEV), want_emb=True)
        assert_close("sigma_raw", out["sigma_raw"], sraw.view(-1), 2e-5)
        assert_close("sigma", out["sigma"], density.view(-1), 2e-5)
        assert_close("emb", out["emb"], emb, 2e-5)
        assert_close("abundances", out["abundances"], outs["abundances"].view(N, C), 2e-5)
        assert_close("spectral", out["spectral"], outs["spectral"].view(N, B), 2e-5)
        if spec:
            assert_close("spectral2", out["spectral2"], outs["spectral2"].view(N, B), 2e-5)
            assert_close("specular", out["specular"], outs["specular"].view(N, B), 2e-5)


def test_density_only():
    ops = _ops()
    p, b, layout, flat, fs = make_case(6, 31, True, 5, 40, log2_T=14)
    pos = T.frustum_positions(b["origins"], b["directions"], b["starts"], b["ends"])
    sig, emb = ops.DensityFn.apply(flat, pos.to(DEV), fs)
    q = (T.scene_contraction_linf(pos) + 2.0) / 4.0
    sel = ((q > 0) & (q < 1)).all(-1)
    h = T.mlp_forward(T.hash_encode(q * sel[:, None], p.hash_table, p.scalings, p.log2_T), list(p.base_w), list(p.base_b))
    assert_close("density_fn", sig, torch.exp(h[:, :1]) * sel[:, None], 2e-5)
    assert_close("density_fn emb", emb, h[:, 1:], 2e-5)


def test_density_query_scratch_and_kept_features():
    """density_fn = hash gather + density-only forward: the same bits whether the features go to the per-device scratch or are kept
    for the caller (the sampler reuses them), at the reference table size and on ragged sample counts; sigma-only callers get no embedding."""
    ops = _ops()
    for n_rays, n_samp, log2_T in ((5, 40, 14), (37, 29, 19), (1, 1, 12)):
        p, b, layout, flat, fs = make_case(6, 31, True, n_rays, n_samp, log2_T=log2_T)
        pos = (T.frustum_positions(b["origins"], b["directions"], b["starts"], b["ends"]) * 1.7).to(DEV)
        keep = {}
        kept = ops.DensityFn.apply(flat, pos, fs, keep)
        scratch = ops.DensityFn.apply(flat, pos, fs)
        assert torch.equal(kept[0], scratch[0]) and torch.equal(kept[1], scratch[1])
        assert keep["enc"].shape == (16, pos.shape[0] * pos.shape[1] if pos.dim() == 3 else pos.shape[0], 2)
        sig, emb = ops.DensityFn.apply(flat, pos, fs, None, False)
        assert torch.equal(sig, kept[0]) and emb.numel() == 0


@pytest.mark.parametrize("C,B,spec,temp", CASES)
def test_field_bwd(C, B, spec, temp):
    ops = _ops()
    p, b, layout, flat, fs = make_case(C, B, spec, 9, 23, log2_T=12, temperature=temp, seed=4)  # N = 207
    pos = T.frustum_positions(b["origins"], b["directions"], b["starts"], b["ends"])
    q = (T.scene_contraction_linf(pos) + 2.0) / 4.0
    sel = ((q > 0) & (q < 1)).all(-1)
    enc = T.hash_encode(q * sel[:, None], p.hash_table, p.scalings, p.log2_T).detach().requires_grad_()
    N = enc.shape[0]
    # oracle forward from enc
    h = T.mlp_forward(enc, list(p.base_w), list(p.base_b))
    sraw, emb = torch.split(h, [1, 15], dim=-1)
    density = T.trunc_exp(sraw) * sel[:, None]
    outs = T.field_outputs(p, b["origins"], b["directions"], b["starts"], b["ends"], emb, temp)
    g = torch.Generator().manual_seed(9)
    cot_s = torch.rand(N, B, generator=g) - 0.3
    cot_d = torch.rand(N, 1, generator=g) - 0.3
    cot_e = torch.rand(N, 15, generator=g) - 0.5
    loss = (outs["spectral"].view(N, B) * cot_s).sum() + (density * cot_d).sum() + (emb * cot_e).sum()
    names = [k for k, _ in p.named_parameters() if k != "hash_table"]
    params = [v for k, v in p.named_parameters() if k != "hash_table"]
    grads = torch.autograd.grad(loss, [enc] + params, allow_unused=True)
    d_enc_ref, pgrads = grads[0], dict(zip(names, grads[1:]))
    d_flat = torch.zeros_like(flat)
    e = enc.detach().view(N, 16, 2).permute(1, 0, 2).contiguous()
    args = (fs, flat, e.to(DEV), True, pos.to(DEV), b["directions"].to(DEV), sel.float().to(DEV))
    # (the backward starts from the feature logits the forward saves: part 0 = mlp_head + mlp_directional + mixing, part 1 = feature_mlp + mlp_base)
    logits = ops.field_fwd(*args, want_emb=True, want_logits=True)["feat_logits"]
    assert logits is not None and logits.shape == (N, 16)
    d_enc = ops.field_bwd(*args, sraw.detach().view(-1).contiguous().to(DEV), emb.detach().contiguous().to(DEV),
                          cot_d.view(-1).to(DEV), cot_s.to(DEV), cot_e.to(DEV), d_flat, feat_logits=logits)
    assert_close("d_enc", d_enc.permute(1, 0, 2).reshape(N, 32), d_enc_ref, 5e-5)
    key = {"base_w": "mlp_base.mlp", "head_w": "mlp_head", "feat_w": "feature_mlp", "dir_w": "mlp_directional",
           "base_b": "mlp_base.mlp", "head_b": "mlp_head", "feat_b": "feature_mlp", "dir_b": "mlp_directional"}
    for k, gref in pgrads.items():
        if k == "endmembers":
            got = layout.view(d_flat, "endmembers")
        else:
            pre, idx = k.rsplit(".", 1)
            got = layout.view(d_flat, f"{key[pre]}.layers.{idx}.{'weight' if pre.endswith('_w') else 'bias'}")
        if gref is None:
            assert float(got.abs().max()) == 0.0, k
        else:
            assert_close(f"grad {k}", got, gref, 5e-5)
    # a missing feature-logits array is an argument error, not a silent other path
    with pytest.raises(RuntimeError, match="umhs_field_bwd"):
        ops.field_bwd(*args, sraw.detach().view(-1).contiguous().to(DEV), emb.detach().contiguous().to(DEV), cot_d.view(-1).to(DEV),
                      cot_s.to(DEV), cot_e.to(DEV), torch.zeros_like(flat), feat_logits=None)


# --------------------------------------------------------------------------------------------- #
@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("S,K", [(64, 31), (100, 5), (7, 141)])
def test_composite_fwd_bwd(ragged, S, K):
    ops = _ops()
    R = 37
    b = T.synthetic_batch(R, S, K, seed=13, ragged=ragged)
    N = b["origins"].shape[0]
    g = torch.Generator().manual_seed(2)
    sigma = torch.exp(torch.randn(N, generator=g) * 1.5 + 2.0).requires_grad_()
    v1 = torch.rand(N, K, generator=g).requires_grad_()
    v2 = torch.rand(N, 3, generator=g)
    t0, t1, ri = b["starts"], b["ends"], b["ray_indices"]
    pinfo = T.pack_info(ri, R)
    got_pinfo = ops.pack_info(ri.to(DEV), R)
    assert torch.equal(got_pinfo.cpu(), pinfo)  # index work: bit-exact (incl. empty rays)
    fo = T.scale_gradients_by_distance_squared({"density": sigma[:, None], "v1": v1}, t0, t1)
    w = T.render_weight_from_density(t0[:, 0], t1[:, 0], fo["density"][:, 0], pinfo)[0]
    o1 = T.accumulate_along_rays(w, fo["v1"], ri, R)
    o2 = T.accumulate_along_rays(w, v2, ri, R)
    acc = T.accumulate_along_rays(w, None, ri, R)
    steps = (t0 + t1) / 2
    depth = T.accumulate_along_rays(w, steps, ri, R) / (acc + 1e-10)
    cot1, cota = torch.rand(R, K, generator=g) - 0.5, torch.rand(R, 1, generator=g) - 0.5
    gs, gv = torch.autograd.grad((o1 * cot1).sum() + (acc * cota).sum(), [sigma, v1])

    sd, vd = sigma.detach().to(DEV).requires_grad_(), v1.detach().to(DEV).requires_grad_()
    outs = ops.CompositeFn.apply(sd.view(-1, 1), t0.to(DEV), t1.to(DEV), got_pinfo, True, vd, v2.to(DEV))
    wg, accg, depthg, o1g, o2g = outs
    assert_close("weights", wg.view(-1), w, 1e-5)
    assert_close("acc", accg, acc, 1e-5)
    assert_close("depth", depthg, depth, 1e-5)
    assert_close("out1", o1g, o1, 1e-5)
    assert_close("out2", o2g, o2, 1e-5)
    ((o1g * cot1.to(DEV)).sum() + (accg * cota.to(DEV)).sum()).backward()
    assert_close("d_sigma", sd.grad, gs, 1e-4)
    assert_close("d_values", vd.grad, gv, 1e-5)


@pytest.mark.parametrize("bands", ["b21", "b31", "b128", "b141"])
def test_spec2rgb_golden_and_grad(golden_dir, bands):
    ops = _ops()
    g = np.load(os.path.join(golden_dir, "g1_colour.npz"))
    M = torch.from_numpy(g[f"{bands}_M"])
    spec = torch.from_numpy(g[f"{bands}_spec"])
    rgb = ops.spec2rgb_fwd(spec.to(DEV), M.to(DEV))
    np.testing.assert_allclose(rgb.cpu().numpy(), g[f"{bands}_rgb"], rtol=0, atol=2e-6)  # reference's own output
    s = spec.clone().requires_grad_()
    cot = torch.rand(spec.shape[0], 3, generator=torch.Generator().manual_seed(1))
    (gref,) = torch.autograd.grad((T.colour_system(s, M) * cot).sum(), s)
    got = ops.spec2rgb_bwd(spec.to(DEV), M.to(DEV), cot.to(DEV))
    assert_close("d_spec", got, gref, 2e-5)


def test_adam_matches_torch_optim():
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    n = 10007
    p0 = torch.rand(n, generator=g)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=2e-2, eps=1e-15)
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pad = (-n) % 4
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * 0.1
        grad[::3] = 0.0  # untouched hash slots still decay m/v
        ref.grad = grad.clone()
        opt.step()
        with torch.no_grad():
            ref[100:200].clamp_(0, 1)  # clamp_endmembers epilogue
        ops.adam_step(p, grad.to(DEV), m, v, step, 2e-2, clamp_range=(100, 200))
        assert_close(f"adam step {step}", p, ref, 2e-6)
    assert pad >= 0


@pytest.mark.parametrize("case,C,B,spec", [("c6b31s", 6, 31, True), ("c9b128s", 9, 128, True), ("c4b141n", 4, 141, False)])
def test_field_against_reference_golden(golden_dir, case, C, B, spec):
    """G4: outputs/grads produced by the reference's own UMHSField.get_density/get_outputs code."""
    ops = _ops()
    g = np.load(os.path.join(golden_dir, f"g4_field_{case}.npz"))
    layout = ops.FieldLayout(C, B, spec, 12)
    flat = torch.zeros(layout.total)
    p = T.FieldParams(C, B, spec, log2_hashmap_size=12, seed=0)
    with torch.no_grad():
        for k, v in p.named_parameters():
            v.copy_(torch.from_numpy(g[f"param_{k}"]))
    for k, v in p.reference_state_dict().items():
        layout.view(flat, k).copy_(v)
    flat = flat.to(DEV).requires_grad_()
    fs = ops.FieldSpec(layout, float(g["temperature"]), True, scalings=ops.hash_scalings().to(DEV))
    o, d, s, e = (torch.from_numpy(g[k]).to(DEV) for k in ("origins", "directions", "starts", "ends"))
    density, emb, spectral, spectral2, specular, abund = ops.FieldFn.apply(flat, o, d, s, e, fs)
    N = o.shape[0]
    gt = lambda k, *shape: torch.from_numpy(g[k]).reshape(*shape)
    assert_close("density", density, gt("density", N, 1), 2e-5)
    assert_close("emb", emb, gt("emb", N, 15), 2e-5)
    assert_close("spectral", spectral, gt("out_spectral", N, B), 2e-5)
    assert_close("abundances", abund, gt("out_abundances", N, C), 2e-5)
    if spec:
        assert_close("spectral2", spectral2, gt("out_spectral2", N, B), 2e-5)
        assert_close("specular", specular, gt("out_specular", N, B), 2e-5)
    loss = (spectral * torch.from_numpy(g["cot_spec"]).to(DEV).view(N, B)).sum() + (density * torch.from_numpy(g["cot_den"]).to(DEV)).sum()
    loss.backward()
    gflat = flat.grad
    tab = layout.view(gflat, "mlp_base.encoder.hash_table").cpu()
    rows = torch.from_numpy(g["grad_hash_rows"])
    assert_close("d_table rows", tab[rows], torch.from_numpy(g["grad_hash_vals"]), 1e-4)
    mask = torch.ones(tab.shape[0], dtype=torch.bool)
    mask[rows] = False
    assert float(tab[mask].abs().max()) == 0.0
    assert_close("d_endmembers", layout.view(gflat, "endmembers"), torch.from_numpy(g["grad_endmembers"]), 1e-4)
    assert_close("d_head_w1", layout.view(gflat, "mlp_head.layers.1.weight"), torch.from_numpy(g["grad_head_w.1"]), 1e-4)
    assert_close("d_base_w0", layout.view(gflat, "mlp_base.mlp.layers.0.weight"), torch.from_numpy(g["grad_base_w.0"]), 1e-4)


@pytest.mark.parametrize("C,B,spec,temp,ragged", [(6, 31, True, 0.4, False), (4, 141, False, 0.7, True)])
def test_end_to_end_loss_and_grads(C, B, spec, temp, ragged):
    """FieldFn -> CompositeFn -> Spec2RgbFn -> loss, against oracle model_outputs/model_loss (radiance 1e-4, PSNR 0.05 dB)."""
    ops = _ops()
    R = 48
    p, b, layout, flat, fs = make_case(C, B, spec, R, 24, ragged=ragged, log2_T=15, temperature=temp, seed=21)
    M = T.colour_matrix(np.linspace(400, 700, B))
    gt_rgb = T.colour_system(b["gt_spectral"], M)
    out = T.model_outputs(p, b["origins"], b["directions"], b["starts"], b["ends"], b["ray_indices"], R, temp, M)
    loss = T.model_loss(out, b["gt_spectral"], gt_rgb, b["bg_random"], "rgb+spectral")
    total = loss["spectral_loss"] + loss["rgb_loss"]
    names = [k for k, _ in p.named_parameters()]
    grads = dict(zip(names, torch.autograd.grad(total, [v for _, v in p.named_parameters()], allow_unused=True)))

    d = dev(b)
    flat = flat.requires_grad_()
    density, emb, spectral, spectral2, specular, abund = ops.FieldFn.apply(flat, d["origins"], d["directions"], d["starts"], d["ends"], fs)
    pinfo = ops.pack_info(d["ray_indices"], R)
    vals = [spectral] + ([spectral2, specular] if spec else []) + [abund]
    w, acc, depth, *outs = ops.CompositeFn.apply(density, d["starts"], d["ends"], pinfo, True, *vals)
    rgb = ops.Spec2RgbFn.apply(outs[0], M.to(DEV))
    pred_rgb = rgb + d["bg_random"] * (1.0 - acc)
    l_spec = 5 * torch.nn.functional.mse_loss(outs[0], d["gt_spectral"])
    l_rgb = torch.nn.functional.mse_loss(pred_rgb, gt_rgb.to(DEV))
    (l_spec + l_rgb).backward()

    for nme, got, want in [("spectral", outs[0], out["spectral"]), ("rgb", rgb, out["rgb"]), ("accumulation", acc, out["accumulation"]),
                           ("abundances", outs[-1], out["abundances"])] + ([("spectral2", outs[1], out["spectral2"]),
                                                                            ("specular", outs[2], out["specular"])] if spec else []):
        assert_close(nme, got, want, 1e-4)
        assert_elementwise(nme, got, want, rtol=1e-4, atol=1e-6)  # every element within 1e-4 of its OWN value
    psnr_ref = float(T.psnr(out["spectral"], b["gt_spectral"]))
    psnr_got = float(T.psnr(outs[0].detach().cpu(), b["gt_spectral"]))
    assert abs(psnr_ref - psnr_got) < 0.05
    assert abs(float(l_spec) - float(loss["spectral_loss"])) <= 1e-4 * abs(float(loss["spectral_loss"]))
    assert abs(float(l_rgb) - float(loss["rgb_loss"])) <= 1e-4 * abs(float(loss["rgb_loss"]))
    g = flat.grad
    assert_close("grad hash_table", layout.view(g, "mlp_base.encoder.hash_table"), grads["hash_table"], 2e-4)
    assert_close("grad endmembers", layout.view(g, "endmembers"), grads["endmembers"], 2e-4)
    assert_close("grad base_w0", layout.view(g, "mlp_base.mlp.layers.0.weight"), grads["base_w.0"], 2e-4)
    assert_close("grad feat_w2", layout.view(g, "feature_mlp.layers.2.weight"), grads["feat_w.2"], 2e-4)
    assert_close("grad head_b0", layout.view(g, "mlp_head.layers.0.bias"), grads["head_b.0"], 2e-4)
    if spec:
        assert_close("grad dir_w1", layout.view(g, "mlp_directional.layers.1.weight"), grads["dir_w.1"], 2e-4)


def test_ray_epilogue_and_fused_loss():
    """R13-R16 fused kernels against the oracle's separate functions (colour system, cluster lookup, depth clip, losses)."""
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    R, B, Cn, N = 300, 31, 6, 5000
    M = T.colour_matrix(np.linspace(400, 700, B))
    spec = torch.rand(R, B, generator=g)
    spec[:5] *= 0.003
    E = torch.rand(Cn, B, generator=g)
    acc = torch.rand(R, 1, generator=g)
    depth = torch.rand(R, 1, generator=g) * 3
    t0 = torch.rand(N, 1, generator=g) * 2 + 0.5
    t1 = t0 + 0.01
    colors = torch.rand(15, 3, generator=g)
    mm = ops.tmid_minmax(t0.to(DEV), t1.to(DEV))
    sd = spec.to(DEV).requires_grad_()
    rgb, dclip, probs, raw, pred = ops.RayEpilogueFn.apply(sd, M.to(DEV), E.to(DEV), acc.to(DEV), depth.to(DEV), mm, colors.to(DEV), 0.2)
    steps = (t0 + t1) / 2
    assert_close("rgb", rgb, T.colour_system(spec, M), 1e-5)
    assert_close("depth clip", dclip, torch.clip(depth, steps.min(), steps.max()), 1e-6)
    ip, pr = T.cluster_lookup(spec, 0.2, E)
    assert_close("seg_probs", probs, pr, 1e-5)
    on = (acc > 0.5).float()
    assert torch.equal(raw.cpu(), pr.argmax(1) * on.squeeze(-1))
    assert_close("seg_pred", pred, colors[pr.argmax(1)] * on, 1e-6)
    # fused losses + gradients (incl. non-unit upstream gradients)
    gt_s, gt_rgb, bg = torch.rand(R, B, generator=g), torch.rand(R, 3, generator=g), torch.rand(R, 3, generator=g)
    sr = spec.clone().requires_grad_()
    ar = acc.clone().requires_grad_()
    rr = T.colour_system(sr, M)
    l_s = 5 * torch.nn.functional.mse_loss(sr, gt_s)
    l_r = 0.7 * torch.nn.functional.mse_loss(rr + bg * (1 - ar), gt_rgb)
    gs_ref, ga_ref = torch.autograd.grad(2.0 * l_s + 3.0 * l_r, [sr, ar])
    ad = acc.to(DEV).requires_grad_()
    ls, lr = ops.LossFn.apply(sd, gt_s.to(DEV), rgb, ad, bg.to(DEV), gt_rgb.to(DEV), 5.0, 0.7)
    assert abs(float(ls) - float(l_s)) < 1e-5 * float(l_s) and abs(float(lr) - float(l_r)) < 1e-5 * float(l_r)
    (2.0 * ls + 3.0 * lr).backward()
    assert_close("d_spectral (loss + rgb chain)", sd.grad, gs_ref, 2e-5)
    assert_close("d_acc", ad.grad, ga_ref, 2e-5)
    l1, _ = ops.LossFn.apply(sd.detach(), gt_s.to(DEV), None, None, None, None, 1.0, 0.0)
    assert abs(float(l1) - float(torch.nn.functional.mse_loss(spec, gt_s))) < 1e-6


@pytest.mark.parametrize("fused", ["0", "1"], ids=["per-sample arrays", "per-ray sums in the field kernels"])
def test_model_train_iteration_matches_oracle_step(fused, monkeypatch):
    """Plugin-surface path (UMHSPipeline.train_iteration): outputs, losses and the parameters after one fused-Adam step -- for both
    forms of the step (composite kernels over [N,B] arrays / two-launch forward + folded compositing backward)."""
    import numpy as np

    monkeypatch.setenv("UMHS_FUSED_BWD", fused)
    from umhsnerf._ns_compat import packed_ray_samples
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    ops = _ops()
    R, S, B, Cn, temp = 40, 20, 31, 6, 0.4
    bands = list(np.linspace(400, 700, B))
    cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=temp, log2_hashmap_size=14, background_color="black")
    pipe = UMHSPipeline.from_packed_samples(cfg, torch.device(DEV), metadata={"wavelengths": bands, "num_classes": Cn}, seed=3)
    p = T.FieldParams(Cn, B, True, log2_hashmap_size=14, table_scale=0.5, seed=9)
    with torch.no_grad():
        p.base_b[1][0] += 1.0
    pipe.model.field.load_state_dict(p.reference_state_dict())
    b = T.synthetic_batch(R, S, B, seed=12)
    M = T.colour_matrix(bands)
    gt_rgb = T.colour_system(b["gt_spectral"], M)
    out = T.model_outputs(p, b["origins"], b["directions"], b["starts"], b["ends"], b["ray_indices"], R, temp, M)
    loss = T.model_loss(out, b["gt_spectral"], gt_rgb, torch.zeros(R, 3), "rgb+spectral")
    params = [v for _, v in p.named_parameters()]
    grads = torch.autograd.grad(sum(loss.values()), params, allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(v) for g, v in zip(grads, params)]
    d = dev(b)
    rs = packed_ray_samples(d["origins"], d["directions"], d["starts"], d["ends"])
    outputs, loss_dict = pipe.train_iteration(rs, d["ray_indices"], R, {"image": gt_rgb.to(DEV), "hs_image": d["gt_spectral"]})
    for k in ("spectral", "spectral2", "specular", "abundances", "rgb", "accumulation", "depth", "seg_probs"):
        assert_close(f"outputs[{k}]", outputs[k], out[k], 1e-4)
        assert_elementwise(f"outputs[{k}]", outputs[k], out[k], rtol=1e-4, atol=1e-6)
    assert set(f"wv_{i}" for i in range(B)) <= set(outputs) and "residual_0" in outputs and "abundances_5" in outputs
    for k in loss:
        assert abs(float(loss_dict[k]) - float(loss[k])) <= 1e-4 * abs(float(loss[k])), k
    with torch.no_grad():
        ms, vs = [torch.zeros_like(v) for v in params], [torch.zeros_like(v) for v in params]
        T.adam_step(params, grads, ms, vs, 1, T.exp_decay_lr(0))
        p.endmembers.clamp_(0, 1)
    sd_new = pipe.model.field.state_dict()
    for k, v in p.reference_state_dict().items():
        assert_close(f"param after step: {k}", sd_new[k], v, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("R,B,C,both", [(4096, 31, 6, True), (1001, 141, 4, True), (37, 3, 9, False), (5000, 128, 16, True)])
def test_fused_training_tail_equals_the_separate_kernels(R, B, C, both):
    """umhs_ray_train_tail == ray_epilogue_fwd + loss_fwd + loss_bwd(unit upstream) + spec2rgb_bwd(accumulate)."""
    from umhsnerf import ops

    g = torch.Generator().manual_seed(R + B)
    dev = "cuda:0"
    rnd = lambda *s: torch.rand(*s, generator=g).to(dev)
    spec, gt = rnd(R, B) * 1.2 - 0.1, rnd(R, B)
    M = (torch.rand(B, 3, generator=g) / B * 2.2).to(dev)  # some rgb values beyond 1 (clamp gradient) and near the gamma knee
    spec[:5] *= 1e-3
    E, acc, depth = rnd(C, B), rnd(R), rnd(R) * 5
    colors, gt_rgb, bg = rnd(C, 3), rnd(R, 3), rnd(R, 3)
    t0 = rnd(777)
    mm = ops.tmid_minmax(t0 * 4, t0 * 4 + 0.1)
    w = (5.0, 0.7) if both else (1.0, 0.0)
    rgb0, dclip0, probs0, raw0, pred0 = ops.ray_epilogue_fwd(spec, M, E, acc, depth, mm, colors, 0.2)
    largs = (spec, gt, rgb0, acc, bg, gt_rgb) if both else (spec, gt, None, None, None, None)
    losses0 = ops.loss_fwd(*largs, *w)
    d_spec0, d_rgb0, d_acc0 = ops.loss_bwd(*largs, *w, torch.ones(2, device=dev))
    if both:
        ops.spec2rgb_bwd(spec, M, d_rgb0, accumulate_into=d_spec0)
    for rep in range(3):  # the arrival counter resets itself: repeated calls give the same sums
        rgb, dclip, probs, raw, pred, losses, d_spec, d_acc = ops.ray_train_tail(spec, M, E, acc, depth, mm, colors, gt, gt_rgb if both else None,
                                                                                 bg if both else None, 0.2, w[0], w[1], both)
        torch.testing.assert_close(rgb, rgb0, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(dclip, dclip0, rtol=0, atol=0)
        torch.testing.assert_close(probs, probs0, rtol=1e-5, atol=1e-6)
        assert float((raw != raw0).float().mean()) < 2e-3  # argmax ties under a different summation order
        torch.testing.assert_close(losses[: 2 if both else 1], losses0[: 2 if both else 1], rtol=2e-5, atol=0)
        torch.testing.assert_close(d_spec, d_spec0, rtol=2e-5, atol=1e-9)
        if both:
            torch.testing.assert_close(d_acc, d_acc0, rtol=1e-5, atol=1e-10)
        else:
            assert d_acc is None
        if rep:
            assert torch.equal(losses, first)
        first = losses.clone()


@pytest.mark.gpu
def test_hashgrid_bwd_prepare_apply_equals_one_call():
    """Histogram/scan from the positions alone (prepare) + per-group scatter/reduce (apply) == umhs_hashgrid_bwd, including rows
    whose gradient is exactly zero (the one-call form skips them, the two-call form cannot know them yet)."""
    from umhsnerf import ops

    dev = "cuda:0"
    g = torch.Generator().manual_seed(12)
    n = 70001
    pos = torch.rand(n, 3, generator=g).to(dev)
    d_enc = torch.randn(16, n, 2, generator=g).to(dev)
    d_enc[:, ::3] = 0.0
    d_enc[5] = 0.0  # a whole level without gradient
    sc = ops.hash_scalings(ops.NUM_LEVELS, 16, 2048).to(dev)
    want = torch.empty(ops.NUM_LEVELS << 19, 2, device=dev)
    ops.hashgrid_bwd(pos, d_enc, sc, 19, want, True, overwrite=True)
    got = torch.full_like(want, float("nan"))
    assert ops.hashgrid_bwd_prepare(pos, sc, 19)
    for l0, cnt in [(12, 4), (0, 5), (5, 7)]:  # any order, any grouping, each level once
        ops.hashgrid_bwd_apply(pos, d_enc, sc, 19, got, True, overwrite=True, level_begin=l0, level_count=cnt)
    # not bit-equal: a zero-gradient sample now sits inside a run of its cell instead of splitting it, so the float partial
    # sums feeding the fixed-point accumulation group differently
    assert not torch.isnan(got).any()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6 * float(want.abs().max()))
    assert torch.equal(got[5 << 19:6 << 19], torch.zeros_like(got[5 << 19:6 << 19]))  # the gradient-free level stays exactly zero
    acc = torch.ones_like(want)  # accumulate mode
    assert ops.hashgrid_bwd_prepare(pos, sc, 19, level_begin=3, level_count=2)
    ops.hashgrid_bwd_apply(pos, d_enc, sc, 19, acc, True, overwrite=False, level_begin=3, level_count=2, ws_range=(3, 2))
    torch.testing.assert_close(acc[3 << 19:5 << 19], want[3 << 19:5 << 19] + 1.0, rtol=1e-5, atol=1e-6 * float(want.abs().max()))
    assert float(acc[: 3 << 19].min()) == 1.0 and float(acc[5 << 19:].max()) == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("C,B,spec", [(6, 31, True), (4, 141, False), (5, 21, False)])
def test_field_bwd_overwrites_every_parameter_gradient(C, B, spec):
    """The persistent gradient buffer is not cleared between steps: every weight / bias / endmember entry must be (over)written
    by each backward, whatever was there before (NaN here)."""
    ops = _ops()
    p, b, layout, flat, fs = make_case(C, B, spec, 9, 23, log2_T=12, seed=4)
    N = b["origins"].shape[0]
    dev_args = [t.to(DEV) for t in (b["origins"], b["directions"], b["starts"].view(-1), b["ends"].view(-1))]
    wpos, pos01, sel = ops.positions_fwd(*dev_args, fs)
    flat = flat.to(DEV)
    enc = ops.hashgrid_fwd(pos01, layout.view(flat, "mlp_base.encoder.hash_table"), fs.scalings, 12, True)
    out = ops.field_fwd(fs, flat, enc, True, wpos, dev_args[1], sel, want_emb=True, want_logits=True)
    g = torch.Generator().manual_seed(1)
    ds, dsp = torch.rand(N, generator=g).to(DEV), torch.rand(N, B, generator=g).to(DEV)
    for logits in (out["feat_logits"],):
        ref = torch.zeros_like(flat)
        ops.field_bwd(fs, flat, enc, True, wpos, dev_args[1], sel, out["sigma_raw"], out["emb"], ds, dsp, None, ref, feat_logits=logits)
        dirty = torch.full_like(flat, float("nan"))
        ops.field_bwd(fs, flat, enc, True, wpos, dev_args[1], sel, out["sigma_raw"], out["emb"], ds, dsp, None, dirty, feat_logits=logits)
        untouched = []
        for name in layout.entries:
            if name == "mlp_base.encoder.hash_table":
                continue
            got, want = layout.view(dirty, name), layout.view(ref, name)
            if torch.isnan(got).any():
                assert torch.isnan(got).all() and float(want.abs().max()) == 0.0, name  # all or nothing, and zero in a clean buffer
                untouched.append(name)
                continue
            assert torch.equal(got, want), name
        # without the specular head nothing reaches mlp_directional: the entries the reduce does not visit stay at their initial zero
        assert all(n.startswith("mlp_directional") for n in untouched) and (not spec or not untouched), untouched


@pytest.mark.parametrize("C,B,spec", [(6, 31, True), (4, 40, False), (9, 128, True), (4, 141, False), (15, 256, True), (1, 3, False)])
def test_two_launch_forward_with_band_sums_in_the_kernel(C, B, spec):
    """umhs_field_base_fwd -> weights -> umhs_field_heads_fwd against umhs_field_fwd + umhs_composite_fwd: per-sample outputs bit for
    bit (same kernels' arithmetic), per-ray sums to rounding (another, fixed, summation order; the mixing term summed per ray as w m
    and multiplied by the endmembers once per ray), run-to-run identical.  Rays with no samples, rays shorter than a 16-sample tile
    (several inside one tile), rays across many tiles, N not a multiple of 16."""
    ops = _ops()
    _, _, layout, flat, fs = make_case(C, B, spec, 8, 8, log2_T=12)
    g = torch.Generator().manual_seed(B)
    counts = torch.tensor([0, 3, 2, 0, 1, 40, 17, 16, 5, 4, 3, 200, 0, 7, 33, 1, 1, 1, 90, 0, 1300, 0, 0, 2], dtype=torch.int64)  # (1300 > 1024)
    R, n = counts.numel(), int(counts.sum())
    assert n % 16 != 0
    starts = torch.cumsum(counts, 0) - counts
    packed_info = torch.stack([starts, counts], 1).contiguous().to(DEV)
    ray_idx = torch.repeat_interleave(torch.arange(R), counts).to(DEV)
    enc = ((torch.rand(16, n, 2, generator=g) - 0.5)).to(DEV)
    wpos = (torch.rand(n, 3, generator=g) * 2 - 1).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(DEV)
    sel = (torch.rand(n, generator=g) > 0.1).float().to(DEV)
    t0 = torch.rand(n, generator=g).to(DEV)
    t1 = t0 + 0.05
    ref = ops.field_fwd(fs, flat, enc, True, wpos, dirs, sel, want_emb=True, want_logits=True)
    vals = [ref["spectral"]] + ([ref["spectral2"], ref["specular"]] if spec else []) + [ref["abundances"]]
    w_ref, acc_ref, depth_ref, comp_ref = ops.composite_fwd(ref["sigma"], t0, t1, packed_info, vals)
    base = ops.field_base_fwd(fs, flat, enc, True, sel)
    for k in ("sigma", "sigma_raw", "emb"):
        assert torch.equal(base[k], ref[k]), k
    w, acc, depth, none = ops.composite_fwd(base["sigma"], t0, t1, packed_info, [])
    assert none == [] and torch.equal(w, w_ref) and torch.equal(acc, acc_ref) and torch.equal(depth, depth_ref)
    runs = []
    for rep in range(2):
        ho = ops.field_heads_fwd(fs, flat, base["emb"], wpos, dirs, w, ray_idx, packed_info, pack_ready=True, release=False, want_abundances=True)
        for k in ("abundances", "feat_logits"):
            assert torch.equal(ho[k], ref[k]), k
        got = ho["comp"] + [ho["comp_abundances"]]
        assert len(got) == len(comp_ref)
        for i, (a, b) in enumerate(zip(got, comp_ref)):
            assert_close(f"comp[{i}]", a, b, 3e-6)
        runs.append([c.clone() for c in got])
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    # the aligned-row form of the base outputs ([N,16], sigma_raw in slot 0) feeds the same kernel: same bits everywhere
    base16 = ops.field_base_fwd(fs, flat, enc, True, sel, pack_ready=True, rows16=True)
    assert base16["emb"] is None and torch.equal(base16["base16"][:, 1:], ref["emb"]) and torch.equal(base16["base16"][:, 0], ref["sigma_raw"])
    h2 = ops.field_heads_fwd(fs, flat, base16["base16"], wpos, dirs, w, ray_idx, packed_info, want_logits=False, pack_ready=True, release=False)
    assert h2["feat_logits"] is None and h2["abundances"] is None
    for a, b in zip(h2["comp"] + [h2["comp_abundances"]], runs[0]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("C,B,spec,gs", [(6, 31, True, True), (4, 40, False, False), (9, 128, True, True), (4, 141, False, True),
                                         (15, 160, True, False), (1, 3, False, True)])
def test_field_backward_with_the_compositing_backward_folded_in(C, B, spec, gs):
    """umhs_field_bwd_composited (d_comp [R,B] in, d_sigma out, no [N,B] array) against umhs_composite_bwd + umhs_field_bwd on the
    same inputs: d_sigma, d_enc and every parameter gradient.  Ragged rays incl. empty and sub-tile ones; with and without
    scale_gradients_by_distance_squared."""
    ops = _ops()
    _, _, layout, flat, fs = make_case(C, B, spec, 8, 8, log2_T=12)
    assert ops.field_bwd_composited_supported(fs)
    # (256 bands with the specular head: the transpose-free kernels' packs no longer fit in LDS -- the query says so and the model keeps
    # the per-sample path, umhs_model.forward_backward_from_samples)
    assert not ops.field_bwd_composited_supported(make_case(15, 256, True, 8, 8, log2_T=12)[4])
    g = torch.Generator().manual_seed(B + 1)
    counts = torch.tensor([0, 3, 2, 0, 1, 40, 17, 16, 5, 4, 3, 200, 0, 7, 33, 1, 1, 1, 90, 0], dtype=torch.int64)
    R, n = counts.numel(), int(counts.sum())
    starts = torch.cumsum(counts, 0) - counts
    packed_info = torch.stack([starts, counts], 1).contiguous().to(DEV)
    ray_idx = torch.repeat_interleave(torch.arange(R), counts).to(DEV)
    enc = ((torch.rand(16, n, 2, generator=g) - 0.5)).to(DEV)
    wpos = (torch.rand(n, 3, generator=g) * 2 - 1).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(DEV)
    sel = (torch.rand(n, generator=g) > 0.1).float().to(DEV)
    t0 = (torch.rand(n, generator=g) * 1.5).to(DEV)
    t1 = t0 + 0.05
    fo = ops.field_fwd(fs, flat, enc, True, wpos, dirs, sel, want_emb=True, want_logits=True)
    w, acc, depth, comp = ops.composite_fwd(fo["sigma"], t0, t1, packed_info, [fo["spectral"]])
    d_comp = torch.randn(R, B, generator=g).to(DEV)
    d_acc = torch.randn(R, generator=g).to(DEV)
    d_sigma_ref, d_values = ops.composite_bwd(fo["sigma"], t0, t1, packed_info, w, [fo["spectral"]], [d_comp], [True], d_acc, gs)
    g_ref = torch.zeros_like(flat)
    d_enc_ref = ops.field_bwd(fs, flat, enc, True, wpos, dirs, sel, fo["sigma_raw"], fo["emb"], d_sigma_ref, d_values[0], None, g_ref,
                              feat_logits=fo["feat_logits"])
    cp = dict(sigma=fo["sigma"], t0=t0, t1=t1, packed_info=packed_info, ray_indices=ray_idx, weights=w, d_comp=d_comp, d_acc=d_acc,
              grad_scaling=gs)
    g_got = torch.zeros_like(flat)
    d_enc = ops.field_bwd(fs, flat, enc, True, wpos, dirs, sel, fo["sigma_raw"], fo["emb"], None, None, None, g_got,
                          feat_logits=fo["feat_logits"], comp=cp)
    assert_close("d_sigma", cp["d_sigma"], d_sigma_ref, 2e-5)
    assert_close("d_enc", d_enc, d_enc_ref, 2e-5)
    tail = layout.tail_offset()
    assert_close("parameter gradients", g_got[tail:], g_ref[tail:], 2e-5)
    # the saved base outputs as aligned [N,16] rows: same bits
    b16 = torch.cat([fo["sigma_raw"][:, None], fo["emb"]], 1).contiguous()
    g2, cp2 = torch.zeros_like(flat), dict(cp)
    d_enc2 = ops.field_bwd(fs, flat, enc, True, wpos, dirs, sel, fo["sigma_raw"], b16, None, None, None, g2, feat_logits=fo["feat_logits"], comp=cp2)
    assert torch.equal(d_enc2, d_enc) and torch.equal(g2, g_got) and torch.equal(cp2["d_sigma"], cp["d_sigma"])
