"""CPU-only checks of the drop-in boundary: the shared library loads, exports every symbol that
include/umhs_hip.h declares, and the Python binding declares a signature for each (no compute calls)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "umhs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(umhs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_library):
    from umhsnerf import _hip

    assert os.path.exists(_hip.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_hip.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in umhs_hip.h but not exported"
        assert s in _hip.SIGNATURES, f"{s} has no ctypes signature in umhsnerf/_hip.py"
    assert set(_hip.SIGNATURES) == set(syms)
    assert _hip.lib().umhs_abi_version() == _hip.ABI_VERSION == 11
    assert _hip.lib().umhs_strerror(-3) == b"workspace missing or too small"


def test_struct_sizes_match_header():
    from umhsnerf import _hip

    assert ctypes.sizeof(_hip.FieldCfg) == 20
    assert ctypes.sizeof(_hip.FieldParams) == 21 * 8 == ctypes.sizeof(_hip.FieldGrads)
    assert ctypes.sizeof(_hip.ValueStreams) == 4 + 16 + 4 + 32 + 32  # n, k[4], pad, values[4], out[4]
    assert ctypes.sizeof(_hip.ValueGrads) == 4 + 16 + 4 + 32 * 3


def test_no_cpu_fallback():
    from umhsnerf import ops

    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.spec2rgb_fwd(torch.rand(4, 21), torch.rand(21, 3))


def test_flat_layout_names_and_alignment():
    from umhsnerf import ops

    L = ops.FieldLayout(6, 31, True, 19)
    assert L.entries["mlp_base.encoder.hash_table"] == (0, (16 << 19, 2))
    assert L.entries["feature_mlp.layers.2.weight"][1] == (7, 64)
    assert L.entries["mlp_head.layers.2.weight"][1] == (6, 64)
    assert L.entries["mlp_directional.layers.1.weight"][1] == (31, 16)
    assert L.entries["endmembers"][1] == (6, 31)
    assert all(off % 4 == 0 for off, _ in L.entries.values()) and L.total % 4 == 0
    n_mlp = L.total - (16 << 19) * 2
    assert 16000 < n_mlp < 18500  # ~16.5 k MLP weights + endmembers
    flat = torch.arange(L.total, dtype=torch.float32)
    v = L.view(flat, "endmembers")
    assert v.shape == (6, 31) and v.data_ptr() == flat.data_ptr() + 4 * L.offset("endmembers")
    assert ops.hash_scalings()[15] == 2047 and ops.hash_scalings().dtype == torch.float32


def test_argument_errors_are_reported_before_anything_is_launched():
    """Error behaviour of the boundary: NULL / negative / inconsistent arguments come back as negative codes from the host-side
    checks (no HIP call is made, so this runs without a GPU); empty inputs are a success."""
    from umhsnerf import _hip

    lib = _hip.lib()
    ARG, UNSUP = -1, -2
    dummy = ctypes.c_void_p(4096)  # never dereferenced: every call below returns from its argument checks
    assert lib.umhs_ray_prefix(None, 5, None, None, None) == ARG
    assert lib.umhs_ray_prefix(dummy, -1, dummy, dummy, None) == ARG
    assert lib.umhs_sample_midpoints(None, None, None, None, None, 3, None, None) == ARG
    assert lib.umhs_sample_midpoints(None, None, None, None, None, 0, None, None) == 0  # nothing to do
    assert lib.umhs_compact_samples(*([None] * 3), 4, *([None] * 13)) == ARG
    assert lib.umhs_compact_samples(*([None] * 3), 0, *([None] * 13)) == 0
    assert lib.umhs_visibility_count(dummy, dummy, dummy, dummy, 4, 16, 1e-4, 0.01, dummy, None, None) == ARG  # kept[] missing
    assert lib.umhs_march_scratch(None, None, 8, None, None, 1, 16, 0.05, 1e3, 0.01, 0.0, None, None, None, 0.0, 8, None, None, None, None, 0, None) == ARG
    assert lib.umhs_hashgrid_fwd(None, None, None, 16, 16, 19, None, 2, 0, None) == ARG
    assert lib.umhs_hashgrid_fwd(dummy, ctypes.c_void_p(4096 + 8), dummy, 16, 16, 19, dummy, 2, 0, None) == ARG  # table not 16-byte aligned
    assert lib.umhs_hashgrid_bwd_apply_adam(dummy, dummy, 2, 0, dummy, 0, 0, 16, 0, 16, 19, dummy, dummy, 1 << 20, dummy, dummy, dummy,
                                            1e-2, 0.9, 0.999, 1e-15, 1, 5, None) == UNSUP  # no samples: no reduce pass to ride on
    assert lib.umhs_hashgrid_bwd_apply_adam(dummy, dummy, 2, 0, dummy, 8, 0, 16, 0, 16, 19, dummy, dummy, 1 << 20, None, dummy, dummy,
                                            1e-2, 0.9, 0.999, 1e-15, 1, 5, None) == ARG
    assert lib.umhs_adam_step(dummy, dummy, dummy, dummy, 16, 1e-2, 0.9, 0.999, 1e-15, 0, 1.0, 0, 0, None) == ARG  # step < 1
    assert lib.umhs_hashgrid_bwd_workspace_bytes(1000, 16, 22) == 0  # 2^22 table: only the atomic path (more than 128 buckets)
    assert lib.umhs_hashgrid_bwd_workspace_bytes(1000, 16, 20) > 0
    # round 4: the gather that also takes the backward's histogram, and the walk of the marcher on its own
    WS = -3
    assert lib.umhs_hashgrid_fwd_count(None, None, None, 16, 16, 19, None, 2, 0, None, 0, None) == ARG
    assert lib.umhs_hashgrid_fwd_count(dummy, dummy, dummy, 0, 16, 19, None, 2, 0, None, 0, None) == 0  # nothing to do
    assert lib.umhs_hashgrid_fwd_count(dummy, dummy, dummy, 16, 16, 19, dummy, 2, 0, None, 0, None) == WS  # no workspace
    assert lib.umhs_hashgrid_fwd_count(dummy, dummy, dummy, 16, 16, 22, dummy, 2, 0, dummy, 1 << 30, None) == UNSUP  # no partitioned path
    assert lib.umhs_hashgrid_bwd_prepare_counted(dummy, dummy, 16, 16, 19, dummy, 64, None) == WS  # workspace too small
    roi = (ctypes.c_float * 6)(-1, -1, -1, 1, 1, 1)
    assert lib.umhs_march_walk_workspace_bytes(0) == 0 and lib.umhs_march_walk_workspace_bytes(4096) >= 4096 * 512 * 9
    assert lib.umhs_march_walk(None, None, 8, None, roi, 4, 128, 0.05, 1e3, None, None, None, 0.0, None, 0, None) == ARG
    assert lib.umhs_march_walk(dummy, dummy, 0, dummy, roi, 4, 128, 0.05, 1e3, None, None, None, 0.0, None, 0, None) == 0
    assert lib.umhs_march_walk(dummy, dummy, 8, dummy, roi, 4, 128, 0.05, 1e3, None, None, None, 0.0, dummy, 16, None) == WS
    assert lib.umhs_march_walk(dummy, dummy, 8, dummy, roi, 9, 128, 0.05, 1e3, None, None, None, 0.0, dummy, 1 << 20, None) == UNSUP  # levels > 8
    assert lib.umhs_march_count(dummy, dummy, 8, dummy, roi, 4, 128, 0.05, 1e3, 0.01, 0.0, None, None, None, 0.0, dummy, dummy, 16, None) == WS  # lists too small


def test_null_pointers_of_the_field_and_hashgrid_entries_are_argument_errors():
    """Every pointer a given configuration dereferences is checked on the host (the one GPU fault of round 1
    was a NULL ``d_enc`` reaching the bucket-histogram kernel).  Nothing below gets as far as a launch."""
    from umhsnerf import _hip

    lib = _hip.lib()
    ARG = -1
    d = ctypes.c_void_p(4096)
    cfg = _hip.FieldCfg(31, 6, 1, 0, 0.4)
    pp = _hip.FieldParams(*([4096] * 21))
    gp = _hip.FieldGrads(*([4096] * 21))
    fwd = lambda **kw: lib.umhs_field_fwd(*[kw.get(k, v) for k, v in dict(
        cfg=ctypes.byref(cfg), params=ctypes.byref(pp), enc=d, sn=2, sl=2 * 64, wpos=d, dirs=d, sel=d, n=64, sigma=d, sigma_raw=d, emb=d,
        spectral=d, spectral2=d, specular=d, abund=d, logits=d, ws=None, wsb=0, ready=0, stream=None).items()])
    for missing in ("params", "enc", "wpos", "dirs", "sel", "sigma", "spectral"):
        assert fwd(**{missing: None}) == ARG, missing
    assert fwd(sn=3) == ARG  # odd strides cannot be float2 rows
    no_w = _hip.FieldParams(*([4096] * 21))
    no_w.dir_w1 = None  # a layer this cfg uses
    assert fwd(params=ctypes.byref(no_w)) == ARG
    bwd = lambda **kw: lib.umhs_field_bwd(*[kw.get(k, v) for k, v in dict(
        cfg=ctypes.byref(cfg), params=ctypes.byref(pp), enc=d, sn=2, sl=2 * 64, wpos=d, dirs=d, sel=d, sigma_raw=d, emb=d, logits=d, n=64,
        d_sigma=d, d_spectral=d, d_emb=None, d_enc=d, grads=ctypes.byref(gp), ws=d, wsb=1 << 30, ready=0, stream=None).items()])
    for missing in ("params", "enc", "wpos", "dirs", "sel", "sigma_raw", "emb", "d_sigma", "d_spectral", "grads"):
        assert bwd(**{missing: None}) == ARG, missing
    assert bwd(ws=None) == -3 and bwd(wsb=16) == -3  # workspace missing / too small
    # the two-launch forward and the backward with the compositing backward folded in (none of these has a fallback for a missing input)
    base = lambda **kw: lib.umhs_field_base_fwd(*[kw.get(k, v) for k, v in dict(
        cfg=ctypes.byref(cfg), params=ctypes.byref(pp), enc=d, sn=2, sl=2 * 64, sel=d, n=64, sigma=d, sigma_raw=d, emb=d, b16=None, ws=d,
        wsb=1 << 30, ready=1, stream=None).items()])
    for missing in ("params", "enc", "sel", "sigma", "ws"):
        assert base(**{missing: None}) == ARG, missing
    assert base(wsb=16) == -3
    heads = lambda **kw: lib.umhs_field_heads_fwd(*[kw.get(k, v) for k, v in dict(
        cfg=ctypes.byref(cfg), params=ctypes.byref(pp), emb=d, es=15, wpos=d, dirs=d, n=64, weights=d, ray=d, pinfo=d, R=4, abund=d,
        logits=d, c0=d, c1=d, c2=d, cab=d, scratch=d, sb=1 << 30, ws=d, wsb=1 << 30, ready=1, stream=None).items()])
    assert heads(es=14) == ARG and heads(es=16, emb=ctypes.c_void_p(4100)) == ARG  # 15 or 16; the row form is 16-byte aligned
    for missing in ("params", "emb", "wpos", "dirs", "weights", "ray", "pinfo", "c0", "c1", "c2", "scratch", "ws"):
        assert heads(**{missing: None}) == ARG, missing
    assert heads(sb=16) == -3 and heads(wsb=16) == -3
    assert lib.umhs_field_heads_fwd_scratch_bytes(ctypes.byref(cfg), 64, 4) >= (4 * 2 * (32 + 16 + 16) + 4 * 16) * 4
    assert lib.umhs_field_bwd_composited_scratch_bytes(ctypes.byref(cfg), 64, 4) >= (4 * 32 + 4 * 32 + 6 * 31) * 4
    assert lib.umhs_field_bwd_composited_supported(ctypes.byref(cfg)) == 1
    bwdc = lambda **kw: lib.umhs_field_bwd_composited(*[kw.get(k, v) for k, v in dict(
        cfg=ctypes.byref(cfg), params=ctypes.byref(pp), enc=d, sn=2, sl=2 * 64, wpos=d, dirs=d, sel=d, sigma_raw=d, emb=d, es=15, logits=d, n=64,
        sigma=d, t0=d, t1=d, pinfo=d, R=4, ray=d, weights=d, d_comp=d, d_acc=None, gs=1, d_sigma=d, d_enc=d, grads=ctypes.byref(gp), sc=d,
        scb=1 << 30, ws=d, wsb=1 << 30, ready=0, stream=None).items()])
    assert bwdc(scb=16) == -3
    for missing in ("params", "enc", "wpos", "dirs", "sel", "sigma_raw", "emb", "logits", "sigma", "t0", "t1", "pinfo", "ray", "weights", "d_comp",
                    "d_sigma", "grads", "sc"):
        assert bwdc(**{missing: None}) == ARG, missing
    assert bwdc(ws=None) == -3
    dots = lambda **kw: lib.umhs_composite_bwd_dots(*[kw.get(k, v) for k, v in dict(
        sigma=d, t0=d, t1=d, pinfo=d, R=4, n=64, weights=d, dots=d, d_acc=None, gs=0, d_sigma=d, stream=None).items()])
    for missing in ("sigma", "t0", "t1", "pinfo", "weights", "dots", "d_sigma"):
        assert dots(**{missing: None}) == ARG, missing
    # hash-grid backward halves: positions, scalings, gradient and destination are all required before anything is launched
    prep = lambda pos=d, sc=d: lib.umhs_hashgrid_bwd_prepare(pos, sc, 512, 0, 16, 19, d, 1 << 30, None)
    assert prep(pos=None) == ARG and prep(sc=None) == ARG
    app = lambda pos=d, g=d, sc=d, tab=d: lib.umhs_hashgrid_bwd_apply(pos, g, 2, 1024, sc, 512, 0, 16, 0, 16, 19, tab, 1, d, 1 << 30, None)
    assert app(pos=None) == ARG and app(g=None) == ARG and app(sc=None) == ARG and app(tab=None) == ARG
    one = lambda g=d, tab=d: lib.umhs_hashgrid_bwd(d, g, 2, 1024, d, 512, 0, 16, 19, tab, 0, d, 1 << 30, None)
    assert one(g=None) == ARG and one(tab=None) == ARG
    # training tail: ground truths / outputs the selected losses need
    tail = lambda **kw: lib.umhs_ray_train_tail(*[kw.get(k, v) for k, v in dict(
        s=d, M=d, E=d, acc=d, depth=d, mm=d, colors=d, gt=d, gt_rgb=d, bg=None, R=64, B=31, C=6, alpha=0.2, ws=5.0, wr=1.0, rgb_loss=1, rgb=d,
        dclip=d, probs=d, raw=d, pred=d, losses=d, d_spec=d, d_acc=d, scratch=d, sb=4096, stream=None).items()])
    for missing in ("s", "M", "acc", "mm", "gt", "gt_rgb", "losses", "d_spec", "d_acc", "scratch", "E", "colors", "depth"):
        assert tail(**{missing: None}) == ARG, missing


def test_workspace_slots_are_not_regrown_or_rebuilt_under_an_outstanding_prepare():
    """ops._workspace hands a *_prepare call and its consumer the same buffer or raises: a slot with an outstanding lease is never
    re-grown (the consumer would read an unfilled buffer) and never rebuilt by another call (it would read somebody else's image)."""
    from umhsnerf import ops

    dev = torch.device("cpu")  # the bookkeeping is device-agnostic; no kernel is called here
    ops._ws_cache.pop((0, 7), None)
    a = ops._workspace(1 << 20, dev, slot=7)
    ops._lease(dev, 7, "some_prepare")
    assert ops._workspace(1 << 19, dev, slot=7) is a  # smaller or equal requests are served from the same buffer
    with pytest.raises(RuntimeError, match="re-grown"):
        ops._workspace(1 << 22, dev, slot=7)
    with pytest.raises(RuntimeError, match="overwrite"):
        ops._require_free(dev, 7, "other_call")
    ops._release(dev, 7)
    assert ops._workspace(1 << 22, dev, slot=7) is not a
    ops._ws_cache.pop((0, 7), None)
