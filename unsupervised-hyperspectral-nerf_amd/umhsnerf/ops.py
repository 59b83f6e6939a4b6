"""Host-side operators of the UMHS hot path: thin tensor wrappers over the C ABI plus the
``torch.autograd.Function`` glue that lets nerfstudio's Trainer drive them.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every number is produced by
libumhs_hip.so.  Nothing in this module has a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _hip
from ._hip import ptr

NUM_LEVELS = 16
FEATURES_PER_LEVEL = 2
GEO_FEAT_DIM = 15
HIDDEN = 64


def hash_scalings(num_levels: int = NUM_LEVELS, min_res: int = 16, max_res: int = 2048) -> torch.Tensor:
    """``scalings`` buffer of nerfstudio's HashEncoding (mlp_base, umhs_field.py:51 / umhs_model.py:183-184).
    Deliberately the same expression: the numpy float64 growth factor is raised to an int64 tensor, which torch
    evaluates in float32, so the top level is floor(2047.99..) = 2047."""
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1.0
    return torch.floor(min_res * growth**levels)


# --------------------------------------------------------------------------------------------- #
# flat parameter layout (one fp32 buffer = param group "fields"; names = reference state-dict keys)
# --------------------------------------------------------------------------------------------- #
@dataclass
class FieldLayout:
    num_classes: int
    wavelengths: int
    pred_specular: bool
    log2_hashmap_size: int = 19
    entries: Dict[str, Tuple[int, Tuple[int, ...]]] = field(default_factory=dict)
    total: int = 0

    def __post_init__(self):
        Cn, B = self.num_classes, self.wavelengths
        T = 1 << self.log2_hashmap_size
        out_feat = Cn + 1 if self.pred_specular else Cn
        shapes = [("mlp_base.encoder.hash_table", (NUM_LEVELS * T, FEATURES_PER_LEVEL))]

        def mlp(prefix, dims):
            for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
                shapes.append((f"{prefix}.layers.{i}.weight", (b, a)))
                shapes.append((f"{prefix}.layers.{i}.bias", (b,)))

        mlp("mlp_base.mlp", [NUM_LEVELS * FEATURES_PER_LEVEL, HIDDEN, 1 + GEO_FEAT_DIM])
        mlp("mlp_head", [12 + GEO_FEAT_DIM, HIDDEN, HIDDEN, Cn])
        mlp("feature_mlp", [12 + GEO_FEAT_DIM, HIDDEN, HIDDEN, out_feat])
        mlp("mlp_directional", [16 + 12, 16, B])
        shapes.append(("endmembers", (Cn, B)))
        off = 0
        for name, shp in shapes:
            self.entries[name] = (off, shp)
            off += (int(np.prod(shp)) + 3) & ~3  # 16-byte aligned segments (float4 Adam, float2 table rows)
        self.total = off

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, shp = self.entries[name]
        return flat[off : off + int(np.prod(shp))].view(shp)

    def offset(self, name: str) -> int:
        return self.entries[name][0]

    def tail_offset(self) -> int:
        """First element behind the hash table: the MLP weights / biases and the endmembers follow."""
        return self.entries["mlp_base.mlp.layers.0.weight"][0]

    # (C struct field, state-dict key)
    C_FIELDS = (
        ("base_w0", "mlp_base.mlp.layers.0.weight"), ("base_b0", "mlp_base.mlp.layers.0.bias"),
        ("base_w1", "mlp_base.mlp.layers.1.weight"), ("base_b1", "mlp_base.mlp.layers.1.bias"),
        ("head_w0", "mlp_head.layers.0.weight"), ("head_b0", "mlp_head.layers.0.bias"),
        ("head_w1", "mlp_head.layers.1.weight"), ("head_b1", "mlp_head.layers.1.bias"),
        ("head_w2", "mlp_head.layers.2.weight"), ("head_b2", "mlp_head.layers.2.bias"),
        ("feat_w0", "feature_mlp.layers.0.weight"), ("feat_b0", "feature_mlp.layers.0.bias"),
        ("feat_w1", "feature_mlp.layers.1.weight"), ("feat_b1", "feature_mlp.layers.1.bias"),
        ("feat_w2", "feature_mlp.layers.2.weight"), ("feat_b2", "feature_mlp.layers.2.bias"),
        ("dir_w0", "mlp_directional.layers.0.weight"), ("dir_b0", "mlp_directional.layers.0.bias"),
        ("dir_w1", "mlp_directional.layers.1.weight"), ("dir_b1", "mlp_directional.layers.1.bias"),
        ("endmembers", "endmembers"),
    )

    def c_struct(self, flat: torch.Tensor, cls, tail_only: Optional[torch.Tensor] = None):
        """``tail_only``: a tensor holding just the MLP / endmember segments (elements [tail_offset(), total) of the flat layout);
        the struct then points into it -- none of its fields lies in the hash table."""
        s = cls()
        base = flat.data_ptr() if tail_only is None else tail_only.data_ptr() - 4 * self.tail_offset()
        for cname, key in self.C_FIELDS:
            setattr(s, cname, base + 4 * self.entries[key][0])
        return s


@dataclass
class FieldSpec:
    """Static configuration the kernels need (everything that is not a tensor)."""

    layout: FieldLayout
    temperature: float
    contraction: bool = True
    aabb: Tuple[float, ...] = (-1.0, -1.0, -1.0, 1.0, 1.0, 1.0)
    scalings: Optional[torch.Tensor] = None  # device [16] float32
    grad_sink: Optional[object] = None  # parallel.FlatGradSink: persistent flat-gradient buffer + early all-reduce

    def cfg(self, density_only: bool = False) -> _hip.FieldCfg:
        return _hip.FieldCfg(self.layout.wavelengths, self.layout.num_classes, int(self.layout.pred_specular),
                             int(density_only), float(self.temperature))


# --------------------------------------------------------------------------------------------- #
# raw operator calls
# --------------------------------------------------------------------------------------------- #
def positions_fwd(origins, directions, starts, ends, spec: FieldSpec, world_pos_in=None):
    src = world_pos_in if world_pos_in is not None else origins
    n = src.shape[0]
    wpos = torch.empty((n, 3), device=src.device, dtype=torch.float32) if world_pos_in is None else world_pos_in
    pos01 = torch.empty((n, 3), device=src.device, dtype=torch.float32)
    sel = torch.empty((n,), device=src.device, dtype=torch.float32)
    aabb = (C.c_float * 6)(*spec.aabb)
    _hip.check(
        _hip.lib().umhs_positions_fwd(ptr(origins), ptr(directions), ptr(starts), ptr(ends), ptr(world_pos_in), n,
                                      int(spec.contraction), aabb, ptr(wpos) if world_pos_in is None else None, ptr(pos01),
                                      ptr(sel), _hip.stream()),
        "umhs_positions_fwd",
    )
    return wpos, pos01, sel


def enc_strides(n: int, level_major: bool) -> Tuple[int, int]:
    return (2, 2 * n) if level_major else (2 * NUM_LEVELS, 2)


def hashgrid_fwd(pos01, table, scalings, log2_T: int, level_major: bool = True, out=None, part=None):
    """part = (offset, count): encode only that range of samples into ``out`` (the full-size enc tensor; level-major)."""
    n = pos01.shape[0]
    shape = (NUM_LEVELS, n, 2) if level_major else (n, NUM_LEVELS * 2)
    enc = out if out is not None else torch.empty(shape, device=pos01.device, dtype=torch.float32)
    sn, sl = enc_strides(n, level_major)
    off, cnt = part if part is not None else (0, n)
    _hip.check(_hip.lib().umhs_hashgrid_fwd(pos01.data_ptr() + 12 * off, ptr(table), ptr(scalings), cnt, NUM_LEVELS, log2_T,
                                            enc.data_ptr() + 4 * sn * off, sn, sl, _hip.stream()), "umhs_hashgrid_fwd")
    return enc


def enc_gather(enc, index):
    """Level-major features of the samples ``index`` (int64 [N], ascending) out of an encoded superset ``enc`` [L, M, 2]."""
    L_, m = enc.shape[0], enc.shape[1]
    n = index.shape[0]
    out = torch.empty((L_, n, 2), device=enc.device, dtype=torch.float32)
    _hip.check(_hip.lib().umhs_enc_gather(ptr(enc), ptr(index), m, n, L_, ptr(out), _hip.stream()), "umhs_enc_gather")
    return out


def hashgrid_bwd(pos01, d_enc, scalings, log2_T: int, d_table, level_major: bool = True, method: str = "auto",
                 overwrite: bool = False, level_begin: int = 0, level_count: int = NUM_LEVELS):
    """d_table (+)= scatter(d_enc) for levels [level_begin, level_begin+level_count).  method: "partition" (atomics-free,
    needs a workspace), "atomic", or "auto".  overwrite=True: every slot of those levels is written (no zeroing needed)."""
    n = pos01.shape[0]
    sn, sl = enc_strides(n, level_major)
    nbytes = _hip.lib().umhs_hashgrid_bwd_workspace_bytes(n, level_count, log2_T) if method != "atomic" else 0
    if method == "partition" and nbytes == 0:
        raise RuntimeError("partitioned hash-grid backward unavailable for this shape")
    ws = _workspace(nbytes, pos01.device, slot=1) if nbytes else None
    _hip.check(_hip.lib().umhs_hashgrid_bwd(ptr(pos01), ptr(d_enc), sn, sl, ptr(scalings), n, level_begin, level_count, log2_T,
                                            ptr(d_table), int(overwrite), ptr(ws), ws.numel() if ws is not None else 0,
                                            _hip.stream()),
               "umhs_hashgrid_bwd")


def hashgrid_bwd_prepare(pos01, scalings, log2_T: int, level_begin: int = 0, level_count: int = NUM_LEVELS) -> bool:
    """Gradient-independent half of the partitioned backward (histogram + scan) into the cached workspace.  False: this shape
    has no partitioned path (the caller then uses hashgrid_bwd)."""
    n = pos01.shape[0]
    nbytes = _hip.lib().umhs_hashgrid_bwd_workspace_bytes(n, level_count, log2_T)
    if nbytes == 0:
        return False
    ws = _workspace(nbytes, pos01.device, slot=1)
    _hip.check(_hip.lib().umhs_hashgrid_bwd_prepare(ptr(pos01), ptr(scalings), n, level_begin, level_count, log2_T, ptr(ws), ws.numel(),
                                                    _hip.stream()), "umhs_hashgrid_bwd_prepare")
    _lease(pos01.device, WS_HASH_BWD, "hashgrid_bwd_prepare")
    return True


def hashgrid_fwd_count(pos01, table, scalings, log2_T: int):
    """hashgrid_fwd (level-major) whose launch also takes the bucket histogram of the partitioned backward for the same positions
    into the cached workspace (reserve_step_workspaces sized it); hashgrid_bwd_prepare_counted finishes the prepare.  None: this
    shape has no partitioned path."""
    n = pos01.shape[0]
    nbytes = _hip.lib().umhs_hashgrid_bwd_workspace_bytes(n, NUM_LEVELS, log2_T)
    if nbytes == 0 or n == 0:
        return None
    ws = _workspace(nbytes, pos01.device, slot=1)
    enc = torch.empty((NUM_LEVELS, n, 2), device=pos01.device, dtype=torch.float32)
    sn, sl = enc_strides(n, True)
    _hip.check(_hip.lib().umhs_hashgrid_fwd_count(ptr(pos01), ptr(table), ptr(scalings), n, NUM_LEVELS, log2_T, ptr(enc), sn, sl, ptr(ws),
                                                  ws.numel(), _hip.stream()), "umhs_hashgrid_fwd_count")
    return enc


def hashgrid_bwd_prepare_counted(pos01, scalings, log2_T: int) -> bool:
    """The two scans of hashgrid_bwd_prepare over the histogram hashgrid_fwd_count left in the workspace (all levels)."""
    n = pos01.shape[0]
    ws = _workspace(_hip.lib().umhs_hashgrid_bwd_workspace_bytes(n, NUM_LEVELS, log2_T), pos01.device, slot=1)
    _hip.check(_hip.lib().umhs_hashgrid_bwd_prepare_counted(ptr(pos01), ptr(scalings), n, NUM_LEVELS, log2_T, ptr(ws), ws.numel(),
                                                            _hip.stream()), "umhs_hashgrid_bwd_prepare_counted")
    _lease(pos01.device, WS_HASH_BWD, "hashgrid_bwd_prepare")
    return True


def hashgrid_bwd_apply(pos01, d_enc, scalings, log2_T: int, d_table, level_major: bool = True, overwrite: bool = False,
                       level_begin: int = 0, level_count: int = NUM_LEVELS, ws_range=(0, NUM_LEVELS), adam=None):
    """Scatter + reduce of levels [level_begin, +level_count) using the workspace hashgrid_bwd_prepare filled for ws_range.
    ``adam`` (overwrite mode, n > 0): dict(table, exp_avg, exp_avg_sq [L*T,2] views, lr, betas, eps, step, level_begin) -- the Adam
    step of the table entries of levels >= level_begin rides in the epilogue of the reduce pass."""
    n = pos01.shape[0]
    sn, sl = enc_strides(n, level_major)
    ws = _workspace(_hip.lib().umhs_hashgrid_bwd_workspace_bytes(n, ws_range[1], log2_T), pos01.device, slot=1)
    if adam is not None:
        assert overwrite and n > 0
        _hip.check(_hip.lib().umhs_hashgrid_bwd_apply_adam(ptr(pos01), ptr(d_enc), sn, sl, ptr(scalings), n, level_begin, level_count,
                                                           ws_range[0], ws_range[1], log2_T, ptr(d_table), ptr(ws), ws.numel(),
                                                           ptr(adam["table"]), ptr(adam["exp_avg"]), ptr(adam["exp_avg_sq"]),
                                                           float(adam["lr"]), float(adam["betas"][0]), float(adam["betas"][1]),
                                                           float(adam["eps"]), int(adam["step"]), int(adam["level_begin"]),
                                                           _hip.stream()), "umhs_hashgrid_bwd_apply_adam")
        if level_begin + level_count >= ws_range[0] + ws_range[1]:
            _release(pos01.device, WS_HASH_BWD)
        return
    _hip.check(_hip.lib().umhs_hashgrid_bwd_apply(ptr(pos01), ptr(d_enc), sn, sl, ptr(scalings), n, level_begin, level_count, ws_range[0],
                                                  ws_range[1], log2_T, ptr(d_table), int(overwrite), ptr(ws), ws.numel(), _hip.stream()),
               "umhs_hashgrid_bwd_apply")
    if level_begin + level_count >= ws_range[0] + ws_range[1]:  # the last level group of this prepare has been launched
        _release(pos01.device, WS_HASH_BWD)


def reserve_step_workspaces(spec: FieldSpec, n: int, device) -> bool:
    """Size the cached workspaces of one training step on the CURRENT stream, so that the *_prepare calls issued on a side
    stream never allocate there (blocks handed out under another stream would need record_stream bookkeeping).
    Returns whether the partitioned hash-grid backward is available for this n."""
    cfg = spec.cfg(False)
    for slot in (WS_FIELD_BWD, WS_HASH_BWD, WS_FIELD_FWD):  # a lease still open here belongs to a step that was abandoned
        _release(device, slot)
    _workspace(_hip.lib().umhs_field_fwd_workspace_bytes(C.byref(cfg)), device, slot=2)
    _workspace(_hip.lib().umhs_field_bwd_workspace_bytes(C.byref(cfg), n), device)
    nbytes = _hip.lib().umhs_hashgrid_bwd_workspace_bytes(n, NUM_LEVELS, spec.layout.log2_hashmap_size)
    if nbytes:
        _workspace(nbytes, device, slot=1)
    return nbytes != 0


def field_fwd_prepare(spec: FieldSpec, flat):
    """Build the forward pack image ahead of field_fwd(pack_ready=True) (same cached workspace)."""
    cfg = spec.cfg(False)
    pp = spec.layout.c_struct(flat, _hip.FieldParams)
    ws = _workspace(_hip.lib().umhs_field_fwd_workspace_bytes(C.byref(cfg)), flat.device, slot=2)
    _hip.check(_hip.lib().umhs_field_fwd_prepare(C.byref(cfg), C.byref(pp), ptr(ws), ws.numel(), _hip.stream()), "umhs_field_fwd_prepare")
    _lease(flat.device, WS_FIELD_FWD, "field_fwd_prepare")


def field_bwd_prepare(spec: FieldSpec, flat, n: int):
    """Build the backward's weight images ahead of field_bwd(packs_ready=True); n sizes the shared workspace once."""
    cfg = spec.cfg(False)
    pp = spec.layout.c_struct(flat, _hip.FieldParams)
    ws = _workspace(_hip.lib().umhs_field_bwd_workspace_bytes(C.byref(cfg), n), flat.device)
    _hip.check(_hip.lib().umhs_field_bwd_prepare(C.byref(cfg), C.byref(pp), ptr(ws), ws.numel(), _hip.stream()), "umhs_field_bwd_prepare")
    _lease(flat.device, WS_FIELD_BWD, "field_bwd_prepare")


def field_fwd_outputs(spec: FieldSpec, n: int, dev, density_only=False, want_emb=None, want_aux=True, want_logits=False):
    """``want_emb``: None = the density-only form returns the embedding, the full form does not; True / False force it."""
    L = spec.layout
    new = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
    want_emb = density_only if want_emb is None else want_emb
    out = dict(sigma=new(n), sigma_raw=new(n), emb=new(n, GEO_FEAT_DIM) if want_emb else None, spectral=None,
               spectral2=None, specular=None, abundances=None,
               feat_logits=new(n, 16) if (want_logits and not density_only) else None)  # saved for the split heads backward
    if not density_only:
        out["spectral"] = new(n, L.wavelengths)
        if want_aux:
            out["abundances"] = new(n, L.num_classes)
            if L.pred_specular:
                out["spectral2"], out["specular"] = new(n, L.wavelengths), new(n, L.wavelengths)
    return out


def field_fwd(spec: FieldSpec, flat, enc, level_major, wpos, dirs, sel, density_only=False, want_emb=None, want_aux=True, pack_ready=False,
              out=None, part=None, want_logits=False):
    """part = (offset, count): evaluate only that range of samples, writing into the full-size tensors of ``out``."""
    n = sel.shape[0]
    L = spec.layout
    dev = sel.device
    cfg = spec.cfg(density_only)
    pp = L.c_struct(flat, _hip.FieldParams)
    sn, sl = enc_strides(n, level_major)
    o = out if out is not None else field_fwd_outputs(spec, n, dev, density_only, want_emb, want_aux, want_logits)
    off, cnt = part if part is not None else (0, n)
    at = lambda t, width: (t.data_ptr() + 4 * width * off) if t is not None else None
    if not pack_ready:
        _require_free(dev, WS_FIELD_FWD, "field_fwd (pack image rebuild)")
    ws = _workspace(_hip.lib().umhs_field_fwd_workspace_bytes(C.byref(cfg)), dev, slot=2)
    _hip.check(_hip.lib().umhs_field_fwd(C.byref(cfg), C.byref(pp), at(enc, sn), sn, sl, at(wpos, 3), at(dirs, 3), at(sel, 1), cnt,
                                         at(o["sigma"], 1), at(o["sigma_raw"], 1), at(o["emb"], GEO_FEAT_DIM), at(o["spectral"], L.wavelengths),
                                         at(o["spectral2"], L.wavelengths), at(o["specular"], L.wavelengths),
                                         at(o["abundances"], L.num_classes), at(o.get("feat_logits"), 16), ptr(ws), ws.numel(), int(pack_ready),
                                         _hip.stream()),
               "umhs_field_fwd")
    if pack_ready:
        _release(dev, WS_FIELD_FWD)
    return o


def field_base_fwd(spec: FieldSpec, flat, enc, level_major, sel, pack_ready=False, rows16=False):
    """mlp_base only, from the full configuration's pack images (first launch of the two-launch training forward):
    -> {"sigma" [N], "sigma_raw" [N], "emb" [N,15]}; ``rows16``: instead of emb, "base16" [N,16] = aligned rows (sigma_raw, emb)."""
    n, dev, L = sel.shape[0], sel.device, spec.layout
    cfg = spec.cfg(False)
    pp = L.c_struct(flat, _hip.FieldParams)
    sn, sl = enc_strides(n, level_major)
    new = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
    o = dict(sigma=new(n), sigma_raw=new(n), emb=None if rows16 else new(n, GEO_FEAT_DIM), base16=new(n, 16) if rows16 else None)
    if not pack_ready:
        _require_free(dev, WS_FIELD_FWD, "field_base_fwd (pack image rebuild)")
    ws = _workspace(_hip.lib().umhs_field_fwd_workspace_bytes(C.byref(cfg)), dev, slot=2)
    _hip.check(_hip.lib().umhs_field_base_fwd(C.byref(cfg), C.byref(pp), ptr(enc), sn, sl, ptr(sel), n, ptr(o["sigma"]), ptr(o["sigma_raw"]),
                                              ptr(o["emb"]), ptr(o["base16"]), ptr(ws), ws.numel(), int(pack_ready), _hip.stream()),
               "umhs_field_base_fwd")
    return o


_step_scratch: Dict[Tuple[int, str], torch.Tensor] = {}


def _scratch(dev, key: str, nbytes: int) -> torch.Tensor:
    """Grow-only per-device scratch of the two-launch step (tile partials, per-ray sums); contents are dead after each call."""
    k = (dev.index or 0, key)
    sc = _step_scratch.get(k)
    if sc is None or sc.numel() < nbytes:
        sc = _step_scratch[k] = torch.empty(max(nbytes, 1 << 16), device=dev, dtype=torch.uint8)
    return sc


def field_heads_fwd(spec: FieldSpec, flat, emb, wpos, dirs, weights, ray_indices, packed_info, want_logits=True, pack_ready=True,
                    release=True, want_abundances=False):
    """Second launch of the two-launch training forward: heads from ``emb`` ([N,15], or the [N,16] aligned rows of
    field_base_fwd(rows16=True)) with the per-ray sums formed in the kernel; no [N,B] array is written (the mixing term is summed
    per ray as w m and multiplied by the endmembers once per ray).
    -> {"abundances" [N,C] | None, "feat_logits" [N,16] | None, "comp": [spectral, spectral2, specular] ([R,B]; one entry without the
        specular head), "comp_abundances" [R,C]}.
    ``pack_ready``: the images field_fwd_prepare / field_base_fwd left in the forward workspace are reused."""
    n, dev, L = emb.shape[0], emb.device, spec.layout
    R = packed_info.shape[0]
    cfg = spec.cfg(False)
    pp = L.c_struct(flat, _hip.FieldParams)
    new = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
    B = L.wavelengths
    o = dict(abundances=new(n, L.num_classes) if want_abundances else None, feat_logits=new(n, 16) if want_logits else None,
             comp_abundances=new(R, L.num_classes))
    comp = [new(R, B)] + ([new(R, B), new(R, B)] if L.pred_specular else [])
    sc = _scratch(dev, "heads_fwd", _hip.lib().umhs_field_heads_fwd_scratch_bytes(C.byref(cfg), n, R))
    ws = _workspace(_hip.lib().umhs_field_fwd_workspace_bytes(C.byref(cfg)), dev, slot=2)
    _hip.check(_hip.lib().umhs_field_heads_fwd(C.byref(cfg), C.byref(pp), ptr(emb), emb.shape[1], ptr(wpos), ptr(dirs), n, ptr(weights),
                                               ptr(ray_indices), ptr(packed_info), R, ptr(o["abundances"]), ptr(o["feat_logits"]),
                                               ptr(comp[0]), ptr(comp[1]) if len(comp) > 1 else None,
                                               ptr(comp[2]) if len(comp) > 1 else None, ptr(o["comp_abundances"]), ptr(sc), sc.numel(), ptr(ws),
                                               ws.numel(), int(pack_ready), _hip.stream()), "umhs_field_heads_fwd")
    if pack_ready and release:
        _release(dev, WS_FIELD_FWD)
    o["comp"] = comp
    return o


def field_density(spec: FieldSpec, flat, pos01, sel, want_emb: bool = True, keep=None):
    """density_fn: hash-grid gather + density-only field forward (two launches; the level-major features live in a per-device scratch,
    or in ``keep["enc"]`` for a caller that reuses them) -> {"sigma" [N], "sigma_raw" [N], "emb" [N,15] | None}."""
    n, L = sel.shape[0], spec.layout
    if keep is None:
        enc = _scratch(sel.device, "density_enc", NUM_LEVELS * n * 8)[: NUM_LEVELS * n * 8].view(torch.float32).view(NUM_LEVELS, n, 2)
    else:
        enc = keep["enc"] = torch.empty((NUM_LEVELS, n, 2), device=sel.device, dtype=torch.float32)
    hashgrid_fwd(pos01, L.view(flat, "mlp_base.encoder.hash_table"), spec.scalings, L.log2_hashmap_size, True, out=enc)
    return field_fwd(spec, flat, enc, True, None, None, sel, density_only=True, want_emb=want_emb)


_ws_cache: Dict[Tuple[int, int], torch.Tensor] = {}
_ws_leases: Dict[Tuple[int, int], str] = {}  # (device, slot) -> the *_prepare call whose consumer has not been launched yet

WS_FIELD_BWD, WS_HASH_BWD, WS_FIELD_FWD = 0, 1, 2  # slots of the per-device workspace cache


def _workspace(nbytes: int, device, slot: int = 0) -> torch.Tensor:
    """Cached per-(device, slot) scratch.  A slot is never re-grown while a ``*_prepare`` call has parked data in it for a consumer
    that has not been launched yet (``_lease`` / ``_release``): the consumer would be handed a fresh, unfilled buffer."""
    key = (device.index or 0, slot)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        if key in _ws_leases:
            raise RuntimeError(f"workspace slot {slot} would be re-grown ({0 if ws is None else ws.numel()} -> {nbytes} bytes) while "
                               f"{_ws_leases[key]} holds data in it for a consumer that has not run; size it first "
                               "(ops.reserve_step_workspaces)")
        ws = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
        _ws_cache[key] = ws
    return ws


def _lease(device, slot: int, who: str) -> None:
    _ws_leases[(device.index or 0, slot)] = who


def _release(device, slot: int) -> None:
    _ws_leases.pop((device.index or 0, slot), None)


def _require_free(device, slot: int, who: str) -> None:
    """A call that rebuilds a slot's contents must not run between a ``*_prepare`` and its consumer."""
    held = _ws_leases.get((device.index or 0, slot))
    if held is not None:
        raise RuntimeError(f"{who} would overwrite workspace slot {slot} while {held} holds data in it for a consumer that has not run")


def field_heads_fwd_supported(spec: FieldSpec) -> bool:
    """Can field_base_fwd / field_heads_fwd (the two-launch forward with the per-ray sums in the kernel) serve this configuration?"""
    ok = getattr(spec, "_heads_fwd_ok", None)
    if ok is None:  # depends on the configuration only: asked once
        cfg = spec.cfg(False)
        ok = bool(_hip.lib().umhs_field_heads_fwd_supported(C.byref(cfg)))
        try:
            spec._heads_fwd_ok = ok
        except AttributeError:
            pass
    return ok


def field_bwd_composited_supported(spec: FieldSpec) -> bool:
    ok = getattr(spec, "_composited_ok", None)
    if ok is None:  # depends on the configuration only: asked once
        cfg = spec.cfg(False)
        ok = bool(_hip.lib().umhs_field_bwd_composited_supported(C.byref(cfg)))
        try:
            spec._composited_ok = ok
        except AttributeError:
            pass
    return ok


def field_bwd(spec: FieldSpec, flat, enc, level_major, wpos, dirs, sel, sigma_raw, emb, d_sigma, d_spectral, d_emb, d_flat,
              packs_ready=False, feat_logits=None, d_tail=None, comp=None):
    """Writes d_enc (returned) and the MLP / endmember gradients straight into ``d_flat`` (flat layout) -- or, when ``d_tail`` is
    given (a tensor of ``total - tail_offset()`` floats), into that instead (gradient accumulation adds it to the flat gradient).
    ``comp``: dict(sigma, t0, t1, packed_info, ray_indices, weights, d_comp, d_acc, grad_scaling) -- the value half of the compositing
    backward runs inside (umhs_field_bwd_composited); d_sigma / d_spectral are then ignored and comp["d_sigma"] receives d_sigma."""
    n = sel.shape[0]
    L = spec.layout
    cfg = spec.cfg(False)
    pp = L.c_struct(flat, _hip.FieldParams)
    gp = L.c_struct(d_flat, _hip.FieldGrads, tail_only=d_tail)
    sn, sl = enc_strides(n, level_major)
    d_enc = torch.empty_like(enc)
    nbytes = _hip.lib().umhs_field_bwd_workspace_bytes(C.byref(cfg), n)
    ws = _workspace(nbytes, sel.device)
    if comp is not None:
        comp["d_sigma"] = torch.empty(n, device=sel.device, dtype=torch.float32)
        pi = comp["packed_info"]
        sc = _scratch(sel.device, "bwd_comp", _hip.lib().umhs_field_bwd_composited_scratch_bytes(C.byref(cfg), n, pi.shape[0]))
        _hip.check(_hip.lib().umhs_field_bwd_composited(
            C.byref(cfg), C.byref(pp), ptr(enc), sn, sl, ptr(wpos), ptr(dirs), ptr(sel), ptr(sigma_raw), ptr(emb), emb.shape[1], ptr(feat_logits), n,
            ptr(comp["sigma"]), ptr(comp["t0"]), ptr(comp["t1"]), ptr(pi), pi.shape[0], ptr(comp["ray_indices"]), ptr(comp["weights"]),
            ptr(comp["d_comp"]), ptr(comp["d_acc"]), int(bool(comp["grad_scaling"])), ptr(comp["d_sigma"]), ptr(d_enc), C.byref(gp), ptr(sc),
            sc.numel(), ptr(ws), ws.numel(), int(packs_ready), _hip.stream()), "umhs_field_bwd_composited")
        if packs_ready:
            _release(sel.device, WS_FIELD_BWD)
        return d_enc
    _hip.check(_hip.lib().umhs_field_bwd(C.byref(cfg), C.byref(pp), ptr(enc), sn, sl, ptr(wpos), ptr(dirs), ptr(sel),
                                         ptr(sigma_raw), ptr(emb), ptr(feat_logits), n,
                                         ptr(d_sigma), ptr(d_spectral), ptr(d_emb), ptr(d_enc), C.byref(gp), ptr(ws),
                                         ws.numel(), int(packs_ready), _hip.stream()), "umhs_field_bwd")
    if packs_ready:
        _release(sel.device, WS_FIELD_BWD)
    return d_enc


def pack_info(ray_indices: torch.Tensor, num_rays: int) -> torch.Tensor:
    out = torch.empty((num_rays, 2), device=ray_indices.device, dtype=torch.int64)
    _hip.check(_hip.lib().umhs_pack_info(ptr(ray_indices), ray_indices.shape[0], num_rays, ptr(out), _hip.stream()),
               "umhs_pack_info")
    return out


def composite_fwd(sigma, t0, t1, packed_info, values: Sequence[torch.Tensor]):
    R, n, dev = packed_info.shape[0], sigma.shape[0], sigma.device
    st = _hip.ValueStreams()
    st.n_streams = len(values)
    outs = []
    for i, v in enumerate(values):
        o = torch.empty((R, v.shape[-1]), device=dev, dtype=torch.float32)
        st.k[i], st.values[i], st.out[i] = v.shape[-1], v.data_ptr(), o.data_ptr()
        outs.append(o)
    weights = torch.empty((n,), device=dev, dtype=torch.float32)
    acc = torch.empty((R,), device=dev, dtype=torch.float32)
    depth = torch.empty((R,), device=dev, dtype=torch.float32)
    _hip.check(_hip.lib().umhs_composite_fwd(ptr(sigma), ptr(t0), ptr(t1), ptr(packed_info), R, n, C.byref(st), ptr(weights),
                                             ptr(acc), ptr(depth), _hip.stream()), "umhs_composite_fwd")
    return weights, acc, depth, outs


def composite_bwd(sigma, t0, t1, packed_info, weights, values, d_outs, want_dvalues, d_acc, grad_scaling: bool):
    R, n, dev = packed_info.shape[0], sigma.shape[0], sigma.device
    g = _hip.ValueGrads()
    d_values: List[Optional[torch.Tensor]] = []
    k = 0
    for v, do, want in zip(values, d_outs, want_dvalues):
        if do is None:
            d_values.append(None)
            continue
        dv = torch.empty_like(v) if want else None
        g.k[k], g.values[k], g.d_out[k] = v.shape[-1], v.data_ptr(), do.data_ptr()
        g.d_values[k] = dv.data_ptr() if dv is not None else None
        d_values.append(dv)
        k += 1
    g.n_streams = k
    d_sigma = torch.empty((n,), device=dev, dtype=torch.float32)  # every packed sample belongs to a ray: fully written
    _hip.check(_hip.lib().umhs_composite_bwd(ptr(sigma), ptr(t0), ptr(t1), ptr(packed_info), R, n, ptr(weights), C.byref(g),
                                             ptr(d_acc), int(grad_scaling), ptr(d_sigma), _hip.stream()), "umhs_composite_bwd")
    return d_sigma, d_values


def spec2rgb_fwd(spec_t, M):
    R, B = spec_t.shape
    rgb = torch.empty((R, 3), device=spec_t.device, dtype=torch.float32)
    _hip.check(_hip.lib().umhs_spec2rgb_fwd(ptr(spec_t), ptr(M), R, B, ptr(rgb), _hip.stream()), "umhs_spec2rgb_fwd")
    return rgb


def spec2rgb_bwd(spec_t, M, d_rgb, accumulate_into=None):
    R, B = spec_t.shape
    d_spec = torch.empty_like(spec_t) if accumulate_into is None else accumulate_into
    _hip.check(_hip.lib().umhs_spec2rgb_bwd(ptr(spec_t), ptr(M), ptr(d_rgb), R, B, ptr(d_spec), int(accumulate_into is not None), _hip.stream()),
               "umhs_spec2rgb_bwd")
    return d_spec


def adam_step(p, g, m, v, step: int, lr: float, betas=(0.9, 0.999), eps=1e-15, grad_scale=1.0, clamp_range=(0, 0)):
    _hip.check(_hip.lib().umhs_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps, step,
                                         grad_scale, clamp_range[0], clamp_range[1], _hip.stream()), "umhs_adam_step")


def pixel_indices(uniform, n_images: int, height: int, width: int):
    """PixelSampler.sample_method: long(uniform[R,3] * (n, H, W)) -> [R,3] int64 rows (camera, y, x)."""
    u = _hip.f32c(uniform)
    out = torch.empty(u.shape, dtype=torch.int64, device=u.device)
    _hip.check(_hip.lib().umhs_pixel_indices(ptr(u), u.shape[0], n_images, height, width, ptr(out), _hip.stream()), "umhs_pixel_indices")
    return out


def raygen(indices, c2w, intrinsics, want_area: bool = True, want_norm: bool = False):
    """Cameras.generate_rays for perspective cameras: -> origins [R,3], directions [R,3], pixel_area [R,1] | None, norm | None."""
    r, dev = indices.shape[0], indices.device
    o, d = torch.empty(r, 3, device=dev), torch.empty(r, 3, device=dev)
    area = torch.empty(r, 1, device=dev) if want_area else None
    nrm = torch.empty(r, 1, device=dev) if want_norm else None
    _hip.check(_hip.lib().umhs_raygen(ptr(indices), ptr(c2w), ptr(intrinsics), r, c2w.shape[0], ptr(o), ptr(d), ptr(area), ptr(nrm),
                                      _hip.stream()), "umhs_raygen")
    return o, d, area, nrm


def pixel_gather(indices, stack):
    """batch[key] = stack[c, y, x] for a resident image stack [n,H,W,K] (fp32, or uint8 -> /255)."""
    assert stack.dim() == 4 and stack.is_contiguous() and stack.dtype in (torch.float32, torch.uint8)
    n, h, w, k = stack.shape
    out = torch.empty(indices.shape[0], k, device=indices.device)
    _hip.check(_hip.lib().umhs_pixel_gather(ptr(indices), ptr(stack), int(stack.dtype == torch.uint8), n, h, w, k, indices.shape[0],
                                            ptr(out), _hip.stream()), "umhs_pixel_gather")
    return out


def pixel_metrics(pred, gt):
    """(sum of squared errors, sum of finite spectral angles, number of finite angles) over channel-last images [...,K]; float64."""
    k = pred.shape[-1]
    p, g = _hip.f32c(pred).view(-1, k), _hip.f32c(gt).view(-1, k)
    assert p.shape == g.shape
    nb = max(1, min(1024, (p.shape[0] + 255) // 256))
    part = torch.empty(nb, 3, dtype=torch.float64, device=p.device)
    _hip.check(_hip.lib().umhs_pixel_metrics(ptr(p), ptr(g), p.shape[0], k, ptr(part), nb, _hip.stream()), "umhs_pixel_metrics")
    return part.sum(0)


def ssim(a, b, data_range=None):
    """torchmetrics structural_similarity_index_measure (gaussian 11x11, sigma 1.5) of channel-last images [H,W,K] -> 0-dim float64."""
    a, b = _hip.f32c(a), _hip.f32c(b)
    assert a.dim() == 3 and a.shape == b.shape
    h, w, k = a.shape
    n = _hip.lib().umhs_ssim_partials(h, w, k)
    if n == 0:
        raise ValueError(f"SSIM needs images of at least 11x11 pixels, got {h}x{w}")
    if data_range is None:
        (alo, ahi), (blo, bhi) = torch.aminmax(a), torch.aminmax(b)
        dr = torch.maximum(ahi - alo, bhi - blo).reshape(1)
    else:
        dr = torch.full((1,), float(data_range), device=a.device)
    part = torch.empty(n, dtype=torch.float64, device=a.device)
    _hip.check(_hip.lib().umhs_ssim(ptr(a), ptr(b), h, w, k, ptr(dr), ptr(part), n, _hip.stream()), "umhs_ssim")
    return part.sum() / float((h - 10) * (w - 10) * k)


def field_backward_into(spec: FieldSpec, flat, pos01, sel, wpos, d, enc, sigma_raw, emb, d_sigma, d_spectral, d_emb,
                        prepared: bool = False, feat_logits=None, hash_ready=None, comp=None):
    """``prepared``: field_bwd_prepare and hashgrid_bwd_prepare (all levels) already ran for this step's parameters/positions;
    ``hash_ready``: event of the stream hashgrid_bwd_prepare was issued on -- waited for only in front of the scatter pass, so the
    histogram may still be running under the field backward.
    Backward of the field (field_bwd + hash-grid scatter) into the flat gradient.  With a gradient sink that owns the next
    backward the buffer becomes ``param.grad`` directly, finished segments start their all-reduce, and None is returned;
    otherwise the freshly written flat gradient is returned (autograd accumulates it)."""
    L = spec.layout
    sink = spec.grad_sink
    own = sink is not None and sink.param is flat and sink.owns_next_backward()
    # the 64 MiB table segment is fully written by hashgrid_bwd(overwrite=True): no memset of the flat gradient
    d_flat = sink.begin() if own else torch.empty_like(flat)
    acc = own and sink.accumulating  # a further micro-step of an accumulation window: everything below ADDS to the buffer
    tail = L.tail_offset()
    if not own:  # field_reduce overwrites every weight / bias / endmember entry; only the alignment padding between the segments is
        d_flat[tail:].zero_()  # never written -- the sink's persistent buffer has it zeroed once, a fresh tensor needs it now
    d_tail = torch.zeros(L.total - tail, device=flat.device, dtype=torch.float32) if acc else None
    d_enc = field_bwd(spec, flat.detach(), enc, True, wpos, d, sel, sigma_raw, emb, d_sigma, d_spectral, d_emb, d_flat,
                      packs_ready=prepared, feat_logits=feat_logits, d_tail=d_tail, comp=comp)
    if acc:
        d_flat[tail:].add_(d_tail)
    if own:
        sink.segment_done(d_flat[tail:])
    table = L.view(d_flat, "mlp_base.encoder.hash_table")
    T = 1 << L.log2_hashmap_size
    if prepared and hash_ready is not None:
        torch.cuda.current_stream(flat.device).wait_event(hash_ready)
    groups = sink.groups(NUM_LEVELS) if own else [(0, NUM_LEVELS)]
    # armed by the trainer (one GPU, optimizer.step() follows): the table's Adam step rides in the reduce pass (UMHSAdam.arm_fused)
    fused = sink.take_fused_adam(flat) if (own and not acc and prepared and len(groups) == 1 and pos01.shape[0] > 0) else None
    ow = not acc  # overwrite mode writes every slot of the levels; accumulation runs the same kernels in their += mode
    for l0, cnt in groups:
        if prepared and fused is not None:
            lv = lambda t: L.view(t, "mlp_base.encoder.hash_table")
            hashgrid_bwd_apply(pos01, d_enc, spec.scalings, L.log2_hashmap_size, table, True, overwrite=True, level_begin=l0, level_count=cnt,
                               adam=dict(fused, table=lv(flat.data), exp_avg=lv(fused["exp_avg"]), exp_avg_sq=lv(fused["exp_avg_sq"])))
            tb = L.offset("mlp_base.encoder.hash_table")
            sink.adam_done = (fused["step"], tb + fused["level_begin"] * T * FEATURES_PER_LEVEL, tb + NUM_LEVELS * T * FEATURES_PER_LEVEL)
        elif prepared:  # histogram + scan of all levels were done ahead of time (hashgrid_bwd_prepare)
            hashgrid_bwd_apply(pos01, d_enc, spec.scalings, L.log2_hashmap_size, table, True, overwrite=ow, level_begin=l0, level_count=cnt)
        else:
            hashgrid_bwd(pos01, d_enc, spec.scalings, L.log2_hashmap_size, table, True, overwrite=ow, level_begin=l0, level_count=cnt)
        if own:
            sink.table_levels_done(table, l0, cnt)
    if own:  # the buffer becomes param.grad directly (autograd gets None: nothing to accumulate or copy)
        sink.commit()
        return None
    return d_flat


def ray_epilogue_fwd(s, m, E, accumulation, depth, tminmax, colors, alpha: float):
    R, B = s.shape
    Cn = E.shape[0]
    new = lambda *shp: torch.empty(shp, device=s.device, dtype=torch.float32)
    rgb, dclip, probs, raw, pred = new(R, 3), new(R, 1), new(R, Cn), new(R), new(R, 3)
    _hip.check(_hip.lib().umhs_ray_epilogue_fwd(ptr(s), ptr(m), ptr(E), ptr(accumulation), ptr(depth), ptr(tminmax), ptr(colors), R, B, Cn,
                                                float(alpha), ptr(rgb), ptr(dclip), ptr(probs), ptr(raw), ptr(pred), _hip.stream()),
               "umhs_ray_epilogue_fwd")
    return rgb, dclip, probs, raw, pred


_tail_scratch: Dict[int, torch.Tensor] = {}


def ray_train_tail(s, m, E, accumulation, depth, tminmax, colors, gt_spec, gt_rgb, bg, alpha: float, w_spec: float, w_rgb: float,
                   rgb_loss: bool):
    """Per-ray tail of a training step in one launch: ray_epilogue_fwd + loss_fwd + loss_bwd(unit upstream) + spec2rgb_bwd.
    -> (rgb, depth_clipped, seg_probs, seg_raw, seg_pred, losses[2], d_spectral, d_accumulation | None)"""
    R, B = s.shape
    Cn = E.shape[0]
    dev = s.device
    # ONE allocation for the nine outputs, carved into 16-byte-aligned views (fresh every call: callers keep what they get).  Nine
    # torch.empty calls were ~40 us of host time between a profiler's event pair around this operator -- more than the kernel's 19 us,
    # so whenever the GPU ran dry the "operator time" was the host's (BENCH_r03's kernels_ms row).
    sizes = [R * 3, R, R * Cn, R, R * 3, 2, R * B, R if rgb_loss else 0]
    offs, total = [], 0
    for n_ in sizes:
        offs.append(total)
        total += (n_ + 3) & ~3
    buf = torch.empty(total, device=dev, dtype=torch.float32)
    view = lambda i, *shp: buf[offs[i]:offs[i] + sizes[i]].view(*shp)
    rgb, dclip, probs, raw, pred = view(0, R, 3), view(1, R, 1), view(2, R, Cn), view(3, R), view(4, R, 3)
    losses, d_spec = view(5, 2), view(6, R, B)
    d_acc = view(7, R) if rgb_loss else None
    sc = _tail_scratch.get(dev.index or 0)
    if sc is None:
        sc = _tail_scratch[dev.index or 0] = torch.zeros(_hip.lib().umhs_ray_train_tail_scratch_bytes(), dtype=torch.uint8, device=dev)
    _hip.check(_hip.lib().umhs_ray_train_tail(ptr(s), ptr(m), ptr(E), ptr(accumulation), ptr(depth), ptr(tminmax), ptr(colors), ptr(gt_spec),
                                              ptr(gt_rgb), ptr(bg), R, B, Cn, float(alpha), float(w_spec), float(w_rgb), int(rgb_loss),
                                              ptr(rgb), ptr(dclip), ptr(probs), ptr(raw), ptr(pred), ptr(losses), ptr(d_spec), ptr(d_acc),
                                              ptr(sc), sc.numel(), _hip.stream()), "umhs_ray_train_tail")
    return rgb, dclip, probs, raw, pred, losses, d_spec, d_acc


def loss_fwd(s, g, r, a, bg, gr, w_spec: float, w_rgb: float):
    R, B = s.shape
    losses = torch.empty(2, device=s.device, dtype=torch.float32)
    _hip.check(_hip.lib().umhs_loss_fwd(ptr(s), ptr(g), ptr(r), ptr(a), ptr(bg), ptr(gr), R, B, float(w_spec), float(w_rgb),
                                        ptr(losses), _hip.stream()), "umhs_loss_fwd")
    return losses


def loss_bwd(s, g, r, a, bg, gr, w_spec: float, w_rgb: float, grad_losses):
    R, B = s.shape
    d_spec = torch.empty_like(s)
    d_rgb = torch.empty_like(r) if r is not None else None
    d_acc = torch.empty_like(a) if r is not None else None
    _hip.check(_hip.lib().umhs_loss_bwd(ptr(s), ptr(g), ptr(r), ptr(a), ptr(bg), ptr(gr), R, B, float(w_spec), float(w_rgb), ptr(grad_losses),
                                        ptr(d_spec), ptr(d_rgb), ptr(d_acc), _hip.stream()), "umhs_loss_bwd")
    return d_spec, d_rgb, d_acc


def adam_step_rows(p, g, m, v, rows, step: int, lr: float, betas=(0.9, 0.999), eps=1e-15, grad_scale=1.0):
    """Adam on the 2-float rows ``rows`` (int64, device) of the flat buffers only."""
    _hip.check(_hip.lib().umhs_adam_step_rows(ptr(p), ptr(g), ptr(m), ptr(v), ptr(rows), rows.numel(), lr, betas[0], betas[1], eps, step,
                                              grad_scale, _hip.stream()), "umhs_adam_step_rows")


def adam_step_rows_range(p, g, m, v, rows, begin: int, end: int, step: int, lr: float, betas=(0.9, 0.999), eps=1e-15, grad_scale=1.0,
                         clamp_range=(0, 0)):
    """adam_step_rows(rows) + adam_step on elements [begin, end) of the same flat buffers (clamp_range in absolute elements): one launch."""
    _hip.check(_hip.lib().umhs_adam_step_rows_range(ptr(p), ptr(g), ptr(m), ptr(v), ptr(rows), rows.numel(), begin, end - begin, lr, betas[0],
                                                    betas[1], eps, step, grad_scale, clamp_range[0], clamp_range[1], _hip.stream()),
               "umhs_adam_step_rows_range")


# --------------------------------------------------------------------------------------------- #
# method="rgb": the two MLPs of NerfactoField as HIP kernels (csrc/umhs_rgb.hip)
# --------------------------------------------------------------------------------------------- #
def rgb_base_fwd(enc, sel, w0, b0, w1, b1, want_emb: bool = True, want_raw: bool = False):
    """mlp_base of the rgb field: enc [N,32] (sample-major) -> (density [N] = trunc_exp(out0) * sel, emb [N,15] | None, sigma_raw [N] | None)."""
    n = enc.shape[0]
    new = lambda *shp: torch.empty(shp, device=enc.device, dtype=torch.float32)
    density, emb, raw = new(n), (new(n, 15) if want_emb else None), (new(n) if want_raw else None)
    _hip.check(_hip.lib().umhs_rgb_base_fwd(ptr(enc), ptr(sel), ptr(w0), ptr(b0), ptr(w1), ptr(b1), n, ptr(density), ptr(emb), ptr(raw),
                                            _hip.stream()), "umhs_rgb_base_fwd")
    return density, emb, raw


def rgb_head_fwd(dirs, emb, w0, b0, w1, b1, w2, b2):
    """NerfactoField.mlp_head: [SH16((d + 1) / 2) | emb15] -> 64 -> 64 -> 3, sigmoid -> rgb [N,3]."""
    n = dirs.shape[0]
    rgb = torch.empty((n, 3), device=dirs.device, dtype=torch.float32)
    _hip.check(_hip.lib().umhs_rgb_head_fwd(ptr(dirs), ptr(emb), ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(w2), ptr(b2), n, ptr(rgb),
                                            _hip.stream()), "umhs_rgb_head_fwd")
    return rgb


class RgbBaseFn(torch.autograd.Function):
    """(enc [N,32], sel [N], w0, b0, w1, b1) -> (density [N], emb [N,15]); backward recomputes the forward inside the kernel."""

    @staticmethod
    def forward(ctx, enc, sel, w0, b0, w1, b1):
        enc, sel = enc.contiguous(), sel.contiguous()
        w = [t.detach().contiguous() for t in (w0, b0, w1, b1)]
        density, emb, _ = rgb_base_fwd(enc, sel, *w)
        ctx.save_for_backward(enc, sel, *w)
        return density, emb

    @staticmethod
    def backward(ctx, d_density, d_emb):
        enc, sel, w0, b0, w1, b1 = ctx.saved_tensors
        n = enc.shape[0]
        d_enc = torch.empty_like(enc)
        g = [torch.empty_like(t) for t in (w0, b0, w1, b1)]
        if n == 0:
            return (d_enc, None, *[t.zero_() for t in g])
        dd = d_density.contiguous().float() if d_density is not None else None
        de = d_emb.contiguous().float() if d_emb is not None else None
        if dd is None and de is None:
            dd = torch.zeros(n, device=enc.device)
        ws = _scratch(enc.device, "rgb_mlp_bwd", _hip.lib().umhs_rgb_mlp_bwd_workspace_bytes(0, n))
        _hip.check(_hip.lib().umhs_rgb_base_bwd(ptr(enc), ptr(sel), ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(dd), ptr(de), n, ptr(d_enc),
                                                ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(g[3]), 0, ptr(ws), ws.numel(), _hip.stream()),
                   "umhs_rgb_base_bwd")
        return (d_enc, None, *g)


class RgbHeadFn(torch.autograd.Function):
    """(directions [N,3], emb [N,15], w0, b0, w1, b1, w2, b2) -> rgb [N,3]."""

    @staticmethod
    def forward(ctx, dirs, emb, w0, b0, w1, b1, w2, b2):
        dirs, emb = dirs.contiguous(), emb.contiguous()
        w = [t.detach().contiguous() for t in (w0, b0, w1, b1, w2, b2)]
        rgb = rgb_head_fwd(dirs, emb, *w)
        ctx.save_for_backward(dirs, emb, *w)
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        dirs, emb, w0, b0, w1, b1, w2, b2 = ctx.saved_tensors
        n = dirs.shape[0]
        d_emb = torch.empty_like(emb)
        g = [torch.empty_like(t) for t in (w0, b0, w1, b1, w2, b2)]
        if n == 0:
            return (None, d_emb, *[t.zero_() for t in g])
        dr = d_rgb.contiguous().float()
        ws = _scratch(dirs.device, "rgb_mlp_bwd", _hip.lib().umhs_rgb_mlp_bwd_workspace_bytes(1, n))
        _hip.check(_hip.lib().umhs_rgb_head_bwd(ptr(dirs), ptr(emb), ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(dr), n,
                                                ptr(d_emb), *[ptr(t) for t in g], 0, ptr(ws), ws.numel(), _hip.stream()), "umhs_rgb_head_bwd")
        return (None, d_emb, *g)


# --------------------------------------------------------------------------------------------- #
# autograd glue
# --------------------------------------------------------------------------------------------- #
class FieldFn(torch.autograd.Function):
    """UMHSField.forward (get_density + get_outputs, umhs_field.py:151-329) on packed samples.

    inputs : flat params, origins [N,3], directions [N,3], starts [N,1], ends [N,1]
    outputs: density [N,1], emb [N,15], spectral [N,B], spectral2 | None, specular | None, abundances [N,C]
    """

    @staticmethod
    def forward(ctx, flat, origins, directions, starts, ends, spec: FieldSpec):
        ctx.set_materialize_grads(False)  # unused outputs arrive as None instead of freshly filled zero tensors
        L = spec.layout
        o, d = _hip.f32c(origins), _hip.f32c(directions)
        s, e = _hip.f32c(starts).view(-1), _hip.f32c(ends).view(-1)
        wpos, pos01, sel = positions_fwd(o, d, s, e, spec)
        table = L.view(flat.detach(), "mlp_base.encoder.hash_table")
        enc = hashgrid_fwd(pos01, table, spec.scalings, L.log2_hashmap_size, True)
        out = field_fwd(spec, flat.detach(), enc, True, wpos, d, sel, want_emb=True, want_logits=flat.requires_grad)
        ctx.spec = spec
        ctx.has_logits = out["feat_logits"] is not None
        ctx.save_for_backward(flat, pos01, sel, wpos, d, enc, out["sigma_raw"], out["emb"], *([out["feat_logits"]] if ctx.has_logits else []))
        n = o.shape[0]
        res = [out["sigma"].view(n, 1), out["emb"], out["spectral"], out["spectral2"], out["specular"], out["abundances"]]
        ctx.mark_non_differentiable(*[t for t in res[3:] if t is not None])
        return tuple(res)

    @staticmethod
    def backward(ctx, d_sigma, d_emb, d_spectral, *_):
        flat, pos01, sel, wpos, d, enc, sigma_raw, emb, *rest = ctx.saved_tensors
        logits = rest[0] if ctx.has_logits else None
        spec: FieldSpec = ctx.spec
        L = spec.layout
        n = sel.shape[0]
        zeros = lambda *s: torch.zeros(s, device=sel.device, dtype=torch.float32)
        d_sigma = _hip.f32c(d_sigma).view(-1) if d_sigma is not None else zeros(n)
        d_spectral = _hip.f32c(d_spectral) if d_spectral is not None else zeros(n, L.wavelengths)
        d_emb = _hip.f32c(d_emb) if d_emb is not None else None
        d_flat = field_backward_into(spec, flat, pos01, sel, wpos, d, enc, sigma_raw, emb, d_sigma, d_spectral, d_emb, feat_logits=logits)
        return d_flat, None, None, None, None, None


class DensityFn(torch.autograd.Function):
    """density_fn / get_density forward only (no-grad users: occupancy grid, sampler; umhs_model.py:208,553).
    ``keep``: a dict that receives the level-major hash features of the queried positions (``keep["enc"]``)."""

    @staticmethod
    def forward(ctx, flat, positions, spec: FieldSpec, keep=None, want_emb=True):
        L = spec.layout
        p = _hip.f32c(positions).view(-1, 3)
        _, pos01, sel = positions_fwd(None, None, None, None, spec, world_pos_in=p)
        out = field_density(spec, flat.detach(), pos01, sel, want_emb, keep)
        if out["emb"] is None:  # density_fn callers (sampler, occupancy grid) only want sigma: 60 B per sample not written
            out["emb"] = out["sigma"].new_empty(0)
        ctx.mark_non_differentiable(out["sigma"], out["emb"])
        return out["sigma"].view(-1, 1), out["emb"]

    @staticmethod
    def backward(ctx, *_):
        raise RuntimeError("density_fn is a no-grad path (umhs_model.py:229-237); use UMHSField.forward for gradients")


class CompositeFn(torch.autograd.Function):
    """nerfacc.render_weight_from_density + accumulate_along_rays for every composited stream at once
    (umhs_model.py:245-304 -> umhs_renderer.py:28-30).  values: tuple of [N,k] tensors; returns
    (weights [N,1], accumulation [R,1], depth_unclipped [R,1], *composited [R,k])."""

    @staticmethod
    def forward(ctx, sigma, starts, ends, packed_info, grad_scaling: bool, *values):
        ctx.set_materialize_grads(False)
        s = _hip.f32c(sigma).view(-1)
        t0, t1 = _hip.f32c(starts).view(-1), _hip.f32c(ends).view(-1)
        vals = [_hip.f32c(v).view(v.shape[-2], v.shape[-1]) for v in values]
        weights, acc, depth, outs = composite_fwd(s, t0, t1, packed_info, vals)
        ctx.save_for_backward(s, t0, t1, packed_info, weights, *vals)
        ctx.grad_scaling = grad_scaling
        ctx.needs = [v.requires_grad for v in values]
        ctx.vshapes = [v.shape for v in values]
        ctx.mark_non_differentiable(weights, depth)
        return (weights.view(-1, 1), acc.view(-1, 1), depth.view(-1, 1), *outs)

    @staticmethod
    def backward(ctx, _dw, d_acc, _dd, *d_outs):
        s, t0, t1, pinfo, weights, *vals = ctx.saved_tensors
        d_outs = [(_hip.f32c(g) if g is not None else None) for g in d_outs]
        d_acc = _hip.f32c(d_acc).view(-1) if d_acc is not None else None
        d_sigma, d_values = composite_bwd(s, t0, t1, pinfo, weights, vals, d_outs, ctx.needs, d_acc, ctx.grad_scaling)
        d_values = [(g.view(shp) if g is not None else None) for g, shp in zip(d_values, ctx.vshapes)]
        return (d_sigma.view(-1, 1), None, None, None, None, *d_values)


class Spec2RgbFn(torch.autograd.Function):
    """ColourSystem.forward (utils/spec_to_rgb.py:112-127)."""

    @staticmethod
    def forward(ctx, spec_t, M):
        s, m = _hip.f32c(spec_t), _hip.f32c(M)
        ctx.save_for_backward(s, m)
        return spec2rgb_fwd(s, m)

    @staticmethod
    def backward(ctx, d_rgb):
        s, m = ctx.saved_tensors
        return spec2rgb_bwd(s, m, _hip.f32c(d_rgb)), None


def accumulate_fwd(weights, values, packed_info):
    """out[r] = sum over the samples n of ray r of weights[n] * values[n] (no autograd)."""
    R = packed_info.shape[0]
    out = torch.empty((R, values.shape[1]), device=values.device, dtype=torch.float32)
    st = _hip.ValueStreams()
    st.n_streams, st.k[0], st.values[0], st.out[0] = 1, values.shape[1], values.data_ptr(), out.data_ptr()
    _hip.check(_hip.lib().umhs_accumulate_fwd(ptr(weights), ptr(packed_info), R, weights.shape[0], C.byref(st), _hip.stream()),
               "umhs_accumulate_fwd")
    return out


class AccumulateFn(torch.autograd.Function):
    """nerfacc.accumulate_along_rays with caller-provided weights (SpectralRenderer.forward stand-alone,
    umhs_renderer.py:28-30): out[r] = sum_n w[n] v[n].  Differentiable in both weights and values."""

    @staticmethod
    def forward(ctx, weights, values, packed_info):
        w, v = _hip.f32c(weights).view(-1), _hip.f32c(values)
        v = v.view(v.shape[-2], v.shape[-1])
        R = packed_info.shape[0]
        out = torch.empty((R, v.shape[1]), device=v.device, dtype=torch.float32)
        st = _hip.ValueStreams()
        st.n_streams, st.k[0], st.values[0], st.out[0] = 1, v.shape[1], v.data_ptr(), out.data_ptr()
        _hip.check(_hip.lib().umhs_accumulate_fwd(ptr(w), ptr(packed_info), R, w.shape[0], C.byref(st), _hip.stream()),
                   "umhs_accumulate_fwd")
        ctx.save_for_backward(w, v, packed_info)
        ctx.shapes = (weights.shape, values.shape)
        return out

    @staticmethod
    def backward(ctx, d_out):
        w, v, pinfo = ctx.saved_tensors
        d_out = _hip.f32c(d_out)
        d_w, d_v = torch.zeros_like(w), torch.empty_like(v)
        g = _hip.ValueGrads()
        g.n_streams, g.k[0], g.values[0], g.d_out[0], g.d_values[0] = 1, v.shape[1], v.data_ptr(), d_out.data_ptr(), d_v.data_ptr()
        _hip.check(_hip.lib().umhs_accumulate_bwd(ptr(w), ptr(pinfo), pinfo.shape[0], w.shape[0], C.byref(g), ptr(d_w),
                                                  _hip.stream()), "umhs_accumulate_bwd")
        return d_w.view(ctx.shapes[0]), d_v.view(ctx.shapes[1]), None


def tmid_minmax(starts, ends, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Encoded (min, max) of the sample mid-points over the batch; consumed by RayEpilogueFn only."""
    t0, t1 = _hip.f32c(starts).view(-1), _hip.f32c(ends).view(-1)
    mm = out if out is not None else torch.empty(2, device=t0.device, dtype=torch.float32)
    _hip.check(_hip.lib().umhs_tmid_minmax(ptr(t0), ptr(t1), t0.shape[0], ptr(mm), _hip.stream()), "umhs_tmid_minmax")
    return mm


class RayEpilogueFn(torch.autograd.Function):
    """One launch for the per-ray tail of UMHSModel.get_outputs (umhs_model.py:254-313): ColourSystem, depth clip,
    ClusterLookup(alpha) + argmax + class colours.  Returns (rgb, depth, seg_probs, seg_raw, seg_pred); gradient flows
    through rgb only (no loss of the reference uses the segmentation outputs)."""

    @staticmethod
    def forward(ctx, spectral, M, endmembers, accumulation, depth, tminmax, colors, alpha: float):
        ctx.set_materialize_grads(False)
        s, m = _hip.f32c(spectral), _hip.f32c(M)
        E = _hip.f32c(endmembers)
        rgb, dclip, probs, raw, pred = ray_epilogue_fwd(s, m, E, _hip.f32c(accumulation).view(-1), _hip.f32c(depth).view(-1), tminmax,
                                                        _hip.f32c(colors), alpha)
        ctx.save_for_backward(s, m)
        ctx.mark_non_differentiable(dclip, probs, raw, pred)
        return rgb, dclip, probs, raw, pred

    @staticmethod
    def backward(ctx, d_rgb, *_):
        s, m = ctx.saved_tensors
        d_spec = spec2rgb_bwd(s, m, _hip.f32c(d_rgb)) if d_rgb is not None else None
        return d_spec, None, None, None, None, None, None, None


class LossFn(torch.autograd.Function):
    """(w_spec * MSE(spectral, gt), w_rgb * MSE(rgb + bg*(1-acc), gt_rgb)) in one launch; backward in one launch
    (umhs_model.py:358-370).  rgb/acc/bg/gt_rgb may be None (method == "spectral")."""

    @staticmethod
    def forward(ctx, spectral, gt_spectral, rgb, accumulation, background, gt_rgb, w_spec: float, w_rgb: float):
        ctx.set_materialize_grads(False)
        c = lambda t: _hip.f32c(t) if t is not None else None
        s, g, r, bg, gr = c(spectral), c(gt_spectral), c(rgb), c(background), c(gt_rgb)
        a = c(accumulation).view(-1) if accumulation is not None else None
        losses = loss_fwd(s, g, r, a, bg, gr, w_spec, w_rgb)
        ctx.save_for_backward(*[t for t in (s, g, r, a, bg, gr) if t is not None])
        ctx.has = [t is not None for t in (s, g, r, a, bg, gr)]
        ctx.w = (float(w_spec), float(w_rgb))
        ctx.acc_shape = accumulation.shape if accumulation is not None else None
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_spec, g_rgb):
        it = iter(ctx.saved_tensors)
        s, g, r, a, bg, gr = [(next(it) if h else None) for h in ctx.has]
        R, B = s.shape
        zero = lambda: torch.zeros((), device=s.device, dtype=torch.float32)
        gl = torch.stack([g_spec if g_spec is not None else zero(), g_rgb if g_rgb is not None else zero()]).to(torch.float32)
        d_spec, d_rgb, d_acc = loss_bwd(s, g, r, a, bg, gr, ctx.w[0], ctx.w[1], gl)
        return d_spec, None, d_rgb, (d_acc.view(ctx.acc_shape) if d_acc is not None else None), None, None, None, None
