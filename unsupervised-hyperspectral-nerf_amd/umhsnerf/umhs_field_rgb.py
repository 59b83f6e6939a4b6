"""UMHSField with ``method="rgb"`` -- the reference's default method and its scripts/rgb.sh (BASELINE configs[0], "plumbing").

``umhs_field.py:280-294``: with ``method="rgb"`` the reference's field IS nerfstudio's NerfactoField: ``mlp_base`` (hash grid ->
64 -> 1 + 15) and NerfactoField's own colour head ``mlp_head(cat[SH16(d), emb15]) -> 64 -> 64 -> 3`` with a Sigmoid output
[upstream-recalled: nerfstudio 1.1.5 ``NerfactoField.__init__``; the appearance embedding has width 0 with the reference's default
``use_appearance_embedding=True``, umhs_model.py:181].  None of the spectral heads exists.

This is NOT the hot path (SURVEY section 8 scopes the spectral methods).  Positions / contraction, the hash-grid gather and its
atomics-free backward, the sampler, the per-ray compositing and the fused Adam step are the HIP operators of the spectral path; the
two small MLPs (32 -> 64 -> 16 with trunc_exp, and SH16 | emb15 -> 64 -> 64 -> 3 with the sigmoid) are the fp32-MFMA kernels of
``csrc/umhs_rgb.hip`` (``ops.RgbBaseFn`` / ``ops.RgbHeadFn``): since round 4 no library GEMM and no torch elementwise kernel is left
in this field.  No CPU fallback: the HIP library is required exactly as for the spectral methods.

Parameters live in one flat fp32 buffer under the reference's state-dict key names, as in :class:`UMHSField`.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor, nn

from . import ops
from ._ns_compat import FieldHeadNames, RaySamples


class HashEncodeFn(torch.autograd.Function):
    """``mlp_base.encoder``: HIP hash-grid gather [N,3] -> [N,32] (sample-major) and its atomics-free scatter backward."""

    @staticmethod
    def forward(ctx, table, pos01, scalings, log2_T: int):
        enc = ops.hashgrid_fwd(pos01, table.detach(), scalings, log2_T, level_major=False)
        ctx.save_for_backward(pos01, scalings)
        ctx.log2_T, ctx.table_shape = log2_T, tuple(table.shape)
        return enc

    @staticmethod
    def backward(ctx, d_enc):
        pos01, scalings = ctx.saved_tensors
        d_table = torch.zeros(ctx.table_shape, device=d_enc.device, dtype=torch.float32)
        ops.hashgrid_bwd(pos01, d_enc.contiguous().float(), scalings, ctx.log2_T, d_table, level_major=False)
        return d_table, None, None, None


class RGBLayout:
    """Flat layout of the rgb field (names = the reference's state-dict keys for ``method="rgb"``)."""

    def __init__(self, log2_hashmap_size: int = 19):
        self.log2_hashmap_size = log2_hashmap_size
        T = 1 << log2_hashmap_size
        shapes = [("mlp_base.encoder.hash_table", (ops.NUM_LEVELS * T, ops.FEATURES_PER_LEVEL))]
        for prefix, dims in (("mlp_base.mlp", [ops.NUM_LEVELS * ops.FEATURES_PER_LEVEL, ops.HIDDEN, 1 + ops.GEO_FEAT_DIM]),
                             ("mlp_head", [16 + ops.GEO_FEAT_DIM, ops.HIDDEN, ops.HIDDEN, 3])):
            for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
                shapes += [(f"{prefix}.layers.{i}.weight", (b, a)), (f"{prefix}.layers.{i}.bias", (b,))]
        self.entries: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, shp in shapes:
            self.entries[name] = (off, shp)
            off += (int(np.prod(shp)) + 3) & ~3
        self.total = off

    def view(self, flat: Tensor, name: str) -> Tensor:
        off, shp = self.entries[name]
        return flat[off:off + int(np.prod(shp))].view(shp)


class UMHSRGBField(nn.Module):
    """``UMHSField(method="rgb")``: density from the hash grid + mlp_base, colour from NerfactoField's mlp_head."""

    aabb: Tensor

    def __init__(self, aabb: Tensor, num_images: int = 1, log2_hashmap_size: int = 19, max_res: int = 2048, spatial_distortion: Any = "linf",
                 appearance_embedding_dim: int = 0, seed: Optional[int] = None, **kwargs) -> None:
        super().__init__()
        if appearance_embedding_dim != 0 or max_res != 2048:
            raise NotImplementedError("the rgb field is built for the reference's defaults (no appearance embedding, max_res 2048)")
        self.method, self.geo_feat_dim, self.appearance_embedding_dim = "rgb", ops.GEO_FEAT_DIM, 0
        self.register_buffer("aabb", torch.as_tensor(aabb, dtype=torch.float32).reshape(2, 3))
        self._aabb_host = tuple(float(v) for v in self.aabb.flatten().tolist())
        self.spatial_distortion = spatial_distortion
        self.layout = RGBLayout(log2_hashmap_size)
        self.register_buffer("scalings", ops.hash_scalings(ops.NUM_LEVELS, 16, max_res), persistent=False)
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        flat, L = torch.zeros(self.layout.total), self.layout
        tab = L.view(flat, "mlp_base.encoder.hash_table")
        tab.copy_((torch.rand(tab.shape, generator=g) * 2 - 1) * 1e-3)
        for name, (_, shp) in L.entries.items():
            if name.endswith(".weight"):
                w = torch.empty(shp)
                nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=g)
                L.view(flat, name).copy_(w)
                L.view(flat, name[:-6] + "bias").copy_((torch.rand(shp[0], generator=g) * 2 - 1) / math.sqrt(shp[1]))
        self.flat = nn.Parameter(flat)
        self.use_grad_sink = False  # (the spectral field's in-place gradient sink: this field's gradient comes from autograd)
        self._enc_capture = None

    # ---- checkpoints under the reference's key names (as UMHSField) ----------------------------------------------------------
    def named_views(self) -> Dict[str, Tensor]:
        return {k: self.layout.view(self.flat.detach(), k) for k in self.layout.entries}

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k in self.layout.entries:
            destination[prefix + k] = self.layout.view(self.flat if keep_vars else self.flat.detach(), k)
        destination[prefix + "aabb"] = self.aabb if keep_vars else self.aabb.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        with torch.no_grad():
            for k, (_, shp) in self.layout.entries.items():
                src = state_dict.get(prefix + k)
                if src is None:
                    missing_keys.append(prefix + k)
                elif tuple(src.shape) != tuple(shp):
                    error_msgs.append(f"size mismatch for {prefix + k}: checkpoint {tuple(src.shape)} vs field {tuple(shp)}")
                else:
                    self.layout.view(self.flat, k).copy_(src.to(self.flat.device, torch.float32))
            if prefix + "aabb" in state_dict:
                self.aabb.copy_(state_dict[prefix + "aabb"].to(self.aabb.device))
                self._aabb_host = tuple(float(v) for v in self.aabb.flatten().tolist())
        mine = {prefix + k for k in self.layout.entries} | {prefix + "aabb"}
        for key in state_dict:
            if key.startswith(prefix) and key not in mine:
                unexpected_keys.append(key)

    def _spec(self):
        """What UMHSPipeline asks every field for: this one has no gradient sink (``param.grad`` comes from autograd, the exchange
        between ranks is the optimizer's one all-reduce of the flat gradient)."""
        return SimpleNamespace(grad_sink=SimpleNamespace(defer_reduce=False, accumulating=False))

    # ---- arithmetic ------------------------------------------------------------------------------------------------------------
    def _geom(self):
        return SimpleNamespace(aabb=self._aabb_host, contraction=self.spatial_distortion is not None)

    def _weights(self, prefix: str, n_layers: int, flat: Optional[Tensor] = None):
        flat = self.flat if flat is None else flat
        out = []
        for i in range(n_layers):
            out += [self.layout.view(flat, f"{prefix}.layers.{i}.weight"), self.layout.view(flat, f"{prefix}.layers.{i}.bias")]
        return out

    def _density_from_pos01(self, pos01: Tensor, sel: Tensor) -> Tuple[Tensor, Tensor]:
        """mlp_base: hash-grid gather (HashEncodeFn) -> 64 -> 16 with trunc_exp on output 0 and the selector product, all in
        ``umhs_rgb_base_fwd`` / ``_bwd`` (average_init_density = 1, umhs_field.py:57)."""
        table = self.layout.view(self.flat, "mlp_base.encoder.hash_table")
        enc = HashEncodeFn.apply(table, pos01, self.scalings, self.layout.log2_hashmap_size)
        density, emb = ops.RgbBaseFn.apply(enc, sel, *self._weights("mlp_base.mlp", 2))
        return density[:, None], emb

    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
        fr = ray_samples.frustums
        shp = fr.origins.shape[:-1]
        n = int(np.prod(shp))
        _, pos01, sel = ops.positions_fwd(fr.origins.reshape(n, 3).float().contiguous(), fr.directions.reshape(n, 3).float().contiguous(),
                                          fr.starts.reshape(n).float().contiguous(), fr.ends.reshape(n).float().contiguous(), self._geom())
        density, emb = self._density_from_pos01(pos01, sel)
        return density.view(*shp, 1), emb.view(*shp, self.geo_feat_dim)

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None) -> Dict[Any, Tensor]:
        assert density_embedding is not None
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        dirs = ray_samples.frustums.directions
        shp = dirs.shape[:-1]
        # get_normalized_directions + SHEncoding(levels=4) + cat + mlp_head + Sigmoid: one kernel (umhs_rgb_head_fwd)
        rgb = ops.RgbHeadFn.apply(dirs.reshape(-1, 3).float(), density_embedding.reshape(-1, self.geo_feat_dim).float(),
                                  *self._weights("mlp_head", 3)).view(*shp, 3)
        return {FieldHeadNames.RGB: rgb}

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False) -> Dict[Any, Tensor]:
        density, emb = self.get_density(ray_samples)
        out = self.get_outputs(ray_samples, density_embedding=emb)
        out[FieldHeadNames.DENSITY] = density
        return out

    def density_fn(self, positions: Tensor, times: Optional[Tensor] = None) -> Tensor:
        """Density at raw positions [*,3] (occupancy grid / sampler); no-grad path."""
        shp = positions.shape[:-1]
        with torch.no_grad():
            p = positions.reshape(-1, 3).float().contiguous()
            _, pos01, sel = ops.positions_fwd(None, None, None, None, self._geom(), world_pos_in=p)
            flat = self.flat.detach()
            enc = ops.hashgrid_fwd(pos01, self.layout.view(flat, "mlp_base.encoder.hash_table"), self.scalings, self.layout.log2_hashmap_size,
                                   level_major=False)
            density, _, _ = ops.rgb_base_fwd(enc, sel, *self._weights("mlp_base.mlp", 2, flat), want_emb=False)
        return density.view(*shp, 1)
