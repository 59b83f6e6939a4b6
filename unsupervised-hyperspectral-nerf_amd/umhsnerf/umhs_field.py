"""UMHSField -- spectral-unmixing radiance field, HIP implementation.

Mirror of the reference's ``umhsnerf/umhs_field.py`` (class ``UMHSField(NerfactoField)``, ``:28-329``): same
constructor arguments, ``get_density`` / ``get_outputs`` / ``forward`` / ``density_fn`` signatures, output keys
and output shapes, same state-dict key names.  The arithmetic (hash grid, MLPs, encodings, mixing) runs in
libumhs_hip.so through ``ops.FieldFn``; there is no torch or CPU implementation behind it.

All field parameters live in ONE flat fp32 buffer (``self.flat``) -- the unit the fused Adam step and the RCCL
gradient all-reduce work on; the reference's per-tensor names are views into it (``state_dict()`` /
``load_state_dict()`` speak the reference's names, so checkpoints interchange).
"""
from __future__ import annotations

import math
import os
from typing import Any, Dict, Literal, Optional, Tuple

import numpy as np
import torch
from torch import Tensor, nn

from . import ops
from ._ns_compat import FieldHeadNames, RaySamples

_REF_KEYS_DOC = "mlp_base.encoder.hash_table, mlp_base.mlp.layers.*, mlp_head.layers.*, feature_mlp.layers.*, mlp_directional.layers.*, endmembers"


class UMHSField(nn.Module):
    """UMHS field with spectral unmixing (hash grid -> density/embedding; heads -> abundances x endmembers)."""

    aabb: Tensor

    def __init__(
        self,
        aabb: Tensor,
        num_images: int,
        implementation: Literal["hip", "tcnn", "torch"] = "hip",
        num_layers_color: int = 3,
        hidden_dim_color: int = 64,
        wavelengths: int = 128,
        method: Literal["rgb", "spectral", "rgb+spectral"] = "rgb+spectral",
        num_classes: int = 4,
        feature_dim: int = 256,
        temperature: float = 0.5,
        converter=None,
        pred_dino: bool = False,
        pred_specular: bool = False,
        load_vca: bool = False,
        log2_hashmap_size: int = 19,
        max_res: int = 2048,
        spatial_distortion: Any = "linf",
        appearance_embedding_dim: int = 0,
        seed: Optional[int] = None,
        **kwargs,
    ) -> None:
        super().__init__()
        # "tcnn"/"torch" in existing scripts (scripts/hotdog.sh:7) select this HIP implementation too
        if method == "rgb":
            raise NotImplementedError("method='rgb' (umhs_field.py:280-294) is NerfactoField's colour head: umhs_field_rgb.UMHSRGBField "
                                      "(UMHSModel builds it); this class is the field of the two spectral methods")
        if pred_dino:
            raise NotImplementedError("pred_dino needs the reference's missing dino modules (SURVEY §2); out of scope")
        if num_layers_color != 3 or hidden_dim_color != 64 or max_res != 2048 or appearance_embedding_dim != 0:
            raise NotImplementedError("HIP kernels are built for the reference's fixed widths (3x64 heads, max_res 2048)")
        self.register_buffer("aabb", torch.as_tensor(aabb, dtype=torch.float32).reshape(2, 3))
        self._aabb_host = tuple(float(v) for v in self.aabb.flatten().tolist())  # read once: no per-step host sync
        self._spec_cache = None
        self.method, self.num_classes, self.wavelengths = method, num_classes, wavelengths
        self.feature_dim, self.pred_specular, self.pred_dino = feature_dim, pred_specular, pred_dino
        self.average_init_density, self.geo_feat_dim, self.appearance_embedding_dim = 1, ops.GEO_FEAT_DIM, 0
        self.temperature, self.converter, self.use_scalar = temperature, converter, True
        self.spatial_distortion = spatial_distortion
        self.layout = ops.FieldLayout(num_classes, wavelengths, pred_specular, log2_hashmap_size)
        self.register_buffer("scalings", ops.hash_scalings(ops.NUM_LEVELS, 16, max_res), persistent=False)  # a plain attribute upstream
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        flat = torch.zeros(self.layout.total)
        L = self.layout
        # nerfstudio defaults: hash table U(-1,1)*1e-3, nn.Linear init, endmembers randn or vca.npy (umhs_field.py:78-85)
        tab = L.view(flat, "mlp_base.encoder.hash_table")
        tab.copy_((torch.rand(tab.shape, generator=g) * 2 - 1) * 1e-3)
        for name, (_, shp) in L.entries.items():
            if name.endswith(".weight"):
                w = torch.empty(shp)
                nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=g)
                L.view(flat, name).copy_(w)
                bound = 1 / math.sqrt(shp[1])
                L.view(flat, name[:-6] + "bias").copy_((torch.rand(shp[0], generator=g) * 2 - 1) * bound)
        if load_vca and os.path.exists("vca.npy"):
            E = torch.tensor(np.load("vca.npy"), dtype=torch.float32)
        else:
            E = torch.randn(num_classes, wavelengths, generator=g)
        L.view(flat, "endmembers").copy_(E)
        self.flat = nn.Parameter(flat)
        # coarse hash levels use a small, fixed subset of their 2^T slots: the optimizer and the gradient all-reduce skip the rest
        from .parallel import live_hash_rows

        n_sparse, rows = live_hash_rows(self.scalings, log2_hashmap_size)
        self.register_buffer("live_rows", rows, persistent=False)
        self._sparse_end = n_sparse * (1 << log2_hashmap_size) * ops.FEATURES_PER_LEVEL
        self._cache: Optional[Tuple] = None
        self._enc_capture = None
        self.use_grad_sink = False  # UMHSPipeline turns this on: backward writes param.grad in place (+ early all-reduce)
        self._grad_sink = None

    # ---- parameter views under the reference's names ------------------------------------------------
    @property
    def endmembers(self) -> Tensor:
        return self.layout.view(self.flat, "endmembers")

    def named_views(self) -> Dict[str, Tensor]:
        return {k: self.layout.view(self.flat.detach(), k) for k in self.layout.entries}

    # Checkpoints speak the reference's key names (``_model.field.mlp_base.encoder.hash_table`` ...): the flat buffer is saved as
    # its per-tensor views and loaded back through them, at every level of the module tree (nn.Module recurses through
    # ``_save_to_state_dict`` / ``_load_from_state_dict``, so ``UMHSModel`` / ``UMHSPipeline`` checkpoints work too).
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k in self.layout.entries:
            v = self.layout.view(self.flat if keep_vars else self.flat.detach(), k)
            destination[prefix + k] = v
        destination[prefix + "aabb"] = self.aabb if keep_vars else self.aabb.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        with torch.no_grad():
            for k, (_, shp) in self.layout.entries.items():
                src = state_dict.get(prefix + k)
                if src is None:
                    missing_keys.append(prefix + k)  # reported even when not strict (load_state_dict's return value lists them)
                    continue
                if tuple(src.shape) != tuple(shp):
                    error_msgs.append(f"size mismatch for {prefix + k}: checkpoint {tuple(src.shape)} vs field {tuple(shp)}")
                    continue
                self.layout.view(self.flat, k).copy_(src.to(self.flat.device, torch.float32))
            if prefix + "aabb" in state_dict:
                self.aabb.copy_(state_dict[prefix + "aabb"].to(self.aabb.device))
                self._aabb_host = tuple(float(v) for v in self.aabb.flatten().tolist())
                self._spec_cache = None  # (a dict without "aabb" keeps the constructed box: parameter-only dicts load too)
        mine = {prefix + k for k in self.layout.entries} | {prefix + "aabb"}
        kids = tuple(prefix + name + "." for name, m in self._modules.items() if m is not None)
        for key in state_dict:
            if key.startswith(prefix) and key not in mine and not key.startswith(kids):
                unexpected_keys.append(key)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Called on the field directly, a dict of just the reference-named parameters loads too (``strict`` then concerns those
        names only: buffers -- ``aabb``, ``converter.transform_matrix`` -- keep their constructed values when absent)."""
        res = super().load_state_dict(state_dict, strict=False)
        if strict:
            missing = [k for k in res.missing_keys if k in self.layout.entries]  # (prefix is empty here)
            if missing or res.unexpected_keys:
                raise RuntimeError(f"Error(s) in loading state_dict for UMHSField: missing {missing}, unexpected {list(res.unexpected_keys)}; "
                                   f"expected {_REF_KEYS_DOC}")
        return res

    def _spec(self) -> ops.FieldSpec:
        if self._sparse_end:
            self.flat._umhs_live_rows = (self.live_rows, self._sparse_end)
        c = self._spec_cache
        if c is None or c.scalings.device != self.scalings.device or c.temperature != float(self.temperature):
            c = ops.FieldSpec(self.layout, float(self.temperature), self.spatial_distortion is not None, self._aabb_host, self.scalings)
            self._spec_cache = c
        if self.use_grad_sink and (self._grad_sink is None or self._grad_sink.param is not self.flat):
            from .parallel import FlatGradSink

            self._grad_sink = FlatGradSink(self.flat)
            self._grad_sink.set_sparse_levels(self.scalings, self.layout.log2_hashmap_size)
            self.flat._umhs_grad_sink = self._grad_sink
        c.grad_sink = self._grad_sink if self.use_grad_sink else None
        return c

    # ---- reference API ------------------------------------------------------------------------------
    def _run(self, ray_samples: RaySamples):
        fr = ray_samples.frustums
        n = int(np.prod(fr.origins.shape[:-1]))
        outs = ops.FieldFn.apply(self.flat, fr.origins.reshape(n, 3), fr.directions.reshape(n, 3), fr.starts.reshape(n, 1),
                                 fr.ends.reshape(n, 1), self._spec())
        self._cache = (ray_samples, outs)
        return outs

    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
        """(density [*,1], base_mlp_out [*,15]) -- umhs_field.py:300-329.  The fused kernel evaluates the heads in the
        same launch; ``get_outputs`` on the same ray_samples picks them up without recomputation."""
        density, emb = self._run(ray_samples)[:2]
        shp = ray_samples.frustums.origins.shape[:-1]
        return density.view(*shp, 1), emb.view(*shp, self.geo_feat_dim)

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None) -> Dict[Any, Tensor]:
        assert density_embedding is not None
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        if self._cache is None or self._cache[0] is not ray_samples:
            raise NotImplementedError("get_outputs() must follow get_density() on the same ray_samples (as Field.forward "
                                      "does): the HIP kernel fuses both and a foreign density_embedding is not supported")
        _, _, spectral, spectral2, specular, abund = self._cache[1]
        self._cache = None
        n = spectral.shape[0]
        out: Dict[Any, Tensor] = {}
        if self.pred_specular:  # reference shape quirks, umhs_field.py:253-261
            out["spectral"], out["spectral2"] = spectral.view(1, n, -1), spectral2
            out["specular"] = specular.view(1, n, -1)
            out["abundances"] = abund.view(1, n, -1)
        else:
            out["spectral"] = spectral
            out["abundances"] = abund.view(1, n, -1)
        return out

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False) -> Dict[Any, Tensor]:
        density, emb = self.get_density(ray_samples)
        out = self.get_outputs(ray_samples, density_embedding=emb)
        out[FieldHeadNames.DENSITY] = density
        return out

    def density_fn(self, positions: Tensor, times: Optional[Tensor] = None) -> Tensor:
        """Density at raw positions [*,3] (occupancy grid / sampler, umhs_model.py:208,553); no-grad path."""
        shp = positions.shape[:-1]
        keep = self._enc_capture  # a dict while the model's sampler wants the hash features of its candidates back, else None
        with torch.no_grad():
            sigma, _ = ops.DensityFn.apply(self.flat, positions.reshape(-1, 3), self._spec(), keep, False)
        return sigma.view(*shp, 1)
