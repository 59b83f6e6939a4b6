"""UMHSModel -- mirror of the reference's ``umhsnerf/umhs_model.py`` (``UMHSConfig`` :61-119, ``UMHSModel`` :122-620)
driving the HIP hot path: field -> packed transmittance/compositing over all bands -> spectrum->sRGB -> losses.

Output keys, shapes, loss weights and callbacks follow the reference.  What differs by design:
  * the four per-stream renderer calls of ``get_outputs`` (:270-304) are ONE compositing launch;
  * ``scale_gradients_by_distance_squared`` (:241-242) is applied inside the compositing backward;
  * nerfacc's CUDA occupancy marcher does not exist on ROCm: ``get_outputs`` samples through the HIP marcher
    (``sampler.OccGridEstimator`` / ``VolumetricSampler``); ``get_outputs_from_samples`` takes any packed samples (this is
    what the benchmark and the parity tests feed).
"""
from __future__ import annotations

import os

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Literal, Optional, Tuple, Type, Union

import numpy as np
import torch
from torch import Tensor, nn

from . import _hip, ops
from ._ns_compat import FieldHeadNames, ModelBase, ModelConfigBase, RayBundle, RaySamples, TrainingCallback, TrainingCallbackLocation
from .optim import UMHSAdam
from .sampler import OccGridEstimator, VolumetricSampler
from .umhs_field import UMHSField
from .umhs_renderer import SpectralRenderer
from .utils.clusterprobe import ClusterLookup
from .umhs_field_rgb import UMHSRGBField
from .utils.spec_to_rgb import ColourSystem

CLASS_COLORS = torch.tensor([
    [0.49, 0.29, 0.95], [0.29, 0.95, 0.30], [0.95, 0.29, 0.47], [0.29, 0.66, 0.95], [0.86, 0.95, 0.29],
    [0.85, 0.29, 0.95], [0.29, 0.95, 0.66], [0.95, 0.46, 0.29], [0.29, 0.30, 0.95], [0.50, 0.95, 0.29],
    [0.95, 0.29, 0.69], [0.29, 0.88, 0.95], [0.95, 0.82, 0.29], [0.63, 0.29, 0.95], [0.29, 0.95, 0.43],
])  # umhs_model.py:146-162


@dataclass
class UMHSConfig(ModelConfigBase):
    """``UMHSConfig(InstantNGPModelConfig)``, umhs_model.py:61-119: same names, same defaults (``method="rgb"``, :108, included: the
    reference's default method is NerfactoField's own colour head, served by umhs_field_rgb.py; every script of the reference but
    scripts/rgb.sh passes ``--pipeline.model.method rgb+spectral``, the path this package exists for).
    ``implementation`` keeps the reference's default ``"torch"`` (:104) and its ``"tcnn"`` (scripts/hotdog.sh:7): all values select the
    HIP kernels, whose arithmetic is the torch path's.  A nerfstudio ``ModelConfig`` when nerfstudio is importable
    -- not an ``InstantNGPModelConfig``, whose model would build tcnn / nerfacc modules first."""

    _target: Type = field(default_factory=lambda: UMHSModel)
    enable_collider: bool = False
    collider_params: Optional[Dict[str, float]] = field(default_factory=lambda: {"near_plane": 2.0, "far_plane": 6.0})
    grid_resolution: Union[int, List[int]] = 128
    grid_levels: int = 4
    max_res: int = 2048
    log2_hashmap_size: int = 19
    alpha_thre: float = 0.01
    cone_angle: float = 0.004
    render_step_size: Optional[float] = None
    near_plane: float = 0.05
    far_plane: float = 1e3
    use_gradient_scaling: bool = True
    use_appearance_embedding: bool = True
    # "last_sample" is what scripts/spectral.sh:6 passes; for the loss blend it is black (RGBRenderer.blend_background_for_loss_computation,
    # umhs_renderer.py:108-109), and this model's rgb is converter(spectral), never a renderer's last-sample composite
    background_color: Literal["random", "last_sample", "black", "white"] = "random"
    disable_scene_contraction: bool = False
    implementation: Literal["hip", "tcnn", "torch"] = "torch"
    method: Literal["rgb", "spectral", "rgb+spectral"] = "rgb"
    rgb_loss_weight: float = 1.0
    spectral_loss_weight: float = 1.0  # unused by the reference's loss too (hard-coded 5, umhs_model.py:369)
    temperature: float = 0.2
    pred_dino: bool = False
    pred_specular: bool = False
    load_vca: bool = False
    eval_num_rays_per_chunk: int = 512
    per_band_outputs: bool = True  # wv_i / residual_i / abundances_i views (umhs_model.py:273-304)


class BandOutputs(dict):
    """Output dict whose per-band entries (``wv_i`` / ``residual_i`` / ``abundances_i``) are column views made on first access."""

    bands: Dict[str, Tensor] = {}

    def __missing__(self, key):
        stem, _, i = str(key).rpartition("_")
        src = self.bands.get(stem)
        if src is None or not i.isdigit() or int(i) >= src.shape[-1]:
            raise KeyError(key)
        v = self[key] = src[..., int(i)]
        return v

    def __contains__(self, key):
        if dict.__contains__(self, key):
            return True
        stem, _, i = str(key).rpartition("_")
        src = self.bands.get(stem)
        return src is not None and i.isdigit() and int(i) < src.shape[-1]

    def materialize(self) -> "BandOutputs":
        for stem, src in self.bands.items():
            for i in range(src.shape[-1]):
                self[f"{stem}_{i}"]
        return self

    # anything that enumerates the dict sees the full key set of the reference
    def __iter__(self):
        return dict.__iter__(self.materialize())

    def __len__(self):
        return dict.__len__(self.materialize())

    def keys(self):
        return dict.keys(self.materialize())

    def items(self):
        return dict.items(self.materialize())

    def values(self):
        return dict.values(self.materialize())


class LazyMetrics(dict):
    """Metrics dict whose entries are computed on first access.  nerfstudio's trainer reads the training metrics every
    ``steps_per_log`` steps only; the ~15 small reduction kernels behind them (1.4 % of a sampler-driven step) then run only when
    somebody looks.  The thunks hold the step's outputs / batch tensors, which nothing modifies in place afterwards."""

    def __init__(self, thunks: Dict[str, Callable[[], Tensor]]):
        super().__init__()
        self._thunks = dict(thunks)

    def __missing__(self, key):
        fn = self._thunks.pop(key, None)
        if fn is None:
            raise KeyError(key)
        v = self[key] = fn()
        return v

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._thunks

    def get(self, key, default=None):
        return self[key] if key in self else default

    def materialize(self) -> "LazyMetrics":
        for k in list(self._thunks):
            self[k]
        return self

    def __iter__(self):
        return dict.__iter__(self.materialize())

    def __len__(self):
        return dict.__len__(self.materialize())

    def keys(self):
        return dict.keys(self.materialize())

    def items(self):
        return dict.items(self.materialize())

    def values(self):
        return dict.values(self.materialize())


class UMHSModel(ModelBase):
    """UMHS model (``UMHSModel(NGPModel)``); built by ``config.setup(scene_box=, num_train_data=, metadata=, grad_scaler=,
    num_classes=, wavelengths=)`` exactly as the reference pipeline does (umhs_pipeline.py:98-105)."""

    def __init__(self, config: UMHSConfig, scene_box=None, num_train_data: int = 1, metadata: Optional[Dict] = None,
                 seed: Optional[int] = None, grad_scaler=None, num_classes: Optional[int] = None, wavelengths=None, **kwargs):
        nn.Module.__init__(self)  # nerfstudio's Model.__init__ would call populate_modules() before the fields below exist
        self.config = config
        self.scene_box, self.render_aabb, self.collider, self.callbacks = scene_box, None, None, None
        aabb = getattr(scene_box, "aabb", scene_box)
        self.scene_aabb_t = torch.as_tensor(aabb if aabb is not None else [[-1, -1, -1], [1, 1, 1]], dtype=torch.float32).reshape(2, 3)
        self.num_train_data = num_train_data
        self.grad_scaler = grad_scaler  # fp32 hot path: carried for interface parity, never used
        # the reference reads self.kwargs["metadata"]["wavelengths" | "num_classes"] (umhs_model.py:171-172,188-189); the two
        # explicit kwargs of its pipeline (:103-104) fill in what a metadata dict lacks
        self.kwargs = dict(metadata or {})
        if wavelengths is not None:
            self.kwargs.setdefault("wavelengths", wavelengths)
        if num_classes is not None:
            self.kwargs.setdefault("num_classes", num_classes)
        if self.kwargs.get("wavelengths") is None or "num_classes" not in self.kwargs:
            raise KeyError('metadata must carry "wavelengths" and "num_classes" (umhs_model.py:171-172,188-189)')
        self._seed = seed
        self.populate_modules()

    def populate_modules(self):
        c = self.config
        wl = self.kwargs["wavelengths"]
        self.step = 0
        self.register_buffer("class_colors", CLASS_COLORS.clone())
        if "spectral" in c.method:
            self.renderer_spectral = SpectralRenderer()
        self.converter = ColourSystem(bands=wl, cs="sRGB")
        if not c.use_appearance_embedding:
            # umhs_model.py:181: the reference's flag is inverted -- False is what switches its 32-d per-image embedding ON (no script
            # does); that input of the rgb / specular heads is not built here, and ignoring the flag would silently train another model
            raise NotImplementedError("use_appearance_embedding=False (= the reference's 32-d appearance embedding, umhs_model.py:181) is not supported")
        if c.method == "rgb":  # the reference's default method = NerfactoField's own colour head (umhs_field.py:280-294): umhs_field_rgb.py
            self.field = UMHSRGBField(aabb=self.scene_aabb_t, num_images=self.num_train_data, log2_hashmap_size=c.log2_hashmap_size,
                                      max_res=c.max_res, spatial_distortion=None if c.disable_scene_contraction else "linf",
                                      appearance_embedding_dim=0, seed=self._seed)
        else:
            self.field = self._spectral_field(c, wl)
        self.scene_aabb = nn.Parameter(self.scene_aabb_t.flatten(), requires_grad=False)
        if c.render_step_size is None:
            c.render_step_size = float(((self.scene_aabb_t[1] - self.scene_aabb_t[0]) ** 2).sum().sqrt() / 1000)
        # umhs_model.py:201-209: nerfacc.OccGridEstimator(roi_aabb, resolution, levels) + VolumetricSampler(grid, density_fn)
        self.occupancy_grid = OccGridEstimator(self.scene_aabb_t.flatten(), resolution=int(c.grid_resolution), levels=c.grid_levels)
        self.sampler = VolumetricSampler(self.occupancy_grid, density_fn=self.field.density_fn)
        self.cluster_probe = ClusterLookup(len(wl), self.kwargs["num_classes"])
        self.background_color = c.background_color

    def _spectral_field(self, c, wl) -> UMHSField:
        return UMHSField(
            aabb=self.scene_aabb_t, num_images=self.num_train_data, implementation=c.implementation,
            log2_hashmap_size=c.log2_hashmap_size, max_res=c.max_res,
            spatial_distortion=None if c.disable_scene_contraction else "linf",
            appearance_embedding_dim=0,  # the reference's inverted flag yields 0 with the default config (:181)
            method=c.method, wavelengths=len(wl) if "spectral" in c.method else 0, num_classes=self.kwargs["num_classes"],
            temperature=c.temperature, converter=self.converter, pred_dino=c.pred_dino, pred_specular=c.pred_specular,
            load_vca=c.load_vca, seed=self._seed,
        )

    @property
    def device(self):
        return self.field.flat.device

    def label_to_rgb(self, labels: Tensor) -> Tensor:
        return self.class_colors[labels.long().squeeze(-1)]

    # ---- optimisation surface --------------------------------------------------------------------
    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        return {"fields": [self.field.flat]}

    def make_optimizer(self, lr: float = 2e-2, eps: float = 1e-15, lr_final: Optional[float] = 1e-5, max_steps: int = 30000) -> UMHSAdam:
        L = self.field.layout
        off, shp = L.entries.get("endmembers", (0, ()))  # (method="rgb" has no endmembers: nothing to clamp)
        return UMHSAdam(self.get_param_groups()["fields"], lr=lr, eps=eps, clamp_range=(off, off + (int(np.prod(shp)) if shp else 0)),
                        lr_final=lr_final, max_steps=max_steps)

    def clamp_endmembers(self, step: int = 0) -> None:
        """AFTER_TRAIN_ITERATION callback of the reference (umhs_model.py:568-572); UMHSAdam already fuses it."""
        with torch.no_grad():
            self.field.endmembers[:] = self.field.endmembers.clamp(0, 1)

    def update_occupancy_grid(self, step: int) -> None:
        """BEFORE_TRAIN_ITERATION callback of the reference (umhs_model.py:549-554)."""
        self.step = step
        self.occupancy_grid.update_every_n_steps(step=step, occ_eval_fn=lambda x: self.field.density_fn(x) * self.config.render_step_size)

    def get_training_callbacks(self, training_callback_attributes=None) -> List[TrainingCallback]:
        """umhs_model.py:542-591: clamp_endmembers AFTER every iteration (spectral methods), update_occupancy_grid BEFORE."""
        callbacks = []
        if self.config.method != "rgb":
            callbacks.append(TrainingCallback(where_to_run=[TrainingCallbackLocation.AFTER_TRAIN_ITERATION], update_every_num_iters=1,
                                              func=self.clamp_endmembers))
        callbacks.append(TrainingCallback(where_to_run=[TrainingCallbackLocation.BEFORE_TRAIN_ITERATION], update_every_num_iters=1,
                                          func=self.update_occupancy_grid))
        return callbacks

    def update_to_step(self, step: int) -> None:
        """nerfstudio Model.update_to_step (called by load_pipeline, umhs_pipeline.py:167): nothing here depends on the step."""
        self.step = step

    # ---- forward -----------------------------------------------------------------------------------
    def get_outputs(self, ray_bundle: RayBundle) -> Dict[str, Tensor]:
        ray_samples, ray_indices = self.sample(ray_bundle)
        return self.get_outputs_from_samples(ray_samples, ray_indices, len(ray_bundle))

    def sample(self, ray_bundle: RayBundle):
        """The sampler call of umhs_model.py:229-237 (no-grad): packed ray samples + ray_indices."""
        c = self.config
        # The training forward gathers the survivors' hash features out of the encoding the sampler's density query produced for every
        # marched candidate (UMHS_REUSE_ENC=0: hashes them again; 0.07 ms per step slower at 3.5 M candidates / 0.9 M survivors).
        # Gradient-free rendering reuses them as well -- an eval image keeps nearly every marched candidate (21.1 -> 20.9 ms per 256x256
        # image, the heads kernel is what its time is made of).
        reuse = isinstance(self.sampler, VolumetricSampler) and os.environ.get("UMHS_REUSE_ENC", "1") != "0" and (
            self.training or (not torch.is_grad_enabled() and os.environ.get("UMHS_RENDER_PER_RAY", "1") != "0"))
        self.field._enc_capture = {} if reuse else None
        try:
            ray_samples, ray_indices = self._sample(ray_bundle)
            cap = self.field._enc_capture
        finally:
            self.field._enc_capture = None
        keep = getattr(self.sampler.occupancy_grid, "last_keep_index", None) if reuse else None
        if reuse and keep is not None and "enc" in cap and keep.numel() == ray_indices.numel():
            # the sampler's density query encoded every marched candidate: the survivors' features are a row gather away
            ray_samples.metadata = {**(getattr(ray_samples, "metadata", None) or {}), "umhs_enc": (cap["enc"], keep)}
        pinfo = getattr(self.sampler, "last_packed_info", None)
        if pinfo is not None and pinfo.shape[0] == len(ray_bundle):  # (start, count) of every ray: the sampler has it already
            ray_samples.metadata = {**(getattr(ray_samples, "metadata", None) or {}), "umhs_packed_info": pinfo}
        return ray_samples, ray_indices

    def prefetch_sample(self, ray_bundle: RayBundle) -> bool:
        """Start the occupancy-grid march of a coming ``sample(ray_bundle)`` on the current stream (it depends on the rays and
        the grid only).  The trainer calls this one step ahead on a side stream; ``sample`` picks the result up."""
        c = self.config
        if not isinstance(self.sampler, VolumetricSampler):
            return False
        with torch.no_grad():
            self.sampler.prefetch(ray_bundle=ray_bundle, near_plane=c.near_plane, far_plane=c.far_plane,
                                  render_step_size=c.render_step_size, alpha_thre=c.alpha_thre, cone_angle=c.cone_angle)
        return True

    def occupancy_update_due(self, step: int) -> bool:
        """Will ``update_occupancy_grid(step)`` rewrite the grid (OccGridEstimator.update_every_n_steps, n=16)?"""
        return step % 16 == 0

    def _sample(self, ray_bundle: RayBundle):
        c = self.config
        with torch.no_grad():
            ray_samples, ray_indices = self.sampler(ray_bundle=ray_bundle, near_plane=c.near_plane, far_plane=c.far_plane,
                                                    render_step_size=c.render_step_size, alpha_thre=c.alpha_thre,
                                                    cone_angle=c.cone_angle)
        return ray_samples, ray_indices

    def forward(self, ray_bundle: RayBundle) -> Dict[str, Tensor]:
        return self.get_outputs(ray_bundle)

    def get_outputs_from_samples(self, ray_samples: RaySamples, ray_indices: Tensor, num_rays: int,
                                 packed_info: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """Body of ``get_outputs`` after the sampler, umhs_model.py:239-327."""
        c = self.config
        fr = ray_samples.frustums
        if packed_info is None:
            packed_info = (getattr(ray_samples, "metadata", None) or {}).get("umhs_packed_info")
        if packed_info is None:
            packed_info = ops.pack_info(ray_indices, num_rays)
        if (not torch.is_grad_enabled() and c.method != "rgb" and fr.origins.numel() > 0
                and os.environ.get("UMHS_RENDER_PER_RAY", "1") != "0" and ops.field_heads_fwd_supported(self.field._spec())):
            return self._render_outputs_from_samples(ray_samples, ray_indices, num_rays, packed_info)
        if c.method == "rgb":
            return self._rgb_outputs_from_samples(ray_samples, packed_info)
        fo = self.field(ray_samples)
        values = [fo["spectral"]]
        if c.pred_specular:
            values += [fo["spectral2"].detach(), fo["specular"]]  # spectral2 carries no loss in the reference (:373-374)
        values.append(fo["abundances"])
        weights, accumulation, depth, *comp = ops.CompositeFn.apply(fo[FieldHeadNames.DENSITY], fr.starts, fr.ends, packed_info,
                                                                     bool(c.use_gradient_scaling), *values)
        mm = ops.tmid_minmax(fr.starts, fr.ends)  # DepthRenderer clips to the batch-wide [min, max] of t_mid
        spectral = comp[0]
        spec_for_rgb = spectral.detach() if c.method == "spectral" else spectral  # no_grad pseudo-rgb (:289-293)
        rgb, depth_c, seg_probs, seg_raw, seg_pred = ops.RayEpilogueFn.apply(
            spec_for_rgb, self.converter.transform_matrix, self.field.endmembers.detach(), accumulation, depth, mm,
            self.class_colors, 0.2)
        return self._assemble_outputs(accumulation, depth_c, comp, rgb, packed_info, seg_probs, seg_raw, seg_pred, weights)

    def _rgb_outputs_from_samples(self, ray_samples: RaySamples, packed_info: Tensor) -> Dict[str, Tensor]:
        """``method="rgb"`` (umhs_model.py:265-267): the field's colour composited per ray -> rgb, accumulation, depth.
        Deliberate deviation: the reference calls ``renderer_rgb(rgb=, weights=)`` WITHOUT ``ray_indices`` / ``num_rays`` on packed
        samples, which makes nerfstudio's RGBRenderer sum over ALL samples of the batch (one colour for every ray); this composites
        per ray, as the reference does for every other output.  Background: "random" leaves the colour unblended here (it is blended
        in the loss, ``blend_background_for_loss_computation``), "white" / "black" add ``bg (1 - accumulation)``, as RGBRenderer does."""
        c, fr = self.config, ray_samples.frustums
        if fr.origins.numel() == 0:  # no sample survived the march: every ray shows the background (and the step has a zero gradient)
            R, dev_ = packed_info.shape[0], packed_info.device
            zero = (self.field.flat[:1] * 0.0).sum() if torch.is_grad_enabled() else torch.zeros((), device=dev_)  # keeps loss.backward() legal
            acc = torch.zeros(R, 1, device=dev_) + zero
            rgb = torch.zeros(R, 3, device=dev_) + zero + (1.0 if self.background_color == "white" else 0.0)
            return {"accumulation": acc, "depth": torch.zeros(R, 1, device=dev_), "rgb": rgb, "num_samples_per_ray": packed_info[:, 1],
                    "weights": torch.zeros(0, 1, device=dev_)}
        fo = self.field(ray_samples)
        weights, accumulation, depth, rgb = ops.CompositeFn.apply(fo[FieldHeadNames.DENSITY], fr.starts, fr.ends, packed_info,
                                                                  bool(c.use_gradient_scaling), fo[FieldHeadNames.RGB])
        if self.background_color in ("white", "black"):
            rgb = rgb + (1.0 if self.background_color == "white" else 0.0) * (1.0 - accumulation)
        if not self.training:
            rgb = rgb.clamp(0.0, 1.0)
        tmid = (fr.starts + fr.ends) / 2  # DepthRenderer: expected depth (the compositing kernel's), clipped to the batch-wide [min, max] of t_mid
        depth_c = torch.minimum(torch.maximum(depth, tmid.min()), tmid.max()) if tmid.numel() else depth
        return {"accumulation": accumulation, "depth": depth_c, "rgb": rgb, "num_samples_per_ray": packed_info[:, 1], "weights": weights}

    def _render_outputs_from_samples(self, ray_samples, ray_indices, num_rays: int, packed_info) -> Dict[str, Tensor]:
        """The same outputs without gradients (eval images, ``ns-render``): mlp_base -> transmittance weights -> heads with the per-ray
        sums formed inside the kernel -- an image's samples never exist as [N, bands] arrays (an eval chunk of 32 k rays
        x 529 samples x 31 bands x 3 streams is 6.4 GB written and read back otherwise)."""
        c, f = self.config, self.field
        spec = f._spec()
        L = spec.layout
        fr = ray_samples.frustums
        n = fr.origins.numel() // 3
        o, d = _hip.f32c(fr.origins).view(n, 3), _hip.f32c(fr.directions).view(n, 3)
        t0, t1 = _hip.f32c(fr.starts).view(-1), _hip.f32c(fr.ends).view(-1)
        flat = f.flat.detach()
        wpos, pos01, sel = ops.positions_fwd(o, d, t0, t1, spec)
        cached = (getattr(ray_samples, "metadata", None) or {}).get("umhs_enc")
        enc, counted = None, False
        if cached is not None and cached[1].numel() == n:
            enc = ops.enc_gather(cached[0], cached[1])  # encoded once, by the sampler's density query (same positions, same table)
        else:
            enc = ops.hashgrid_fwd(pos01, L.view(flat, "mlp_base.encoder.hash_table"), spec.scalings, L.log2_hashmap_size, True)
        fo = ops.field_base_fwd(spec, flat, enc, True, sel, rows16=True)
        del enc, cached
        weights, acc, depth, _ = ops.composite_fwd(fo["sigma"], t0, t1, packed_info, [])
        ri = (ray_indices if ray_indices.dtype == torch.int64 else ray_indices.long()).contiguous()
        ho = ops.field_heads_fwd(spec, flat, fo["base16"], wpos, d, weights, ri, packed_info, want_logits=False, pack_ready=True, release=False)
        comp = ho["comp"] + [ho["comp_abundances"]]
        mm = ops.tmid_minmax(t0, t1)
        rgb, depth_c, seg_probs, seg_raw, seg_pred = ops.ray_epilogue_fwd(
            comp[0], _hip.f32c(self.converter.transform_matrix), f.endmembers.detach(), acc, depth, mm, _hip.f32c(self.class_colors), 0.2)
        return self._assemble_outputs(acc.view(-1, 1), depth_c, comp, rgb, packed_info, seg_probs, seg_raw, seg_pred, weights.view(-1, 1))

    def _assemble_outputs(self, accumulation, depth_c, comp, rgb, packed_info, seg_probs, seg_raw, seg_pred, weights, lazy_bands=False):
        """The output dict of umhs_model.py:260-327 (same keys).  ``lazy_bands``: the per-band entries (``wv_i``, ``residual_i``,
        ``abundances_i``; 2B+C slicing ops per step in the reference) are created on first access instead of eagerly."""
        c = self.config
        spectral, abund = comp[0], comp[-1]
        outputs: Dict[str, Tensor] = BandOutputs() if lazy_bands else {}
        outputs.update({"accumulation": accumulation, "depth": depth_c, "spectral": spectral})
        if c.pred_specular:
            outputs["spectral2"], outputs["specular"] = comp[1], comp[2]
        outputs["rgb"] = rgb
        outputs["num_samples_per_ray"] = packed_info[:, 1]
        outputs["abundances"] = abund
        if c.per_band_outputs:
            if lazy_bands:
                outputs.bands = {"wv": spectral, "abundances": abund, **({"residual": comp[2]} if c.pred_specular else {})}
            else:
                for i in range(spectral.shape[-1]):
                    outputs[f"wv_{i}"] = spectral[..., i]
                if c.pred_specular:
                    for i in range(spectral.shape[-1]):
                        outputs[f"residual_{i}"] = comp[2][..., i]
                for i in range(abund.shape[-1]):
                    outputs[f"abundances_{i}"] = abund[:, i]
        outputs["seg_probs"], outputs["seg_raw"], outputs["seg_pred"] = seg_probs, seg_raw, seg_pred
        outputs["weights"] = weights
        return outputs

    # ---- training step without autograd -------------------------------------------------------------
    def direct_step_supported(self, batch) -> bool:
        if not self.field.use_grad_sink:
            return False
        sink = self.field._spec().grad_sink  # created on first use
        return (self.training and self.config.method in ("spectral", "rgb+spectral") and batch["image"].shape[-1] == 3
                and os.environ.get("UMHS_DIRECT_STEP", "1") != "0" and sink.owns_next_backward())

    def draw_training_background(self, batch: Dict) -> Optional[Tensor]:
        """The random background of this step's rgb loss (RGBRenderer.blend_background_for_loss_computation), drawn by the caller
        when it wants to fix the position of the draw in the generator's sequence (UMHSPipeline does, in front of its prefetch)."""
        if self.config.method != "rgb+spectral" or self.background_color != "random":
            return None
        return torch.rand_like(_hip.f32c(batch["image"].to(self.device)))

    def forward_backward_from_samples(self, ray_samples: RaySamples, ray_indices: Tensor, num_rays: int, batch: Dict,
                                      packed_info: Optional[Tensor] = None, background: Optional[Tensor] = None):
        """get_outputs (after the sampler) + get_loss_dict + backward of the summed loss, as one straight launch sequence with no
        autograd graph: the same kernels with the same arguments in the same order as the autograd path (which remains the general
        one), minus ~0.6 ms of host time per step.  Gradients land in the field's gradient sink (= ``field.flat.grad``).
        Returns (outputs, loss_dict); both are detached."""
        c, f = self.config, self.field
        spec = f._spec()
        L = spec.layout
        fr = ray_samples.frustums
        n = fr.origins.numel() // 3
        o, d = _hip.f32c(fr.origins).view(n, 3), _hip.f32c(fr.directions).view(n, 3)
        t0, t1 = _hip.f32c(fr.starts).view(-1), _hip.f32c(fr.ends).view(-1)
        flat = f.flat.detach()
        if packed_info is None:
            packed_info = (getattr(ray_samples, "metadata", None) or {}).get("umhs_packed_info")
        if packed_info is None:
            packed_info = ops.pack_info(ray_indices, num_rays)
        # forward (FieldFn.forward -> CompositeFn.forward -> RayEpilogueFn.forward -> LossFn.forward)
        wpos, pos01, sel = ops.positions_fwd(o, d, t0, t1, spec)
        # Everything of this step that depends on positions / parameters only -- the t_mid clip bounds, the weight pack images of
        # the forward and the backward, the bucket histogram + scan of the hash-grid backward (~80 us of small launches) -- runs
        # on a side stream in the shadow of the two big forward kernels.
        side = self._side_stream() if os.environ.get("UMHS_SIDE_STREAM", "1") != "0" else None
        prepared = False
        if side is not None:
            main = torch.cuda.current_stream(self.device)
            mm = torch.empty(2, device=self.device, dtype=torch.float32)  # every allocation happens on the main stream
            can_partition = ops.reserve_step_workspaces(spec, n, self.device)
            ev_in, ev_ready, ev_done = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
            ev_in.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev_in)
                ops.field_fwd_prepare(spec, flat)
                ops.tmid_minmax(t0, t1, out=mm)
                if can_partition and n > 0:
                    ops.field_bwd_prepare(spec, flat, n)
                ev_ready.record(side)  # one event for all three: every cross-stream wait is a barrier packet (~5 us) on main
        cached = (getattr(ray_samples, "metadata", None) or {}).get("umhs_enc")
        enc, counted = None, False
        if cached is not None and cached[1].numel() == n:
            enc = ops.enc_gather(cached[0], cached[1])  # encoded once, by the sampler's density query (same positions, same table)
        else:
            # The gather kernel has every (sample, level) hashed and its vector ALU idle: the bucket histogram of the hash-grid
            # backward rides in the same launch (as a kernel of its own it cost 16 us of the step even hidden on the side stream).
            if side is not None and can_partition and n > 0 and os.environ.get("UMHS_FUSED_COUNT", "1") != "0":
                enc = ops.hashgrid_fwd_count(pos01, L.view(flat, "mlp_base.encoder.hash_table"), spec.scalings, L.log2_hashmap_size)
            if enc is None:
                enc = ops.hashgrid_fwd(pos01, L.view(flat, "mlp_base.encoder.hash_table"), spec.scalings, L.log2_hashmap_size, True)
            else:
                counted = True
        if side is not None:
            # What is left of the backward's prepare (the scans; with an encoding taken from the sampler's cache the histogram too,
            # a few workgroups per level) trickles along under everything up to the scatter pass, the first kernel that needs it.
            ev_hash = torch.cuda.Event()
            ev_hash.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev_hash)
                if counted:
                    prepared = ops.hashgrid_bwd_prepare_counted(pos01, spec.scalings, L.log2_hashmap_size)
                elif can_partition and n > 0:
                    prepared = ops.hashgrid_bwd_prepare(pos01, spec.scalings, L.log2_hashmap_size)
                ev_done.record(side)
            main.wait_event(ev_ready)
        # Above 32 bands the step runs without any per-sample [N,B] array: forward as two launches with the rendering
        # weights known in between, per-ray sums inside the heads kernel, the mixing product once per RAY; the value half of the
        # compositing backward folded into the field backward.  At 31 bands both forms take the same time (C2 0.82 ms), so the
        # per-sample form stays the default there.  UMHS_FUSED_BWD=0 / 1 forces either.
        knob_b = os.environ.get("UMHS_FUSED_BWD", "")
        fused_bwd = (n > 0 and knob_b != "0" and (knob_b == "1" or L.wavelengths > 32) and ops.field_bwd_composited_supported(spec)
                     and ops.field_heads_fwd_supported(spec))
        split_fwd = fused_bwd
        if split_fwd:
            # mlp_base -> weights (transmittance scan) -> heads, whose kernel forms the per-ray sums itself
            fo = ops.field_base_fwd(spec, flat, enc, True, sel, pack_ready=side is not None, rows16=True)
            fo["emb"] = fo["base16"]  # the aligned-row form feeds the heads kernel and the composited backward
            weights, acc, depth, _ = ops.composite_fwd(fo["sigma"], t0, t1, packed_info, [])
            ri = ray_indices if ray_indices.dtype == torch.int64 else ray_indices.long()
            ri = ri.contiguous()
            ho = ops.field_heads_fwd(spec, flat, fo["emb"], wpos, d, weights, ri, packed_info, pack_ready=True, release=side is not None)
            fo.update(feat_logits=ho["feat_logits"])
            comp = ho["comp"] + [ho["comp_abundances"]]
            values = None
        else:
            fo = ops.field_fwd(spec, flat, enc, True, wpos, d, sel, want_emb=True, pack_ready=side is not None, want_logits=True)
            values = [fo["spectral"]] + ([fo["spectral2"], fo["specular"]] if c.pred_specular else []) + [fo["abundances"]]
        if side is None:
            mm = ops.tmid_minmax(t0, t1)
        M = _hip.f32c(self.converter.transform_matrix)
        hs, image = _hip.f32c(batch["hs_image"].to(self.device)), _hip.f32c(batch["image"].to(self.device))
        both = c.method == "rgb+spectral"
        bg = (background if background is not None else torch.rand_like(image)) if (both and self.background_color == "random") else None
        w = (5.0, float(c.rgb_loss_weight)) if both else (1.0, 0.0)
        bwd_comp = None
        if not split_fwd:
            weights, acc, depth, comp = ops.composite_fwd(fo["sigma"], t0, t1, packed_info, values)
        # ray epilogue + both losses + their backward down to d_spectral / d_accumulation: one launch
        rgb, depth_c, seg_probs, seg_raw, seg_pred, losses, d_spec, d_acc = ops.ray_train_tail(
            comp[0], M, f.endmembers.detach(), acc, depth, mm, _hip.f32c(self.class_colors), hs, image if both else None, bg, 0.2,
            w[0], w[1], both)
        if fused_bwd:
            d_sigma = d_spectral_samples = None
            bwd_comp = dict(sigma=fo["sigma"], t0=t0, t1=t1, packed_info=packed_info, ray_indices=ri, weights=weights, d_comp=d_spec,
                            d_acc=d_acc, grad_scaling=bool(c.use_gradient_scaling))
        else:
            d_sigma, d_values = ops.composite_bwd(fo["sigma"], t0, t1, packed_info, weights, values[:1], [d_spec], [True], d_acc,
                                                  bool(c.use_gradient_scaling))
            d_spectral_samples = d_values[0]
        left = ops.field_backward_into(spec, f.flat, pos01, sel, wpos, d, enc, fo["sigma_raw"], fo["emb"], d_sigma, d_spectral_samples, None,
                                       prepared=prepared, feat_logits=fo["feat_logits"], hash_ready=ev_done if side is not None else None,
                                       comp=bwd_comp)
        assert left is None  # direct_step_supported() guarantees the sink owned this backward
        outputs = self._assemble_outputs(acc.view(-1, 1), depth_c, comp, rgb, packed_info, seg_probs, seg_raw, seg_pred, weights.view(-1, 1),
                                         lazy_bands=True)
        loss_dict = {"spectral_loss": losses[0]}
        if both:
            loss_dict["rgb_loss"] = losses[1]
        return outputs, loss_dict

    def _side_stream(self):
        s = getattr(self, "_side", None)
        if s is None or s.device != self.device:
            s = self._side = torch.cuda.Stream(device=self.device)
        return s

    def _ones2(self) -> Tensor:
        t = getattr(self, "_ones2_t", None)
        if t is None or t.device != self.device:
            t = self._ones2_t = torch.ones(2, device=self.device)
        return t

    # ---- losses / metrics ----------------------------------------------------------------------------
    def blend_background_for_loss_computation(self, pred_image, pred_accumulation, gt_image, background: Optional[Tensor] = None):
        """nerfstudio RGBRenderer.blend_background_for_loss_computation as called at umhs_model.py:358-362 (``background``: a fixed
        draw of the random background instead of a fresh one -- parity tests)."""
        if self.background_color == "random":
            bg = background if background is not None else torch.rand_like(pred_image)
            pred_image = pred_image + bg * (1.0 - pred_accumulation)
        if gt_image.shape[-1] == 4:
            bgc = bg if self.background_color == "random" else torch.full_like(pred_image, 1.0 if self.background_color == "white" else 0.0)
            gt_image = gt_image[..., :3] * gt_image[..., 3:] + bgc * (1 - gt_image[..., 3:])
        return pred_image, gt_image

    def get_loss_dict(self, outputs, batch, metrics_dict=None, background: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """umhs_model.py:329-383: 5*MSE(spectral) + rgb_loss_weight*MSE(rgb blended with a random background).
        ``background`` (not in the reference): the [R,3] random-background draw, for callers that fix it (parity tests)."""
        loss_dict = {}
        image = batch["image"].to(self.device)
        m = self.config.method
        if m != "rgb" and image.shape[-1] == 3:  # fused path: both MSEs (+ random-background blend) in one launch
            hs = batch["hs_image"].to(self.device)
            if m == "spectral":
                loss_dict["spectral_loss"], _ = ops.LossFn.apply(outputs["spectral"], hs, None, None, None, None, 1.0, 0.0)
            else:
                bg = (background if background is not None else torch.rand_like(outputs["rgb"])) if self.background_color == "random" else None
                loss_dict["spectral_loss"], loss_dict["rgb_loss"] = ops.LossFn.apply(
                    outputs["spectral"], hs, outputs["rgb"], outputs["accumulation"], bg, image, 5.0, float(self.config.rgb_loss_weight))
            return loss_dict
        pred_rgb, gt_rgb = self.blend_background_for_loss_computation(outputs["rgb"], outputs["accumulation"], image, background)
        if m == "rgb":
            loss_dict["rgb_loss"] = torch.nn.functional.mse_loss(pred_rgb, gt_rgb)
        elif m == "spectral":
            loss_dict["spectral_loss"] = torch.nn.functional.mse_loss(outputs["spectral"], batch["hs_image"].to(self.device))
        else:
            loss_dict["spectral_loss"] = 5 * torch.nn.functional.mse_loss(outputs["spectral"], batch["hs_image"].to(self.device))
            loss_dict["rgb_loss"] = self.config.rgb_loss_weight * torch.nn.functional.mse_loss(pred_rgb, gt_rgb)
        return loss_dict

    @staticmethod
    def psnr(pred: Tensor, gt: Tensor) -> Tensor:
        return 10.0 * torch.log10(1.0 / torch.mean((pred - gt) ** 2))

    def get_metrics_dict(self, outputs, batch) -> Dict[str, Tensor]:
        """umhs_model.py:385-405.  Values stay device tensors (the reference's ``.item()`` calls would sync the stream)."""
        rgb, gt_rgb, nspr = outputs["rgb"].detach(), batch["image"].to(self.device)[..., :3], outputs["num_samples_per_ray"]
        md = {"psnr": lambda: self.psnr(rgb, gt_rgb), "rmse": lambda: torch.sqrt(torch.nn.functional.mse_loss(rgb, gt_rgb))}
        if "spectral" in self.config.method:
            spec, gt = outputs["spectral"].detach(), batch["hs_image"].to(self.device)
            md["psnr_spectral"] = lambda: self.psnr(spec, gt)
            md["rmse_spectral"] = lambda: torch.sqrt(torch.nn.functional.mse_loss(spec, gt))
        md["num_samples_per_batch"] = lambda: nspr.sum()
        lazy = LazyMetrics(md)
        return lazy if self.training else dict(lazy.materialize())

    @torch.no_grad()  # as nerfstudio's Model.get_outputs_for_camera_ray_bundle
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """umhs_model.py:593-620.  The reference walks the image in 512-ray chunks (its kernels are launch-bound there); here a
        chunk is ``max(eval_num_rays_per_chunk, 32768)`` rays, i.e. a 128x128 image or a quarter-megapixel strip is ONE fused
        inference launch per kernel, and outputs stay on the model device."""
        o = camera_ray_bundle.origins
        hw = o.shape[:-1]
        origins, directions = o.reshape(-1, 3).to(self.device), camera_ray_bundle.directions.reshape(-1, 3).to(self.device)
        outs: Dict[str, List[Tensor]] = {}
        n, ch = origins.shape[0], max(int(self.config.eval_num_rays_per_chunk), 32768)
        for i in range(0, n, ch):
            rb = RayBundle(origins=origins[i:i + ch], directions=directions[i:i + ch])
            for k, v in self.forward(rb).items():
                if isinstance(v, Tensor) and v.shape[:1] == (len(rb),):
                    outs.setdefault(k, []).append(v)
        return {k: (v[0] if len(v) == 1 else torch.cat(v)).view(*hw, -1) for k, v in outs.items()}

    def get_outputs_for_camera(self, camera, obb_box=None) -> Dict[str, Tensor]:
        """umhs_model.py:527-539: ``camera`` is anything with ``generate_rays(camera_indices=0, keep_shape=True)`` or a RayBundle."""
        rb = camera if isinstance(camera, RayBundle) else camera.generate_rays(camera_indices=0, keep_shape=True)
        return self.get_outputs_for_camera_ray_bundle(rb)

    @torch.no_grad()
    def get_image_metrics_and_images(self, outputs: Dict[str, Tensor], batch: Dict[str, Tensor]):
        """umhs_model.py:407-512: psnr / ssim on rgb, psnr / ssim / sam / rmse on the spectral image; lpips is omitted (its
        pretrained weights cannot be fetched offline).  One host read at the end instead of one ``.item()`` per metric."""
        gt_rgb = batch["image"].to(self.device)
        if gt_rgb.shape[-1] == 4:  # renderer_rgb.blend_background: composite RGBA ground truth over the background colour
            bgv = 1.0 if self.background_color == "white" else 0.0
            gt_rgb = gt_rgb[..., :3] * gt_rgb[..., 3:] + bgv * (1 - gt_rgb[..., 3:])
        pred_rgb = outputs["rgb"]
        vals = {}
        sse, _, _ = ops.pixel_metrics(pred_rgb, gt_rgb)
        vals["psnr"] = 10.0 * torch.log10(pred_rgb.numel() / sse)
        vals["ssim"] = ops.ssim(gt_rgb, pred_rgb)
        if "spectral" in self.config.method:
            gt_s, pred_s = batch["hs_image"].to(self.device), outputs["spectral"]
            sse, sam, cnt = ops.pixel_metrics(pred_s, gt_s)
            mse = sse / pred_s.numel()
            vals["psnr_spectral"] = 10.0 * torch.log10(1.0 / mse)
            vals["ssim_spectral"] = ops.ssim(gt_s, pred_s)
            vals["sam_spectral"] = sam / cnt
            vals["rmse_spectral"] = torch.sqrt(mse)
        keys = list(vals)
        host = torch.stack([vals[k].double() for k in keys]).tolist()
        metrics_dict = dict(zip(keys, host))
        acc, depth = outputs["accumulation"], outputs["depth"]
        near, far = depth.min(), depth.max()
        images_dict = {"img": torch.cat([gt_rgb, pred_rgb], dim=1), "accumulation": acc.clamp(0, 1).expand(*acc.shape[:-1], 3),
                       "depth": ((depth - near) / (far - near + 1e-10)).clamp(0, 1).expand(*depth.shape[:-1], 3),
                       "se_per_pixel": ((gt_rgb - pred_rgb) ** 2).mean(dim=-1, keepdim=True)}
        return metrics_dict, images_dict

    @staticmethod
    def compute_sam(pred: Tensor, gt: Tensor, eps: float = 1e-8) -> Tensor:
        """umhs_model.py:514-525 (helper; mean spectral angle in radians, eps in the denominator)."""
        cos = (pred * gt).sum(dim=-1) / (torch.norm(pred, dim=-1) * torch.norm(gt, dim=-1) + eps)
        return torch.acos(torch.clamp(cos, -1, 1)).mean()
