"""Occupancy-grid volumetric sampler for ROCm.

Mirror of what the reference gets from nerfacc==0.5.2 ``OccGridEstimator`` + nerfstudio's ``VolumetricSampler``
(``umhs_model.py:201-209`` construction, ``:229-237`` sampling, ``:549-554`` grid update): same constructor arguments,
buffers (``aabbs``, ``occs``, ``binaries``) and method names.  nerfacc ships CUDA kernels only; the ray marching and the
visibility pruning run in libumhs_hip.so (``umhs_march_walk`` + ``umhs_march_scratch/count/write``, ``umhs_visibility``), the density queries in the
density-only field kernel.  The grid bookkeeping of ``update_every_n_steps`` is index arithmetic in torch (runs every 16
steps).  nerfacc's source is not available offline: behaviour is restated from its published algorithm (parity unpinned).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _hip
from ._hip import ptr
from ._ns_compat import RayBundle, RaySamples, packed_ray_samples


_scratch_pool = {}  # device index -> [(t_starts rows, t_ends rows, voxel lists, event of the last reader)]


def _scratch_acquire(n: int, nwalk: int, dev) -> Tuple[Tensor, Tensor, Tensor]:
    """[R, cap] scratch rows of the single-pass march + the ``nwalk`` bytes of voxel lists of ``umhs_march_walk``.  A march in flight
    owns its set (a prefetched march and an eval-time march may be outstanding together); ``_scratch_release`` hands it back once
    the compaction that read it has been issued."""
    pool = _scratch_pool.setdefault(dev.index or 0, [])
    cur = torch.cuda.current_stream(dev)
    for i, (a, b, w, ev) in enumerate(pool):
        if a.numel() >= n and w.numel() >= nwalk:
            pool.pop(i)
            cur.wait_event(ev)
            _use_on(cur, a, b, w)
            return a, b, w
    pool.clear()  # too small for this batch size: let them go
    return torch.empty(n, device=dev), torch.empty(n, device=dev), torch.empty(nwalk, device=dev, dtype=torch.uint8)


def _scratch_release(trio, dev) -> None:
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    _scratch_pool.setdefault(dev.index or 0, []).append((trio[0], trio[1], trio[2], ev))


class MarchHandle:
    """A ray march in flight (``march_begin``): the walk has been issued, the sample count is on its way to the host."""

    __slots__ = ("o", "d", "R", "bin", "args", "nears", "fars", "jitter", "cap", "scratch", "walked", "counts", "packed_info", "stats",
                 "host", "event", "stream")


def _use_on(stream, *tensors) -> None:
    """Tensors produced on another stream are about to be used on ``stream``: keep the allocator from recycling them early."""
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(stream)


def ray_prefix(counts: Tensor) -> Tuple[Tensor, Tensor]:
    """Per-ray counts [R] int64 -> (packed_info [R,2] = (exclusive prefix, count), stats [2] = (total, longest)); one launch."""
    R = counts.shape[0]
    packed_info = torch.empty((R, 2), device=counts.device, dtype=torch.int64)
    stats = torch.empty((2,), device=counts.device, dtype=torch.int64)
    _hip.check(_hip.lib().umhs_ray_prefix(ptr(counts), R, ptr(packed_info), ptr(stats), _hip.stream()), "umhs_ray_prefix")
    return packed_info, stats


def sample_midpoints(origins: Tensor, directions: Tensor, ray_indices: Tensor, t_starts: Tensor, t_ends: Tensor) -> Tensor:
    """origins[ray_indices] + directions[ray_indices] * (t_starts + t_ends)[:, None] / 2.0 (same bits), one launch."""
    n = ray_indices.shape[0]
    pos = torch.empty((n, 3), device=origins.device, dtype=torch.float32)
    _hip.check(_hip.lib().umhs_sample_midpoints(ptr(_hip.f32c(origins)), ptr(_hip.f32c(directions)), ptr(ray_indices.contiguous()),
                                                ptr(_hip.f32c(t_starts)), ptr(_hip.f32c(t_ends)), n, ptr(pos), _hip.stream()),
               "umhs_sample_midpoints")
    return pos


def march_begin(origins: Tensor, directions: Tensor, binaries_u8: Tensor, roi_aabb, levels: int, resolution: int, near: float,
                far: float, step: float, cone: float, nears: Optional[Tensor] = None, fars: Optional[Tensor] = None,
                jitter: Optional[Tensor] = None, jitter_step: float = 0.0) -> MarchHandle:
    """First half of ``march_rays`` -- everything before the host sync: the walk (single pass, samples parked in the [R, cap]
    scratch rows), the per-ray prefix and an asynchronous copy of (total, longest row) to pinned host memory.  Issued on the
    current stream; a trainer may issue it one step ahead on a side stream (``OccGridEstimator.prefetch_march``), provided
    that stream is ordered after the previous ``march_finish`` (both use the one scratch buffer of the device)."""
    h = MarchHandle()
    h.o, h.d = _hip.f32c(origins), _hip.f32c(directions)
    h.R, dev = h.o.shape[0], h.o.device
    h.bin = binaries_u8
    h.args = ((C.c_float * 6)(*[float(v) for v in roi_aabb]), levels, resolution, near, far, step, cone)
    h.nears = _hip.f32c(nears) if nears is not None else None
    h.fars = _hip.f32c(fars) if fars is not None else None
    h.jitter = (_hip.f32c(jitter), float(jitter_step)) if jitter is not None else (None, 0.0)
    h.counts = torch.empty((h.R,), device=dev, dtype=torch.int64)
    h.cap = int(os.environ.get("UMHS_MARCH_CAP", "1024"))  # scratch row per ray of the single-pass form (0: always two passes)
    h.stream = torch.cuda.current_stream(dev)
    lib = _hip.lib()
    roi = h.args[0]
    # the walk on its own, one wave per ray (UMHS_MARCH_SERIAL=1: the emission kernel walks the grid itself, one thread per ray)
    nwalk = lib.umhs_march_walk_workspace_bytes(h.R) if os.environ.get("UMHS_MARCH_SERIAL", "0") != "1" else 0
    h.scratch = _scratch_acquire(h.R * h.cap, nwalk, dev) if h.R > 0 else None
    h.walked = (h.scratch[2], nwalk) if (h.scratch is not None and nwalk > 0) else (None, 0)
    if h.walked[0] is not None:
        _hip.check(lib.umhs_march_walk(ptr(h.o), ptr(h.d), h.R, ptr(h.bin), roi, levels, resolution, near, far, ptr(h.nears), ptr(h.fars),
                                       ptr(h.jitter[0]), h.jitter[1], ptr(h.walked[0]), h.walked[1], _hip.stream()), "umhs_march_walk")
    if h.scratch is not None and h.cap > 0:  # one emission pass: counts + the samples themselves parked in [R, cap] rows
        _hip.check(lib.umhs_march_scratch(ptr(h.o), ptr(h.d), h.R, ptr(h.bin), roi, levels, resolution, near, far, step, cone, ptr(h.nears),
                                          ptr(h.fars), ptr(h.jitter[0]), h.jitter[1], h.cap, ptr(h.counts), ptr(h.scratch[0]), ptr(h.scratch[1]),
                                          ptr(h.walked[0]), h.walked[1], _hip.stream()), "umhs_march_scratch")
    else:
        _hip.check(lib.umhs_march_count(ptr(h.o), ptr(h.d), h.R, ptr(h.bin), roi, levels, resolution, near, far, step, cone,
                                        ptr(h.nears), ptr(h.fars), ptr(h.jitter[0]), h.jitter[1], ptr(h.counts), ptr(h.walked[0]), h.walked[1],
                                        _hip.stream()), "umhs_march_count")
    h.packed_info, h.stats = ray_prefix(h.counts)
    # the sample count sizes the outputs: one host sync per batch, as in nerfacc (the row-overflow flag rides along)
    h.host = torch.zeros(2, dtype=torch.int64).pin_memory()
    h.host.copy_(h.stats, non_blocking=True)
    h.event = torch.cuda.Event()
    h.event.record(h.stream)
    return h


def march_finish(h: MarchHandle):
    """Second half of ``march_rays``: wait for the count, size the outputs, move the samples from the scratch rows to their
    packed places.  -> (ray_indices int64 [N], t_starts [N], t_ends [N], packed_info [R,2]) on the current stream."""
    dev = h.o.device
    h.event.synchronize()
    n, cmax = (int(v) for v in h.host.tolist())
    cur = torch.cuda.current_stream(dev)
    if cur != h.stream:  # marched ahead of time on another stream
        cur.wait_event(h.event)
        _use_on(cur, h.o, h.d, h.nears, h.fars, h.jitter[0], h.counts, h.packed_info, h.stats, *(h.scratch or ()))
    t0 = torch.empty((n,), device=dev, dtype=torch.float32)
    t1 = torch.empty((n,), device=dev, dtype=torch.float32)
    ri = torch.empty((n,), device=dev, dtype=torch.int64)
    if n > 0:
        lib = _hip.lib()
        roi, levels, resolution, near, far, step, cone = h.args
        if h.scratch is not None and h.cap > 0 and cmax <= h.cap:
            _hip.check(lib.umhs_march_compact(ptr(h.packed_info), h.R, h.cap, ptr(h.scratch[0]), ptr(h.scratch[1]), ptr(t0), ptr(t1), ptr(ri),
                                              _hip.stream()), "umhs_march_compact")
        else:  # some ray overflowed its scratch row: second emission pass writing straight to the packed places
            _hip.check(lib.umhs_march_write(ptr(h.o), ptr(h.d), h.R, ptr(h.bin), roi, levels, resolution, near, far, step, cone,
                                            ptr(h.nears), ptr(h.fars), ptr(h.jitter[0]), h.jitter[1], ptr(h.packed_info), ptr(t0),
                                            ptr(t1), ptr(ri), ptr(h.walked[0]), h.walked[1], _hip.stream()),
                       "umhs_march_write")
    if h.scratch is not None:
        _scratch_release(h.scratch, dev)
        h.scratch = None
    return ri, t0, t1, h.packed_info


def march_rays(origins: Tensor, directions: Tensor, binaries_u8: Tensor, roi_aabb, levels: int, resolution: int, near: float,
               far: float, step: float, cone: float, nears: Optional[Tensor] = None, fars: Optional[Tensor] = None,
               jitter: Optional[Tensor] = None, jitter_step: float = 0.0):
    """-> (ray_indices int64 [N], t_starts [N], t_ends [N], packed_info [R,2]) on the device."""
    return march_finish(march_begin(origins, directions, binaries_u8, roi_aabb, levels, resolution, near, far, step, cone, nears, fars,
                                    jitter, jitter_step))


def visibility_mask(sigma: Tensor, t_starts: Tensor, t_ends: Tensor, packed_info: Tensor, early_stop_eps: float, alpha_thre: float,
                    with_counts: bool = False):
    """nerfacc render_visibility_from_density -> bool mask [N]; ``with_counts``: (uint8 mask [N], survivors per ray int64 [R])."""
    s = _hip.f32c(sigma).view(-1)
    mask = torch.empty(s.shape, device=s.device, dtype=torch.uint8)
    R = packed_info.shape[0]
    if with_counts:
        kept = torch.empty((R,), device=s.device, dtype=torch.int64)
        _hip.check(_hip.lib().umhs_visibility_count(ptr(s), ptr(t_starts), ptr(t_ends), ptr(packed_info), R, s.shape[0], float(early_stop_eps),
                                                    float(alpha_thre), ptr(mask), ptr(kept), _hip.stream()), "umhs_visibility_count")
        return mask, kept
    _hip.check(_hip.lib().umhs_visibility(ptr(s), ptr(t_starts), ptr(t_ends), ptr(packed_info), R, s.shape[0],
                                          float(early_stop_eps), float(alpha_thre), ptr(mask), _hip.stream()), "umhs_visibility")
    return mask.bool()


def compact_samples(mask_u8: Tensor, packed_in: Tensor, packed_out: Tensor, n_out: int, t_starts: Tensor, t_ends: Tensor, origins: Tensor,
                    directions: Tensor, camera_indices: Optional[Tensor] = None):
    """Survivors of ``mask_u8`` in packed order -> dict(ray_indices, t_starts, t_ends, origins [n,3], directions [n,3],
    camera_indices [n,1] | None, sel [n]); one launch instead of nonzero + index_selects + gathers."""
    dev, R = mask_u8.device, packed_in.shape[0]
    out = {"ray_indices": torch.empty((n_out,), device=dev, dtype=torch.int64), "t_starts": torch.empty((n_out,), device=dev),
           "t_ends": torch.empty((n_out,), device=dev), "origins": torch.empty((n_out, 3), device=dev),
           "directions": torch.empty((n_out, 3), device=dev), "sel": torch.empty((n_out,), device=dev, dtype=torch.int64),
           "camera_indices": None}
    cam = None
    if camera_indices is not None:
        cam = camera_indices.reshape(-1).to(torch.int64).contiguous()
        out["camera_indices"] = torch.empty((n_out, 1), device=dev, dtype=torch.int64)
    if n_out > 0:
        _hip.check(_hip.lib().umhs_compact_samples(ptr(mask_u8), ptr(packed_in), ptr(packed_out), R, ptr(t_starts), ptr(t_ends),
                                                   ptr(_hip.f32c(origins)), ptr(_hip.f32c(directions)), ptr(cam), ptr(out["ray_indices"]),
                                                   ptr(out["t_starts"]), ptr(out["t_ends"]), ptr(out["origins"]), ptr(out["directions"]),
                                                   ptr(out["camera_indices"]), ptr(out["sel"]), _hip.stream()), "umhs_compact_samples")
    return out


class OccGridEstimator(nn.Module):
    """Multi-level occupancy grid: level l covers ``roi_aabb`` enlarged 2^l about its centre (nerfacc ``OccGridEstimator``)."""

    def __init__(self, roi_aabb, resolution: int = 128, levels: int = 1):
        super().__init__()
        roi = torch.as_tensor(roi_aabb, dtype=torch.float32).flatten()
        assert roi.numel() == 6
        self.levels, self.res = int(levels), int(resolution)
        self.cells_per_lvl = self.res**3
        c, h = (roi[:3] + roi[3:]) / 2, (roi[3:] - roi[:3]) / 2
        self.register_buffer("aabbs", torch.stack([torch.cat([c - h * 2**l, c + h * 2**l]) for l in range(levels)]))
        self.register_buffer("occs", torch.zeros(self.levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros((self.levels, self.res, self.res, self.res), dtype=torch.bool))
        g = torch.arange(self.res)
        self.register_buffer("grid_coords", torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).reshape(-1, 3))
        self._roi = [float(v) for v in roi.tolist()]

    # ---- sampling ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sampling(self, rays_o: Tensor, rays_d: Tensor, sigma_fn: Optional[Callable] = None, near_plane: float = 0.0,
                 far_plane: float = 1e10, t_min: Optional[Tensor] = None, t_max: Optional[Tensor] = None,
                 render_step_size: float = 1e-3, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, stratified: bool = False,
                 cone_angle: float = 0.0, camera_indices: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
        """nerfacc ``OccGridEstimator.sampling``.  ``camera_indices`` [R(,1)] (extension): gathered per surviving sample together with
        the ray origins / directions; the gathered rows and the survivors' packed_info are left in ``last_pruned`` /
        ``last_packed_info`` for the caller (VolumetricSampler) so that it does not have to index them again."""
        self.last_keep_index = None  # index of the survivors among the marched candidates, when a density pruning pass ran
        pre = getattr(self, "_prefetched", None)
        key = self._march_key(rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified, cone_angle)
        if pre is not None and pre[0] == key and pre[1] is self.binaries and pre[2] == self.binaries._version:
            h, self._prefetched = pre[3], None  # this very march was issued ahead of time: same rays, same grid, jitter drawn
        else:
            h = self._march_begin(rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified, cone_angle)
        ri, t0, t1, pinfo = march_finish(h)
        self.last_packed_info, self.last_pruned, self.last_candidates = pinfo, None, int(t0.numel())
        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and sigma_fn is not None and t0.numel() > 0:
            alpha_thre = min(alpha_thre, self._occs_mean())  # nerfacc reads occs.mean() per batch; it only changes in _update()
            sigmas = sigma_fn(t0, t1, ri)
            # survivors: mask + per-ray counts -> their packed_info -> ONE host sync for the total -> one compaction launch (order kept)
            mask, kept = visibility_mask(sigmas, t0, t1, pinfo, early_stop_eps, alpha_thre, with_counts=True)
            pinfo2, stats = ray_prefix(kept)
            hook, self.pre_sync_hook = getattr(self, "pre_sync_hook", None), None
            if hook is not None:  # host work a trainer wants done while the GPU is busy with the density query (one shot)
                hook()
            n2 = int(stats[0])
            out = compact_samples(mask, pinfo, pinfo2, n2, t0, t1, rays_o, rays_d, camera_indices)
            ri, t0, t1 = out["ray_indices"], out["t_starts"], out["t_ends"]
            self.last_keep_index, self.last_packed_info, self.last_pruned = out["sel"], pinfo2, out
        return ri, t0, t1

    @staticmethod
    def _march_key(rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified, cone_angle):
        return (rays_o.data_ptr(), rays_d.data_ptr(), tuple(rays_o.shape), float(near_plane), float(far_plane),
                None if t_min is None else t_min.data_ptr(), None if t_max is None else t_max.data_ptr(), float(render_step_size),
                bool(stratified), float(cone_angle))

    def _march_begin(self, rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified, cone_angle) -> MarchHandle:
        # per-ray planes only when the caller has them; the stratified start (nears + rand * step) is applied inside the walk
        jitter = torch.rand_like(rays_o[..., 0]) if stratified else None
        return march_begin(rays_o, rays_d, self.binaries.view(torch.uint8), self._roi, self.levels, self.res, near_plane, far_plane,
                           render_step_size, cone_angle, t_min, t_max, jitter, render_step_size)

    def prefetch_march(self, rays_o: Tensor, rays_d: Tensor, near_plane: float = 0.0, far_plane: float = 1e10,
                       t_min: Optional[Tensor] = None, t_max: Optional[Tensor] = None, render_step_size: float = 1e-3,
                       stratified: bool = False, cone_angle: float = 0.0) -> None:
        """Issue the ray march of a coming ``sampling()`` call now, on the current stream (a trainer's side stream, while the
        previous step is still computing): the walk depends on the rays and the grid only, not on the field.  ``sampling()``
        picks the result up when it is called with the same rays and arguments and the grid has not changed since; otherwise
        the prefetched march is dropped.  The stratified jitter is drawn here, i.e. in the same order as without prefetch as
        long as nothing else draws from the device generator in between."""
        key = self._march_key(rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified, cone_angle)
        h = self._march_begin(rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified, cone_angle)
        self._prefetched = (key, self.binaries, self.binaries._version, h)

    def _occs_mean(self) -> float:
        m = getattr(self, "_occs_mean_cache", None)
        key = (self.occs.data_ptr(), self.occs._version)  # any in-place write to ``occs`` bumps its version counter
        if m is None or m[0] != key:
            m = self._occs_mean_cache = (key, float(self.occs.mean()))
        return m[1]

    # ---- grid update (nerfacc OccGridEstimator._update) -----------------------------------------------------------
    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16) -> None:
        if not self.training:
            raise RuntimeError("update_every_n_steps() is a training-time call")
        if step % n == 0:
            self._update(step, occ_eval_fn, occ_thre, ema_decay, warmup_steps)

    def _sample_cells(self, step: int, warmup_steps: int):
        dev = self.occs.device
        if step < warmup_steps:
            every = torch.arange(self.cells_per_lvl, device=dev)
            return [every] * self.levels
        n = self.cells_per_lvl // 4
        out = []
        for lvl in range(self.levels):
            uniform = torch.randint(self.cells_per_lvl, (n,), device=dev)
            occupied = torch.nonzero(self.binaries[lvl].flatten())[:, 0]
            if n < occupied.numel():
                occupied = occupied[torch.randint(occupied.numel(), (n,), device=dev)]
            out.append(torch.cat([uniform, occupied]))
        return out

    @torch.no_grad()
    def _update(self, step, occ_eval_fn, occ_thre, ema_decay, warmup_steps):
        for lvl, idx in enumerate(self._sample_cells(step, warmup_steps)):
            coords = self.grid_coords[idx].to(torch.float32)
            x = (coords + torch.rand_like(coords)) / self.res
            x = self.aabbs[lvl, :3] + x * (self.aabbs[lvl, 3:] - self.aabbs[lvl, :3])
            occ = occ_eval_fn(x).squeeze(-1)
            cell = lvl * self.cells_per_lvl + idx
            self.occs[cell] = torch.maximum(self.occs[cell] * ema_decay, occ)
        thre = torch.clamp(self.occs[self.occs >= 0].mean(), max=occ_thre)
        self.binaries = (self.occs > thre).view(self.binaries.shape)

    def mark_all_occupied(self) -> None:
        self.occs.fill_(1.0)
        self.binaries = torch.ones_like(self.binaries)


class VolumetricSampler(nn.Module):
    """nerfstudio ``VolumetricSampler``: marches rays through the occupancy grid and returns packed RaySamples."""

    def __init__(self, occupancy_grid: OccGridEstimator, density_fn: Optional[Callable] = None):
        super().__init__()
        self.occupancy_grid, self.density_fn = occupancy_grid, density_fn

    def get_sigma_fn(self, origins: Tensor, directions: Tensor) -> Optional[Callable]:
        if self.density_fn is None or not self.training:
            return None
        density_fn = self.density_fn

        def sigma_fn(t_starts, t_ends, ray_indices):
            return density_fn(sample_midpoints(origins, directions, ray_indices, t_starts, t_ends)).squeeze(-1)

        return sigma_fn

    @staticmethod
    def _march_inputs(ray_bundle: RayBundle, far_plane: Optional[float]):
        rays_o, rays_d = ray_bundle.origins.contiguous(), ray_bundle.directions.contiguous()
        t_min = ray_bundle.nears.contiguous().reshape(-1) if ray_bundle.nears is not None and ray_bundle.fars is not None else None
        t_max = ray_bundle.fars.contiguous().reshape(-1) if t_min is not None else None
        return rays_o, rays_d, t_min, t_max, (1e10 if far_plane is None else far_plane)

    def prefetch(self, ray_bundle: RayBundle, render_step_size: float, near_plane: float = 0.0, far_plane: Optional[float] = None,
                 alpha_thre: float = 0.01, cone_angle: float = 0.0) -> None:
        """Start the march of a coming ``forward(ray_bundle, ...)`` on the current stream (see OccGridEstimator.prefetch_march)."""
        rays_o, rays_d, t_min, t_max, far_plane = self._march_inputs(ray_bundle, far_plane)
        self.occupancy_grid.prefetch_march(rays_o, rays_d, near_plane=near_plane, far_plane=far_plane, t_min=t_min, t_max=t_max,
                                           render_step_size=render_step_size, stratified=self.training, cone_angle=cone_angle)

    def forward(self, ray_bundle: RayBundle, render_step_size: float, near_plane: float = 0.0, far_plane: Optional[float] = None,
                alpha_thre: float = 0.01, cone_angle: float = 0.0) -> Tuple[RaySamples, Tensor]:
        rays_o, rays_d, t_min, t_max, far_plane = self._march_inputs(ray_bundle, far_plane)
        g = self.occupancy_grid
        ri, t0, t1 = g.sampling(rays_o, rays_d, sigma_fn=self.get_sigma_fn(rays_o, rays_d), near_plane=near_plane,
                                far_plane=far_plane, t_min=t_min, t_max=t_max, render_step_size=render_step_size,
                                stratified=self.training, cone_angle=cone_angle, alpha_thre=alpha_thre,
                                camera_indices=ray_bundle.camera_indices)
        self.last_packed_info = g.last_packed_info
        if t0.shape[0] == 0:  # nerfstudio: one fake sample so that downstream shapes stay valid
            ri = torch.zeros((1,), dtype=torch.long, device=rays_o.device)
            t0 = torch.ones((1,), dtype=torch.float32, device=rays_o.device)
            t1 = torch.ones((1,), dtype=torch.float32, device=rays_o.device)
            self.last_packed_info = None
        elif g.last_pruned is not None:  # the compaction launch gathered the rays' rows along with the survivors
            p = g.last_pruned
            return packed_ray_samples(p["origins"], p["directions"], t0, t1, p["camera_indices"]), ri
        cam = ray_bundle.camera_indices[ri] if ray_bundle.camera_indices is not None else None
        return packed_ray_samples(rays_o[ri], rays_d[ri], t0, t1, cam), ri
