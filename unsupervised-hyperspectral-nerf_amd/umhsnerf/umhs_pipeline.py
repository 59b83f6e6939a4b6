"""Pipeline glue (mirror of ``umhsnerf/umhs_pipeline.py``).  The reference forces ``world_size = 1``
(``:86,108-109``), which silently disables its DDP wrap; here ray batches are sharded one rank per GPU and the flat
"fields" gradient is all-reduced over RCCL/xGMI inside ``UMHSAdam.step``.

Data managers / parsers / checkpoint I/O are nerfstudio's and out of scope for the hot path."""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.distributed as dist

from ._ns_compat import RaySamples
from .umhs_model import UMHSConfig, UMHSModel


class UMHSPipeline(torch.nn.Module):
    """Minimal pipeline: model + one train iteration on packed samples (what ``Trainer.train_iteration`` does around
    ``pipeline.get_train_loss_dict``: forward, loss, backward, optimizer step)."""

    def __init__(self, config: UMHSConfig, device, metadata: Optional[Dict] = None, world_size: int = 1, local_rank: int = 0,
                 seed: Optional[int] = 42, scene_box=None, datamanager=None):
        super().__init__()
        self.world_size, self.local_rank = world_size, local_rank
        self.device = torch.device(device)
        self.datamanager = datamanager  # data.umhs_datamanager.UMHSDataManager (umhs_pipeline.py:87-94) or None (packed-sample callers)
        if datamanager is not None:
            metadata = {**(datamanager.metadata or {}), **(metadata or {})}
            scene_box = scene_box if scene_box is not None else getattr(datamanager, "scene_box", None)
        self._model = UMHSModel(config, scene_box=scene_box, metadata=metadata, seed=seed).to(device)
        if world_size > 1:  # identical parameters on every rank (DDP's initial broadcast)
            dist.broadcast(self._model.field.flat.data, src=0)
        self._model.field.use_grad_sink = True  # backward fills param.grad in place and reduces finished segments early
        self.optimizer = self._model.make_optimizer()

    @property
    def model(self) -> UMHSModel:
        return self._model

    def train_iteration(self, ray_samples: RaySamples, ray_indices, num_rays: int, batch: Dict, packed_info=None):
        self.optimizer.zero_grad(set_to_none=True)
        if self._model.direct_step_supported(batch):  # straight launch sequence, no autograd graph
            self.optimizer.arm_fused()  # optimizer.step() follows unconditionally: the table's Adam step may ride in the backward
            outputs, loss_dict = self._model.forward_backward_from_samples(ray_samples, ray_indices, num_rays, batch, packed_info)
        else:
            outputs = self._model.get_outputs_from_samples(ray_samples, ray_indices, num_rays, packed_info)
            loss_dict = self._model.get_loss_dict(outputs, batch)
            sum(loss_dict.values()).backward()
        self.optimizer.step()
        return outputs, loss_dict

    # ---- the reference's pipeline surface (umhs_pipeline.py:115-150 and VanillaPipeline.get_train_loss_dict) -----------
    def get_train_loss_dict(self, step: int):
        """One ``Trainer.train_iteration``: BEFORE callbacks (occupancy grid), next_train, forward, loss, backward, Adam (+ gradient
        reduction), AFTER callbacks (clamp_endmembers, fused into the Adam kernel).  Returns (outputs, loss_dict, metrics_dict)."""
        self._model.update_occupancy_grid(step)
        ray_bundle, batch = self._next_train(step)
        self.optimizer.zero_grad(set_to_none=True)
        if self._model.direct_step_supported(batch):
            sampled = torch.cuda.Event() if self.device.type == "cuda" else None
            if sampled is not None:
                sampled.record(torch.cuda.current_stream(self.device))  # the occupancy grid is final for this step
            ahead = {}

            def while_gpu_busy():
                # Host work with no dependence on this step's samples, done while the GPU runs the sampler's density query (the host
                # would otherwise just wait for the survivor count): this step's random background -- drawn here so that the device
                # generator sees jitter(step), background(step), jitter(step + 1) with or without prefetch -- and the next batch.
                ahead["bg"] = self._model.draw_training_background(batch)
                self._prefetch_next(step + 1, sampled)
                ahead["done"] = True

            grid = getattr(getattr(self._model, "sampler", None), "occupancy_grid", None)
            if grid is not None:
                grid.pre_sync_hook = while_gpu_busy
            ray_samples, ray_indices = self._model.sample(ray_bundle)
            if grid is not None:
                grid.pre_sync_hook = None
            if not ahead.get("done"):  # the sampler had nothing to prune (no host sync there): same work, now
                while_gpu_busy()
            self.optimizer.arm_fused()  # optimizer.step() follows unconditionally: the table's Adam step may ride in the backward
            outputs, loss_dict = self._model.forward_backward_from_samples(ray_samples, ray_indices, len(ray_bundle), batch,
                                                                           background=ahead["bg"])
            metrics_dict = self._model.get_metrics_dict(outputs, batch)
        else:
            outputs = self._model(ray_bundle)
            metrics_dict = self._model.get_metrics_dict(outputs, batch)
            loss_dict = self._model.get_loss_dict(outputs, batch, metrics_dict)
            sum(loss_dict.values()).backward()
        self.optimizer.step()
        return outputs, loss_dict, metrics_dict

    # ---- one-step-ahead ray batch + occupancy march -------------------------------------------------------------------
    # The march of a batch needs the rays and the occupancy grid only -- not the field -- and it is one latency-bound
    # dependency chain per ray (a few hundred waves, ~0.7 ms): issued on its own stream while the current step's sampler is
    # still at its density query, it runs in the shadow of that query and of the forward pass instead of in front of the next
    # step, and the host issues it while it would otherwise wait for the survivor count.  (Issued behind the step's launches it
    # ran beside the first field-backward kernel, whose 231-VGPR waves leave no room on a SIMD for a 76-VGPR marcher wave: every
    # CU that held one could not take a backward workgroup, and that kernel ran at half speed.  The density, forward and
    # compositing kernels co-reside with it.)  Skipped when the next step rewrites the grid.  The batch is drawn from the data manager's own generator and the stratified jitter from
    # the device generator in the same order as without prefetch (nothing else draws between the two), so the training
    # trajectory is bit-identical either way (tests/test_hip_sampler.py).
    def _next_train(self, step: int):
        pre, self._ahead = getattr(self, "_ahead", None), None
        if pre is None or pre[0] != step:
            return self.datamanager.next_train(step)
        _, ray_bundle, batch, ready = pre
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ready)
        for t in _tensors_of(ray_bundle) + _tensors_of(batch):
            t.record_stream(main)
        return ray_bundle, batch

    def _prefetch_next(self, step: int, sampled) -> None:
        if (sampled is None or self.device.type != "cuda" or os.environ.get("UMHS_PREFETCH_MARCH", "1") == "0" or not self._model.training
                or self._model.occupancy_update_due(step)):
            return
        side = getattr(self, "_ahead_stream", None)
        if side is None:
            side = self._ahead_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(side):
            side.wait_event(sampled)
            ray_bundle, batch = self.datamanager.next_train(step)
            self._model.prefetch_sample(ray_bundle)
            ready = torch.cuda.Event()
            ready.record(side)
        self._ahead = (step, ray_bundle, batch, ready)

    @torch.no_grad()
    def get_eval_loss_dict(self, step: int):
        self.eval()
        ray_bundle, batch = self.datamanager.next_eval(step)
        outputs = self._model(ray_bundle)
        metrics_dict = self._model.get_metrics_dict(outputs, batch)
        loss_dict = self._model.get_loss_dict(outputs, batch, metrics_dict)
        self.train()
        return outputs, loss_dict, metrics_dict

    @torch.no_grad()
    def get_eval_image_metrics_and_images(self, step: int):
        self.eval()
        camera_ray_bundle, batch = self.datamanager.next_eval_image(step)
        outputs = self._model.get_outputs_for_camera_ray_bundle(camera_ray_bundle)
        metrics_dict, images_dict = self._model.get_image_metrics_and_images(outputs, batch)
        metrics_dict["num_rays"] = int(camera_ray_bundle.origins.shape[0] * camera_ray_bundle.origins.shape[1])
        self.train()
        return metrics_dict, images_dict


def _tensors_of(obj) -> list:
    """Every CUDA tensor reachable from a batch dict / RayBundle (for stream bookkeeping)."""
    if torch.is_tensor(obj):
        return [obj] if obj.is_cuda else []
    if isinstance(obj, dict):
        return [t for v in obj.values() for t in _tensors_of(v)]
    if isinstance(obj, (list, tuple)):
        return [t for v in obj for t in _tensors_of(v)]
    if hasattr(obj, "__dict__"):
        return [t for v in vars(obj).values() for t in _tensors_of(v)]
    return []


def make_nerfstudio_trainer_config(defaults):  # pragma: no cover - needs nerfstudio
    raise NotImplementedError("nerfstudio TrainerConfig wiring is exercised only where nerfstudio is installed")
