"""Pipeline (mirror of ``umhsnerf/umhs_pipeline.py``): ``UMHSPipelineConfig`` (``:30-45``), ``UMHSPipeline`` with the
reference's constructor ``(config, device, test_mode, world_size, local_rank, grad_scaler)`` (``:62-113``), its eval entry
points (``:115-154``) and ``load_pipeline`` (``:157-175``).

What differs by design: the reference forces ``world_size = 1`` (``:86,108-109``), which silently disables its DDP wrap; here
ray batches are sharded one rank per GPU and the flat "fields" gradient is all-reduced over RCCL/xGMI by the gradient sink
(``parallel.FlatGradSink``) while the backward is still running, so there is no DDP wrapper around the model."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Dict, Literal, Mapping, Optional, Type

import torch
import torch.distributed as dist

from ._ns_compat import PipelineBase, PipelineConfigBase, RaySamples
from .data.umhs_datamanager import UMHSDataManagerConfig
from .umhs_model import UMHSConfig, UMHSModel


@dataclass
class UMHSPipelineConfig(PipelineConfigBase):
    """``UMHSPipelineConfig(VanillaPipelineConfig)``, umhs_pipeline.py:30-45 (same names and defaults)."""

    _target: Type = field(default_factory=lambda: UMHSPipeline)
    datamanager: Any = field(default_factory=UMHSDataManagerConfig)
    model: Any = field(default_factory=UMHSConfig)
    check_nan: bool = False
    num_classes: int = 5
    gradient_accumulation_steps: int = 1
    """Not in the reference (its scripts pass ``--gradient-accumulation_steps`` to the trainer, scripts/rgb+spectral.sh:5): set it to
    the trainer's value when running on more than one GPU, so that the gradient exchange starts in the LAST micro-step's backward."""


class _DepositedGrad(torch.autograd.Function):
    """Loss values whose summed gradient w.r.t. the flat parameter has ALREADY been written to ``param.grad`` by the launch-sequence
    step (UMHSModel.forward_backward_from_samples).  nerfstudio's Trainer calls ``backward()`` on the sum of what get_train_loss_dict
    returns (through ``grad_scaler.scale``); this turns that call into the one thing left to do: what was deposited is the gradient of
    the PLAIN sum of the losses, so an upstream gradient g != 1 -- a GradScaler's loss scale, a division by the accumulation steps --
    multiplies the step's contribution to ``param.grad`` by g (once per step: every loss of the step must arrive with the same g; a
    trainer that weights the losses differently needs the autograd path, UMHS_DIRECT_STEP=0, and gets an error here, not a silently
    wrong step).  Reading g is one host sync per loss, so it is skipped where g is known to be 1: a pipeline whose trainer handed it
    no grad scaler or a disabled one (the shipped TrainerConfig: mixed_precision False) and that does no gradient accumulation
    (``step_state["unit_scale"]``), and under UMHS_TRUST_UNIT_LOSS_SCALE=1."""

    @staticmethod
    def forward(ctx, value, flat, step_state):
        ctx.flat, ctx.step_state = flat, step_state
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.step_state.get("unit_scale") or os.environ.get("UMHS_TRUST_UNIT_LOSS_SCALE", "0") == "1":
            return None, None, None
        st, gv = ctx.step_state, float(g)
        if "g" not in st:
            st["g"] = gv
            if gv != 1.0:
                if st.get("accumulated"):  # earlier micro-steps of the window are in the same buffer: their share must not be rescaled
                    raise RuntimeError(f"loss scaled by {gv:g} inside a gradient-accumulation window: the deposited gradient cannot be "
                                       "rescaled per micro-step (run with UMHS_DIRECT_STEP=0)")
                sink = st.get("sink")
                if sink is not None and (sink.reducing() or sink.works):
                    # the backward has already handed segments of this buffer to asynchronous all-reduces: an in-place multiply
                    # would race with the transfers and leave the ranks with differently scaled sums
                    raise RuntimeError(f"loss scaled by {gv:g} while the gradient exchange of this step is in flight: a loss scale on "
                                       "more than one rank needs the autograd path (UMHS_DIRECT_STEP=0)")
                ctx.flat.grad.mul_(gv)
        elif st["g"] != gv:
            raise RuntimeError(f"the trainer weights this step's losses differently ({st['g']:g} vs {gv:g}): the launch-sequence step has "
                               "deposited the gradient of their plain sum (run with UMHS_DIRECT_STEP=0)")
        return None, None, None


class UMHSPipeline(PipelineBase):
    """UMHS pipeline.  Built by nerfstudio's Trainer through ``config.pipeline.setup(device=..., test_mode=..., world_size=...,
    local_rank=..., grad_scaler=...)`` -- the trainer then owns zero_grad / backward / optimizer.step (``trainer_driven``) -- or
    by ``from_packed_samples`` (benchmark, tests), which owns a fused-Adam optimizer and steps it itself."""

    def __init__(self, config: UMHSPipelineConfig, device, test_mode: Literal["test", "val", "inference"] = "val", world_size: int = 1,
                 local_rank: int = 0, grad_scaler=None):
        torch.nn.Module.__init__(self)  # like the reference's super(VanillaPipeline, self).__init__(): no base-class construction
        self.config = config
        if config.check_nan:
            torch.autograd.set_detect_anomaly(True)
        self.test_mode = test_mode
        datamanager = config.datamanager.setup(device=device, test_mode=test_mode, world_size=world_size, local_rank=local_rank,
                                               num_classes=config.num_classes)
        if hasattr(datamanager, "to"):
            datamanager.to(device)
        assert datamanager.train_dataset is not None, "Missing input dataset"
        model = config.model.setup(
            scene_box=datamanager.train_dataset.scene_box, num_train_data=len(datamanager.train_dataset),
            metadata=datamanager.train_dataset.metadata, grad_scaler=grad_scaler, num_classes=config.num_classes,
            wavelengths=datamanager.train_dataparser_outputs.metadata.get("wavelengths", None))
        self.grad_scaler = grad_scaler  # (None or disabled: the trainer's backward() arrives unscaled, see _deposit)
        self._init_common(model.to(device), datamanager, device, world_size, local_rank, trainer_driven=True,
                          gradient_accumulation_steps=config.gradient_accumulation_steps)

    @classmethod
    def from_packed_samples(cls, config: UMHSConfig, device, metadata: Optional[Dict] = None, world_size: int = 1, local_rank: int = 0,
                            seed: Optional[int] = 42, scene_box=None, datamanager=None, gradient_accumulation_steps: int = 1):
        """Model config + metadata -> pipeline that owns its optimizer (``train_iteration`` / ``get_train_loss_dict`` step it).
        ``datamanager``: a ``UMHSDataManager`` or None (callers that feed packed samples: bench.py, the parity tests)."""
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        self.config = UMHSPipelineConfig(model=config, num_classes=int((metadata or {}).get("num_classes", 5)),
                                         gradient_accumulation_steps=gradient_accumulation_steps)
        self.test_mode = "val"
        self.grad_scaler = None
        if datamanager is not None:
            metadata = {**(datamanager.metadata or {}), **(metadata or {})}
            scene_box = scene_box if scene_box is not None else getattr(datamanager, "scene_box", None)
        model = UMHSModel(config, scene_box=scene_box, metadata=metadata, seed=seed).to(device)
        self._init_common(model, datamanager, device, world_size, local_rank, trainer_driven=False,
                          gradient_accumulation_steps=gradient_accumulation_steps)
        return self

    def _init_common(self, model, datamanager, device, world_size, local_rank, trainer_driven, gradient_accumulation_steps):
        self.world_size, self.local_rank = world_size, local_rank
        self._device = torch.device(device)  # (nerfstudio's Pipeline.device is a read-only property: own attribute + property below)
        self.datamanager = datamanager
        self._model = model
        self.trainer_driven = trainer_driven
        self.gradient_accumulation_steps = max(1, int(gradient_accumulation_steps))
        self._micro = 0  # micro-steps accumulated since the last optimizer step (train_iteration bookkeeping)
        if world_size > 1:  # identical parameters on every rank (DDP's initial broadcast)
            dist.broadcast(self._model.field.flat.data, src=0)
        self._model.field.use_grad_sink = True  # backward fills param.grad in place and reduces finished segments early
        self.optimizer = None if trainer_driven else self._model.make_optimizer()

    @property
    def model(self) -> UMHSModel:
        return self._model

    @property
    def device(self) -> torch.device:
        return self._device

    # ---- micro-step bookkeeping (gradient accumulation) --------------------------------------------------------------
    def _begin_micro_step(self, micro: int) -> bool:
        """Called in front of a forward/backward: ``micro`` = index inside the accumulation window.  Zeroes the gradient at the
        window's start when this pipeline owns the optimizer (the nerfstudio trainer does it itself, Optimizers.zero_grad_some) and
        tells the gradient sink whether this backward is the one that exchanges gradients.  Returns "last micro-step"."""
        last = micro == self.gradient_accumulation_steps - 1
        if not self.trainer_driven and micro == 0:
            self.optimizer.zero_grad(set_to_none=True)
        sink = self._model.field._spec().grad_sink
        sink.defer_reduce = not last
        if last and not self.trainer_driven and self.gradient_accumulation_steps == 1:
            self.optimizer.arm_fused()  # optimizer.step() follows unconditionally: the table's Adam step may ride in the backward
        return last

    def _deposit(self, loss_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Trainer-driven mode: the trainer will call ``.backward()`` on the summed losses -- the gradient is in ``param.grad`` already."""
        flat = self._model.field.flat
        sink = self._model.field._spec().grad_sink
        scaler = getattr(self, "grad_scaler", None)
        scaler_on = scaler is not None and (scaler.is_enabled() if hasattr(scaler, "is_enabled") else True)
        # no scaler (or a disabled one) and no accumulation window: the trainer's backward() arrives with g = 1 -- nothing to read back
        unit = (not scaler_on) and self.gradient_accumulation_steps == 1 and os.environ.get("UMHS_CHECK_LOSS_SCALE", "0") != "1"
        step_state = {"accumulated": sink.accumulating, "sink": sink, "unit_scale": unit}
        return {k: _DepositedGrad.apply(v, flat, step_state) for k, v in loss_dict.items()}

    def train_iteration(self, ray_samples: RaySamples, ray_indices, num_rays: int, batch: Dict, packed_info=None, background=None):
        """One micro-step on caller-provided packed samples (bench.py, parity tests): forward, losses, backward and -- on the last
        micro-step of the accumulation window -- the optimizer step.  Needs ``from_packed_samples`` (an own optimizer).
        ``background``: the [R,3] draw of the random-background blend of the rgb loss (a parity test hands the oracle's draw in;
        None = drawn from the device generator, as nerfstudio's renderer does)."""
        if self.trainer_driven:
            raise RuntimeError("train_iteration() steps the pipeline's own optimizer; a trainer-built pipeline is driven through "
                               "get_train_loss_dict()")
        last = self._begin_micro_step(self._micro)
        if self._model.direct_step_supported(batch):  # straight launch sequence, no autograd graph
            outputs, loss_dict = self._model.forward_backward_from_samples(ray_samples, ray_indices, num_rays, batch, packed_info,
                                                                           background=background)
        else:
            outputs = self._model.get_outputs_from_samples(ray_samples, ray_indices, num_rays, packed_info)
            loss_dict = self._model.get_loss_dict(outputs, batch, background=background)
            sum(loss_dict.values()).backward()
        self._micro = 0 if last else self._micro + 1
        if last:
            self.optimizer.step()
        return outputs, loss_dict

    # ---- the reference's pipeline surface (umhs_pipeline.py:115-150 and VanillaPipeline.get_train_loss_dict) -----------
    def get_train_loss_dict(self, step: int):
        """``VanillaPipeline.get_train_loss_dict``: next_train, forward, losses, metrics -> (outputs, loss_dict, metrics_dict).

        Trainer-driven (built by nerfstudio's Trainer): the trainer has run the BEFORE_TRAIN_ITERATION callbacks (occupancy grid)
        and zeroed the gradients, and will call ``backward()`` on the summed losses and step its optimizers.  The launch-sequence
        step below has then already left the gradient in ``param.grad`` (``_DepositedGrad`` makes the trainer's backward a no-op);
        the general autograd path returns ordinary differentiable losses.
        Stand-alone (``from_packed_samples``): this call is the whole ``Trainer.train_iteration`` -- callbacks, backward, fused
        Adam + clamp (+ gradient exchange) included."""
        if not self.trainer_driven:
            self._model.update_occupancy_grid(step)
        ray_bundle, batch = self._next_train(step)
        last = self._begin_micro_step(step % self.gradient_accumulation_steps)
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
            outputs, loss_dict = self._model.forward_backward_from_samples(ray_samples, ray_indices, len(ray_bundle), batch,
                                                                           background=ahead["bg"])
            metrics_dict = self._model.get_metrics_dict(outputs, batch)
            if self.trainer_driven:
                loss_dict = self._deposit(loss_dict)
        else:
            outputs = self._model(ray_bundle)
            metrics_dict = self._model.get_metrics_dict(outputs, batch)
            loss_dict = self._model.get_loss_dict(outputs, batch, metrics_dict)
            if not self.trainer_driven:
                sum(loss_dict.values()).backward()
        if last and not self.trainer_driven:
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

    def load_pipeline(self, loaded_state: Mapping[str, Any], step: int) -> None:
        """umhs_pipeline.py:157-175: strip DDP's ``module.`` prefix, ``model.update_to_step(step)``, ``load_state_dict``.  Checkpoints
        carry the reference's key names (``_model.field.mlp_base.encoder.hash_table`` ...).  Keys of modules this build does not hold
        (the reference's lpips network, ``field.mlp_base`` of the discarded NGP parent ...) are ignored; a missing field key is an
        error.  (The reference's debugging side effects -- printing the dict, np.save of the endmembers -- are not reproduced.)"""
        state = _strip_module_prefix(loaded_state)
        self._model.update_to_step(step)
        # nn.Module's loader, called explicitly: nerfstudio's Pipeline overrides load_state_dict (it splits off the "_model." keys,
        # loads the model strictly and returns None), and a checkpoint of the reference carries modules this build does not hold
        result = torch.nn.Module.load_state_dict(self, state, strict=False)
        missing = [k for k in result.missing_keys if ".field." in k or k.startswith("field.")]
        if missing:
            raise RuntimeError(f"checkpoint lacks field parameters: {missing}")

    def get_param_groups(self) -> Dict[str, list]:
        """VanillaPipeline.get_param_groups: datamanager groups (none here) + the model's ``{"fields": [...]}``."""
        return self._model.get_param_groups()

    def get_training_callbacks(self, training_callback_attributes=None) -> list:
        # the trainer's --gradient-accumulation_steps (scripts/rgb+spectral.sh:5) lives on the Trainer, keyed by parameter group:
        # read it here, so that on more than one rank only the window's last micro-step exchanges gradients without a second flag
        steps = getattr(getattr(training_callback_attributes, "trainer", None), "gradient_accumulation_steps", None)
        if steps is not None:
            try:
                self.gradient_accumulation_steps = max(1, int(steps["fields"]))
            except (KeyError, TypeError):
                pass
        return self._model.get_training_callbacks(training_callback_attributes)

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


def _strip_module_prefix(loaded_state: Mapping[str, Any]) -> Dict[str, Any]:
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in loaded_state.items()}


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


