"""The nerfstudio types the plugin surface touches.  With nerfstudio importable the real classes are used: ray/sample
containers, ``FieldHeadNames`` and the config / model / pipeline base classes, so that ``UMHSConfig`` is a ``ModelConfig``,
``UMHSPipelineConfig`` a ``VanillaPipelineConfig`` and ``umhs_config.umhs_method`` a real ``MethodSpecification``.  Without it
(it is not installable offline) minimal stand-ins with the same constructor / ``setup()`` behaviour keep the hot path and its
tests independent of nerfstudio.  What has and has not been exercised against real nerfstudio is listed in INTEGRATION.md."""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Optional, Type

import torch
from torch import Tensor, nn

try:  # pragma: no cover - exercised only where nerfstudio is installed
    from nerfstudio.cameras.rays import Frustums, RayBundle, RaySamples  # type: ignore
    from nerfstudio.field_components.field_heads import FieldHeadNames  # type: ignore

    HAVE_NERFSTUDIO = True
except Exception:  # ModuleNotFoundError offline
    HAVE_NERFSTUDIO = False

    class FieldHeadNames(Enum):
        RGB = "rgb"
        DENSITY = "density"

    @dataclass
    class Frustums:
        origins: Tensor  # [N,3]
        directions: Tensor  # [N,3]
        starts: Tensor  # [N,1]
        ends: Tensor  # [N,1]
        pixel_area: Optional[Tensor] = None
        offsets: Optional[Tensor] = None

        @property
        def shape(self):
            return self.origins.shape[:-1]

        def get_positions(self) -> Tensor:
            pos = self.origins + self.directions * (self.starts + self.ends) / 2
            return pos if self.offsets is None else pos + self.offsets

    @dataclass
    class RaySamples:
        frustums: Frustums
        camera_indices: Optional[Tensor] = None
        deltas: Optional[Tensor] = None
        metadata: Optional[dict] = None

    @dataclass
    class RayBundle:
        origins: Tensor  # [R,3]
        directions: Tensor  # [R,3]
        pixel_area: Optional[Tensor] = None
        camera_indices: Optional[Tensor] = None
        nears: Optional[Tensor] = None
        fars: Optional[Tensor] = None
        metadata: Optional[dict] = None

        def __len__(self):
            return self.origins.shape[0]


try:  # base classes of the plugin surface
    from nerfstudio.configs.base_config import InstantiateConfig  # type: ignore
    from nerfstudio.models.base_model import Model as ModelBase, ModelConfig as ModelConfigBase  # type: ignore
    from nerfstudio.pipelines.base_pipeline import VanillaPipeline as PipelineBase  # type: ignore
    from nerfstudio.pipelines.base_pipeline import VanillaPipelineConfig as PipelineConfigBase  # type: ignore

    HAVE_NERFSTUDIO_BASES = True
except Exception:  # ModuleNotFoundError offline
    HAVE_NERFSTUDIO_BASES = False

    @dataclass
    class InstantiateConfig:
        """nerfstudio.configs.base_config.InstantiateConfig: ``setup(**kwargs)`` instantiates ``_target(self, **kwargs)``."""

        _target: Type = None

        def setup(self, **kwargs) -> Any:
            return self._target(self, **kwargs)

    @dataclass
    class ModelConfigBase(InstantiateConfig):
        pass

    @dataclass
    class PipelineConfigBase(InstantiateConfig):
        pass

    class ModelBase(nn.Module):
        pass

    class PipelineBase(nn.Module):
        pass


try:
    from nerfstudio.engine.callbacks import TrainingCallback, TrainingCallbackAttributes, TrainingCallbackLocation  # type: ignore
except Exception:

    class TrainingCallbackLocation(Enum):
        BEFORE_TRAIN_ITERATION = "before_train_iteration"
        AFTER_TRAIN_ITERATION = "after_train_iteration"
        AFTER_TRAIN = "after_train"

    TrainingCallbackAttributes = Any

    class TrainingCallback:
        """nerfstudio.engine.callbacks.TrainingCallback: ``func(step)`` at the given locations every ``update_every_num_iters``."""

        def __init__(self, where_to_run, func, update_every_num_iters=None, iters=None, args=None, kwargs=None):
            self.where_to_run, self.func, self.update_every_num_iters, self.iters = where_to_run, func, update_every_num_iters, iters
            self.args, self.kwargs = list(args or []), dict(kwargs or {})

        def run_callback(self, step: int) -> None:
            if self.update_every_num_iters is not None:
                if step % self.update_every_num_iters == 0:
                    self.func(*self.args, **self.kwargs, step=step)
            elif self.iters is not None:
                if step in self.iters:
                    self.func(*self.args, **self.kwargs, step=step)

        def run_callback_at_location(self, step: int, location) -> None:
            if location in self.where_to_run:
                self.run_callback(step=step)


def packed_ray_samples(origins, directions, starts, ends, camera_indices=None) -> "RaySamples":
    n = origins.shape[0]
    if camera_indices is None:
        camera_indices = torch.zeros((n, 1), dtype=torch.long, device=origins.device)
    fr = Frustums(origins=origins, directions=directions, starts=starts.view(n, 1), ends=ends.view(n, 1),
                  pixel_area=torch.ones((n, 1), device=origins.device))
    return RaySamples(frustums=fr, camera_indices=camera_indices)
