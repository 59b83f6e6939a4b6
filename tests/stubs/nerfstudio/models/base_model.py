from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Type

import torch
from torch import nn

from nerfstudio.configs.base_config import InstantiateConfig


@dataclass
class ModelConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: Model)
    enable_collider: bool = True
    collider_params: Optional[Dict[str, float]] = None
    loss_coefficients: Dict[str, float] = field(default_factory=lambda: {"rgb_loss_coarse": 1.0, "rgb_loss_fine": 1.0})
    eval_num_rays_per_chunk: int = 4096
    prompt: Optional[str] = None


class Model(nn.Module):
    config: ModelConfig

    def __init__(self, config: ModelConfig, scene_box: Any, num_train_data: int, **kwargs) -> None:
        super().__init__()
        self.config, self.scene_box, self.render_aabb, self.num_train_data, self.kwargs = config, scene_box, None, num_train_data, kwargs
        self.collider = None
        self.populate_modules()
        self.callbacks = None
        self.device_indicator_param = nn.Parameter(torch.empty(0))

    @property
    def device(self):
        return self.device_indicator_param.device

    def populate_modules(self):
        pass
