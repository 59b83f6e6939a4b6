from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Type

from torch import nn

from nerfstudio.configs.base_config import InstantiateConfig


class Pipeline(nn.Module):
    @property
    def model(self):
        return self._model

    @property
    def device(self):
        return self.model.device


@dataclass
class VanillaPipelineConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: VanillaPipeline)
    datamanager: Any = None
    model: Any = None


class VanillaPipeline(Pipeline):
    def __init__(self, config, device, test_mode="val", world_size=1, local_rank=0, grad_scaler=None):
        super().__init__()
        self.config, self.test_mode = config, test_mode
