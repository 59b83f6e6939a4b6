from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Mapping, Optional, Type

from torch import nn

from nerfstudio.configs.base_config import InstantiateConfig


class Pipeline(nn.Module):
    @property
    def model(self):
        return self._model

    @property
    def device(self):
        return self.model.device

    def load_state_dict(self, state_dict: Mapping[str, Any], strict: Optional[bool] = None):
        """Restated from nerfstudio 1.1.5: the "_model." keys go to the model (strictly first), the rest to nn.Module's loader with
        strict=False -- and the method returns None, not nn.Module's (missing, unexpected) result."""
        model_state = {k[len("_model."):]: v for k, v in state_dict.items() if k.startswith("_model.")}
        if model_state and all(k.startswith("module.") for k in model_state):  # DDP's prefix
            model_state = {k[len("module."):]: v for k, v in model_state.items()}
        rest = {k: v for k, v in state_dict.items() if not k.startswith("_model.")}
        try:
            self.model.load_state_dict(model_state, strict=True)
        except RuntimeError:
            if strict:
                raise
            self.model.load_state_dict(model_state, strict=False)
        super().load_state_dict(rest, strict=False)


@dataclass
class VanillaPipelineConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: VanillaPipeline)
    datamanager: Any = None
    model: Any = None


class VanillaPipeline(Pipeline):
    def __init__(self, config, device, test_mode="val", world_size=1, local_rank=0, grad_scaler=None):
        super().__init__()
        self.config, self.test_mode = config, test_mode
