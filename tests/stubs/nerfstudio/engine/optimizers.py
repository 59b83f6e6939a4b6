from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Type

import torch

from nerfstudio.configs.base_config import PrintableConfig


@dataclass
class OptimizerConfig(PrintableConfig):
    _target: Type = torch.optim.Adam
    lr: float = 0.0005
    eps: float = 1e-08
    max_norm: Optional[float] = None

    def setup(self, params) -> torch.optim.Optimizer:
        kwargs = vars(self).copy()
        kwargs.pop("_target")
        kwargs.pop("max_norm")
        return self._target(params, **kwargs)


@dataclass
class AdamOptimizerConfig(OptimizerConfig):
    _target: Type = torch.optim.Adam
    weight_decay: float = 0


class Optimizers:
    """A set of optimizers (+ schedulers), one per parameter group name."""

    def __init__(self, config: Dict[str, Any], param_groups: Dict[str, List[torch.nn.Parameter]]) -> None:
        self.config, self.optimizers, self.schedulers, self.parameters = config, {}, {}, {}
        for name, params in param_groups.items():
            lr_init = config[name]["optimizer"].lr
            self.optimizers[name] = config[name]["optimizer"].setup(params=params)
            self.parameters[name] = params
            if config[name]["scheduler"]:
                self.schedulers[name] = config[name]["scheduler"].setup().get_scheduler(optimizer=self.optimizers[name], lr_init=lr_init)

    def zero_grad_some(self, names: List[str]) -> None:
        for n in names:
            self.optimizers[n].zero_grad()

    def optimizer_scaler_step_some(self, grad_scaler, names: List[str]) -> None:
        for n in names:
            max_norm = self.config[n]["optimizer"].max_norm
            if max_norm is not None:
                grad_scaler.unscale_(self.optimizers[n])
                torch.nn.utils.clip_grad_norm_(self.parameters[n], max_norm)
            if any(any(p.grad is not None for p in g["params"]) for g in self.optimizers[n].param_groups):
                grad_scaler.step(self.optimizers[n])

    def scheduler_step_all(self, step: int) -> None:
        for s in self.schedulers.values():
            s.step()
