from __future__ import annotations

from dataclasses import dataclass, field
from typing import Literal, Optional, Type

import numpy as np
from torch.optim import Optimizer, lr_scheduler

from nerfstudio.configs.base_config import InstantiateConfig


@dataclass
class SchedulerConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: Scheduler)


class Scheduler:
    def __init__(self, config) -> None:
        self.config = config


@dataclass
class ExponentialDecaySchedulerConfig(SchedulerConfig):
    _target: Type = field(default_factory=lambda: ExponentialDecayScheduler)
    lr_pre_warmup: float = 1e-8
    lr_final: Optional[float] = None
    warmup_steps: int = 0
    max_steps: int = 100000
    ramp: Literal["linear", "cosine"] = "cosine"


class ExponentialDecayScheduler(Scheduler):
    def get_scheduler(self, optimizer: Optimizer, lr_init: float):
        lr_final = lr_init if self.config.lr_final is None else self.config.lr_final

        def func(step):
            if step < self.config.warmup_steps:
                if self.config.ramp == "cosine":
                    lr = self.config.lr_pre_warmup + (lr_init - self.config.lr_pre_warmup) * np.sin(
                        0.5 * np.pi * np.clip(step / self.config.warmup_steps, 0, 1))
                else:
                    lr = self.config.lr_pre_warmup + (lr_init - self.config.lr_pre_warmup) * step / self.config.warmup_steps
            else:
                t = np.clip((step - self.config.warmup_steps) / (self.config.max_steps - self.config.warmup_steps), 0, 1)
                lr = np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
            return lr / lr_init  # divided by lr_init because the multiplier is with the initial learning rate

        return lr_scheduler.LambdaLR(optimizer, lr_lambda=func)
