from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Literal, Optional, Type

from nerfstudio.configs.base_config import InstantiateConfig, MachineConfig, ViewerConfig


@dataclass
class TrainerConfig(InstantiateConfig):
    """ExperimentConfig + TrainerConfig fields the method registration sets or the scripts override."""

    _target: Type = field(default_factory=lambda: Trainer)
    method_name: Optional[str] = None
    experiment_name: Optional[str] = None
    machine: MachineConfig = field(default_factory=MachineConfig)
    viewer: ViewerConfig = field(default_factory=ViewerConfig)
    pipeline: Any = None
    optimizers: Dict[str, Any] = field(default_factory=dict)
    vis: str = "wandb"
    steps_per_save: int = 1000
    steps_per_eval_batch: int = 500
    steps_per_eval_image: int = 500
    steps_per_eval_all_images: int = 25000
    max_num_iterations: int = 1000000
    mixed_precision: bool = False
    use_grad_scaler: bool = False
    save_only_latest_checkpoint: bool = True
    log_gradients: bool = False
    gradient_accumulation_steps: Dict[str, int] = field(default_factory=dict)


class Trainer:
    pass
