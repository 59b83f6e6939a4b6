from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, Literal, Optional, Type

from nerfstudio.configs.base_config import InstantiateConfig, MachineConfig, ViewerConfig
from nerfstudio.engine.callbacks import TrainingCallbackAttributes
from nerfstudio.engine.optimizers import Optimizers


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
    """What nerfstudio 1.1.5's Trainer touches of a pipeline BEFORE the first iteration (restated): ``setup()`` builds the pipeline
    and the optimizers and collects the callbacks, ``train()`` saves the dataparser transform and hands the datasets to the viewer."""

    def __init__(self, config: TrainerConfig, local_rank: int = 0, world_size: int = 1, device: str = "cpu", base_dir: str = ".") -> None:
        self.config, self.local_rank, self.world_size, self.device = config, local_rank, world_size, device
        self.base_dir = Path(base_dir)
        self.gradient_accumulation_steps: Dict[str, int] = defaultdict(lambda: 1)
        self.gradient_accumulation_steps.update(config.gradient_accumulation_steps)
        self.grad_scaler = None

    def setup(self, test_mode: Literal["test", "val", "inference"] = "val") -> None:
        self.pipeline = self.config.pipeline.setup(device=self.device, test_mode=test_mode, world_size=self.world_size,
                                                   local_rank=self.local_rank, grad_scaler=self.grad_scaler)
        self.optimizers = Optimizers(self.config.optimizers.copy(), self.pipeline.get_param_groups())
        self.callbacks = self.pipeline.get_training_callbacks(
            TrainingCallbackAttributes(optimizers=self.optimizers, grad_scaler=self.grad_scaler, pipeline=self.pipeline, trainer=self))

    def train_prologue(self) -> Dict[str, Any]:
        """The lines of ``train()`` in front of the step loop."""
        dm = self.pipeline.datamanager
        if hasattr(dm, "train_dataparser_outputs"):
            dm.train_dataparser_outputs.save_dataparser_transform(self.base_dir / "dataparser_transforms.json")
        # viewer.init_scene(train_dataset=..., eval_dataset=...): one thumbnail + camera per training image
        seen = {"cameras": len(dm.train_dataset.cameras), "thumbnails": [tuple(dm.train_dataset[i]["image"].shape) for i in range(len(dm.train_dataset))],
                "names": len(dm.train_dataset.image_filenames)}
        if dm.eval_dataset is not None:
            seen["eval_cameras"] = len(dm.eval_dataset.cameras)
        return seen
