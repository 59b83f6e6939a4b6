from dataclasses import dataclass

from nerfstudio.engine.trainer import TrainerConfig


@dataclass
class MethodSpecification:
    """Method specification class used to register custom methods with Nerfstudio."""

    config: TrainerConfig
    description: str
