from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Optional, Tuple, Type


class PrintableConfig:
    pass


@dataclass
class InstantiateConfig(PrintableConfig):
    """Config class for instantiating the class specified in the _target attribute."""

    _target: Type

    def setup(self, **kwargs) -> Any:
        return self._target(self, **kwargs)


@dataclass
class MachineConfig(PrintableConfig):
    seed: int = 42
    num_devices: int = 1
    num_machines: int = 1
    machine_rank: int = 0
    dist_url: str = "auto"
    device_type: str = "cuda"


@dataclass
class ViewerConfig(PrintableConfig):
    relative_log_filename: str = "viewer_log_filename.txt"
    websocket_port: Optional[int] = None
    websocket_port_default: int = 7007
    websocket_host: str = "0.0.0.0"
    num_rays_per_chunk: int = 32768
    max_num_display_images: int = 512
    quit_on_train_completion: bool = False
