"""Minimal stand-ins for the nerfstudio types the hot path touches, used ONLY when nerfstudio itself is not
importable (it is not installable offline).  With nerfstudio present the real classes are used, so the
plugin registers and trains through the stock ``ns-train`` machinery."""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum
from typing import Optional

import torch
from torch import Tensor

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


def packed_ray_samples(origins, directions, starts, ends, camera_indices=None) -> "RaySamples":
    n = origins.shape[0]
    if camera_indices is None:
        camera_indices = torch.zeros((n, 1), dtype=torch.long, device=origins.device)
    fr = Frustums(origins=origins, directions=directions, starts=starts.view(n, 1), ends=ends.view(n, 1),
                  pixel_area=torch.ones((n, 1), device=origins.device))
    return RaySamples(frustums=fr, camera_indices=camera_indices)
