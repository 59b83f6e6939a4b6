"""Hyperspectral frames (mirror of ``umhsnerf/data/utils/hs_dataloader.py``): ``hyperspectral_file_path`` points to an
``.npy`` cube H x W x B; values are converted to float32 and clamped to [0, 1] (``:49-50``).  The VCA endmember
initialisation the reference triggers from here (``:52-58``) is an initialiser, not on the hot path: pass ``load_vca``
endmembers to the field directly."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch


def load_hs_image(path) -> torch.Tensor:
    cube = np.load(path)  # H, W, B
    if cube.ndim != 3:
        raise ValueError(f"{path}: expected an H x W x B cube, got shape {cube.shape}")
    return torch.from_numpy(np.ascontiguousarray(cube)).float().clamp(0, 1)


def load_image(path) -> torch.Tensor:
    """RGB(A) frame as float32 in [0,1] (InputDataset.get_image_float32); ``.npy`` or anything PIL opens."""
    path = str(path)
    if path.endswith(".npy"):
        arr = np.load(path)
    else:
        from PIL import Image

        arr = np.array(Image.open(path))
    if arr.ndim == 2:
        arr = np.repeat(arr[:, :, None], 3, axis=2)
    if arr.dtype == np.uint8:
        return torch.from_numpy(arr.astype(np.float32) / 255.0)
    return torch.from_numpy(arr.astype(np.float32))


def stack_frames(frames: Sequence[torch.Tensor]) -> torch.Tensor:
    """[n,H,W,K] contiguous stack -- the layout ``umhs_pixel_gather`` reads (one K-float row per pixel)."""
    shapes = {tuple(f.shape) for f in frames}
    if len(shapes) != 1:
        raise ValueError(f"frames differ in shape: {sorted(shapes)} (the resident stack needs one H x W x K)")
    return torch.stack(list(frames)).contiguous()


class HyperspectralDataset:
    """image + hs_image per frame, read once and kept (``--images-on-gpu``)."""

    def __init__(self, outputs, device="cpu"):
        if not outputs.metadata.get("hs_filenames"):
            raise AssertionError("hs_filenames missing: every frame needs hyperspectral_file_path")
        self.outputs, self.cameras, self.metadata = outputs, outputs.cameras, outputs.metadata
        self.image = stack_frames([load_image(p) for p in outputs.image_filenames]).to(device)
        self.hs_image = stack_frames([load_hs_image(p) for p in outputs.metadata["hs_filenames"]]).to(device)
        if self.image.shape[:3] != self.hs_image.shape[:3]:
            raise ValueError(f"image {tuple(self.image.shape)} and hs_image {tuple(self.hs_image.shape)} differ in n/H/W")

    def __len__(self) -> int:
        return self.image.shape[0]

    def __getitem__(self, i: int):
        return {"image_idx": i, "image": self.image[i], "hs_image": self.hs_image[i]}
