"""Spectral renderer (mirror of the reference's ``umhsnerf/umhs_renderer.py``): packed per-ray accumulation of
B-band radiance on the HIP kernels, plus the background-blending helpers and the dense weight function."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import Tensor, nn

from . import ops

COLORS_DICT = {
    "white": torch.tensor([1.0, 1.0, 1.0]), "black": torch.tensor([0.0, 0.0, 0.0]), "red": torch.tensor([1.0, 0.0, 0.0]),
    "green": torch.tensor([0.0, 1.0, 0.0]), "blue": torch.tensor([0.0, 0.0, 1.0]),
}


class SpectralRenderer(nn.Module):
    """Calculate spectral radiance along the ray (``umhs_renderer.py:10-30``)."""

    background_color: Optional[str] = "random"

    @classmethod
    def forward(cls, spectral: Tensor, weights: Tensor, ray_indices: Optional[Tensor] = None,
                num_rays: Optional[int] = None, packed_info: Optional[Tensor] = None) -> Tensor:
        """spectral [*,N,K] (a leading 1-dim is squeezed, as the reference does), weights [N,1] -> [R,K].
        Without ray_indices the samples of ONE ray/batch are summed over dim -2 (nerfacc's dense branch)."""
        if spectral.dim() == 3:
            spectral = spectral.squeeze(0)
        elif spectral.dim() == 1:
            spectral = spectral.unsqueeze(0)
        if ray_indices is None:
            n = spectral.shape[-2]
            packed_info = torch.tensor([[0, n]], dtype=torch.int64, device=spectral.device)
            return ops.AccumulateFn.apply(weights[..., 0], spectral, packed_info)[0]
        assert num_rays is not None
        if packed_info is None:
            packed_info = ops.pack_info(ray_indices, num_rays)
        return ops.AccumulateFn.apply(weights[..., 0], spectral, packed_info)

    def __call__(self, *args, **kwargs):  # classmethod forward, like nerfstudio's renderers
        return type(self).forward(*args, **kwargs)

    @classmethod
    def get_background_color(cls, background_color, shape: Tuple[int, ...], device) -> Tensor:
        assert background_color not in {"last_sample", "random"}
        if isinstance(background_color, str) and background_color in COLORS_DICT:
            background_color = COLORS_DICT[background_color]
        assert isinstance(background_color, Tensor)
        return background_color.expand(shape).to(device)

    def blend_background(self, image: Tensor, rgba: Tensor, background_color=None) -> Tensor:
        """``umhs_renderer.py:58-86``: no-op unless ``rgba`` carries an alpha channel."""
        if rgba.size(-1) < 4:
            return image
        opacity = rgba[..., 3:]
        if background_color is None:
            background_color = self.background_color
            if background_color in {"last_sample", "random"}:
                background_color = "black"
        background_color = self.get_background_color(background_color, shape=image.shape, device=image.device)
        return image * opacity + background_color.to(image.device) * (1 - opacity)

    def blend_background_for_loss_computation(self, pred_image: Tensor, pred_accumulation: Tensor, gt_image: Tensor,
                                              rgba_image: Tensor) -> Tuple[Tensor, Tensor]:
        """``umhs_renderer.py:89-114``: random background added to the prediction where accumulation < 1."""
        background_color = self.background_color
        if background_color == "last_sample":
            background_color = "black"
        elif background_color == "random":
            background_color = torch.rand_like(pred_image)
            pred_image = pred_image + background_color * (1.0 - pred_accumulation)
        gt_image = self.blend_background(gt_image, rgba_image, background_color=background_color)
        return pred_image, gt_image


def get_weights_spectral(deltas: Tensor, densities: Tensor) -> Tensor:
    """Dense-layout weights [..., S, 1] (``umhs_renderer.py:117-139``) through the packed transmittance kernel."""
    S = deltas.shape[-2]
    d = deltas.reshape(-1)
    R = d.numel() // S
    pinfo = torch.stack([torch.arange(R, device=d.device) * S, torch.full((R,), S, device=d.device)], -1).to(torch.int64)
    w = ops.CompositeFn.apply(densities.reshape(-1, 1), torch.zeros_like(d), d, pinfo, False)[0]
    return torch.nan_to_num(w.view(densities.shape))
