"""Spectrum -> sRGB projection (mirror of the reference's ``umhsnerf/utils/spec_to_rgb.py``).

The B x 3 matrix is host-side setup (same analytic CIE-1931 fit, ``spec_to_rgb.py:6-21,62-90``); the per-ray
``clamp(gamma(spec @ M), 0, 1)`` runs in ``umhs_spec2rgb_fwd/bwd`` (fp32, like the reference on CPU)."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops


def _g(x, alpha, mu, s1, s2):
    sigma = np.clip((x < mu) * s1 + (x >= mu) * s2, a_min=1e-6, a_max=None)
    return alpha * np.exp((x - mu) ** 2 / (-2 * (sigma**2)))


def _xyz(a, b):
    return np.array((a, b, 1 - a - b))


_D65, _E = _xyz(0.3127, 0.3291), _xyz(1 / 3, 1 / 3)
COLOR_SPACE = {
    "sRGB": (_xyz(0.64, 0.33), _xyz(0.30, 0.60), _xyz(0.15, 0.06), _D65),
    "AdobeRGB": (_xyz(0.64, 0.33), _xyz(0.21, 0.71), _xyz(0.15, 0.06), _D65),
    "AppleRGB": (_xyz(0.625, 0.34), _xyz(0.28, 0.595), _xyz(0.155, 0.07), _D65),
    "UHDTV": (_xyz(0.708, 0.292), _xyz(0.170, 0.797), _xyz(0.131, 0.046), _D65),
    "CIERGB": (_xyz(0.7347, 0.2653), _xyz(0.2738, 0.7174), _xyz(0.1666, 0.0089), _E),
}


class ColourSystem(nn.Module):
    def __init__(self, bands, cs="sRGB", device="cuda"):
        super().__init__()
        x = np.array(bands) * 10  # nm -> Angstrom
        cmf = np.array([
            _g(x, 1.056, 5998, 379, 310) + _g(x, 0.362, 4420, 160, 267) + _g(x, -0.065, 5011, 204, 262),
            _g(x, 0.821, 5688, 469, 405) + _g(x, 0.286, 5309, 163, 311),
            _g(x, 1.217, 4370, 118, 360) + _g(x, 0.681, 4590, 260, 138),
        ])
        red, green, blue, white = COLOR_SPACE[cs]
        MI = np.linalg.inv(np.vstack((red, green, blue)).T)
        A = MI / MI.dot(white)[:, np.newaxis]
        RGB = cmf.T @ A.T
        RGB = RGB / np.sum(RGB, axis=0, keepdims=True)
        self.register_buffer("transform_matrix", torch.from_numpy(RGB).float())

    def forward(self, spec):
        shape = spec.shape
        rgb = ops.Spec2RgbFn.apply(spec.reshape(-1, shape[-1]), self.transform_matrix)
        return rgb.view(*shape[:-1], 3)
