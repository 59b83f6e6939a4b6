"""PyTorch-CPU restatement of the UMHS hot path (``implementation="torch"``).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Works in float32 (the
reference's CPU arithmetic) or float64 (tolerance reference) -- every function
follows the dtype of its inputs.

Citations are relative to ``/root/reference``.  Functions tagged
``[upstream-recalled]`` restate nerfstudio==1.1.5 / nerfacc==0.5.2 code that the
reference calls but that is not vendored (``nerfstudioa100_environment.yml:188-189``);
their parity is UNPINNED (no reference fixture exists for them).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn

# --------------------------------------------------------------------------- #
# nerfstudio field components  [upstream-recalled]
# --------------------------------------------------------------------------- #

HASH_PRIMES = (1, 2654435761, 805459861)


def hash_scalings(num_levels: int = 16, min_res: int = 16, max_res: int = 2048) -> Tensor:
    """``HashEncoding.__init__`` scalings buffer  [upstream-recalled].

    ``growth = exp((ln max - ln min)/(L-1))`` is a numpy float64 scalar but
    ``growth ** torch.arange(L)`` is evaluated by torch in float32, so
    ``floor(16 * g**15)`` is 2047, not 2048.  Used via ``umhs_model.py:183-184``
    (max_res=2048, log2_hashmap_size=19) and NerfactoField defaults (L=16, base 16).
    """
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1.0
    return torch.floor(min_res * growth**levels)  # float32 [L]


def hash_fn(coords: Tensor, table_size: int, level_offset: Tensor) -> Tensor:
    """``HashEncoding.hash_fn``  [upstream-recalled]: int32 coords * int64 primes, xor, mod T, + l*T."""
    c = coords.to(torch.int64) * torch.tensor(HASH_PRIMES, dtype=torch.int64)
    x = torch.bitwise_xor(c[..., 0], c[..., 1])
    x = torch.bitwise_xor(x, c[..., 2])
    x = x % table_size
    return x + level_offset


def hash_encode(x: Tensor, table: Tensor, scalings: Tensor, log2_T: int) -> Tensor:
    """``HashEncoding.pytorch_fwd``  [upstream-recalled]; call site ``umhs_field.py:320`` (inside mlp_base).

    x [N,3] in [0,1]; table [L*T, F]; returns [N, L*F].  All levels hashed, ceil/floor corners,
    ``offset = scaled - floor`` and the blend order f03,f12,f56,f47 -> f0312,f4756 -> out.
    """
    L = scalings.numel()
    T = 1 << log2_T
    level_offset = torch.arange(L, dtype=torch.int64) * T
    xs = x[..., None, :]
    scaled = xs * scalings.to(x.dtype).view(-1, 1)
    sc = torch.ceil(scaled).to(torch.int32)
    sf = torch.floor(scaled).to(torch.int32)
    offset = scaled - sf

    def cat(a, b, c):
        return torch.cat([a[..., 0:1], b[..., 1:2], c[..., 2:3]], dim=-1)

    h0 = hash_fn(sc, T, level_offset)
    h1 = hash_fn(cat(sc, sf, sc), T, level_offset)
    h2 = hash_fn(cat(sf, sf, sc), T, level_offset)
    h3 = hash_fn(cat(sf, sc, sc), T, level_offset)
    h4 = hash_fn(cat(sc, sc, sf), T, level_offset)
    h5 = hash_fn(cat(sc, sf, sf), T, level_offset)
    h6 = hash_fn(sf, T, level_offset)
    h7 = hash_fn(cat(sf, sc, sf), T, level_offset)
    f0, f1, f2, f3 = table[h0], table[h1], table[h2], table[h3]
    f4, f5, f6, f7 = table[h4], table[h5], table[h6], table[h7]
    ox, oy, oz = offset[..., 0:1], offset[..., 1:2], offset[..., 2:3]
    f03 = f0 * ox + f3 * (1 - ox)
    f12 = f1 * ox + f2 * (1 - ox)
    f56 = f5 * ox + f6 * (1 - ox)
    f47 = f4 * ox + f7 * (1 - ox)
    f0312 = f03 * oy + f12 * (1 - oy)
    f4756 = f47 * oy + f56 * (1 - oy)
    enc = f0312 * oz + f4756 * (1 - oz)
    return torch.flatten(enc, start_dim=-2, end_dim=-1)


def mlp_forward(x: Tensor, weights: List[Tensor], biases: List[Tensor], out_activation: Optional[str] = None) -> Tensor:
    """nerfstudio ``MLP.pytorch_fwd``  [upstream-recalled]: Linear(+bias) with ReLU between, none at the end.

    Built at ``umhs_field.py:67-75`` (feature_mlp), ``:95-103`` (mlp_head), ``:105-113`` (mlp_directional,
    out_activation Sigmoid) and inside ``MLPWithHashEncoding`` for mlp_base.
    """
    for i, (w, b) in enumerate(zip(weights, biases)):
        x = F.linear(x, w, b)
        if i < len(weights) - 1:
            x = torch.relu(x)
    if out_activation == "sigmoid":
        x = torch.sigmoid(x)
    return x


def nerf_encoding(x: Tensor, num_frequencies: int = 2, min_freq: float = 0.0, max_freq: float = 1.0) -> Tensor:
    """``NeRFEncoding.pytorch_fwd``  [upstream-recalled]; built by NerfactoField as
    ``NeRFEncoding(in_dim=3, num_frequencies=2, min_freq_exp=0, max_freq_exp=1)``; call ``umhs_field.py:184``.
    Output order: [x f0, x f1, y f0, y f1, z f0, z f1 | same + pi/2]."""
    scaled = 2 * torch.pi * x
    freqs = 2 ** torch.linspace(min_freq, max_freq, num_frequencies).to(x.dtype)
    si = scaled[..., None] * freqs
    si = si.view(*si.shape[:-2], -1)
    return torch.sin(torch.cat([si, si + torch.pi / 2.0], dim=-1))


def sh_encoding_deg4(d: Tensor) -> Tensor:
    """``components_from_spherical_harmonics(degree=4)``  [upstream-recalled]; ``SHEncoding(levels=4)`` is
    applied at ``umhs_field.py:160-162`` to ``(dir+1)/2`` with NO rescale back to [-1,1] in the torch path."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz = x**2, y**2, z**2
    c = torch.zeros((*d.shape[:-1], 16), dtype=d.dtype)
    c[..., 0] = 0.28209479177387814
    c[..., 1] = 0.4886025119029199 * y
    c[..., 2] = 0.4886025119029199 * z
    c[..., 3] = 0.4886025119029199 * x
    c[..., 4] = 1.0925484305920792 * x * y
    c[..., 5] = 1.0925484305920792 * y * z
    c[..., 6] = 0.9461746957575601 * zz - 0.31539156525251999
    c[..., 7] = 1.0925484305920792 * x * z
    c[..., 8] = 0.5462742152960396 * (xx - yy)
    c[..., 9] = 0.5900435899266435 * y * (3 * xx - yy)
    c[..., 10] = 2.890611442640554 * x * y * z
    c[..., 11] = 0.4570457994644658 * y * (5 * zz - 1)
    c[..., 12] = 0.3731763325901154 * z * (5 * zz - 3)
    c[..., 13] = 0.4570457994644658 * x * (5 * zz - 1)
    c[..., 14] = 1.445305721320277 * z * (xx - yy)
    c[..., 15] = 0.5900435899266435 * x * (xx - 3 * yy)
    return c


class _TruncExp(torch.autograd.Function):
    """nerfstudio ``trunc_exp``  [upstream-recalled] (``umhs_field.py:17,327``): fwd exp(x); bwd g*exp(clamp(x,-15,15))."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def scene_contraction_linf(x: Tensor) -> Tensor:
    """``SceneContraction(order=inf)``  [upstream-recalled]; ``umhs_model.py:133``, applied ``umhs_field.py:304``."""
    mag = torch.linalg.norm(x, ord=float("inf"), dim=-1)[..., None]
    return torch.where(mag < 1, x, (2 - (1 / mag)) * (x / mag))


def frustum_positions(origins: Tensor, directions: Tensor, starts: Tensor, ends: Tensor) -> Tensor:
    """``Frustums.get_positions``  [upstream-recalled]: o + d*(start+end)/2 (starts/ends [N,1])."""
    return origins + directions * (starts + ends) / 2


class _GradientScaler(torch.autograd.Function):
    """``scale_gradients_by_distance_squared``  [upstream-recalled] (``umhs_model.py:241-242``)."""

    @staticmethod
    def forward(ctx, value, scaling):
        ctx.save_for_backward(scaling)
        return value, scaling

    @staticmethod
    def backward(ctx, output_grad, grad_scaling):
        (scaling,) = ctx.saved_tensors
        return output_grad * scaling, grad_scaling


def scale_gradients_by_distance_squared(field_outputs: Dict, starts: Tensor, ends: Tensor) -> Dict:
    ray_dist = (starts + ends) / 2
    scaling = torch.square(ray_dist).clamp(0, 1)
    return {k: _GradientScaler.apply(v, scaling)[0] for k, v in field_outputs.items()}


# --------------------------------------------------------------------------- #
# nerfacc packed rendering  [upstream-recalled]
# --------------------------------------------------------------------------- #


def pack_info(ray_indices: Tensor, n_rays: int) -> Tensor:
    """``nerfacc.pack_info``  [upstream-recalled] (``umhs_model.py:245``): [R,2] (start, count), int64."""
    cnts = torch.zeros((n_rays,), dtype=torch.long)
    cnts.index_add_(0, ray_indices, torch.ones_like(ray_indices))
    starts = cnts.cumsum(0) - cnts
    return torch.stack([starts, cnts], dim=-1)


def exclusive_sum_packed(x: Tensor, packed_info: Tensor) -> Tensor:
    """Per-ray exclusive cumulative sum over a packed [N] tensor (nerfacc ``exclusive_sum``)."""
    out = torch.zeros_like(x)
    for s, c in packed_info.tolist():
        if c > 0:
            seg = x[s : s + c]
            out[s : s + c] = torch.cumsum(seg, 0) - seg
    return out


def render_weight_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info: Tensor):
    """``nerfacc.render_weight_from_density``  [upstream-recalled] (``umhs_model.py:246-251``).

    alpha = 1-exp(-sigma*dt); trans = exp(-exclusive_sum(sigma*dt)); w = alpha*trans.
    The reference's own dense twin is ``get_weights_spectral`` (``umhs_renderer.py:117-139``) -- pinned by G3."""
    sigmas_dt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sigmas_dt)
    trans = torch.exp(-exclusive_sum_packed(sigmas_dt, packed_info))
    return trans * alphas, trans, alphas


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor], ray_indices: Optional[Tensor], n_rays: Optional[int]) -> Tensor:
    """``nerfacc.accumulate_along_rays``  [upstream-recalled] (``umhs_renderer.py:28-30``)."""
    src = weights[..., None] if values is None else weights[..., None] * values
    if ray_indices is not None:
        out = torch.zeros((n_rays, src.shape[-1]), dtype=src.dtype)
        out.index_add_(0, ray_indices, src)
        return out
    return torch.sum(src, dim=-2)


def get_weights_spectral(deltas: Tensor, densities: Tensor) -> Tensor:
    """Dense-layout weights, ``umhs_renderer.py:117-139`` (pinned by golden G3)."""
    delta_density = deltas * densities
    alphas = 1 - torch.exp(-delta_density)
    transmittance = torch.cumsum(delta_density[..., :-1, :], dim=-2)
    transmittance = torch.cat(
        [torch.zeros((*transmittance.shape[:1], 1, transmittance.shape[-1]), dtype=densities.dtype), transmittance], dim=-2
    )
    transmittance = torch.exp(-transmittance)
    return torch.nan_to_num(alphas * transmittance)


def spectral_renderer(spectral: Tensor, weights: Tensor, ray_indices: Optional[Tensor] = None, num_rays: Optional[int] = None) -> Tensor:
    """``SpectralRenderer.forward``, ``umhs_renderer.py:15-30`` (squeezes a leading 1-dim)."""
    if spectral.dim() == 3:
        spectral = spectral.squeeze(0)
    elif spectral.dim() == 1:
        spectral = spectral.unsqueeze(0)
    return accumulate_along_rays(weights[..., 0], spectral, ray_indices, num_rays)


def render_depth_expected(weights: Tensor, starts: Tensor, ends: Tensor, ray_indices: Tensor, num_rays: int, clip_range=None) -> Tensor:
    """nerfstudio ``DepthRenderer(method="expected")`` packed branch  [upstream-recalled] (``umhs_model.py:254-256``).
    ``clip_range`` = (min, max) of the WHOLE batch's sample midpoints when the caller hands in a ray chunk of it (tests only)."""
    eps = 1e-10
    steps = (starts + ends) / 2
    depth = accumulate_along_rays(weights[..., 0], steps, ray_indices, num_rays)
    acc = accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays)
    depth = depth / (acc + eps)
    lo, hi = (steps.min(), steps.max()) if clip_range is None else clip_range
    return torch.clip(depth, lo, hi)


# --------------------------------------------------------------------------- #
# reference utils (pinned by goldens G1/G2/G5)
# --------------------------------------------------------------------------- #


def _g(x, alpha, mu, sigma1, sigma2):
    sigma = np.clip((x < mu) * sigma1 + (x >= mu) * sigma2, a_min=1e-6, a_max=None)
    return alpha * np.exp((x - mu) ** 2 / (-2 * (sigma**2)))


def colour_matrix(bands) -> Tensor:
    """``ColourSystem.__init__`` for cs='sRGB', ``utils/spec_to_rgb.py:6-21,34-38,62-90``: [B,3] float32."""
    x = np.array(bands) * 10
    cmf = np.array(
        [
            _g(x, 1.056, 5998, 379, 310) + _g(x, 0.362, 4420, 160, 267) + _g(x, -0.065, 5011, 204, 262),
            _g(x, 0.821, 5688, 469, 405) + _g(x, 0.286, 5309, 163, 311),
            _g(x, 1.217, 4370, 118, 360) + _g(x, 0.681, 4590, 260, 138),
        ]
    )

    def xyz(a, b):
        return np.array((a, b, 1 - a - b))

    red, green, blue, white = xyz(0.64, 0.33), xyz(0.30, 0.60), xyz(0.15, 0.06), xyz(0.3127, 0.3291)
    M = np.vstack((red, green, blue)).T
    MI = np.linalg.inv(M)
    wscale = MI.dot(white)
    A = MI / wscale[:, np.newaxis]
    RGB = cmf.T @ A.T
    RGB = RGB / np.sum(RGB, axis=0, keepdims=True)
    return torch.from_numpy(RGB).float()


def colour_system(spec: Tensor, M: Tensor) -> Tensor:
    """``ColourSystem.forward`` + ``gamma_correction``, ``utils/spec_to_rgb.py:103-127`` (fp32 on CPU)."""
    rgb = torch.matmul(spec, M.to(spec.dtype))
    rgb = torch.where(rgb < 0.0031308, 12.92 * rgb, 1.055 * (rgb.clamp(min=1e-6).pow(1 / 2.4)) - 0.055)
    return rgb.clamp(0, 1)


def cluster_lookup(x: Tensor, alpha: Optional[float], clusters: Tensor):
    """``ClusterLookup.forward``, ``utils/clusterprobe.py:17-38``."""
    nc = F.normalize(clusters, dim=1)
    nf = F.normalize(x, dim=1)
    ip = torch.matmul(nf, nc.t())
    if alpha is None:
        probs = F.one_hot(torch.argmax(ip, dim=1), clusters.shape[0]).to(torch.float32)
    else:
        probs = F.softmax(ip * alpha, dim=1)
    return ip, probs


def blend_background_for_loss(pred_image: Tensor, pred_accumulation: Tensor, gt_image: Tensor, bg_random: Tensor):
    """nerfstudio ``RGBRenderer.blend_background_for_loss_computation`` with background_color="random"
    [upstream-recalled] (``umhs_model.py:358-362``; the reference's spectral twin is ``umhs_renderer.py:89-114``,
    pinned by G5).  ``bg_random`` replaces ``torch.rand_like(pred_image)`` so runs are reproducible.  GT has no
    alpha channel in the synthetic batches, so it is returned unchanged (``umhs_renderer.py:75-76``)."""
    return pred_image + bg_random * (1.0 - pred_accumulation), gt_image


# --------------------------------------------------------------------------- #
# UMHSField / UMHSModel restatement
# --------------------------------------------------------------------------- #


class FieldParams(nn.Module):
    """Parameters of ``UMHSField`` with the reference's state-dict key names (``umhs_field.py:67,81-85,95,105`` +
    NerfactoField's ``mlp_base`` = MLPWithHashEncoding{encoder.hash_table, mlp.layers.*}  [upstream-recalled])."""

    def __init__(
        self,
        num_classes: int,
        wavelengths: int,
        pred_specular: bool,
        method: str = "rgb+spectral",
        log2_hashmap_size: int = 19,
        num_levels: int = 16,
        features_per_level: int = 2,
        hidden_dim: int = 64,
        geo_feat_dim: int = 15,
        hidden_dim_color: int = 64,
        table_scale: float = 1e-3,
        seed: int = 42,
        dtype=torch.float32,
    ):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.C, self.B, self.pred_specular, self.method = num_classes, wavelengths, pred_specular, method
        self.log2_T, self.L, self.Fdim, self.geo = log2_hashmap_size, num_levels, features_per_level, geo_feat_dim
        T = 1 << log2_hashmap_size
        self.hash_table = nn.Parameter(((torch.rand((T * num_levels, features_per_level), generator=g) * 2 - 1) * table_scale).to(dtype))
        self.register_buffer("scalings", hash_scalings(num_levels, 16, 2048))

        def lin(i, o):
            w = torch.empty(o, i)
            nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=g)
            bound = 1 / math.sqrt(i)
            b = (torch.rand(o, generator=g) * 2 - 1) * bound
            return nn.Parameter(w.to(dtype)), nn.Parameter(b.to(dtype))

        def mlp(dims):
            ws, bs = nn.ParameterList(), nn.ParameterList()
            for i, o in zip(dims[:-1], dims[1:]):
                w, b = lin(i, o)
                ws.append(w)
                bs.append(b)
            return ws, bs

        self.base_w, self.base_b = mlp([num_levels * features_per_level, hidden_dim, 1 + geo_feat_dim])
        if "spectral" in method:
            in_dim = 12 + geo_feat_dim
            out_feat = num_classes + 1 if pred_specular else num_classes
            self.feat_w, self.feat_b = mlp([in_dim, hidden_dim_color, hidden_dim_color, out_feat])
            self.head_w, self.head_b = mlp([in_dim, hidden_dim_color, hidden_dim_color, num_classes])
            self.dir_w, self.dir_b = mlp([16 + 12, 16, wavelengths])
            self.endmembers = nn.Parameter(torch.rand(num_classes, wavelengths, generator=g).to(dtype))
        else:  # method == "rgb": mlp_head(cat[d, emb]) -> 3   (umhs_field.py:280-294)
            self.head_w, self.head_b = mlp([16 + geo_feat_dim, hidden_dim_color, hidden_dim_color, 3])

    def reference_state_dict(self) -> Dict[str, Tensor]:
        """Map to the reference's key names (checkpoint compat surface)."""
        sd = {"mlp_base.encoder.hash_table": self.hash_table.detach()}
        for name, ws, bs in (
            ("mlp_base.mlp", self.base_w, self.base_b),
            ("mlp_head", self.head_w, self.head_b),
        ):
            for i, (w, b) in enumerate(zip(ws, bs)):
                sd[f"{name}.layers.{i}.weight"], sd[f"{name}.layers.{i}.bias"] = w.detach(), b.detach()
        if "spectral" in self.method:
            for name, ws, bs in (("feature_mlp", self.feat_w, self.feat_b), ("mlp_directional", self.dir_w, self.dir_b)):
                for i, (w, b) in enumerate(zip(ws, bs)):
                    sd[f"{name}.layers.{i}.weight"], sd[f"{name}.layers.{i}.bias"] = w.detach(), b.detach()
            sd["endmembers"] = self.endmembers.detach()
        return sd


def field_density(p: FieldParams, origins, directions, starts, ends, contraction: bool = True, aabb: Optional[Tensor] = None):
    """``UMHSField.get_density``, ``umhs_field.py:300-329``.  Returns (density[N,1], emb[N,15], sigma_raw[N,1], selector[N])."""
    pos = frustum_positions(origins, directions, starts, ends)
    if contraction:
        pos = scene_contraction_linf(pos)
        pos = (pos + 2.0) / 4.0
    else:  # SceneBox.get_normalized_positions [upstream-recalled]: (x - aabb[0]) / (aabb[1]-aabb[0])
        pos = (pos - aabb[0]) / (aabb[1] - aabb[0])
    selector = ((pos > 0.0) & (pos < 1.0)).all(dim=-1)
    pos = pos * selector[..., None]
    enc = hash_encode(pos, p.hash_table, p.scalings, p.log2_T)
    h = mlp_forward(enc, list(p.base_w), list(p.base_b))
    sigma_raw, emb = torch.split(h, [1, p.geo], dim=-1)
    density = 1 * trunc_exp(sigma_raw)
    density = density * selector[..., None]
    return density, emb, sigma_raw, selector


def field_outputs(p: FieldParams, origins, directions, starts, ends, emb: Tensor, temperature: float) -> Dict[str, Tensor]:
    """``UMHSField.get_outputs`` for method in {"spectral","rgb+spectral"} (``umhs_field.py:151-261``) and "rgb" (``:280-294``).

    Keeps the reference's output shapes: with pred_specular "spectral"/"specular"/"abundances" are [1,N,*] and
    "spectral2" is [N,B]; without it "spectral" is [N,B]."""
    out: Dict[str, Tensor] = {}
    N = origins.shape[0]
    dn = (directions + 1.0) / 2.0  # get_normalized_directions [upstream-recalled]
    d = sh_encoding_deg4(dn.view(-1, 3))
    if "spectral" in p.method:
        pos = frustum_positions(origins, directions, starts, ends)
        pe = nerf_encoding(pos.view(-1, 3)).unsqueeze(0)  # [1,N,12]
        emb3 = emb.unsqueeze(0)
        h1 = torch.cat([pe.view(-1, 12), emb3.view(-1, p.geo)], dim=-1)
        scalar = torch.sigmoid(mlp_forward(h1, list(p.head_w), list(p.head_b)).view(N, -1, p.C))  # [N,1,C]
        fin = torch.cat([pe, emb3], dim=-1)
        logits = mlp_forward(fin.view(-1, fin.size(-1)), list(p.feat_w), list(p.feat_b)).view(1, N, -1)
        if p.pred_specular:
            logits, s1 = torch.split(logits, [p.C, 1], dim=-1)
            s1 = torch.sigmoid(s1)
        abund = F.softmax(logits / temperature, dim=-1)  # [1,N,C]
        E = p.endmembers.unsqueeze(0).unsqueeze(0)
        E = E.expand(abund.shape[0], abund.shape[1], -1, -1).transpose(2, 3).squeeze(0)  # [N,B,C]
        adapted = scalar * E
        spec = (adapted @ abund.unsqueeze(-1)).squeeze()  # [N,B]
        if p.pred_specular:
            specular = mlp_forward(torch.cat([d, pe.view(-1, 12)], dim=-1), list(p.dir_w), list(p.dir_b), "sigmoid").view(N, p.B)
            spec2 = spec + (s1 * specular)  # [1,N,B]
            out["spectral"] = spec2
            out["spectral2"] = spec
            with torch.no_grad():
                out["specular"] = s1 * specular
        else:
            out["spectral"] = spec
        out["abundances"] = abund
    else:
        # NerfactoField.mlp_head: out_activation=nn.Sigmoid()  [upstream-recalled: nerfstudio 1.1.5 fields/nerfacto_field.py]
        h = torch.cat([d, emb.view(-1, p.geo)], dim=-1)
        out["rgb"] = mlp_forward(h, list(p.head_w), list(p.head_b), "sigmoid").view(N, 3)
    return out


def model_outputs(
    p: FieldParams,
    origins,
    directions,
    starts,
    ends,
    ray_indices: Tensor,
    num_rays: int,
    temperature: float,
    colour_M: Tensor,
    use_gradient_scaling: bool = True,
    contraction: bool = True,
    depth_clip_range=None,
) -> Dict[str, Tensor]:
    """``UMHSModel.get_outputs`` after the sampler, ``umhs_model.py:239-313`` (spectral methods; ``method="rgb"``: rgb / accumulation / depth)."""
    density, emb, _, _ = field_density(p, origins, directions, starts, ends, contraction)
    fo = field_outputs(p, origins, directions, starts, ends, emb, temperature)
    fo["density"] = density
    if use_gradient_scaling:
        fo = scale_gradients_by_distance_squared(fo, starts, ends)
    pinfo = pack_info(ray_indices, num_rays)
    weights = render_weight_from_density(starts[..., 0], ends[..., 0], fo["density"][..., 0], pinfo)[0][..., None]
    out = {
        "depth": render_depth_expected(weights, starts, ends, ray_indices, num_rays, depth_clip_range),
        "accumulation": accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays),
        "weights": weights,
    }
    if p.method == "rgb":
        # umhs_model.py:265-267.  The reference omits ray_indices / num_rays here, so nerfstudio's RGBRenderer sums over ALL packed
        # samples of the batch (one colour for every ray) -- a defect, not a behaviour to restate: per-ray compositing, as for every
        # other output.  background_color="random": no blend in the forward  [upstream-recalled].
        out["rgb"] = accumulate_along_rays(weights[..., 0], fo["rgb"], ray_indices, num_rays)
        out["num_samples_per_ray"] = pinfo[:, 1]
        return out
    spectral = spectral_renderer(fo["spectral"], weights, ray_indices, num_rays)
    out["spectral"] = spectral
    if p.pred_specular:
        out["spectral2"] = spectral_renderer(fo["spectral2"], weights, ray_indices, num_rays)
        with torch.no_grad():
            out["specular"] = spectral_renderer(fo["specular"], weights, ray_indices, num_rays)
    if p.method == "spectral":
        with torch.no_grad():
            out["rgb"] = colour_system(spectral, colour_M)
    else:
        out["rgb"] = colour_system(spectral, colour_M)
    out["num_samples_per_ray"] = pinfo[:, 1]
    with torch.no_grad():
        out["abundances"] = spectral_renderer(fo["abundances"], weights, ray_indices, num_rays)
    ip, probs = cluster_lookup(spectral, 0.2, p.endmembers)
    out["seg_probs"] = probs
    with torch.no_grad():
        acc_if = (out["accumulation"] > 0.5).to(spectral.dtype)
        out["seg_raw"] = probs.argmax(1) * acc_if.squeeze()
    return out


def model_loss(out: Dict[str, Tensor], gt_spectral: Tensor, gt_rgb: Tensor, bg_random: Tensor, method: str, rgb_loss_weight: float = 1.0):
    """``UMHSModel.get_loss_dict``, ``umhs_model.py:358-370`` (the factor 5 is hard-coded at ``:369``)."""
    pred_rgb, gt = blend_background_for_loss(out["rgb"], out["accumulation"], gt_rgb, bg_random)
    loss = {}
    if method == "rgb":
        loss["rgb_loss"] = F.mse_loss(pred_rgb, gt)
    elif method == "spectral":
        loss["spectral_loss"] = F.mse_loss(out["spectral"], gt_spectral)
    else:
        loss["spectral_loss"] = 5 * F.mse_loss(out["spectral"], gt_spectral)
        loss["rgb_loss"] = rgb_loss_weight * F.mse_loss(pred_rgb, gt)
    return loss


def psnr(pred: Tensor, gt: Tensor) -> Tensor:
    """torchmetrics ``PeakSignalNoiseRatio(data_range=1.0)``  [upstream-recalled] (``umhs_model.py:391,398``)."""
    return 10.0 * torch.log10(1.0 / torch.mean((pred - gt) ** 2))


def adam_step(params: List[Tensor], grads: List[Tensor], ms: List[Tensor], vs: List[Tensor], step: int, lr: float, b1=0.9, b2=0.999, eps=1e-15):
    """torch.optim.Adam single step (nerfstudio ``AdamOptimizerConfig(lr=2e-2, eps=1e-15)``, ``umhs_config.py:59-64``)."""
    bc1, bc2 = 1 - b1**step, 1 - b2**step
    for p_, g_, m_, v_ in zip(params, grads, ms, vs):
        m_.mul_(b1).add_(g_, alpha=1 - b1)
        v_.mul_(b2).addcmul_(g_, g_, value=1 - b2)
        denom = (v_.sqrt() / math.sqrt(bc2)).add_(eps)
        p_.addcdiv_(m_, denom, value=-lr / bc1)


def exp_decay_lr(step: int, lr_init: float = 2e-2, lr_final: float = 1e-5, max_steps: int = 30000) -> float:
    """nerfstudio ``ExponentialDecayScheduler`` without warmup  [upstream-recalled] (``umhs_config.py:63``)."""
    t = min(max(step / max_steps, 0.0), 1.0)
    return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


# --------------------------------------------------------------------------- #
# synthetic batches
# --------------------------------------------------------------------------- #


def synthetic_batch(R: int, S: int, B: int, seed: int = 42, ragged: bool = False, dtype=torch.float32) -> Dict[str, Tensor]:
    """Packed synthetic ray batch: cameras on a radius-2.5 sphere looking at the unit box, S samples per ray
    (or Poisson(S) clipped to [1,256] when ragged), step = sqrt(12)/1000*(1+0.004 t) (``umhs_model.py:84,199-200``)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(R, 3, generator=g)
    u = u / u.norm(dim=-1, keepdim=True)
    o = (torch.rand(R, 3, generator=g) * 2 - 1) * 0.5 + 2.5 * u
    d = -o + torch.randn(R, 3, generator=g) * 0.1
    d = d / d.norm(dim=-1, keepdim=True)
    if ragged:
        cnt = torch.poisson(torch.full((R,), float(S)), generator=g).clamp(1, 256).long()
        cnt[torch.rand(R, generator=g) < 0.05] = 0  # some empty rays (nerfacc allows them)
    else:
        cnt = torch.full((R,), S, dtype=torch.long)
    ray_indices = torch.repeat_interleave(torch.arange(R), cnt)
    N = int(cnt.sum())
    step = math.sqrt(12.0) / 1000.0
    # march through the unit box: start where the ray is ~1.3 from the origin, jittered
    t_near = (o.norm(dim=-1) - 1.3).clamp(min=0.05) + torch.rand(R, generator=g) * step
    k = torch.arange(N) - torch.repeat_interleave(cnt.cumsum(0) - cnt, cnt)
    stride = 2.6 / max(S, 1)  # cover the box diameter with S samples
    t0 = t_near[ray_indices] + k * stride
    t1 = t0 + step * (1 + 0.004 * t0)
    return {
        "origins": o[ray_indices].to(dtype).contiguous(),
        "directions": d[ray_indices].to(dtype).contiguous(),
        "starts": t0[:, None].to(dtype).contiguous(),
        "ends": t1[:, None].to(dtype).contiguous(),
        "ray_indices": ray_indices,
        "num_rays": R,
        "gt_spectral": torch.rand(R, B, generator=g).to(dtype),
        "bg_random": torch.rand(R, 3, generator=g).to(dtype),
    }


# --------------------------------------------------------------------------- #
# occupancy-grid ray marcher  [upstream-recalled, parity unpinned]
# nerfacc==0.5.2 OccGridEstimator.sampling / traverse_grids / render_visibility_from_density, called through nerfstudio's
# VolumetricSampler at umhs_model.py:201-209,229-237.  nerfacc's CUDA source is not available offline; this restates its
# published behaviour: multi-level nested grids (level l = roi enlarged 2^l about its centre), per-voxel traversal, samples of
# size dt = clamp(t*cone_angle, step, inf) emitted while their MID-POINT is inside an occupied voxel, a new run of samples
# restarting at the voxel entry after empty space.  All arithmetic in float32 in a fixed order so that the HIP kernel (built
# with fp contraction off) reproduces it exactly.
# --------------------------------------------------------------------------- #
F32 = np.float32


def occ_grid_aabbs(roi_aabb, levels: int) -> np.ndarray:
    a = np.asarray(roi_aabb, dtype=np.float32).reshape(2, 3)
    c, h = (a[0] + a[1]) / F32(2), (a[1] - a[0]) / F32(2)
    return np.stack([np.concatenate([c - h * F32(2**l), c + h * F32(2**l)]) for l in range(levels)]).astype(np.float32)


def march_ray_ref(o, d, binaries: np.ndarray, roi_aabb, near, far, step, cone):
    """One ray.  binaries: bool [levels, res, res, res].  Returns (t_starts, t_ends) float32 lists."""
    levels, res = binaries.shape[0], binaries.shape[1]
    a = np.asarray(roi_aabb, dtype=np.float32).reshape(2, 3)
    c, h = (a[0] + a[1]) / F32(2), (a[1] - a[0]) / F32(2)
    o, d = np.asarray(o, np.float32), np.asarray(d, np.float32)
    near, far, step, cone = F32(near), F32(far), F32(step), F32(cone)
    BIG = F32(1e30)
    inv = np.where(d != 0, F32(1) / np.where(d != 0, d, F32(1)), np.where(np.signbit(d), -BIG, BIG)).astype(np.float32)
    inv_h = (F32(1) / h).astype(np.float32)  # positions are normalised with reciprocals (one multiply per axis and voxel, not a divide)
    ho = h * F32(2 ** (levels - 1))
    t0, t1 = (c - ho - o) * inv, (c + ho - o) * inv
    tn, tf = np.max(np.minimum(t0, t1)), np.min(np.maximum(t0, t1))
    t, t_end = max(tn, near), min(tf, far)
    ts, te = [], []
    if not (t < t_end):
        return ts, te
    continuous, t_last = False, t
    guard = 0
    while t < t_end and guard < 100000:
        guard += 1
        tm = t + F32(1e-5) * max(F32(1), abs(t))  # a point just inside the voxel being entered
        p = o + d * tm
        m = np.max(np.abs(p - c) * inv_h)
        if not (m < F32(2 ** (levels - 1))):
            break
        lvl = 0 if m < F32(1) else int(np.frexp(m)[1])  # m in [2^(e-1), 2^e) -> level e
        lvl = min(max(lvl, 0), levels - 1)
        hl = h * F32(2**lvl)
        vmin_l = c - hl
        vs = (hl * F32(2)) / F32(res)
        inv_vs = F32(1) / vs
        idx = np.clip(np.floor((p - vmin_l) * inv_vs).astype(np.int64), 0, res - 1)
        lo, hi = vmin_l + idx.astype(np.float32) * vs, vmin_l + (idx + 1).astype(np.float32) * vs
        tx = (np.where(d >= 0, hi, lo) - o) * inv
        t_exit = np.min(np.where(d != 0, tx, BIG))
        if not (t_exit > t):
            t_exit = np.nextafter(t, BIG, dtype=np.float32)
        t_clip = min(t_exit, t_end)
        if binaries[lvl, idx[0], idx[1], idx[2]]:
            if not continuous:
                t_last = t
            while True:
                dt = min(max(t_last * cone, step), BIG)
                if not (t_last + dt * F32(0.5) < t_clip):
                    break
                ts.append(t_last)
                te.append(t_last + dt)
                t_last = t_last + dt
            continuous = True
        else:
            continuous = False
        t = t_clip
    return ts, te


def march_rays_ref(origins, directions, binaries, roi_aabb, near, far, step, cone):
    """Packed (ray_indices int64 [N], t_starts [N], t_ends [N]) for a batch of rays (python loop: small cases only)."""
    ri, s, e = [], [], []
    for r in range(origins.shape[0]):
        ts, te = march_ray_ref(origins[r].numpy(), directions[r].numpy(), binaries, roi_aabb, near, far, step, cone)
        ri += [r] * len(ts)
        s += ts
        e += te
    return (torch.tensor(ri, dtype=torch.int64), torch.tensor(np.array(s, dtype=np.float32)), torch.tensor(np.array(e, dtype=np.float32)))


def render_visibility_from_density(t_starts, t_ends, sigmas, packed_info, early_stop_eps: float, alpha_thre: float) -> Tensor:
    """nerfacc.render_visibility_from_density  [upstream-recalled]: keep samples whose transmittance >= eps and alpha >= thre."""
    sdt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sdt)
    trans = torch.exp(-exclusive_sum_packed(sdt, packed_info))
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis


# ------------------------------------------------------------------------------------------------ #
# pixel sampler / ray generator / GT gather (nerfstudio==1.1.5 behind umhs_datamanager.py:95-108).
# nerfstudio is not available offline: [upstream-recalled], parity unpinned for these three.
# ------------------------------------------------------------------------------------------------ #
def pixel_sample_indices(uniform: torch.Tensor, n_images: int, height: int, width: int) -> torch.Tensor:
    """PixelSampler.sample_method: ``(rand((R,3)) * tensor([n, H, W])).long()`` -> rows (camera, y, x)."""
    return (uniform.float() * torch.tensor([n_images, height, width], dtype=torch.float32)).long()


def generate_rays(indices: torch.Tensor, c2w: torch.Tensor, intrinsics: torch.Tensor):
    """RayGenerator.forward + Cameras._generate_rays_from_coords, perspective, no distortion.

    indices [R,3] (camera, y, x); c2w [n,3,4]; intrinsics [n,4] = fx, fy, cx, cy.  Pixel centres (+0.5), camera looks down -z,
    y up; returns origins, unit directions, pixel_area [R,1] (|d - d_x+1| * |d - d_y+1|), directions_norm [R,1]."""
    c, y, x = indices[:, 0], indices[:, 1].float() + 0.5, indices[:, 2].float() + 0.5
    fx, fy, cx, cy = (intrinsics[c, k] for k in range(4))
    coord = torch.stack([(x - cx) / fx, -(y - cy) / fy], -1)
    coord_x = torch.stack([(x - cx + 1) / fx, -(y - cy) / fy], -1)
    coord_y = torch.stack([(x - cx) / fx, -(y - cy + 1) / fy], -1)
    cs = torch.stack([coord, coord_x, coord_y], 0)  # [3,R,2]
    ds = torch.cat([cs, -torch.ones_like(cs[..., :1])], -1)  # [3,R,3]
    rot = c2w[c][:, :3, :3]
    ds = torch.sum(ds[..., None, :] * rot, dim=-1)
    nrm = torch.maximum(torch.linalg.vector_norm(ds, dim=-1, keepdim=True), torch.tensor([torch.finfo(torch.float32).eps]))
    ds = ds / nrm
    dx = torch.sqrt(torch.sum((ds[0] - ds[1]) ** 2, dim=-1))
    dy = torch.sqrt(torch.sum((ds[0] - ds[2]) ** 2, dim=-1))
    return c2w[c][:, :3, 3], ds[0], (dx * dy)[:, None], nrm[0]


def gather_pixels(indices: torch.Tensor, stack: torch.Tensor) -> torch.Tensor:
    """collate_image_dataset_batch: ``stack[c, y, x]`` (uint8 stacks are float32/255 as in InputDataset.get_image_float32)."""
    v = stack[indices[:, 0], indices[:, 1], indices[:, 2]]
    return v.float() / 255.0 if stack.dtype == torch.uint8 else v


def auto_orient_and_center_poses(poses: torch.Tensor, method: str = "up", center_method: str = "poses"):
    """nerfstudio camera_utils.auto_orient_and_center_poses for the settings the reference's dataparser defaults to
    (umhs_dataparser.py:86-89: orientation "up", center "poses"; "none" supported).  poses [n,3|4,4] float32."""
    origins = poses[:, :3, 3]
    translation = origins.mean(0) if center_method == "poses" else torch.zeros(3)
    if center_method not in ("poses", "none") or method not in ("up", "none"):
        raise NotImplementedError((method, center_method))
    if method == "up":
        up = poses[:, :3, 1].mean(0)
        up = up / torch.linalg.norm(up)
        a, b = up, torch.tensor([0.0, 0.0, 1.0])
        v = torch.linalg.cross(a, b)
        cth = torch.dot(a, b)
        if float(cth) < -1 + 1e-8:  # opposite vectors: nerfstudio perturbs a and retries
            a = a + (torch.rand(3) - 0.5) * 0.01
            a = a / torch.linalg.norm(a)
            v, cth = torch.linalg.cross(a, b), torch.dot(a, b)
        s = torch.linalg.norm(v)
        K = torch.tensor([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        R = torch.eye(3) + K + K @ K * ((1 - cth) / (s ** 2 + 1e-8))
        transform = torch.cat([R, R @ -translation[..., None]], dim=-1)
    else:
        transform = torch.eye(4)[:3].clone()
        transform[:3, 3] = -translation
    p4 = poses if poses.shape[1] == 4 else torch.cat([poses, torch.tensor([[[0.0, 0, 0, 1]]]).expand(poses.shape[0], 1, 4)], 1)
    return transform @ p4, transform


# ------------------------------------------------------------------------------------------------ #
# image metrics of get_image_metrics_and_images (umhs_model.py:407-453).  torchmetrics==1.5.2 is not
# available offline: [upstream-recalled] restatement of structural_similarity_index_measure and SpectralAngleMapper.
# ------------------------------------------------------------------------------------------------ #
def ssim_ref(preds: torch.Tensor, target: torch.Tensor, data_range=None) -> torch.Tensor:
    """preds/target [1,C,H,W].  gaussian_kernel=True, sigma=1.5 (kernel 11), k1=.01, k2=.03, reduction elementwise_mean."""
    if data_range is None:
        data_range = max(preds.max() - preds.min(), target.max() - target.min())
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    ch, ks, pad = preds.shape[1], 11, 5
    dist = torch.arange((1 - ks) / 2, (1 + ks) / 2, 1, dtype=preds.dtype)
    g = torch.exp(-((dist / 1.5) ** 2) / 2)
    g = (g / g.sum()).unsqueeze(0)
    kernel = torch.matmul(g.t(), g).expand(ch, 1, ks, ks)
    p = torch.nn.functional.pad(preds, (pad, pad, pad, pad), mode="reflect")
    t = torch.nn.functional.pad(target, (pad, pad, pad, pad), mode="reflect")
    out = torch.nn.functional.conv2d(torch.cat((p, t, p * p, t * t, p * t)), kernel, groups=ch).split(preds.shape[0])
    mpp, mtt, mpt = out[0].pow(2), out[1].pow(2), out[0] * out[1]
    vp, vt = torch.clamp(out[2] - mpp, min=0.0), torch.clamp(out[3] - mtt, min=0.0)
    cov = out[4] - mpt
    full = ((2 * mpt + c1) * (2 * cov + c2)) / ((mpp + mtt + c1) * (vp + vt + c2))
    idx = full[..., pad:-pad, pad:-pad]
    return idx.reshape(idx.shape[0], -1).mean(-1).mean()


def sam_ref(preds: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """SpectralAngleMapper(reduction="none") on [1,C,H,W] -> [1,H,W] angles (NaN where a spectrum is all zero)."""
    dot = (preds * target).sum(dim=1)
    return torch.clamp(dot / (preds.norm(dim=1) * target.norm(dim=1)), -1, 1).acos()


def image_metrics_ref(pred_rgb, gt_rgb, pred_spec=None, gt_spec=None):
    """Metric values of umhs_model.py:430-452 for channel-last images [H,W,K] (lpips omitted: pretrained weights)."""
    chw = lambda x: torch.moveaxis(x, -1, 0)[None]
    g, p = chw(gt_rgb), chw(pred_rgb)
    md = {"psnr": float(10 * torch.log10(1.0 / torch.mean((g - p) ** 2))), "ssim": float(ssim_ref(g, p))}
    if pred_spec is not None:
        gs, ps = chw(gt_spec), chw(pred_spec)
        md["psnr_spectral"] = float(10 * torch.log10(1.0 / torch.mean((gs - ps) ** 2)))
        md["ssim_spectral"] = float(ssim_ref(gs, ps))
        md["sam_spectral"] = float(torch.nanmean(sam_ref(ps, gs)))
        md["rmse_spectral"] = float(torch.sqrt(torch.nn.functional.mse_loss(ps, gs)))
    return md
