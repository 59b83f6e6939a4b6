#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE's own code.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

What is pinned:
  G1  umhsnerf.utils.spec_to_rgb.ColourSystem      -- imported natively
  G2  umhsnerf.utils.clusterprobe.ClusterLookup    -- imported natively
  G3  umhsnerf.umhs_renderer.get_weights_spectral  -- imported under third-party stubs
  G4  umhsnerf.umhs_field.UMHSField.get_outputs / get_density -- executed UNMODIFIED on a
      hand-assembled instance under third-party stubs; the nerfstudio encoders / MLPs plugged into
      that instance are oracle/torch_ref.py's restatements (nerfstudio itself is not installable
      offline), so G4 pins the reference's glue arithmetic, shapes and autograd wiring -- NOT the
      third-party pieces (those stay "parity unpinned").
  G5  umhsnerf.umhs_renderer.SpectralRenderer.blend_background_for_loss_computation

The stubs below are empty stand-ins for *uninstalled third-party libraries* (jaxtyping, nerfacc,
tinycudann, nerfstudio); no reference source is modified or copied.  Fixtures are data only.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import torch_ref as T  # noqa: E402


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Sub:
        def __class_getitem__(cls, _):
            return torch.Tensor

    mod("jaxtyping", Float=_Sub, Int=_Sub, Shaped=_Sub)
    mod("nerfacc", accumulate_along_rays=T.accumulate_along_rays)
    mod("tinycudann")

    class NerfactoField(nn.Module):
        pass

    class SemanticRenderer(nn.Module):
        pass

    class FieldHeadNames:
        RGB = "rgb"
        DENSITY = "density"

    class SceneBox:
        @staticmethod
        def get_normalized_positions(positions, aabb):
            return (positions - aabb[0]) / (aabb[1] - aabb[0])

    mod("nerfstudio")
    mod("nerfstudio.fields")
    mod("nerfstudio.fields.nerfacto_field", NerfactoField=NerfactoField)
    mod("nerfstudio.fields.base_field", get_normalized_directions=lambda d: (d + 1.0) / 2.0)
    mod("nerfstudio.cameras")
    mod("nerfstudio.cameras.rays", RaySamples=object)
    mod("nerfstudio.field_components")
    mod("nerfstudio.field_components.mlp", MLP=object, MLPWithHashEncoding=object)
    mod("nerfstudio.field_components.field_heads", FieldHeadNames=FieldHeadNames)
    mod("nerfstudio.field_components.activations", trunc_exp=T.trunc_exp)
    mod("nerfstudio.field_components.encodings", NeRFEncoding=object, SHEncoding=object)
    mod("nerfstudio.data")
    mod("nerfstudio.data.scene_box", SceneBox=SceneBox)
    mod("nerfstudio.model_components")
    mod("nerfstudio.model_components.renderers", SemanticRenderer=SemanticRenderer)
    mod("nerfstudio.utils")
    mod("nerfstudio.utils.colors", COLORS_DICT={"black": torch.tensor([0.0, 0.0, 0.0]), "white": torch.tensor([1.0, 1.0, 1.0])})


BAND_SETS = {
    "b21": list(range(450, 651, 10)),
    "b31": list(range(400, 701, 10)),
    "b128": np.linspace(400, 1000, 128).tolist(),
    "b141": np.linspace(440, 720, 141).tolist(),
}


def g1_colour():
    from umhsnerf.utils.spec_to_rgb import ColourSystem

    out = {}
    for name, bands in BAND_SETS.items():
        cs = ColourSystem(bands=bands, cs="sRGB", device="cpu")
        g = torch.Generator().manual_seed(1)
        spec = torch.rand(64, len(bands), generator=g)
        spec[:8] *= 0.004  # rows whose rgb straddles the 0.0031308 gamma knee / goes negative
        spec[8:12] *= 3.0  # rows that clamp at 1
        for ch in range(3):  # one-hot spectra on the most negative matrix entry -> negative rgb -> clamps at 0
            spec[12 + ch] = 0.0
            spec[12 + ch, int(cs.transform_matrix[:, ch].argmin())] = 1.0
        out[f"{name}_bands"] = np.asarray(bands, dtype=np.float64)
        out[f"{name}_M"] = cs.transform_matrix.numpy()
        out[f"{name}_spec"] = spec.numpy()
        out[f"{name}_rgb"] = cs(spec).numpy()
    np.savez_compressed(os.path.join(HERE, "g1_colour.npz"), **out)


def g2_cluster():
    from umhsnerf.utils.clusterprobe import ClusterLookup

    E = np.load("/root/reference/endmembers_hotdog.npy")
    out = {"endmembers_hotdog": E}
    g = torch.Generator().manual_seed(2)
    x = torch.rand(48, E.shape[1], generator=g)
    cl = ClusterLookup(E.shape[1], E.shape[0])
    ip, pr = cl(x, alpha=0.2, clusters=torch.from_numpy(E))
    ip2, pr2 = cl(x, alpha=None, clusters=torch.from_numpy(E))
    out.update(x=x.numpy(), ip=ip.numpy(), probs_a02=pr.numpy(), probs_none=pr2.numpy())
    np.savez_compressed(os.path.join(HERE, "g2_cluster.npz"), **out)


def g3_weights():
    from umhsnerf.umhs_renderer import get_weights_spectral

    g = torch.Generator().manual_seed(3)
    R, S = 12, 40
    deltas = torch.rand(R, S, 1, generator=g) * 0.05
    dens = torch.exp(torch.randn(R, S, 1, generator=g) * 2.0)
    dens[0] = 0.0  # zero-density ray
    dens[1] *= 1e4  # saturating ray
    w = get_weights_spectral(deltas, dens)
    np.savez_compressed(os.path.join(HERE, "g3_weights.npz"), deltas=deltas.numpy(), densities=dens.numpy(), weights=w.numpy())


class _Enc(nn.Module):
    def __init__(self, fn, out_dim):
        super().__init__()
        self.fn, self._out = fn, out_dim

    def get_out_dim(self):
        return self._out

    def forward(self, x):
        return self.fn(x)


class _MLP(nn.Module):
    def __init__(self, ws, bs, out_act=None):
        super().__init__()
        self.ws, self.bs, self.out_act = ws, bs, out_act

    def forward(self, x):
        return T.mlp_forward(x, list(self.ws), list(self.bs), self.out_act)


class _Base(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        return T.mlp_forward(T.hash_encode(x, self.p.hash_table, self.p.scalings, self.p.log2_T), list(self.p.base_w), list(self.p.base_b))


class _Frustums:
    def __init__(self, o, d, s, e):
        self.origins, self.directions, self.starts, self.ends = o, d, s, e
        self.shape = o.shape[:-1]

    def get_positions(self):
        return T.frustum_positions(self.origins, self.directions, self.starts, self.ends)


class _RaySamples:
    def __init__(self, o, d, s, e):
        self.frustums = _Frustums(o, d, s, e)
        self.camera_indices = torch.zeros(o.shape[0], 1, dtype=torch.long)


def _assemble_field(p: T.FieldParams, temperature: float):
    from umhsnerf.umhs_field import UMHSField

    f = UMHSField.__new__(UMHSField)
    nn.Module.__init__(f)
    f.method, f.num_classes, f.wavelengths = p.method, p.C, p.B
    f.pred_specular, f.pred_dino, f.use_scalar = p.pred_specular, False, True
    f.temperature, f.geo_feat_dim, f.appearance_embedding_dim = temperature, p.geo, 0
    f.embedding_appearance, f.average_init_density = None, 1
    f.direction_encoding = _Enc(T.sh_encoding_deg4, 16)
    f.position_encoding = _Enc(T.nerf_encoding, 12)
    f.mlp_base = _Base(p)
    f.mlp_head = _MLP(p.head_w, p.head_b)
    f.feature_mlp = _MLP(p.feat_w, p.feat_b)
    f.mlp_directional = _MLP(p.dir_w, p.dir_b, "sigmoid")
    f.endmembers = p.endmembers
    f.spatial_distortion = T.scene_contraction_linf
    f.aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    return f


def g4_field():
    cases = {
        "c6b31s": dict(C=6, B=31, spec=True, temp=0.4),
        "c9b128s": dict(C=9, B=128, spec=True, temp=0.3),
        "c4b141n": dict(C=4, B=141, spec=False, temp=0.7),
    }
    E141 = np.load("/root/reference/endmembers_hotdog.npy")
    for name, c in cases.items():
        p = T.FieldParams(c["C"], c["B"], c["spec"], log2_hashmap_size=12, table_scale=0.5, seed=7)
        if c["B"] == 141:
            with torch.no_grad():
                p.endmembers.copy_(torch.from_numpy(E141))
        batch = T.synthetic_batch(R=6, S=16, B=c["B"], seed=11)
        o, d, s, e = batch["origins"], batch["directions"], batch["starts"], batch["ends"]
        field = _assemble_field(p, c["temp"])
        rs = _RaySamples(o, d, s, e)
        density, emb = field.get_density(rs)  # reference code, umhs_field.py:300-329
        outs = field.get_outputs(rs, density_embedding=emb)  # reference code, umhs_field.py:151-296
        # a scalar functional of every differentiable output, so parameter grads are pinned too
        g = torch.Generator().manual_seed(5)
        cot_spec = torch.rand(outs["spectral"].shape, generator=g)
        cot_den = torch.rand(density.shape, generator=g)
        loss = (outs["spectral"] * cot_spec).sum() + (density * cot_den).sum()
        names = [k for k, _ in p.named_parameters()]
        grads = torch.autograd.grad(loss, [v for _, v in p.named_parameters()], allow_unused=True)
        out = dict(origins=o.numpy(), directions=d.numpy(), starts=s.numpy(), ends=e.numpy(), temperature=np.float64(c["temp"]),
                   density=density.detach().numpy(), emb=emb.detach().numpy(), cot_spec=cot_spec.numpy(), cot_den=cot_den.numpy())
        for k, v in outs.items():
            out[f"out_{k}"] = v.detach().numpy()
        for k, v in p.named_parameters():
            out[f"param_{k}"] = v.detach().numpy()
        for k, gv in zip(names, grads):
            if gv is not None:
                if k == "hash_table":  # sparse: store touched rows only
                    nz = gv.abs().sum(-1).nonzero()[:, 0]
                    out["grad_hash_rows"] = nz.numpy()
                    out["grad_hash_vals"] = gv[nz].numpy()
                else:
                    out[f"grad_{k}"] = gv.numpy()
        np.savez_compressed(os.path.join(HERE, f"g4_field_{name}.npz"), **out)


def g5_blend():
    from umhsnerf.umhs_renderer import SpectralRenderer

    r = SpectralRenderer()
    g = torch.Generator().manual_seed(6)
    pred = torch.rand(32, 3, generator=g)
    acc = torch.rand(32, 1, generator=g)
    gt = torch.rand(32, 3, generator=g)
    torch.manual_seed(1234)
    bg = torch.rand_like(pred)
    torch.manual_seed(1234)
    p2, g2 = r.blend_background_for_loss_computation(pred, acc, gt, gt)  # umhs_renderer.py:89-114 (rgba has 3 ch -> GT unchanged)
    np.savez_compressed(os.path.join(HERE, "g5_blend.npz"), pred=pred.numpy(), acc=acc.numpy(), gt=gt.numpy(), bg=bg.numpy(),
                        pred_out=p2.numpy(), gt_out=g2.numpy())


if __name__ == "__main__":
    torch.set_num_threads(4)
    g1_colour()
    g2_cluster()
    _install_stubs()
    g3_weights()
    g4_field()
    g5_blend()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
