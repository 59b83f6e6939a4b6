"""On-disk format reader (mirror of ``umhsnerf/data/umhs_dataparser.py``): nerfstudio ``transforms.json`` with a
``hyperspectral_file_path`` per frame and top-level ``wavelengths`` (``:222-228,318-320``).

Same config fields and defaults as ``UMHSDataParserConfig`` (``:68-112``); frames sorted by file name (``:160-167``),
train/eval split by "train"/"eval" in the parent folder name (``:43-65``) or fraction / interval / all, poses oriented
("up") and centred ("poses") then scaled into the +/-1 box (``:296-311``), scene box +/- ``scene_scale`` (``:324-333``).
Perspective cameras without distortion; masks / depth / dino / 3D points / image downscaling are not part of the hot
path and raise if requested."""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Literal, Optional

import numpy as np
import torch


@dataclass
class Cameras:
    """Perspective pinhole cameras (the subset of nerfstudio's ``Cameras`` the ray generator needs)."""

    camera_to_worlds: torch.Tensor  # [n,3,4]
    fx: torch.Tensor  # [n]
    fy: torch.Tensor
    cx: torch.Tensor
    cy: torch.Tensor
    height: int
    width: int

    def __len__(self) -> int:
        return self.camera_to_worlds.shape[0]

    @property
    def intrinsics(self) -> torch.Tensor:
        return torch.stack([self.fx, self.fy, self.cx, self.cy], -1).float().contiguous()

    def to(self, device) -> "Cameras":
        return Cameras(self.camera_to_worlds.to(device), self.fx.to(device), self.fy.to(device), self.cx.to(device), self.cy.to(device),
                       self.height, self.width)


@dataclass
class SceneBox:
    aabb: torch.Tensor  # [2,3]


@dataclass
class DataparserOutputs:
    image_filenames: List[Path]
    cameras: Cameras
    scene_box: SceneBox
    dataparser_scale: float
    dataparser_transform: torch.Tensor  # [3,4]
    metadata: Dict = field(default_factory=dict)

    def save_dataparser_transform(self, path) -> None:
        """nerfstudio ``DataparserOutputs.save_dataparser_transform``: ``Trainer.train()`` writes ``dataparser_transforms.json`` next to
        the checkpoints before step 0 (ns-export / ns-render read the world transform back from it)."""
        save_dataparser_transform(self.dataparser_transform, self.dataparser_scale, path)


def save_dataparser_transform(transform: torch.Tensor, scale: float, path) -> None:
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_text(json.dumps({"transform": torch.as_tensor(transform).tolist(), "scale": float(scale)}, indent=4))


@dataclass
class UMHSDataParserConfig:
    data: Path = Path()
    scale_factor: float = 1.0
    downscale_factor: Optional[int] = None
    scene_scale: float = 1.0
    orientation_method: Literal["pca", "up", "vertical", "none"] = "up"
    center_method: Literal["poses", "focus", "none"] = "poses"
    auto_scale_poses: bool = True
    eval_mode: Literal["fraction", "filename", "interval", "all"] = "filename"
    train_split_fraction: float = 0.9
    eval_interval: int = 8
    depth_unit_scale_factor: float = 1e-3
    mask_color: Optional[tuple] = None
    load_3D_points: bool = False
    num_classes: int = 5

    def setup(self) -> "UMHSDataParser":
        return UMHSDataParser(self)


def _rotation_between(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a, b = a / torch.linalg.norm(a), b / torch.linalg.norm(b)
    v, c = torch.linalg.cross(a, b), torch.dot(a, b)
    if float(c) < -1 + 1e-8:  # opposite: perturb and retry (camera_utils.rotation_matrix_between)
        return _rotation_between(a + (torch.rand(3) - 0.5) * 0.01, b)
    s = torch.linalg.norm(v)
    K = torch.tensor([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    return torch.eye(3) + K + K @ K * ((1 - c) / (s ** 2 + 1e-8))


def auto_orient_and_center_poses(poses: torch.Tensor, method: str = "up", center_method: str = "poses"):
    """camera_utils.auto_orient_and_center_poses for orientation "up"/"none" and centring "poses"/"none"."""
    if method not in ("up", "none") or center_method not in ("poses", "none"):
        raise NotImplementedError(f"orientation_method={method!r} / center_method={center_method!r}: only up|none and poses|none")
    translation = poses[:, :3, 3].mean(0) if center_method == "poses" else torch.zeros(3)
    if method == "up":
        up = poses[:, :3, 1].mean(0)
        R = _rotation_between(up / torch.linalg.norm(up), torch.tensor([0.0, 0.0, 1.0]))
        transform = torch.cat([R, R @ -translation[..., None]], dim=-1)
    else:
        transform = torch.eye(4)[:3].clone()
        transform[:3, 3] = -translation
    return transform @ poses, transform


def split_by_filename(image_filenames):
    i_train, i_eval = [], []
    for i, f in enumerate(image_filenames):
        base = os.path.basename(os.path.dirname(f))
        if "train" in base:
            i_train.append(i)
        elif "eval" in base:
            i_eval.append(i)
        else:
            raise ValueError("frame should contain train/eval in its name to use this eval-frame-index eval mode")
    return np.array(i_train, dtype=np.int64), np.array(i_eval, dtype=np.int64)


def split_fraction(n: int, train_fraction: float):
    n_train = math.ceil(n * train_fraction)
    i_all = np.arange(n)
    i_train = np.linspace(0, n - 1, n_train, dtype=int)
    return i_train, np.setdiff1d(i_all, i_train)


def split_interval(n: int, interval: int):
    i_all = np.arange(n)
    return i_all[i_all % interval != 0], i_all[i_all % interval == 0]


class UMHSDataParser:
    def __init__(self, config: UMHSDataParserConfig):
        self.config = config

    def get_dataparser_outputs(self, split: str = "train") -> DataparserOutputs:
        return self._generate_dataparser_outputs(split)

    def _generate_dataparser_outputs(self, split="train") -> DataparserOutputs:
        c = self.config
        data = Path(c.data)
        assert data.exists(), f"Data directory {data} does not exist."
        meta_path, data_dir = (data, data.parent) if data.suffix == ".json" else (data / "transforms.json", data)
        with open(meta_path) as f:
            meta = json.load(f)
        if c.downscale_factor not in (None, 1):
            raise NotImplementedError("scale factors are not supported for hyperspectral data (hs_dataloader.py:37)")
        for key in ("k1", "k2", "k3", "p1", "p2", "distortion_params"):
            if key in meta and np.any(np.asarray(meta[key], dtype=np.float64) != 0):
                raise NotImplementedError("lens distortion is not handled by the HIP ray generator")
        if meta.get("camera_model", "OPENCV") not in ("OPENCV", "PINHOLE", "SIMPLE_PINHOLE"):
            raise NotImplementedError(f"camera_model {meta['camera_model']}: perspective cameras only")
        if c.load_3D_points:
            raise NotImplementedError("load_3D_points is not used by this method")

        frames = sorted(meta["frames"], key=lambda fr: str(data_dir / Path(fr["file_path"])))
        per = {k: (k not in meta) for k in ("fl_x", "fl_y", "cx", "cy", "h", "w")}
        vals = {k: [] for k in per}
        image_filenames, hs_filenames, poses = [], [], []
        for fr in frames:
            for k, per_frame in per.items():
                if per_frame:
                    assert k in fr, f"{k} not specified in frame"
                    vals[k].append(float(fr[k]))
            image_filenames.append(data_dir / Path(fr["file_path"]))
            poses.append(np.array(fr["transform_matrix"]))
            if "hyperspectral_file_path" in fr:
                hs_filenames.append(data_dir / Path(fr["hyperspectral_file_path"]))
        assert len(hs_filenames) in (0, len(image_filenames)), \
            "Different number of image and hyperspectral filenames: hyperspectral_file_path must be on every frame or none"

        n = len(image_filenames)
        if f"{split}_filenames" in meta:
            wanted = {data_dir / Path(x) for x in meta[f"{split}_filenames"]}
            missing = wanted.difference(image_filenames)
            if missing:
                raise RuntimeError(f"Some filenames for split {split} were not found: {missing}.")
            indices = np.array([i for i, p in enumerate(image_filenames) if p in wanted], dtype=np.int64)
        elif any(f"{s}_filenames" in meta for s in ("train", "val", "test")):
            raise RuntimeError(f"The dataset's list of filenames for split {split} is missing.")
        else:
            if c.eval_mode == "filename":
                i_train, i_eval = split_by_filename(image_filenames)
            elif c.eval_mode == "fraction":
                i_train, i_eval = split_fraction(n, c.train_split_fraction)
            elif c.eval_mode == "interval":
                i_train, i_eval = split_interval(n, c.eval_interval)
            elif c.eval_mode == "all":
                i_train = i_eval = np.arange(n)
            else:
                raise ValueError(f"Unknown eval mode {c.eval_mode}")
            if split == "train":
                indices = i_train
            elif split in ("val", "test"):
                indices = i_eval
            else:
                raise ValueError(f"Unknown dataparser split {split}")

        poses_t = torch.from_numpy(np.array(poses).astype(np.float32))
        poses_t, transform = auto_orient_and_center_poses(poses_t, meta.get("orientation_override", c.orientation_method), c.center_method)
        scale = 1.0
        if c.auto_scale_poses:
            scale /= float(torch.max(torch.abs(poses_t[:, :3, 3])))
        scale *= c.scale_factor
        poses_t[:, :3, 3] *= scale

        idx = torch.as_tensor(indices, dtype=torch.long)
        pick = lambda k: (torch.full((len(idx),), float(meta[k])) if not per[k] else torch.tensor(vals[k], dtype=torch.float32)[idx])
        hs, ws = pick("h"), pick("w")
        if len(idx) and (hs.min() != hs.max() or ws.min() != ws.max()):
            raise NotImplementedError("frames of different size cannot share one resident stack")
        cameras = Cameras(poses_t[idx][:, :3, :4].contiguous(), pick("fl_x"), pick("fl_y"), pick("cx"), pick("cy"),
                          int(hs[0]) if len(idx) else 0, int(ws[0]) if len(idx) else 0)
        wavelengths = None
        if hs_filenames:
            assert "wavelengths" in meta, "Wavelengths not specified in metadata"
            wavelengths = [int(x) for x in meta["wavelengths"]]
        if "applied_scale" in meta:
            scale *= float(meta["applied_scale"])
        s = c.scene_scale
        return DataparserOutputs(
            image_filenames=[image_filenames[i] for i in indices], cameras=cameras,
            scene_box=SceneBox(torch.tensor([[-s, -s, -s], [s, s, s]], dtype=torch.float32)), dataparser_scale=scale,
            dataparser_transform=transform,
            metadata={"hs_filenames": [hs_filenames[i] for i in indices] if hs_filenames else None, "split": split,
                      "num_classes": c.num_classes, "wavelengths": wavelengths, "height": cameras.height, "width": cameras.width})
