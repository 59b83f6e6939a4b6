"""UMHSDataManager (mirror of ``umhsnerf/data/umhs_datamanager.py``) with the image stacks resident in HBM.

``next_train`` (``:95-108``) is: draw pixel indices, gather their rows from the image / hs_image stacks, generate the rays.
In the reference those are nerfstudio's PixelSampler, a fancy-index gather per key and ``Cameras.generate_rays``; here each
is one kernel of libumhs_hip.so (``umhs_pixel_indices`` / ``umhs_pixel_gather`` / ``umhs_raygen``) on tensors that never
leave the GPU.  The uniform draws come from ``torch.rand`` on the device generator, so runs are seeded the same way."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple, Type

import torch

from .. import ops
from .._ns_compat import InstantiateConfig, RayBundle
from .umhs_dataparser import Cameras, DataparserOutputs, UMHSDataParserConfig, save_dataparser_transform
from .utils.hs_dataloader import HyperspectralDataset


@dataclass
class UMHSDataManagerConfig(InstantiateConfig):
    """``UMHSDataManagerConfig`` (umhs_datamanager.py:36-47): ``setup(device=, test_mode=, world_size=, local_rank=, num_classes=)``."""

    _target: Type = field(default_factory=lambda: UMHSDataManager)
    dataparser: Any = field(default_factory=UMHSDataParserConfig)
    train_num_rays_per_batch: int = 4096
    eval_num_rays_per_batch: int = 4096
    images_on_gpu: bool = True
    patch_size: int = 1


class _DatasetView:
    """What is read off ``datamanager.train_dataset`` / ``eval_dataset``: by the pipeline (umhs_pipeline.py:96-104: scene_box,
    metadata, len()) and by nerfstudio's Trainer / viewer before step 0 (``cameras``, ``dataset[i]["image"]``, ``image_filenames``:
    the viewer's init_scene draws the training cameras with their thumbnails)."""

    def __init__(self, split, scene_box, metadata, image_filenames=None):
        self._split, self.scene_box, self.metadata = split, scene_box, metadata
        self.image_filenames = list(image_filenames) if image_filenames is not None else [f"frame_{i:05d}" for i in range(len(split))]

    def __len__(self) -> int:
        return len(self._split)

    @property
    def cameras(self):
        return self._split.cameras

    def __getitem__(self, i: int) -> Dict:
        i = int(i)
        if not 0 <= i < len(self):
            raise IndexError(i)
        item = {"image_idx": i, "image": self._split.image[i]}
        if item["image"].dtype == torch.uint8:
            item["image"] = item["image"].float() / 255.0
        if self._split.hs_image is not None:
            item["hs_image"] = self._split.hs_image[i]
        return item


class _ResidentOutputs:
    """``train_dataparser_outputs`` of a datamanager that was handed resident splits directly (tests, benchmarks): the metadata, and an
    identity world transform for the Trainer's ``save_dataparser_transform``."""

    def __init__(self, metadata):
        self.metadata, self.dataparser_scale = metadata, 1.0
        self.dataparser_transform = torch.eye(4)[:3]

    def save_dataparser_transform(self, path) -> None:
        save_dataparser_transform(self.dataparser_transform, self.dataparser_scale, path)


class ResidentSplit:
    """One split: cameras + contiguous [n,H,W,K] stacks -- on the device (``--images-on-gpu True``, scripts/hotdog.sh:8: the batch
    rows are gathered by ``umhs_pixel_gather``), or in host memory (``False``, scripts/pinecone.sh:14: the rows of a batch are indexed
    on the host, as the reference's dataloader does, and only they travel to the device)."""

    def __init__(self, cameras: Cameras, image: torch.Tensor, hs_image: Optional[torch.Tensor], device, on_gpu: bool = True):
        self.device, self.on_gpu = torch.device(device), on_gpu
        self.cameras = cameras.to(device)
        self.c2w = self.cameras.camera_to_worlds.float().contiguous()
        self.intrinsics = self.cameras.intrinsics
        place = (lambda t: t.to(device).contiguous()) if on_gpu else (lambda t: t.cpu().contiguous())
        self.image = place(image)
        self.hs_image = place(hs_image) if hs_image is not None else None
        n, h, w = self.image.shape[:3]
        if (h, w) != (cameras.height, cameras.width) or n != len(cameras):
            raise ValueError(f"stack {tuple(self.image.shape)} does not match {n} cameras of {cameras.height}x{cameras.width}")

    def __len__(self) -> int:
        return self.image.shape[0]

    def sample(self, num_rays: int, generator=None) -> Tuple[RayBundle, Dict]:
        n, h, w = self.image.shape[:3]
        u = torch.rand((num_rays, 3), device=self.device, generator=generator)
        indices = ops.pixel_indices(u, n, h, w)
        return self.rays(indices), self.batch(indices)

    def _rows(self, indices: torch.Tensor, stack: torch.Tensor) -> torch.Tensor:
        if self.on_gpu:
            return ops.pixel_gather(indices, stack)
        n, h, w = stack.shape[:3]
        i = indices.cpu()  # same clamping as umhs_pixel_gather (a uniform draw of exactly 1.0 rounds up to the extent)
        rows = stack[i[:, 0].clamp(0, n - 1), i[:, 1].clamp(0, h - 1), i[:, 2].clamp(0, w - 1)]
        rows = rows.float() / 255.0 if stack.dtype == torch.uint8 else rows.float()
        return rows.to(self.device, non_blocking=True)

    def batch(self, indices: torch.Tensor) -> Dict:
        b = {"image": self._rows(indices, self.image), "indices": indices}
        if self.hs_image is not None:
            b["hs_image"] = self._rows(indices, self.hs_image)
        return b

    def rays(self, indices: torch.Tensor) -> RayBundle:
        o, d, area, nrm = ops.raygen(indices, self.c2w, self.intrinsics, want_area=True, want_norm=True)
        return RayBundle(origins=o, directions=d, pixel_area=area, camera_indices=indices[:, :1].contiguous(),
                         metadata={"directions_norm": nrm})

    def image_rays(self, camera_index: int) -> RayBundle:
        """``cameras.generate_rays(camera_indices=i, keep_shape=True)``: one ray per pixel, [H,W,...]."""
        _, h, w = self.image.shape[:3]
        dev = self.device
        yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
        idx = torch.stack([torch.full_like(yy, camera_index), yy, xx], -1).reshape(-1, 3).contiguous()
        rb = self.rays(idx)
        return RayBundle(origins=rb.origins.view(h, w, 3), directions=rb.directions.view(h, w, 3), pixel_area=rb.pixel_area.view(h, w, 1),
                         camera_indices=rb.camera_indices.view(h, w, 1))


class UMHSDataManager:
    """Train/eval splits of one scene, images on the GPU.  ``world_size``/``local_rank``: every rank keeps the full stacks
    and draws its own full per-rank batch from its own generator (seed + rank), as nerfstudio's data managers do."""

    def __init__(self, config: UMHSDataManagerConfig, device="cpu", test_mode: str = "val", world_size: int = 1, local_rank: int = 0,
                 num_classes: int = 5, seed: int = 42, train: Optional[ResidentSplit] = None, eval: Optional[ResidentSplit] = None,
                 metadata: Optional[Dict] = None, **kwargs):
        if config.patch_size != 1:
            raise NotImplementedError("patch_size > 1 (PatchPixelSampler) is not used by the reference's scripts")
        self.config, self.device, self.world_size, self.local_rank = config, torch.device(device), world_size, local_rank
        config.dataparser.num_classes = num_classes
        eval_names = None
        if train is None:
            parser = config.dataparser.setup()
            self.train_dataparser_outputs: DataparserOutputs = parser.get_dataparser_outputs("train")
            tr = HyperspectralDataset(self.train_dataparser_outputs)
            train = ResidentSplit(tr.cameras, tr.image, tr.hs_image, self.device, on_gpu=config.images_on_gpu)
            ev_out = parser.get_dataparser_outputs("val" if test_mode != "test" else "test")
            if len(ev_out.image_filenames):
                eval_names = ev_out.image_filenames
                ev = HyperspectralDataset(ev_out)
                eval = ResidentSplit(ev.cameras, ev.image, ev.hs_image, self.device, on_gpu=config.images_on_gpu)
            metadata = self.train_dataparser_outputs.metadata
            self.scene_box = self.train_dataparser_outputs.scene_box
        self.train_split, self.eval_split, self.metadata = train, eval, metadata or {}
        names = getattr(getattr(self, "train_dataparser_outputs", None), "image_filenames", None)
        self.train_dataset = _DatasetView(train, getattr(self, "scene_box", None), self.metadata, names)
        self.eval_dataset = _DatasetView(eval, getattr(self, "scene_box", None), self.metadata, eval_names) if eval is not None else None
        if not hasattr(self, "train_dataparser_outputs"):  # resident splits handed in directly (tests): same attribute, no parser behind it
            self.train_dataparser_outputs = _ResidentOutputs(self.metadata)
        self.train_count = self.eval_count = 0
        self.generator = torch.Generator(device=self.device)
        self.generator.manual_seed(seed + local_rank)
        self._eval_cursor = 0

    def to(self, device):  # the stacks were placed at construction (umhs_pipeline.py:94 calls datamanager.to(device))
        return self

    def get_param_groups(self) -> Dict:
        return {}

    def get_train_rays_per_batch(self) -> int:
        return self.config.train_num_rays_per_batch

    def get_eval_rays_per_batch(self) -> int:
        return self.config.eval_num_rays_per_batch

    def next_train(self, step: int) -> Tuple[RayBundle, Dict]:
        self.train_count += 1
        return self.train_split.sample(self.config.train_num_rays_per_batch, self.generator)

    def next_eval(self, step: int) -> Tuple[RayBundle, Dict]:
        self.eval_count += 1
        return (self.eval_split or self.train_split).sample(self.config.eval_num_rays_per_batch, self.generator)

    def next_eval_image(self, step: int):
        split = self.eval_split or self.train_split
        i = self._eval_cursor % len(split)
        self._eval_cursor += 1
        batch = {"image": split.image[i].to(self.device), "image_idx": i}
        if split.hs_image is not None:
            batch["hs_image"] = split.hs_image[i].to(self.device)
        return split.image_rays(i), batch
