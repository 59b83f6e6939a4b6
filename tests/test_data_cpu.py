"""SURVEY 8(f)-3 on CPU: on-disk format reader + the oracle's ray generator (no GPU, no compute calls into the library)."""
import json

import numpy as np
import pytest
import torch

from oracle import torch_ref as T
from umhsnerf.data.umhs_dataparser import UMHSDataParserConfig, auto_orient_and_center_poses, split_fraction, split_interval
from umhsnerf.data.utils.hs_dataloader import HyperspectralDataset


def _pose(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    m = np.eye(4)
    m[:3, :3], m[:3, 3] = q, rng.normal(size=3) * 3
    return m


def make_scene(root, n_train=5, n_eval=2, H=6, W=8, B=7, per_frame_intrinsics=False, seed=0):
    rng = np.random.default_rng(seed)
    frames = []
    for split, cnt in (("train", n_train), ("eval", n_eval)):
        (root / split).mkdir(parents=True)
        (root / f"hs_{split}").mkdir()
        for i in reversed(range(cnt)):  # unsorted on purpose
            np.save(root / split / f"r_{i:03d}.npy", (rng.random((H, W, 4)) * 255).astype(np.uint8))
            np.save(root / f"hs_{split}" / f"r_{i:03d}.npy", (rng.random((H, W, B)) * 1.4 - 0.2).astype(np.float32))  # exercises the clamp
            fr = {"file_path": f"{split}/r_{i:03d}.npy", "hyperspectral_file_path": f"hs_{split}/r_{i:03d}.npy", "transform_matrix": _pose(rng).tolist()}
            if per_frame_intrinsics:
                fr.update(fl_x=10.0 + i, fl_y=11.0 + i, cx=W / 2, cy=H / 2, h=H, w=W)
            frames.append(fr)
    meta = {"frames": frames, "wavelengths": [400 + 10 * k for k in range(B)]}
    if not per_frame_intrinsics:
        meta.update(fl_x=10.0, fl_y=11.0, cx=W / 2, cy=H / 2, h=H, w=W)
    (root / "transforms.json").write_text(json.dumps(meta))
    return meta


def test_dataparser_reads_the_reference_format(tmp_path):
    meta = make_scene(tmp_path)
    parser = UMHSDataParserConfig(data=tmp_path, num_classes=3).setup()
    tr, ev = parser.get_dataparser_outputs("train"), parser.get_dataparser_outputs("val")
    assert [p.name for p in tr.image_filenames] == [f"r_{i:03d}.npy" for i in range(5)] and len(ev.image_filenames) == 2
    assert all("train" in str(p) for p in tr.image_filenames) and all("eval" in str(p) for p in ev.image_filenames)
    assert tr.metadata["wavelengths"] == meta["wavelengths"] and tr.metadata["num_classes"] == 3
    assert tr.cameras.height == 6 and tr.cameras.width == 8 and tuple(tr.cameras.intrinsics[0].tolist()) == (10.0, 11.0, 4.0, 3.0)
    # poses: oriented "up", centred on the mean origin, scaled so the farthest coordinate sits on the +/-1 box -- against the oracle
    frames = sorted(meta["frames"], key=lambda f: str(tmp_path / f["file_path"]))
    raw = torch.tensor(np.array([f["transform_matrix"] for f in frames]), dtype=torch.float32)
    want, transform = T.auto_orient_and_center_poses(raw, "up", "poses")
    scale = 1.0 / float(want[:, :3, 3].abs().max())
    want[:, :3, 3] *= scale
    idx_train = [i for i, f in enumerate(frames) if f["file_path"].startswith("train")]
    torch.testing.assert_close(tr.cameras.camera_to_worlds, want[idx_train][:, :3], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(tr.dataparser_transform, transform, rtol=1e-5, atol=1e-6)
    assert abs(tr.dataparser_scale - scale) < 1e-6
    allc2w = torch.cat([tr.cameras.camera_to_worlds, ev.cameras.camera_to_worlds])
    assert abs(float(allc2w[:, :, 3].abs().max()) - 1.0) < 1e-5
    up = allc2w[:, :3, 1].mean(0)
    assert float(up[2] / up.norm()) > 0.999  # mean camera-up now points along +z
    torch.testing.assert_close(tr.scene_box.aabb, torch.tensor([[-1.0] * 3, [1.0] * 3]))
    ds = HyperspectralDataset(tr)
    assert ds.image.shape == (5, 6, 8, 4) and ds.hs_image.shape == (5, 6, 8, 7) and ds.image.dtype == torch.float32
    assert float(ds.hs_image.min()) == 0.0 and float(ds.hs_image.max()) == 1.0  # clamp(0,1) of hs_dataloader.py:50
    first = np.load(tmp_path / "train" / "r_000.npy").astype(np.float32) / 255.0
    np.testing.assert_array_equal(ds.image[0].numpy(), first)


def test_dataparser_per_frame_intrinsics_splits_and_errors(tmp_path):
    make_scene(tmp_path, per_frame_intrinsics=True)
    out = UMHSDataParserConfig(data=tmp_path / "transforms.json", eval_mode="all", orientation_method="none", center_method="none",
                               auto_scale_poses=False).setup().get_dataparser_outputs("train")
    assert len(out.image_filenames) == 7 and out.dataparser_scale == 1.0
    assert sorted(out.cameras.fx.tolist()) == sorted([10.0, 11.0, 10.0, 11.0, 12.0, 13.0, 14.0])
    meta = json.loads((tmp_path / "transforms.json").read_text())
    frames = sorted(meta["frames"], key=lambda f: str(tmp_path / f["file_path"]))
    np.testing.assert_allclose(out.cameras.camera_to_worlds.numpy(), np.array([f["transform_matrix"] for f in frames])[:, :3], rtol=1e-6)
    i_tr, i_ev = split_fraction(10, 0.9)
    assert len(i_tr) == 9 and len(i_ev) == 1 and set(i_tr) | set(i_ev) == set(range(10))
    i_tr, i_ev = split_interval(17, 8)
    assert list(i_ev) == [0, 8, 16] and len(i_tr) == 14
    del meta["wavelengths"]
    (tmp_path / "transforms.json").write_text(json.dumps(meta))
    with pytest.raises(AssertionError, match="Wavelengths"):
        UMHSDataParserConfig(data=tmp_path).setup().get_dataparser_outputs("train")
    meta["wavelengths"] = list(range(7))
    meta["frames"][0].pop("hyperspectral_file_path")
    (tmp_path / "transforms.json").write_text(json.dumps(meta))
    with pytest.raises(AssertionError, match="hyperspectral"):
        UMHSDataParserConfig(data=tmp_path).setup().get_dataparser_outputs("train")
    with pytest.raises(NotImplementedError):
        auto_orient_and_center_poses(torch.eye(4)[None], "pca", "poses")


def test_oracle_ray_generator_analytic_cases():
    c2w = torch.eye(4)[:3][None].clone()
    c2w[0, :, 3] = torch.tensor([1.0, 2.0, 3.0])
    intr = torch.tensor([[100.0, 50.0, 4.0, 3.0]])
    # half a pixel left of / above the principal point: x < 0, y > 0 (image y runs down, camera y up), looking down -z
    o, d, area, nrm = T.generate_rays(torch.tensor([[0, 2, 3]]), torch.cat([c2w, c2w]), torch.cat([intr, intr + 0.5]))
    assert o.tolist() == [[1.0, 2.0, 3.0]]
    torch.testing.assert_close(d, torch.tensor([[-0.005, 0.01, -1.0]]) / torch.tensor([0.005, 0.01, -1.0]).norm())
    # +x pixel -> +x direction, +y pixel (down the image) -> -y direction; pixel_area ~ 1/(fx fy) near the axis
    o, d, area, nrm = T.generate_rays(torch.tensor([[0, 3, 4], [0, 3, 14], [0, 13, 4]]), c2w, intr)
    assert abs(float(d[0, 0]) - 0.005) < 1e-4 and float(d[1, 0]) > 0.1 and float(d[2, 1]) < -0.15
    assert abs(float(area[0]) * 100 * 50 - 1.0) < 1e-2
    torch.testing.assert_close(d.norm(dim=-1), torch.ones(3))
    torch.testing.assert_close(nrm[1, 0], torch.tensor([0.105, -0.01, -1.0]).norm())
    # rotation: camera x axis -> world y
    rot = torch.tensor([[0.0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0]])[None]
    _, d2, _, _ = T.generate_rays(torch.tensor([[0, 3, 14]]), rot, intr)
    torch.testing.assert_close(d2[0], torch.stack([-d[1, 1], d[1, 0], d[1, 2]]))
    idx = T.pixel_sample_indices(torch.tensor([[0.0, 0.5, 0.999], [0.99, 0.0, 0.25]]), 10, 6, 8)
    assert idx.tolist() == [[0, 3, 7], [9, 0, 2]]
    stack = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).view(2, 3, 4, 5)
    assert T.gather_pixels(torch.tensor([[1, 2, 3]]), stack).tolist() == [stack[1, 2, 3].tolist()]


def test_lazy_training_metrics_behave_like_the_dict_they_replace():
    """UMHSModel.get_metrics_dict hands the trainer a dict whose values are computed when read (host logic only)."""
    import torch
    from umhsnerf.umhs_model import LazyMetrics

    calls = []

    def thunk(name, v):
        def f():
            calls.append(name)
            return torch.tensor(v)
        return f

    m = LazyMetrics({"psnr": thunk("psnr", 30.0), "rmse": thunk("rmse", 0.1)})
    assert "psnr" in m and "nope" not in m and calls == []
    assert float(m["psnr"]) == 30.0 and calls == ["psnr"]
    assert float(m["psnr"]) == 30.0 and calls == ["psnr"]  # computed once
    assert m.get("nope") is None and float(m.get("rmse")) == pytest.approx(0.1) and calls == ["psnr", "rmse"]
    with pytest.raises(KeyError):
        m["nope"]
    m2 = LazyMetrics({"a": thunk("a", 1.0), "b": thunk("b", 2.0)})
    assert sorted(m2.keys()) == ["a", "b"] and len(m2) == 2 and {k: float(v) for k, v in m2.items()} == {"a": 1.0, "b": 2.0}
    assert dict(LazyMetrics({"a": thunk("a", 1.0)})) == {"a": torch.tensor(1.0)}
