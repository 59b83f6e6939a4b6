"""SURVEY 8(f)-3 on the GPU: pixel sampler / ray generator / GT gather kernels against the oracle, the resident data manager,
and training straight from it (sampler -> field -> compositing -> loss -> Adam)."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cams(n, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    q, _ = torch.linalg.qr(torch.randn(n, 3, 3, generator=g))
    c2w = torch.cat([q, torch.randn(n, 3, 1, generator=g)], -1).contiguous()
    intr = torch.stack([torch.rand(n, generator=g) * 500 + 100, torch.rand(n, generator=g) * 500 + 100,
                        torch.full((n,), W / 2) + torch.randn(n, generator=g), torch.full((n,), H / 2) + torch.randn(n, generator=g)], -1)
    return c2w, intr.contiguous()


def test_pixel_indices_and_gather_are_bit_exact():
    from umhsnerf import ops

    g = torch.Generator().manual_seed(1)
    n, H, W = 7, 37, 53
    u = torch.rand(5000, 3, generator=g)
    u[0] = torch.tensor([0.0, 0.0, 0.0])
    u[1] = torch.tensor([np.nextafter(np.float32(1), np.float32(0))] * 3)  # rounds up to (n, H, W): clamped by the gather
    idx = ops.pixel_indices(u.to(DEV), n, H, W)
    want = T.pixel_sample_indices(u, n, H, W)
    assert torch.equal(idx.cpu(), want) and idx.dtype == torch.int64
    ok = idx[2:]
    for K, dt in [(3, torch.float32), (4, torch.uint8), (31, torch.float32), (141, torch.float32), (1, torch.uint8)]:
        stack = (torch.rand(n, H, W, K, generator=g) * 255).to(torch.uint8) if dt == torch.uint8 else torch.rand(n, H, W, K, generator=g)
        got = ops.pixel_gather(ok, stack.to(DEV))
        assert torch.equal(got.cpu(), T.gather_pixels(ok.cpu(), stack)), (K, dt)
    edge = ops.pixel_gather(idx[:2], stack.to(DEV))  # out-of-range rows clamp to the last pixel instead of faulting
    assert torch.equal(edge.cpu(), torch.stack([stack[0, 0, 0], stack[-1, -1, -1]]).float() / 255.0)
    assert ops.pixel_gather(idx[:0], stack.to(DEV)).shape == (0, 1)


def test_raygen_matches_oracle():
    from umhsnerf import ops

    n, H, W = 9, 480, 640
    c2w, intr = _cams(n, H, W)
    g = torch.Generator().manual_seed(2)
    idx = T.pixel_sample_indices(torch.rand(20000, 3, generator=g), n, H, W)
    o, d, area, nrm = ops.raygen(idx.to(DEV), c2w.to(DEV), intr.to(DEV), want_area=True, want_norm=True)
    ro, rd, rarea, rn = T.generate_rays(idx, c2w, intr)
    assert torch.equal(o.cpu(), ro)
    torch.testing.assert_close(d.cpu(), rd, rtol=0, atol=2e-7)
    torch.testing.assert_close(nrm.cpu(), rn, rtol=2e-7, atol=0)
    torch.testing.assert_close(area.cpu(), rarea, rtol=2e-3, atol=0)  # difference of nearly equal unit vectors: cancellation
    o2, d2, a2, n2 = ops.raygen(idx.to(DEV), c2w.to(DEV), intr.to(DEV), want_area=False)
    assert a2 is None and n2 is None and torch.equal(d2, d)


def _split(n=4, H=24, W=32, B=8, seed=3, const=None):
    from umhsnerf.data.umhs_datamanager import ResidentSplit
    from umhsnerf.data.umhs_dataparser import Cameras

    g = torch.Generator().manual_seed(seed)
    # cameras on a sphere of radius 0.9 looking at the origin
    pos = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 0.9
    z = torch.nn.functional.normalize(pos, dim=-1)  # camera looks down -z
    x = torch.nn.functional.normalize(torch.linalg.cross(torch.tensor([[0.0, 0, 1]]).expand(n, 3), z), dim=-1)
    y = torch.linalg.cross(z, x)
    c2w = torch.stack([x, y, z, pos], -1).contiguous()
    cams = Cameras(c2w, torch.full((n,), 30.0), torch.full((n,), 30.0), torch.full((n,), W / 2), torch.full((n,), H / 2), H, W)
    hs = torch.rand(n, H, W, B, generator=g) if const is None else torch.full((n, H, W, B), const)
    rgb = torch.rand(n, H, W, 3, generator=g)
    return ResidentSplit(cams, rgb, hs, DEV), rgb, hs, c2w


def test_datamanager_next_train_matches_oracle_and_is_seeded():
    from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig

    split, rgb, hs, c2w = _split()
    dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=1000), device=DEV, seed=5, train=split)
    rb, batch = dm.next_train(0)
    gen = torch.Generator(device=DEV)
    gen.manual_seed(5)
    u = torch.rand((1000, 3), device=DEV, generator=gen).cpu()
    idx = T.pixel_sample_indices(u, 4, 24, 32)
    assert torch.equal(batch["indices"].cpu(), idx)
    assert torch.equal(batch["image"].cpu(), T.gather_pixels(idx, rgb)) and torch.equal(batch["hs_image"].cpu(), T.gather_pixels(idx, hs))
    ro, rd, rarea, _ = T.generate_rays(idx, c2w, split.intrinsics.cpu())
    assert torch.equal(rb.origins.cpu(), ro) and torch.equal(rb.camera_indices.cpu(), idx[:, :1])
    torch.testing.assert_close(rb.directions.cpu(), rd, rtol=0, atol=2e-7)
    rb2, batch2 = dm.next_train(1)
    assert not torch.equal(batch2["indices"], batch["indices"]) and dm.train_count == 2
    dm_r1 = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=1000), device=DEV, seed=5, train=split, world_size=2, local_rank=1)
    assert not torch.equal(dm_r1.next_train(0)[1]["indices"], batch["indices"])  # seed + rank: ranks draw different rays
    cam, full = dm.next_eval_image(0)
    assert cam.origins.shape == (24, 32, 3) and full["hs_image"].shape == (24, 32, 8)
    # the central rays of every camera point at the scene origin
    ctr = cam.directions[12, 16].cpu()
    assert float(torch.dot(ctr, -torch.nn.functional.normalize(c2w[0, :, 3], dim=0))) > 0.995


def test_training_from_the_datamanager_reduces_the_loss():
    from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    torch.manual_seed(0)
    B = 8
    split, _, _, _ = _split(n=6, B=B, const=0.6)
    with torch.no_grad():
        bands = list(np.linspace(420, 680, B))
    dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=2048), device=DEV, seed=1, train=split)
    cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, background_color="black")
    pipe = UMHSPipeline.from_packed_samples(cfg, DEV, metadata={"wavelengths": bands, "num_classes": 3}, seed=2, datamanager=dm)
    with torch.no_grad():  # a self-consistent target: rgb = converter(hs)
        split.image = pipe.model.converter(split.hs_image.view(-1, B)).view(*split.hs_image.shape[:3], 3).contiguous()
    losses = []
    for step in range(80):
        _, loss_dict, metrics = pipe.get_train_loss_dict(step)
        losses.append(float(sum(loss_dict.values()).detach()))
    assert np.isfinite(losses).all() and np.mean(losses[-10:]) < 0.5 * np.mean(losses[:5]), (losses[:5], losses[-10:])
    assert float(metrics["psnr_spectral"].detach()) > 10.0


@pytest.mark.parametrize("H,W,K", [(37, 53, 3), (64, 96, 31), (11, 11, 1), (23, 150, 141)])
def test_image_metrics_match_the_oracle(H, W, K):
    from umhsnerf import ops

    g = torch.Generator().manual_seed(H * W + K)
    gt = torch.rand(H, W, K, generator=g)
    pred = (gt + 0.1 * torch.randn(H, W, K, generator=g)).clamp(0, 1)
    pred[0, :3] = 0.0  # all-zero spectra: NaN angles, skipped by nanmean
    gt[1, 1] = 0.0
    want = T.image_metrics_ref(pred, gt, pred, gt)
    sse, sam, cnt = ops.pixel_metrics(pred.to(DEV), gt.to(DEV)).tolist()
    chw = lambda x: torch.moveaxis(x, -1, 0)[None]
    n_nan = int(torch.isnan(T.sam_ref(chw(pred), chw(gt))).sum())
    assert cnt == H * W - n_nan and n_nan >= 4
    assert abs(10 * np.log10(pred.numel() / sse) - want["psnr_spectral"]) < 1e-4
    assert abs(np.sqrt(sse / pred.numel()) - want["rmse_spectral"]) < 1e-6
    assert abs(sam / cnt - want["sam_spectral"]) < 2e-6
    got = float(ops.ssim(gt.to(DEV), pred.to(DEV)))
    assert abs(got - want["ssim_spectral"]) < 2e-5, (got, want["ssim_spectral"])
    assert abs(float(ops.ssim(gt.to(DEV), pred.to(DEV), data_range=1.0)) - float(T.ssim_ref(torch.moveaxis(gt, -1, 0)[None], torch.moveaxis(pred, -1, 0)[None], 1.0))) < 2e-5
    assert abs(float(ops.ssim(gt.to(DEV), gt.to(DEV))) - 1.0) < 1e-6
    if H == 11:
        with pytest.raises(ValueError):
            ops.ssim(gt[:10].to(DEV), pred[:10].to(DEV))


def test_eval_image_path_and_metrics(monkeypatch):
    from umhsnerf._ns_compat import RayBundle
    from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    torch.manual_seed(0)
    B = 8
    split, _, _, _ = _split(n=3, B=B)
    dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=1024), device=DEV, seed=1, train=split)
    cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, background_color="black")
    pipe = UMHSPipeline.from_packed_samples(cfg, DEV, metadata={"wavelengths": list(np.linspace(420, 680, B)), "num_classes": 3}, seed=2, datamanager=dm)
    for step in range(3):
        pipe.get_train_loss_dict(step)
    cam, batch = dm.next_eval_image(0)
    pipe.eval()  # no stratified jitter in the marcher
    pipe.model.config.eval_num_rays_per_chunk = 512  # the reference's value: still one launch for this 24x32 image
    out = pipe.model.get_outputs_for_camera_ray_bundle(cam)
    assert out["spectral"].shape == (24, 32, B) and out["rgb"].shape == (24, 32, 3) and out["accumulation"].shape == (24, 32, 1)
    assert out["abundances"].shape == (24, 32, 3) and out["seg_pred"].shape == (24, 32, 3)
    # image == the same rays through forward() as one flat bundle
    with torch.no_grad():
        flat = pipe.model(RayBundle(origins=cam.origins.reshape(-1, 3), directions=cam.directions.reshape(-1, 3)))
    assert torch.equal(flat["spectral"].view(24, 32, B), out["spectral"])
    # gradient-free rendering takes the per-ray path (mlp_base -> weights -> heads with the per-ray sums in the kernel, no [N, bands]
    # array); the per-sample path (what training's autograd form runs) renders the same image
    monkeypatch.setenv("UMHS_RENDER_PER_RAY", "0")
    ref = pipe.model.get_outputs_for_camera_ray_bundle(cam)
    monkeypatch.delenv("UMHS_RENDER_PER_RAY")
    assert set(ref) == set(out)
    for k in ref:
        if k in ("seg_raw", "seg_pred"):  # argmax outputs: equal wherever the two top classes are not within rounding of each other
            assert float((ref[k] != out[k]).float().mean()) < 0.01, k
        else:
            assert float((ref[k].float() - out[k].float()).abs().max()) <= 2e-5 * max(1.0, float(ref[k].float().abs().max())), k
    md, images = pipe.model.get_image_metrics_and_images(out, batch)
    want = T.image_metrics_ref(out["rgb"].cpu(), batch["image"].cpu(), out["spectral"].cpu(), batch["hs_image"].cpu())
    assert set(md) == set(want)
    for k in want:
        assert abs(md[k] - want[k]) < 5e-5 * max(1.0, abs(want[k])), (k, md[k], want[k])
    assert images["img"].shape == (24, 64, 3) and images["depth"].shape == (24, 32, 3) and images["se_per_pixel"].shape == (24, 32, 1)
    pipe.train()
    md2, _ = pipe.get_eval_image_metrics_and_images(1)
    assert md2["num_rays"] == 24 * 32 and np.isfinite(list(md2.values())).all()
    _, loss_dict, metrics = pipe.get_eval_loss_dict(0)
    assert np.isfinite(float(sum(loss_dict.values()))) and pipe.training


def test_host_resident_stacks_give_the_same_batches():
    """--pipeline.datamanager.images-on-gpu False (scripts/pinecone.sh:14, rgb+spectral.sh:15): the stacks stay in host memory and only
    the batch rows travel; same indices, same rows, same rays as the device-resident stacks (fp32 and uint8 images)."""
    from umhsnerf.data.umhs_datamanager import ResidentSplit, UMHSDataManager, UMHSDataManagerConfig

    dev_split, rgb, hs, _ = _split(n=5, B=21, seed=9)
    for img in (rgb, (rgb * 255).to(torch.uint8)):
        a = ResidentSplit(dev_split.cameras, img, hs, DEV, on_gpu=True)
        b = ResidentSplit(dev_split.cameras, img, hs, DEV, on_gpu=False)
        assert not b.image.is_cuda and not b.hs_image.is_cuda and a.image.is_cuda
        dma = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=777), device=DEV, seed=4, train=a)
        dmb = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=777, images_on_gpu=False), device=DEV, seed=4, train=b)
        for step in range(3):
            (ra, ba), (rb, bb) = dma.next_train(step), dmb.next_train(step)
            assert torch.equal(ba["indices"], bb["indices"]) and torch.equal(ra.origins, rb.origins) and torch.equal(ra.directions, rb.directions)
            assert bb["image"].is_cuda and torch.equal(ba["image"], bb["image"]) and torch.equal(ba["hs_image"], bb["hs_image"])
        ca, fa = dma.next_eval_image(0)
        cb, fb = dmb.next_eval_image(0)
        assert torch.equal(fa["hs_image"], fb["hs_image"]) and fb["image"].is_cuda and torch.equal(ca.directions, cb.directions)


def test_the_references_default_batch_of_36864_rays_trains():
    """VERDICT r3 #5a: the reference's own default `train_num_rays_per_batch = 9216 * 4` (umhs_config.py:46) through
    `get_train_loss_dict` -- occupancy-grid march, candidate density query, pruning, hot path, backward, Adam -- on the synthetic scene
    of bench.py's `sampler_step`: finite losses on every step, the loss falls, and the sizes that matter at this batch are recorded
    (gpurun_out/default_batch_36864.json: candidates and survivors per step, the hash-grid backward's record workspace, peak memory)."""
    import json
    import time

    from umhsnerf import _hip, ops
    from umhsnerf.data.umhs_datamanager import UMHSDataManager, UMHSDataManagerConfig
    from umhsnerf.umhs_config import umhs_method
    from umhsnerf.umhs_model import UMHSConfig
    from umhsnerf.umhs_pipeline import UMHSPipeline

    R = 9216 * 4
    dm_cfg = umhs_method.config.pipeline.datamanager
    assert int(dm_cfg.train_num_rays_per_batch) == R, "the registered method carries the reference's default batch"
    torch.manual_seed(0)
    B = 31
    split, _, _, _ = _split(n=6, H=64, W=64, B=B, const=0.6)
    bands = list(np.linspace(400, 700, B))
    dm = UMHSDataManager(UMHSDataManagerConfig(train_num_rays_per_batch=R), device=DEV, seed=1, train=split)
    cfg = UMHSConfig(method="rgb+spectral", pred_specular=True, temperature=0.4, background_color="random")
    pipe = UMHSPipeline.from_packed_samples(cfg, DEV, metadata={"wavelengths": bands, "num_classes": 6}, seed=2, datamanager=dm)
    with torch.no_grad():
        split.image = pipe.model.converter(split.hs_image.view(-1, B)).view(*split.hs_image.shape[:3], 3).contiguous()
    torch.cuda.reset_peak_memory_stats()
    grid = pipe.model.sampler.occupancy_grid
    losses, surv, cand, t_step = [], [], [], []
    for step in range(48):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, loss_dict, _ = pipe.get_train_loss_dict(step)
        torch.cuda.synchronize()
        t_step.append((time.perf_counter() - t0) * 1e3)
        losses.append(float(sum(loss_dict.values()).detach()))
        surv.append(int(out["num_samples_per_ray"].sum()))
        cand.append(int(getattr(grid, "last_candidates", 0)))
    assert np.isfinite(losses).all(), losses
    assert np.mean(losses[-8:]) < 0.7 * np.mean(losses[:4]), (losses[:4], losses[-8:])
    n_max = max(surv)
    ws = int(_hip.lib().umhs_hashgrid_bwd_workspace_bytes(n_max, ops.NUM_LEVELS, 19))
    assert ws > 0, "the partitioned hash-grid backward must be available at this sample count"
    rec = {"rays_per_batch": R, "candidates_per_step": [min(cand), max(cand)], "survivors_per_step": [min(surv), n_max],
           "hashgrid_bwd_workspace_bytes_at_max": ws, "hashgrid_bwd_workspace_bytes_per_sample": round(ws / n_max, 1),
           "peak_memory_allocated_bytes": int(torch.cuda.max_memory_allocated()), "peak_memory_reserved_bytes": int(torch.cuda.max_memory_reserved()),
           "ms_per_step_first_8": [round(t, 2) for t in t_step[:8]], "ms_per_step_last_8_median": round(float(np.median(t_step[-8:])), 2),
           "loss_first_4_mean": float(np.mean(losses[:4])), "loss_last_8_mean": float(np.mean(losses[-8:]))}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "default_batch_36864.json"), "w") as f:
        json.dump(rec, f, indent=1)
