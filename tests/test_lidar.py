"""LiDAR scan -> sparse depth image: oracle sanity on CPU, HIP kernel vs oracle on GPU."""
import numpy as np
import pytest
import torch

from creste_public_amd import synth
from oracle import lidar as ol


def _brute(points, M, H, W):
    img = np.zeros((H, W))
    for p in points:
        c = M[:3] @ np.array([p[0], p[1], p[2], 1.0])
        if c[2] <= 0:
            continue
        u, v = int(np.trunc(c[0] / c[2])), int(np.trunc(c[1] / c[2]))
        if 0 <= u < W and 0 <= v < H:
            img[v, u] = max(img[v, u], c[2])
    return img


def test_oracle_depth_image_and_p2p():
    g = torch.Generator().manual_seed(0)
    H, W = 38, 76
    pts = synth.lidar_scan(1, g)[0, ::37].numpy()
    M = synth.lidar2camrect(1, H, W)[0].numpy()
    img = ol.depth_image(pts, M, H, W)
    assert np.array_equal(img, _brute(pts, M, H, W))
    assert 0 < (img > 0).mean() < 1 and img.max() <= 40.5
    mn = ol.depth_image(pts, M, H, W, reduce="min")
    assert ((mn <= img) | (img == 0)).all() and ((mn > 0) == (img > 0)).all()
    # p2p: pixel (u*d, v*d, d, 1) at full resolution -> the LiDAR point that projected there
    K, T = synth.camera_matrices(H, W)
    p2p = ol.pixel2pts_transform(np.linalg.inv(T.numpy())[:3], np.eye(3), np.hstack([K.numpy(), np.zeros((3, 1))]))
    x = np.array([5.0, 1.0, 0.3, 1.0])
    c = M @ x
    back = p2p @ np.array([c[0], c[1], c[2], 1.0])
    assert np.allclose(back[:3], x[:3], atol=1e-9)
    # the host mirror builds the same matrix
    from creste_public_amd.creste.utils.projection import get_pixel2pts_transform
    calib = dict(lidar2cam=np.linalg.inv(T.numpy()), R=np.eye(3), P=np.hstack([K.numpy(), np.zeros((3, 1))]))
    assert np.allclose(get_pixel2pts_transform(calib), p2p)


@pytest.mark.gpu
@pytest.mark.parametrize("reduce", ["max", "min"])
def test_hip_lidar_depth_image(reduce):
    from creste_public_amd.creste.utils.projection import lidar_depth_images
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 608, 1216
    pts = synth.lidar_scan(B, g)
    pts[0, :50] = torch.tensor([-3.0, 0.2, 0.1])            # behind the camera
    pts[1, 100:120, 0] = 1e-30                              # degenerate range -> huge |u|,|v|
    M = synth.lidar2camrect(B, H, W)
    rgbd = torch.zeros(B, 1, 4, H, W, device="cuda")        # write straight into the depth channel
    out = rgbd[:, 0, 3]
    lidar_depth_images(pts.cuda(), M.cuda(), H, W, out=out, scale=1000.0, depth_priority=reduce)
    for b in range(B):
        ref = ol.depth_image(pts[b].numpy(), M[b].numpy(), H, W, reduce=reduce)
        got = out[b].cpu().numpy().astype(np.float64)
        assert np.array_equal(got > 0, ref > 0)              # identical pixel set (integer work: exact)
        assert np.array_equal(got, (ref * 1000.0).astype(np.float32).astype(np.float64))
    assert (rgbd[:, 0, :3] == 0).all()


@pytest.mark.gpu
def test_hip_projection_matches_reference_golden(golden):
    """`creste_lidar_depth_image_f32`, `creste_lidar_pixels_to_depth_f64` and the host mirror's `pixels_to_depth` against
    the reference's own outputs (projection.npz), exact: every return key, both priorities, fp32 and float64 points."""
    from creste_public_amd.creste.utils import projection as mirror
    g = golden("projection.npz")
    keys = ["image_pts", "image_depth", "depth", "pc_pts", "pc_mask"]
    for tag in "ab":
        H, W = (int(v) for v in g[f"{tag}/hw"])
        pts, l2c = g[f"{tag}/points"], g[f"{tag}/lidar2camrect"]
        for prio in ("max", "min"):
            got = mirror.pixels_to_depth(pts, {"lidar2camrect": l2c}, H, W, return_keys=keys, depth_priority=prio)
            for k, v in zip(keys, got):
                ref = g[f"{tag}/{prio}/{k}"]
                assert v.dtype == ref.dtype and v.shape == ref.shape and np.array_equal(v, ref), (tag, prio, k)
            # default keys / tensor inputs (the reference's own .cpu().numpy() branches)
            ip, idp = mirror.pixels_to_depth(torch.from_numpy(pts), {"lidar2camrect": torch.from_numpy(l2c)}, H, W,
                                             depth_priority=prio)
            assert np.array_equal(ip, g[f"{tag}/{prio}/image_pts"]) and np.array_equal(idp, g[f"{tag}/{prio}/image_depth"])
            if pts.dtype == np.float32:
                # the batched fp32 image kernel that feeds rgbd[:, :, 3]: same pixels, float32(z * scale)
                ref_img = np.zeros((H, W))
                rp = g[f"{tag}/{prio}/image_pts"]
                ref_img[rp[:, 1], rp[:, 0]] = g[f"{tag}/{prio}/image_depth"]
                for scale in (1.0, 1000.0):
                    out = mirror.lidar_depth_images(torch.from_numpy(pts[None, :, :3].copy()).cuda(),
                                                    torch.from_numpy(l2c[None]).cuda(), H, W, scale=scale, depth_priority=prio)
                    assert np.array_equal(out[0].cpu().numpy(), (ref_img * scale).astype(np.float32))
    with pytest.raises(ValueError):
        mirror.pixels_to_depth(pts, {"lidar2camrect": l2c}, H, W, return_keys=["nope"])
