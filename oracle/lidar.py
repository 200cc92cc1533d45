"""Oracle (TEST INFRASTRUCTURE) -- numpy restatement of the LiDAR -> sparse depth image projection and
the pixel->LiDAR transform.

Follows /root/reference/creste/utils/projection.py:11-34 (`get_pixel2pts_transform`) and :64-155
(`pixels_to_depth`).  `pixels_to_depth` calls `torch_scatter.scatter(reduce=...)` (requirements.txt:1, un-vendored,
absent here) and the reference holds no golden vectors, so the function cannot be run: the per-pixel reduction is
restated from torch_scatter's documented semantics (reduce over equal indices, empty slots filled with 0) and
INDEPENDENTLY CROSS-CHECKED bit for bit against torch's own `scatter_reduce('amax'/'amin')`
(tests/test_trunk_hf.py::test_oracle_lidar_depth_image_matches_torch_scatter_reduce).
"""
import numpy as np


def pixel2pts_transform(lidar2cam, R, P):
    """T_cam->lidar @ [R^T] @ [inv(P[:3,:3])] as 4x4 (projection.py:11-34)."""
    T = np.eye(4); T[:3, :] = np.asarray(lidar2cam)[:3, :]
    Tc = np.eye(4); Tc[:3, :3] = np.asarray(R).T
    Pm = np.eye(4); Pm[:3, :3] = np.linalg.inv(np.asarray(P)[:3, :3])
    return np.linalg.inv(T) @ Tc @ Pm


def depth_image(points, lidar2camrect, H, W, reduce="max"):
    """points [N,>=3], lidar2camrect [4,4] (or [3,4]) -> float64 depth image [H,W] in the unit of z_cam."""
    pc = np.asarray(points)[:, :3].astype(np.float64)
    homo = np.hstack((pc, np.ones((pc.shape[0], 1))))
    cam = (np.asarray(lidar2camrect, dtype=np.float64) @ homo.T).T[:, :3]
    with np.errstate(divide="ignore", invalid="ignore"):
        uv = cam / cam[:, -1].reshape(-1, 1)
    i32 = np.iinfo(np.int32)
    uv = np.nan_to_num(np.clip(uv, i32.min, i32.max), nan=-1.0).astype(np.int32)[:, :2]
    ok = (cam[:, 2] > 0) & (uv[:, 0] >= 0) & (uv[:, 0] < W) & (uv[:, 1] >= 0) & (uv[:, 1] < H)
    loc = uv[ok, 1].astype(np.int64) * W + uv[ok, 0]
    z = cam[ok, 2]
    img = np.zeros(H * W, dtype=np.float64)
    if reduce == "max":
        np.maximum.at(img, loc, z)
    elif reduce == "min":
        big = np.full(H * W, np.inf)
        np.minimum.at(big, loc, z)
        img = np.where(np.isinf(big), 0.0, big)
    else:
        raise ValueError(reduce)
    return img.reshape(H, W)
