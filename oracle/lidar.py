"""Oracle (TEST INFRASTRUCTURE) -- numpy restatement of the LiDAR -> sparse depth image projection and
the pixel->LiDAR transform.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; the product path never does.

Follows /root/reference/creste/utils/projection.py:11-34 (`get_pixel2pts_transform`), :37-61
(`get_pts2pixel_transform`) and :64-155 (`pixels_to_depth`).

PINNED (round 4): tests/golden/projection.npz holds the outputs of the reference's own three functions on two synthetic
scans (tests/golden/make_golden.py::gen_projection; behind-camera, z_cam == 0, out-of-image, int32-clip and
duplicate-pixel cases, both `depth_priority` values, every `return_keys` entry), and
tests/test_oracle_golden.py::test_projection_matches_reference checks this module against it exactly.  The one line of
`pixels_to_depth` the generator could not execute as written is its `torch_scatter.scatter` call (requirements.txt:1,
un-vendored, absent here): there it is backed by `torch.scatter_reduce(include_self=False)` on a zero image, which is
torch_scatter's documented result (reduce over equal indices, untouched slots 0).
"""
import numpy as np


def pixel2pts_transform(lidar2cam, R, P):
    """T_cam->lidar @ [R^T] @ [inv(P[:3,:3])] as 4x4 (projection.py:11-34)."""
    T = np.eye(4); T[:3, :] = np.asarray(lidar2cam)[:3, :]
    Tc = np.eye(4); Tc[:3, :3] = np.asarray(R).T
    Pm = np.eye(4); Pm[:3, :3] = np.linalg.inv(np.asarray(P)[:3, :3])
    return np.linalg.inv(T) @ Tc @ Pm


def pts2pixel_transform(lidar2cam, R, P):
    """[P[:3,:3]] @ [R] @ T_lidar->cam as 4x4 (projection.py:37-61): LiDAR xyz1 -> (u z, v z, z, 1)."""
    T = np.eye(4); T[:3, :] = np.asarray(lidar2cam)[:3, :]
    Tc = np.eye(4); Tc[:3, :3] = np.asarray(R)
    Pm = np.eye(4); Pm[:3, :3] = np.asarray(P)[:3, :3]
    return Pm @ Tc @ T


def project(points, lidar2camrect, H, W):
    """The per-point half of `pixels_to_depth` (projection.py:81-108): -> (uv int32 [N,2], z_cam float64 [N], keep bool [N]).
    uv = trunc(clip(xy / z, int32 range)) for EVERY point (nan -> INT32_MIN, as numpy's cast does on this platform);
    keep = z_cam > 0 and the pixel lies inside the image."""
    pc = np.asarray(points)[:, :3].astype(np.float64)
    homo = np.hstack((pc, np.ones((pc.shape[0], 1))))
    cam = (np.asarray(lidar2camrect, dtype=np.float64) @ homo.T).T[:, :3]
    i32 = np.iinfo(np.int32)
    with np.errstate(all="ignore"):
        uv = cam / cam[:, 2:3]
        uv = np.clip(uv, i32.min, i32.max)
        uv = np.where(np.isnan(uv), float(i32.min), uv).astype(np.int32)[:, :2]
    keep = (cam[:, 2] > 0) & (uv[:, 0] >= 0) & (uv[:, 0] < W) & (uv[:, 1] >= 0) & (uv[:, 1] < H)
    return uv, cam[:, 2], keep


def pixels_to_depth(points, lidar2camrect, H, W, reduce="max"):
    """All five `return_keys` of the reference function (projection.py:110-153) as a dict:
    pc_mask [N] bool, pc_pts [K,2] int32 (u, v of the kept points, scan order), depth [H,W] float32 (the LAST kept point
    that lands on a pixel wins: numpy fancy assignment, :116-118), image_pts [M,2] int64 ((u, v) of the pixels whose
    reduced depth is non-zero, row-major pixel order) and image_depth [M] float64 (reduce = 'max' | 'min' of z_cam)."""
    uv, z, keep = project(points, lidar2camrect, H, W)
    kuv, kz = uv[keep], z[keep]
    loc = kuv[:, 1].astype(np.int64) * W + kuv[:, 0]
    last = np.zeros(H * W, dtype=np.float32)
    # last write wins: walk the duplicates backwards and keep the first one seen
    rloc, first = np.unique(loc[::-1], return_index=True)
    last[rloc] = kz[::-1][first].astype(np.float32)
    red = depth_image(points, lidar2camrect, H, W, reduce=reduce)
    vs, us = np.nonzero(red)
    return {"pc_mask": keep, "pc_pts": kuv, "depth": last.reshape(H, W),
            "image_pts": np.stack([us, vs], axis=1).astype(np.int64), "image_depth": red[vs, us]}


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
