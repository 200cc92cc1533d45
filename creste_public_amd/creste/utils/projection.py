"""LiDAR <-> image geometry of the input contract -- mirrors the functions of
/root/reference/creste/utils/projection.py that feed the hot path: `get_pixel2pts_transform` (:11-34,
builds the `p2p` matrix the splat consumes), `get_pts2pixel_transform` (:37-61) and `pixels_to_depth`
(:64-155, LiDAR scan -> sparse depth image; here a HIP kernel, `creste_lidar_depth_image_f32`)."""
import numpy as np
import torch

from ... import ops


def get_pixel2pts_transform(calib_dict):
    T_lidar_cam = np.eye(4)
    T_lidar_cam[:3, :] = np.asarray(calib_dict["lidar2cam"])[:3, :]
    T_canon = np.eye(4)
    T_canon[:3, :3] = np.asarray(calib_dict["R"]).T
    P_pix_cam = np.eye(4)
    P_pix_cam[:3, :3] = np.linalg.inv(np.asarray(calib_dict["P"])[:3, :3])
    return np.linalg.inv(T_lidar_cam) @ T_canon @ P_pix_cam


def get_pts2pixel_transform(calib_dict):
    T_lidar_cam = np.eye(4)
    T_lidar_cam[:3, :] = np.asarray(calib_dict["lidar2cam"])[:3, :]
    T_canon = np.eye(4)
    T_canon[:3, :3] = np.asarray(calib_dict["R"])
    P_pix_cam = np.eye(4)
    P_pix_cam[:3, :3] = np.asarray(calib_dict["P"])[:3, :3]
    return P_pix_cam @ T_canon @ T_lidar_cam


def pixels_to_depth(pc_np, calib, IMG_H, IMG_W, return_keys=['image_pts', 'image_depth'], IMG_DEBUG_FLAG=False,
                    depth_priority="max", device="cuda"):
    """The reference's signature and results (projection.py:64-155), computed by `creste_lidar_pixels_to_depth_f64`:
    pc_np [N,>=3] numpy / tensor (LiDAR frame), calib['lidar2camrect'] [3|4,4] numpy / tensor -> the arrays named by
    `return_keys`, as numpy, in the reference's dtypes: 'image_pts' int64 [M,2] (u, v), 'image_depth' float64 [M] (metres,
    max | min per pixel), 'depth' float32 [H,W] (last write wins), 'pc_pts' int32 [K,2], 'pc_mask' bool [N].
    `IMG_DEBUG_FLAG` is accepted and ignored (the reference writes test.png / pp_depth_max.png with cv2 under it)."""
    for key in return_keys:
        if key not in ("image_pts", "image_depth", "depth", "pc_pts", "pc_mask"):
            raise ValueError(f"Invalid key {key} in return_keys")
    pts = pc_np if isinstance(pc_np, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pc_np))
    if pts.dtype not in (torch.float32, torch.float64):
        pts = pts.to(torch.float64)
    l2c = calib["lidar2camrect"]
    l2c = l2c if isinstance(l2c, torch.Tensor) else torch.from_numpy(np.asarray(l2c))
    uv, mask, reduced, last = ops.lidar_pixels_to_depth(pts[:, :3].contiguous().to(device),
                                                        l2c.to(torch.float64).contiguous().to(device), IMG_H, IMG_W,
                                                        reduce=depth_priority)
    vals = {}
    if "image_pts" in return_keys or "image_depth" in return_keys:
        vu = torch.nonzero(reduced)                               # row-major pixel order, as np.nonzero
        vals["image_pts"] = vu[:, [1, 0]].cpu().numpy()
        vals["image_depth"] = reduced[vu[:, 0], vu[:, 1]].cpu().numpy()
    if "depth" in return_keys:
        vals["depth"] = last.cpu().numpy()
    if "pc_pts" in return_keys:
        vals["pc_pts"] = uv[mask].cpu().numpy()
    if "pc_mask" in return_keys:
        vals["pc_mask"] = mask.cpu().numpy()
    return [vals[k] for k in return_keys]


def lidar_depth_images(points: torch.Tensor, lidar2camrect: torch.Tensor, IMG_H: int, IMG_W: int,
                       out: torch.Tensor = None, scale: float = 1.0, depth_priority: str = "max"):
    """Batched GPU form of `pixels_to_depth(..., return_keys=['image_depth'], depth_priority=...)` -- the per-pixel
    scatter-max / scatter-min path (projection.py:124-128) that `depth_utils.py:27-30` uses, NOT the last-write-wins
    'depth' key (:116-118): points [B,N,>=3] fp32 (CUDA), lidar2camrect [B,4,4] float64 -> depth images
    [B,IMG_H,IMG_W] (z_cam * scale, 0 = no return)."""
    if out is None:
        out = torch.empty((points.shape[0], IMG_H, IMG_W), dtype=torch.float32, device=points.device)
    return ops.lidar_depth_image(points.contiguous(), lidar2camrect.to(torch.float64).contiguous(), IMG_H,
                                 IMG_W, out, scale=scale, reduce=depth_priority)
