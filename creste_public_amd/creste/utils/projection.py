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
