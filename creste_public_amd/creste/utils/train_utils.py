"""Host-side helpers of the hot path -- mirrors the functions of
/root/reference/creste/utils/train_utils.py that the path calls: create_trapezoidal_fov_mask
(:511-557), prefix_dict / merge_dict / merge_loss_dict (:560-599), resize_and_crop (:670-682),
earliest_pose_in_fov (:765-803).  Tiny index/bookkeeping tensors: plain torch (any device)."""
import torch
import torch.nn.functional as F


def create_trapezoidal_fov_mask(H, W, fov_top_angle=50, fov_bottom_angle=40, near=10, far=50):
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    cx, cy = W / 2, H / 2
    dist = torch.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
    ang = torch.atan2(xx - cx, cy - yy) * 180 / torch.pi
    ang[ang < -180] += 360
    top, bot = torch.full_like(dist, fov_top_angle / 2), torch.full_like(dist, fov_bottom_angle / 2)
    lerp = top + (bot - top) * ((dist - near) / (far - near))
    spread = torch.where(dist <= near, top, torch.where(dist >= far, bot, lerp))
    return torch.logical_and(torch.logical_and(dist >= near, dist <= far), torch.abs(ang) <= spread)


def prefix_dict(prefix, d, seprator="/"):
    return {prefix + seprator + k: v for k, v in d.items()}


def merge_dict(*args):
    ret = {}
    for a in args:
        ret.update(a if isinstance(a, dict) else prefix_dict(a[0], a[1]))
    return ret


def merge_loss_dict(full_dict, new_dict):
    full_dict.update(new_dict)
    return full_dict


def resize_and_crop(image, new_size, crop_bounds):
    y1, y2, x1, x2 = crop_bounds
    return F.interpolate(image, size=tuple(new_size), mode="nearest")[:, :, y1:y2, x1:x2].clone()


def earliest_pose_in_fov(expert, fov_mask, return_idx=False):
    """expert [B,T,2] grid poses, fov_mask [1,1,H,W] -> first pose inside the mask, else (H-1, W//2)."""
    B, T, _ = expert.shape
    H, W = fov_mask.shape[-2:]
    dev = expert.device
    r, c = expert[:, :, 0].long(), expert[:, :, 1].long()
    inside = fov_mask.to(dev)[0, 0, r, c] == 1
    t = torch.where(inside, torch.arange(T, device=dev).expand(B, -1), torch.full((B, T), T, device=dev))
    first = t.min(dim=1).values
    t = torch.where(t == T, torch.full_like(t, -1), t)
    last = t.max(dim=1).values
    none = first == T
    first = torch.where(none, torch.zeros_like(first), first)
    ar = torch.arange(B, device=dev)
    pose = torch.stack([r[ar, first], c[ar, first]], dim=1)
    pose = torch.where(none.unsqueeze(1), torch.tensor([H - 1, W // 2], dtype=pose.dtype, device=dev), pose)
    return (pose, first, last) if return_idx else pose
