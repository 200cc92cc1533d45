"""Synthetic inputs of the benchmark / parity workloads: seeded RGB frames, a
128x1024 LiDAR range image projected into a sparse millimetre depth channel, the pixel->LiDAR `p2p`
matrix, expert trajectories and a helper that gives a randomly initialised model realistic BatchNorm
statistics.  Input preparation is not part of the timed path (plain torch, any device)."""
from __future__ import annotations

import math

import torch


def camera_matrices(H: int, W: int, ds: int = 4):
    """Pinhole K for an HxW image (fx=fy=730 px at 1216 px width, principal point at the centre) and
    the fixed camera->LiDAR extrinsic (camera z forward / x right / y down; LiDAR x forward / y left
    / z up; camera 0.1 m ahead, 0.4 m above the LiDAR origin... expressed as T_lidar<-cam)."""
    f = 730.0 * W / 1216.0
    K = torch.tensor([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]], dtype=torch.float64)
    T = torch.tensor([[0, 0, 1, 0.1], [-1, 0, 0, 0.0], [0, -1, 0, -0.4], [0, 0, 0, 1]], dtype=torch.float64)
    return K, T


def make_p2p(B: int, H: int, W: int, ds: int = 4) -> torch.Tensor:
    """p2p [B,1,4,4] = T_lidar<-cam @ inv(K/ds padded to 4x4): maps (u*d, v*d, d, 1) at feature
    resolution to LiDAR xyz (reference projection.py:11-34 / codapefree_dataloader.py:803-841)."""
    K, T = camera_matrices(H, W, ds)
    Kd = K.clone()
    Kd[:2] /= ds
    K4 = torch.eye(4, dtype=torch.float64)
    K4[:3, :3] = Kd
    return (T @ torch.linalg.inv(K4)).float().view(1, 1, 4, 4).repeat(B, 1, 1, 1).contiguous()


def lidar_depth_channel(B: int, H: int, W: int, gen: torch.Generator) -> torch.Tensor:
    """128 beams x 1024 azimuth steps, range ~ U[1,40] m, projected through K into a sparse depth
    image in millimetres (nearest return wins, zeros elsewhere) -> [B,H,W]."""
    K, T = camera_matrices(H, W)
    Tinv = torch.linalg.inv(T)
    az = torch.linspace(-math.pi, math.pi, 1025, dtype=torch.float64)[:-1]
    el = torch.linspace(-22.5, 22.5, 128, dtype=torch.float64) * math.pi / 180
    el, az = torch.meshgrid(el, az, indexing="ij")
    out = torch.zeros(B, H, W)
    for b in range(B):
        rng = torch.rand(128, 1024, generator=gen, dtype=torch.float64) * 39.0 + 1.0
        pts = torch.stack([rng * torch.cos(el) * torch.cos(az), rng * torch.cos(el) * torch.sin(az),
                           rng * torch.sin(el), torch.ones_like(rng)], dim=-1).view(-1, 4)
        cam = (Tinv @ pts.t()).t()[:, :3]
        z = cam[:, 2]
        uv = (K @ cam.t()).t()
        u, v = uv[:, 0] / z, uv[:, 1] / z
        ok = (z > 0.3) & (u >= 0) & (u < W) & (v >= 0) & (v < H)
        ui, vi, zi = u[ok].long(), v[ok].long(), (z[ok] * 1000.0).float()
        flat = torch.full((H * W,), float("inf"))
        flat.scatter_reduce_(0, vi * W + ui, zi, reduce="amin")
        flat[torch.isinf(flat)] = 0.0
        out[b] = flat.view(H, W)
    return out


def lidar_scan(B: int, gen: torch.Generator) -> torch.Tensor:
    """[B, 128*1024, 3] fp32 LiDAR points: 128 beams in +-22.5 deg x 1024 azimuth steps, range ~ U[1,40] m."""
    az = torch.linspace(-math.pi, math.pi, 1025, dtype=torch.float64)[:-1]
    el = torch.linspace(-22.5, 22.5, 128, dtype=torch.float64) * math.pi / 180
    el, az = torch.meshgrid(el, az, indexing="ij")
    rng = torch.rand(B, 128, 1024, generator=gen, dtype=torch.float64) * 39.0 + 1.0
    pts = torch.stack([rng * torch.cos(el) * torch.cos(az), rng * torch.cos(el) * torch.sin(az),
                       rng * torch.sin(el)], dim=-1)
    return pts.view(B, -1, 3).float().contiguous()


def lidar2camrect(B: int, H: int, W: int) -> torch.Tensor:
    """[B,4,4] float64: pixel-homogeneous projection K @ T_cam<-lidar (rows 0..2), last row (0,0,0,1)."""
    K, T = camera_matrices(H, W)
    K4 = torch.eye(4, dtype=torch.float64)
    K4[:3, :3] = K
    return (K4 @ torch.linalg.inv(T)).unsqueeze(0).repeat(B, 1, 1).contiguous()


def make_frames(B: int, H: int, W: int, seed: int = 1337):
    """-> rgbd [B,1,4,H,W] (RGB in [0,1), channel 3 = sparse LiDAR depth in mm), p2p [B,1,4,4]."""
    g = torch.Generator().manual_seed(seed)
    rgb = torch.rand(B, 3, H, W, generator=g)
    depth = lidar_depth_channel(B, H, W, g)
    rgbd = torch.cat([rgb, depth.unsqueeze(1)], dim=1).unsqueeze(1).contiguous()
    return rgbd, make_p2p(B, H, W)


def make_experts(B: int, T: int = 50, grid=256, seed: int = 7) -> torch.Tensor:
    """Straight-ish polylines from the bottom centre of the front half of the BEV map -> [B,T,3,3]
    SE(2) poses in full-resolution BEV cells (row, col in [:, :, :2, 2]).  grid = G or (rows, cols)."""
    g = torch.Generator().manual_seed(seed)
    gh, gw = (grid, grid) if isinstance(grid, int) else grid
    t = torch.linspace(0, 1, T).view(1, T, 1)
    start = torch.tensor([[gh / 2 - 6.0, gw / 2.0]]).repeat(B, 1)
    delta = torch.stack([-(torch.rand(B, generator=g) * 0.3 + 0.5) * gh / 2,
                         (torch.rand(B, generator=g) - 0.5) * gw / 3], dim=1)
    xy = start.unsqueeze(1) + t * delta.unsqueeze(1)
    P = torch.eye(3).repeat(B, T, 1, 1)
    P[:, :, :2, 2] = xy
    return P


@torch.no_grad()
def randomize_bn(model: torch.nn.Module, seed: int = 3):
    """Random affine + running statistics for every BatchNorm (exercises the BN folding; also
    un-zeroes the zero-initialised residual BN of the ResNet blocks)."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


@torch.no_grad()
def calibrate_bn_hip(model: torch.nn.Module, rgbd: torch.Tensor, p2p: torch.Tensor, depth_gain: float = 4.0):
    """Give a randomly initialised MaxEntIRL / TerrainNet the statistics of a trained one, on the GPU: one
    training-mode pass (HIP training path) with BatchNorm momentum 1 stores the batch statistics as running statistics,
    so eval-mode activations are O(1) with spatial variation at every layer; the depth head's last BatchNorm gain is
    then raised so that the per-pixel bin distribution is peaked.  Without this a random-init network predicts ONE
    depth (the mean of the bin values, 12.95 m) for every pixel: all points fall outside the +-12.8 m grid, the BEV map
    is empty and everything downstream runs on zeros (measured: 0 % occupied cells; with it 21 %, depths 0.3-25.4 m)."""
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    saved = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    backbone = getattr(model, "backbone", model)
    was_training = model.training
    from . import hipnn
    prec = hipnn.get_precision()
    hipnn.set_precision("f32")        # exact fp32 engine for the one-off pass (also keeps its launches out of the
    try:                              # timed kernels' rocprof statistics: different kernel symbols)
        backbone.train()
        backbone((rgbd, p2p))
        if backbone is not model:
            model.train()                       # the reward head's BatchNorms (the backbone itself stays in eval here)
            solve = getattr(model, "solve_mdp", False)
            model.solve_mdp = False             # costmap only: no expert needed for the statistics
            try:
                model((rgbd, p2p))
            finally:
                model.solve_mdp = solve
    finally:
        hipnn.set_precision(prec)
        for m, mo in zip(bns, saved):
            m.momentum = mo
        model.train(was_training)
    peak_depth_head(backbone, depth_gain)
    return model


@torch.no_grad()
def peak_depth_head(terrainnet: torch.nn.Module, gain: float = 4.0):
    """raise the depth head's last BatchNorm gain: peaked per-pixel bin distributions -> depths spread over the bin range
    (training-mode BatchNorm normalises by batch statistics, so this alone populates the BEV map there)."""
    dc = terrainnet.depthcomp.depthcomp
    last = [m for m in dc.depth_head.model if isinstance(m, torch.nn.BatchNorm2d)][-1]
    last.weight.mul_(gain)
