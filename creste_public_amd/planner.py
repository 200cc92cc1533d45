"""Candidate trajectories and their cost on the predicted costmap -- the consumer of the hot path's output
(SURVEY.md 8f-4).

Host side mirrors /root/reference/scripts/traversability/planner_utils/control.py (the Ackermann sampler the
reference's labelling tool draws candidates with: `sampleTrajectory` :104-118, `getControls` :12-29,
`transformToBEV` :136-158, `transformToLocal` :120-134, `hausdorffDistance` :36-75 -- numpy + the global numpy RNG, as
there); the SCORING runs on the GPU: `score_trajectories` rasterises K candidate polylines per frame exactly like
`compute_expert_visitation` (creste/utils/loss_utils.py:1054-1116) and sums the costmap over each one's visited cells
(:1197-1258), one workgroup per candidate (`csrc/planner.hip`, creste_trajectory_scores_f32).  `best_trajectory` picks
the cheapest.  There is no CPU fallback for the scoring.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .ops import HipLibraryError, _chk, _stream


def getControls(s, c, v, w, dt=0.1):
    xdot = v * np.cos(s[:, 2])
    ydot = v * np.sin(s[:, 2])
    thetadot = v * c
    return np.stack([xdot * dt, ydot * dt, thetadot * dt], axis=-1)


def sampleRange(size, min=-1, max=1):
    return (np.random.rand(size) * (max - min)) + min


def sampleTrajectory(num_traj, num_iter, cmin, cmax, vmin, vmax, w, dt, epsilon=10):
    """[num_traj, num_iter, 3] (x, y, theta): per step curvatures, then speeds, from the global numpy RNG."""
    trajectories = np.tile(np.array([[0, 0, 0]]), (num_traj, 1, 1))
    for it in range(0, num_iter - 1):
        c = sampleRange(num_traj, cmin, cmax)
        v = sampleRange(num_traj, vmin, vmax)
        nxt = trajectories[:, it] + getControls(trajectories[:, it], c, v, w, dt=dt)
        trajectories = np.concatenate((trajectories, nxt[:, None]), axis=1)
    return trajectories


def _ego_matrix(center):
    M = np.eye(3)
    M[:2, 2] = center
    M[:2, :2] = np.array([[-1, 0], [0, -1]])
    return M


def transformToLocal(trajectories, center=(12.8, 12.8), res=0.1):
    B, T, _ = trajectories.shape
    h = np.ones((B, T, 3))
    h[:, :, :2] = trajectories[:, :, :2]
    out = np.matmul(h, _ego_matrix([c / res for c in center]).T)
    out[:, :, :2] = out[:, :, :2] * res
    return out


def transformToBEV(trajectories, center=(12.8, 12.8), res=0.1):
    B, T, _ = trajectories.shape
    h = np.ones((B, T, 3))
    h[:, :, :2] = trajectories[:, :, :2]
    return (np.matmul(h, _ego_matrix(center).T) / res)[:, :, :2]


def hausdorffDistance(trajectories, expert_idx=0):
    ref = trajectories[expert_idx]
    d = np.linalg.norm(ref[None, :, None, :] - trajectories[:, None, :, :], axis=-1)      # [N, Tref, Tother]
    return np.maximum(d.min(axis=2).max(axis=1), d.min(axis=1).max(axis=1))


def score_trajectories(costmap: torch.Tensor, xy: torch.Tensor, map_ds: float = 2.0, return_visits: bool = False):
    """costmap [B,1,H,W] | [B,H,W] | [H,W] (CUDA fp32: `traversability_preds`), xy [B,K,T,2] | [K,T,2] (row, col) in
    full-resolution BEV cells -> scores [B,K] | [K] (sum of the costmap over each candidate's visited cells); with
    return_visits also the 0/1 visitation maps [..,K,H,W] and the visited-cell counts."""
    lib = _lib.load()
    cm = costmap
    if cm.dim() == 4:
        cm = cm[:, 0]
    batched = xy.dim() == 4
    if cm.dim() == 2:
        cm = cm.unsqueeze(0)
    if not batched:
        xy = xy.unsqueeze(0)
    B, K, T, two = xy.shape
    if two != 2 or cm.shape[0] not in (1, B):
        raise HipLibraryError(f"score_trajectories: costmap {tuple(costmap.shape)} / trajectories {tuple(xy.shape)} disagree")
    cm = _chk(cm.contiguous().float(), name="costmap")
    xy = _chk(xy.contiguous().float(), name="trajectories")
    H, W = cm.shape[-2:]
    dev = cm.device
    scores = torch.empty((B, K), dtype=torch.float32, device=dev)
    visit = torch.empty((B, K, H, W), dtype=torch.float32, device=dev) if return_visits else None
    ncell = torch.empty((B, K), dtype=torch.int32, device=dev) if return_visits else None
    index = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(K) if cm.shape[0] == B and B > 1 else None
    if index is None and cm.shape[0] == B and B == 1:
        stride = 0
    else:
        stride = H * W if cm.shape[0] == B else 0
    work = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(lib.creste_trajectory_scores_f32(
        xy.data_ptr(), B * K, T, float(map_ds), H, W, cm.data_ptr(), index.data_ptr() if index is not None else None,
        stride if index is not None else 0, scores.data_ptr(), visit.data_ptr() if visit is not None else None,
        ncell.data_ptr() if ncell is not None else None, work.data_ptr(), _stream()), "trajectory_scores")
    if not batched:
        scores = scores[0]
        visit = visit[0] if visit is not None else None
        ncell = ncell[0] if ncell is not None else None
    return (scores, visit, ncell) if return_visits else scores


def best_trajectory(costmap: torch.Tensor, xy: torch.Tensor, map_ds: float = 2.0):
    """-> (index [B] of the cheapest candidate per frame, scores [B,K])."""
    s = score_trajectories(costmap, xy, map_ds)
    return torch.argmin(s, dim=-1), s
