#!/usr/bin/env python3
"""Generate tests/golden/planner.npz by running the REFERENCE's own trajectory sampler
(/root/reference/scripts/traversability/planner_utils/control.py) and its polyline rasteriser
(/root/reference/creste/utils/loss_utils.py:1054-1116, through the import shims of make_golden.py).
Build container only; the fixture is data (seeds, inputs, the reference's outputs).

    python tests/golden/make_planner_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs nothing until asked)


def main():
    spec = importlib.util.spec_from_file_location(
        "ref_control", "/root/reference/scripts/traversability/planner_utils/control.py")
    ctl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ctl)
    out = {}
    cfg = dict(num_traj=20, num_iter=50, cmin=-2, cmax=2, vmin=1, vmax=1, w=1.0, dt=0.1)      # control.py main()
    np.random.seed(1337)
    traj = ctl.sampleTrajectory(**cfg)
    out["cfg"] = np.array([cfg[k] for k in ("num_traj", "num_iter", "cmin", "cmax", "vmin", "vmax", "w", "dt")], dtype=np.float64)
    out["traj"] = traj
    cfg2 = dict(num_traj=7, num_iter=30, cmin=-0.5, cmax=1.5, vmin=0.5, vmax=3.0, w=2.0, dt=0.2)
    np.random.seed(7)
    traj2 = ctl.sampleTrajectory(**cfg2)
    out["cfg2"] = np.array([cfg2[k] for k in ("num_traj", "num_iter", "cmin", "cmax", "vmin", "vmax", "w", "dt")], dtype=np.float64)
    out["traj2"] = traj2
    out["bev"] = ctl.transformToBEV(traj, res=0.1)
    out["bev2"] = ctl.transformToBEV(traj2, center=(6.4, 12.8), res=0.05)
    out["local"] = ctl.transformToLocal(np.concatenate([out["bev"], np.zeros((20, 50, 1))], axis=2))
    out["hausdorff"] = ctl.hausdorffDistance(traj, expert_idx=0)
    out["hausdorff2"] = ctl.hausdorffDistance(traj2, expert_idx=3)
    s = np.zeros((4, 3)); s[:, 2] = [0.0, 0.5, -1.0, 3.0]
    out["controls"] = ctl.getControls(s, np.array([0.1, -2.0, 1.0, 0.0]), np.array([1.0, 0.5, 2.0, 1.5]), 1.0, dt=0.1)
    # rasterisation + trajectory reward with the reference's own compute_expert_visitation
    mg.install_shims()
    sys.path.insert(0, mg.REF)
    from creste.utils.loss_utils import MaxEntIRLLoss
    g = torch.Generator().manual_seed(5)
    for tag, (H, W, ds, scale) in {"a": (64, 128, 2, 1.0), "b": (256, 256, 1, 2.5)}.items():
        # candidates in full-resolution BEV cells: the sampled metric trajectories mapped by transformToBEV, plus
        # out-of-grid and degenerate (repeated pose) ones
        xy = torch.from_numpy(ctl.transformToBEV(traj * scale, center=(12.8 * ds * H / 128, 12.8 * ds * W / 256), res=0.1)).float()
        xy[3, 10:] = xy[3, 10]                                    # stops: zero-length segments
        xy[5] = xy[5] * 3.0 - 40.0                                # leaves the grid: clamped
        cm = torch.rand(H, W, generator=g)
        pts, visit = MaxEntIRLLoss.compute_expert_visitation(xy, ds, [H, W])
        out[f"score_{tag}_xy"], out[f"score_{tag}_map"] = xy.numpy(), cm.numpy()
        out[f"score_{tag}_ds"] = np.array([ds])
        out[f"score_{tag}_visit"] = visit.numpy().astype(np.uint8)
        out[f"score_{tag}_scores"] = (visit * cm.unsqueeze(0)).sum(dim=(1, 2)).numpy()
    np.savez_compressed(os.path.join(HERE, "planner.npz"), **out)
    print("wrote planner.npz", os.path.getsize(os.path.join(HERE, "planner.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
