"""candidate-trajectory sampler (host numpy, global RNG) and trajectory scoring on the costmap.
CPU: oracle and the product's host mirror against the reference's own outputs (tests/golden/planner.npz).
GPU: the HIP scoring kernel against the reference fixture (visitation maps bit-exact, scores to fp32 sum order)."""
import numpy as np
import pytest
import torch

CFG_KEYS = ("num_traj", "num_iter", "cmin", "cmax", "vmin", "vmax", "w", "dt")


def _cfg(arr):
    d = dict(zip(CFG_KEYS, arr.tolist()))
    d["num_traj"], d["num_iter"] = int(d["num_traj"]), int(d["num_iter"])
    return d


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_sampler_and_transforms_match_reference(golden, impl):
    g = golden("planner.npz")
    if impl == "oracle":
        from oracle import planner as P
        sample, to_bev, to_local, haus, ctrl = (P.sample_trajectory, P.transform_to_bev, P.transform_to_local,
                                                P.hausdorff_distance, P.get_controls)
    else:
        from creste_public_amd import planner as P
        sample, to_bev, to_local, haus, ctrl = (P.sampleTrajectory, P.transformToBEV, P.transformToLocal,
                                                P.hausdorffDistance, P.getControls)
    np.random.seed(1337)
    traj = sample(**_cfg(g["cfg"]))
    assert np.array_equal(traj, g["traj"])                       # same RNG consumption order, same arithmetic
    np.random.seed(7)
    traj2 = sample(**_cfg(g["cfg2"]))
    assert np.array_equal(traj2, g["traj2"])
    assert np.array_equal(to_bev(traj, res=0.1), g["bev"])
    assert np.array_equal(to_bev(traj2, center=(6.4, 12.8), res=0.05), g["bev2"])
    np.testing.assert_allclose(to_local(np.concatenate([g["bev"], np.zeros((20, 50, 1))], axis=2)), g["local"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(haus(traj, expert_idx=0), g["hausdorff"], rtol=1e-12)
    np.testing.assert_allclose(haus(traj2, expert_idx=3), g["hausdorff2"], rtol=1e-12)
    s = np.zeros((4, 3)); s[:, 2] = [0.0, 0.5, -1.0, 3.0]
    assert np.array_equal(ctrl(s, np.array([0.1, -2.0, 1.0, 0.0]), np.array([1.0, 0.5, 2.0, 1.5]), 1.0, dt=0.1), g["controls"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_scoring_matches_reference(golden, tag):
    from oracle import planner as P
    g = golden("planner.npz")
    scores, visit = P.score_trajectories(g.t(f"score_{tag}_map"), g.t(f"score_{tag}_xy"), int(g[f"score_{tag}_ds"][0]))
    assert torch.equal(visit.to(torch.uint8), g.t(f"score_{tag}_visit"))
    torch.testing.assert_close(scores, g.t(f"score_{tag}_scores"), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_scoring_matches_reference(golden, tag):
    from creste_public_amd import planner
    g = golden("planner.npz")
    cm, xy, ds = g.t(f"score_{tag}_map").cuda(), g.t(f"score_{tag}_xy").cuda(), int(g[f"score_{tag}_ds"][0])
    scores, visit, ncell = planner.score_trajectories(cm, xy, ds, return_visits=True)
    ref_visit = g.t(f"score_{tag}_visit")
    assert torch.equal(visit.cpu().to(torch.uint8), ref_visit)            # rasterisation bit-exact
    assert torch.equal(ncell.cpu().long(), ref_visit.long().sum(dim=(1, 2)))
    torch.testing.assert_close(scores.cpu(), g.t(f"score_{tag}_scores"), rtol=2e-6, atol=1e-6)
    # batched form: B frames with their own costmaps, K candidates each; the cheapest one is picked
    B = 3
    cms = torch.stack([cm, cm * 2.0, cm.flip(0)]).unsqueeze(1)
    xyb = xy.unsqueeze(0).repeat(B, 1, 1, 1)
    idx, s = planner.best_trajectory(cms, xyb, ds)
    torch.testing.assert_close(s[0], scores, rtol=0, atol=0)
    torch.testing.assert_close(s[1], planner.score_trajectories(cm * 2.0, xy, ds), rtol=0, atol=0)
    assert torch.equal(idx, s.argmin(dim=1)) and idx.shape == (B,)


@pytest.mark.gpu
def test_hip_scoring_edge_cases():
    """all poses identical (max_steps 0 -> only the last point), far outside the grid (clamped), grid diagonals."""
    from creste_public_amd import planner
    from oracle import planner as OP
    H, W = 64, 128
    cm = torch.rand(H, W, generator=torch.Generator().manual_seed(0))
    cases = [torch.tensor([[[30.5, 40.5]] * 5]),
             torch.tensor([[[-50.0, -9.0], [500.0, 900.0], [3.0, 3.0]]]),
             torch.tensor([[[0.0, 0.0], [126.0, 254.0]], [[126.0, 0.0], [0.0, 254.0]]])]
    for xy in cases:
        ref_s, ref_v = OP.score_trajectories(cm, xy, 2)
        s, v, n = planner.score_trajectories(cm.cuda(), xy.cuda(), 2, return_visits=True)
        assert torch.equal(v.cpu(), ref_v), xy.shape
        torch.testing.assert_close(s.cpu(), ref_s, rtol=2e-6, atol=1e-6)
