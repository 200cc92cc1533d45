"""GPU: size-independent properties at BASELINE.json's full sizes (batch 16 of 1216x608 -> 256x256 BEV; 8 x 256 x 256
and 4 x 512 x 512 MDP grids), where the CPU oracle is too slow to be the checker."""
import pytest
import torch

import creste_public_amd
from creste_public_amd import MaxEntIRL, maxent_irl_cfg, ops, synth

pytestmark = pytest.mark.gpu

H, W, B = 608, 1216, 16


@pytest.fixture(scope="module")
def full():
    creste_public_amd.set_precision("f16x3")
    torch.manual_seed(0)
    model = MaxEntIRL(maxent_irl_cfg((H, W), solve_mdp=False))
    synth.randomize_bn(model, seed=1)
    model = model.cuda().eval()
    rgbd, p2p = synth.make_frames(B, H, W, seed=1337)
    rgbd, p2p = rgbd.cuda(), p2p.cuda()
    synth.calibrate_bn_hip(model, rgbd[:2], p2p[:2])
    with torch.no_grad():
        out = {k: v.clone() for k, v in model((rgbd, p2p)).items()}
    yield model, rgbd, p2p, out
    creste_public_amd.set_precision("f32")


def test_frames_are_independent_and_the_step_is_deterministic(full):
    """eval-mode inference shards by frame: a repeated step reproduces itself and a permuted batch gives the permuted
    outputs, bit for bit (every kernel reduces within a frame in a fixed order)."""
    model, rgbd, p2p, out = full
    with torch.no_grad():
        again = model((rgbd, p2p))
        perm = torch.arange(B - 1, -1, -1, device="cuda")
        rev = model((rgbd[perm].contiguous(), p2p[perm].contiguous()))
        sub = model((rgbd[5:7].contiguous(), p2p[5:7].contiguous()))
    for k, v in out.items():
        assert torch.equal(again[k], v), k
        assert torch.equal(rev[k][perm], v), k
    # a different batch COMPOSITION changes the per-tensor |max| the f16x3 engine scales by (a power of two), hence
    # which bits the fp16 hi/lo split keeps: frames then agree to fp32 round-off, not bit for bit
    for k in ("depth_preds_logits", "depth_preds_feats", "dino_pe_feats", "depth_preds_metric"):
        a_, b_ = sub[k].double(), out[k][5:7].double()
        assert float((a_ - b_).pow(2).mean().sqrt() / b_.pow(2).mean().sqrt()) < 1e-5, k


def test_splat_conserves_mass_and_is_linear_in_the_features(full):
    model, rgbd, p2p, out = full
    dens, coords = out["bev_densities"], out["bev_coords"]
    X, Y = coords[..., 0].double(), coords[..., 1].double()
    x0, y0 = torch.floor(X), torch.floor(Y)
    rx, ry = X - x0, Y - y0
    mass = torch.zeros(B, dtype=torch.float64, device="cuda")
    for xd in (0, 1):
        for yd in (0, 1):
            w = (rx if xd else 1 - rx) * (ry if yd else 1 - ry)
            ok = (x0 + xd >= 0) & (x0 + xd < 256) & (y0 + yd >= 0) & (y0 + yd < 256)
            mass += (w * ok).sum(1)
    got = dens.double().sum(dim=(1, 2, 3))
    assert torch.allclose(got, mass, rtol=1e-5), (got, mass)             # every valid tap weight lands in exactly one cell
    assert float(got.min()) > 0.2 * coords.shape[1], got                 # a populated map (calibrated network)
    assert 0.1 < float((dens > 0).float().mean()) < 0.5
    # linearity: splat(a*f + b*g) == a*splat(f) + b*splat(g) on the same geometry (weights and densities fixed)
    P = coords.shape[1]
    g = torch.Generator(device="cuda").manual_seed(3)
    f1 = torch.randn(B, 1, P, 96, device="cuda", generator=g)
    f2 = torch.randn(B, 1, P, 96, device="cuda", generator=g)
    xyz = torch.zeros(B, P, 3, device="cuda")
    xyz[..., 0] = 12.8 - coords[..., 1] * 0.1                            # invert the map transform (x <- Y, y <- X)
    xyz[..., 1] = 12.8 - coords[..., 0] * 0.1
    sp = lambda f: ops.bev_splat(xyz, ops.Act(f.contiguous(), 96), (12.8, 12.8), (0.1, 0.1), 256, 256)[1].buf   # noqa: E731
    lhs = sp(2.0 * f1 - 0.5 * f2)
    rhs = 2.0 * sp(f1) - 0.5 * sp(f2)
    assert float((lhs - rhs).abs().max()) < 1e-4 * float(rhs.abs().max())


def test_conv_engine_linearity_at_full_size():
    """the dominant layer (496 -> 496 3x3 at 152x304, batch 16) in the default operand mode: conv(a x + b y) =
    a conv(x) + b conv(y) to fp32 round-off -- operand splitting and power-of-two rescaling do not leak into the result."""
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(16, 152, 304, 496, device="cuda", generator=g)
    y = torch.randn(16, 152, 304, 496, device="cuda", generator=g) * 37.0
    w = torch.randn(496, 496, 3, 3, device="cuda", generator=g) / (496 * 9) ** 0.5
    pc = ops.pack_conv(w, None, None, 1, 1, ops.ACT_NONE, ops.PREC_F16X3)
    conv = lambda t: ops.conv2d(ops.Act(t, 496), pc).buf        # noqa: E731
    lhs = conv(3.0 * x - 0.25 * y)
    rhs = 3.0 * conv(x) - 0.25 * conv(y)
    rel = float((lhs - rhs).pow(2).mean().sqrt() / rhs.pow(2).mean().sqrt())
    assert rel < 2e-6, rel


@pytest.mark.parametrize("shape", [(8, 256, 256), (4, 512, 512)])
def test_value_iteration_fixed_point_and_policy_simplex(shape):
    """BASELINE configs[2] / [4] grids: the returned v is a fixed point of the Bellman backup to the convergence
    threshold, q is the backup of v, the policy is a softmax of q, and a second solve reproduces the first."""
    g = torch.Generator(device="cuda").manual_seed(shape[1])
    r = torch.rand(shape, device="cuda", generator=g)
    v, q, pi, sweeps = ops.value_iteration(r, 0.99, 1e-3)
    v2, q2, pi2, sweeps2 = ops.value_iteration(r, 0.99, 1e-3)
    assert torch.equal(v, v2) and torch.equal(pi, pi2) and int(sweeps) == int(sweeps2)
    assert 600 < int(sweeps) < 760
    assert float((q.max(dim=1).values - v).abs().max()) <= 1.0e-3 + 1e-4          # one more backup moves v by < threshold
    assert torch.allclose(pi.sum(1), torch.ones_like(v), atol=1e-5) and float(pi.min()) >= 0.0
    ref_pi = torch.softmax(q - q.max(dim=1, keepdim=True).values, dim=1)
    assert float((pi - ref_pi).abs().max()) < 1e-5
    assert float(v.min()) >= 0.0 and float(v.max()) <= 1.0 / (1 - 0.99) + 1e-3     # 0 <= v <= r_max / (1 - gamma)


def test_expected_svf_mass_bounds_full_grid():
    Bn, Hh, Ww, T = 8, 256, 256, 50
    g = torch.Generator(device="cuda").manual_seed(2)
    pol = torch.softmax(torch.randn(Bn, 8, Hh, Ww, device="cuda", generator=g) * 2, dim=1)
    t = torch.linspace(0, 1, T, device="cuda").view(1, T, 1)
    xy = torch.tensor([[250.0, 256.0]], device="cuda").repeat(Bn, 1).unsqueeze(1) + t * torch.tensor([[[-200.0, 60.0]]], device="cuda")
    fov = torch.ones(Hh, Ww, dtype=torch.uint8, device="cuda")
    svf, states, grid = ops.expected_svf(pol, xy.contiguous(), fov, T, 2.0, 0.005, True, False)
    tot = svf.sum(dim=(1, 2))
    assert float(svf.min()) >= 0.0 and float(tot.max()) <= T + 1e-3 and float(tot.min()) > 1.0   # mass only leaves the grid
    assert states.shape == (Bn, T, 2) and int(states.min()) >= 0 and int(states[..., 0].max()) < Hh
    assert float(grid.sum(dim=(1, 2)).max()) <= T + 1e-3


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_fused_and_unfused_trunk_paths_agree_at_full_size(prec):
    """the fused kernels of the encoder trunk (stem + block 0, MBConv front halves, LDS-tile depthwise convs, flat 1x1
    re-tiling) against the kernel-per-op path on the same weights at 608x1216: every endpoint within fp32 round-off of
    the other (the fused paths use exact fp32 FMAs and 1-ulp hardware exp / rcp in their activations)"""
    import creste_public_amd
    from creste_public_amd import ops, synth
    from creste_public_amd.creste.models.blocks import effnet as E
    torch.manual_seed(3)
    creste_public_amd.set_precision(prec)
    try:
        trunk = E.EfficientNetB0Trunk(4, (608, 1216))
        synth.randomize_bn(trunk, seed=5)
        trunk = trunk.cuda().eval()
        x = ops.nchw_to_nhwc(torch.rand(2, 4, 608, 1216, device="cuda"))
        x.amax = x.buf.abs().max().reshape(1)
        outs = {}
        with torch.no_grad():
            for fused in (True, False):
                E.FUSE_MBCONV, ops.DW_TILE = fused, fused
                outs[fused] = {k: v.buf.clone() for k, v in trunk.extract_endpoints_act(x).items()}
    finally:
        E.FUSE_MBCONV, ops.DW_TILE = True, True
        creste_public_amd.set_precision("f32")
    assert set(outs[True]) == set(outs[False]) and len(outs[True]) == 5
    for k, a in outs[True].items():
        b = outs[False][k]
        rel = float((a - b).double().pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt())
        assert rel < (3e-6 if prec == "f32" else 2e-5), (k, rel)
