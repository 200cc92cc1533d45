"""Third-party sub-graphs (EfficientNet-B0 MBConv trunk, ResNet-18 BasicBlock trunk) pinned against `transformers`'
independent implementations -- see tests/hf_crosscheck.py.  CPU: oracle vs committed HF outputs (always) and vs a live
HF model incl. the reference resolution 512x612 (when transformers imports).  GPU: the HIP trunks vs the HF outputs."""
import numpy as np
import pytest
import torch

import hf_crosscheck as hc


@pytest.fixture(scope="module")
def fx():
    return hc.load_fixture()


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("case", list(hc.EFFNET_CASES))
def test_oracle_effnet_trunk_matches_hf_fixture(fx, case):
    hw = hc.EFFNET_CASES[case]
    trunk, x = hc.seeded_trunk(hw), hc.trunk_input(hw)
    assert abs(hc.checksum(trunk) - float(fx[f"effnet_{case}_wsum"])) < 1e-6 * float(fx[f"effnet_{case}_wsum"]), \
        "seeded weights differ from the fixture's (torch RNG drift): regenerate with tests/golden/make_trunk_hf.py"
    with torch.no_grad():
        eps = trunk.extract_endpoints(x)
    assert sorted(eps) == [f"reduction_{i}" for i in range(1, 6)]
    for k, v in eps.items():
        ref = torch.from_numpy(fx[f"effnet_{case}_{k}"])
        assert v.shape == ref.shape, (k, v.shape, ref.shape)          # static-padding sides decide the odd sizes
        assert _rel(v, ref) < 2e-5, (k, _rel(v, ref))


def test_oracle_resnet_trunk_matches_hf_fixture(fx):
    bev, x = hc.seeded_bev(), hc.bev_input()
    assert abs(hc.checksum(bev) - float(fx["resnet_wsum"])) < 1e-6 * float(fx["resnet_wsum"])
    x1, x3 = hc.oracle_bev_stages(bev, x)
    assert _rel(x1, torch.from_numpy(fx["resnet_x1"])) < 2e-5
    assert _rel(x3, torch.from_numpy(fx["resnet_x3"])) < 2e-5
    assert float(x3.abs().max()) > 1e-3      # BN gammas randomised: the zero-init residual branches are live


def test_oracle_effnet_trunk_matches_live_hf_at_reference_resolution():
    """512x612 (the reference config's frame): 612 -> 306 -> 153 -> 76 -> 38 -> 19 exercises the static pads on odd
    maps; oracle and HF must agree in SHAPE and value at every endpoint."""
    pytest.importorskip("transformers")
    hw = (512, 612)
    trunk, x = hc.seeded_trunk(hw, seed=51), hc.trunk_input(hw, seed=52)
    hf = hc.hf_effnet_from(trunk)
    ref = hc.hf_effnet_endpoints(hf, x)
    with torch.no_grad():
        eps = trunk.extract_endpoints(x)
    assert [tuple(ref[f"reduction_{i}"].shape[2:]) for i in range(1, 6)] == \
        [(256, 306), (128, 153), (64, 76), (32, 38), (16, 19)]
    for k in ref:
        assert eps[k].shape == ref[k].shape
        assert _rel(eps[k], ref[k]) < 2e-5, k


# --------------------------------------------------------------------------------------------- HIP vs HF (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "bf16x6", "f16x3"])
@pytest.mark.parametrize("case", list(hc.EFFNET_CASES))
def test_hip_effnet_trunk_matches_hf_fixture(fx, case, prec):
    import creste_public_amd
    from creste_public_amd import ops
    from creste_public_amd.creste.models.blocks.effnet import EfficientNetB0Trunk
    hw = hc.EFFNET_CASES[case]
    oracle = hc.seeded_trunk(hw)
    assert abs(hc.checksum(oracle) - float(fx[f"effnet_{case}_wsum"])) < 1e-6 * float(fx[f"effnet_{case}_wsum"])
    creste_public_amd.set_precision(prec)
    try:
        m = EfficientNetB0Trunk(4, hw)
        m.load_state_dict(oracle.state_dict(), strict=True)
        m = m.cuda().eval()
        x = hc.trunk_input(hw).cuda()
        with torch.no_grad():
            eps = m.extract_endpoints_act(ops.nchw_to_nhwc(x.contiguous()))
        torch.cuda.synchronize()
        for k in sorted(eps):
            ref = torch.from_numpy(fx[f"effnet_{case}_{k}"])
            got = eps[k].nchw().cpu()
            assert got.shape == ref.shape, (k, got.shape, ref.shape)
            assert _rel(got, ref) < 5e-5, (k, _rel(got, ref))     # 16 MBConv blocks deep, fp32 re-association
    finally:
        creste_public_amd.set_precision("f32")


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "bf16x6", "f16x3"])
def test_hip_resnet_trunk_matches_hf_fixture(fx, prec):
    import creste_public_amd
    from creste_public_amd import ops
    from creste_public_amd.creste.models.blocks.inpainting import InpaintingResNet18MultiHead
    from creste_public_amd.hipnn import ACT_RELU, ConvUnit
    oracle = hc.seeded_bev()
    creste_public_amd.set_precision(prec)
    try:
        m = InpaintingResNet18MultiHead(96, [32, 6, 2], input_key="bev_features",
                                        output_prefix=["inpainting_sam", "inpainting_sam_dynamic", "elevation"])
        m.load_state_dict(oracle.state_dict(), strict=True)
        m = m.cuda().eval()
        x = ops.nchw_to_nhwc(hc.bev_input().cuda().contiguous())
        with torch.no_grad():
            h = ConvUnit(m.conv1, m.bn1, ACT_RELU)(x)
            for blk in m.layer1:
                h = blk.forward_act(h)
            x1 = h
            for blk in list(m.layer2) + list(m.layer3):
                h = blk.forward_act(h)
        torch.cuda.synchronize()
        assert _rel(x1.nchw().cpu(), torch.from_numpy(fx["resnet_x1"])) < 2e-5
        assert _rel(h.nchw().cpu(), torch.from_numpy(fx["resnet_x3"])) < 2e-5
    finally:
        creste_public_amd.set_precision("f32")


# ------------------------------------------------------ torch_scatter-dependent pieces vs torch.scatter_reduce (CPU)
@pytest.mark.parametrize("reduce", ["max", "min"])
def test_oracle_lidar_depth_image_matches_torch_scatter_reduce(reduce):
    """`pixels_to_depth` reduces with torch_scatter.scatter(reduce=...) (reference projection.py:121-128; package
    absent here).  torch's own scatter_reduce (amax/amin, empty slots left at 0) is an independently written
    implementation of the same documented semantics: the numpy restatement must agree with it bit for bit."""
    from creste_public_amd import synth
    from oracle import lidar as olidar
    H, W = 96, 160
    g = torch.Generator().manual_seed(7)
    pts = synth.lidar_scan(1, g)[0, ::16].numpy()
    l2c = synth.lidar2camrect(1, H, W)[0].numpy()
    img = olidar.depth_image(pts, l2c, H, W, reduce=reduce)
    pc = torch.from_numpy(pts[:, :3]).double()
    cam = (torch.from_numpy(l2c) @ torch.cat([pc, torch.ones(len(pc), 1, dtype=torch.float64)], 1).t()).t()[:, :3]
    uv = (cam[:, :2] / cam[:, 2:3]).to(torch.int32)                  # truncation toward zero, as astype(int)
    ok = (cam[:, 2] > 0) & (uv[:, 0] >= 0) & (uv[:, 0] < W) & (uv[:, 1] >= 0) & (uv[:, 1] < H)
    loc = uv[ok, 1].long() * W + uv[ok, 0].long()
    ref = torch.zeros(H * W, dtype=torch.float64).scatter_reduce(0, loc, cam[ok, 2], reduce="a" + reduce,
                                                                 include_self=False)
    assert int(ok.sum()) > 500 and int((ref > 0).sum()) < int(ok.sum())       # collisions exist
    assert np.array_equal(img.reshape(-1), ref.numpy())
