"""GPU: parity at BASELINE.json configs[1] SIZE (1216x608 frames -> 256x256 BEV costmap) against the CPU oracle.

(a) B=2 frames of 608x1216: the HIP path in every parity-grade operand mode (`f32`, `bf16x6`, `f16x3` -- the mode
    bench.py's headline is measured in) against the fp32 CPU oracle AND its float64 evaluation:
      * stage-wise on identical inputs: costmap within 1e-4 (the north-star tolerance), BEV heads 2e-5, splat exact;
      * end to end: |HIP - float64| within a small factor of the fp32 CPU reference's own |fp32 - float64|
        (same criterion as tests/test_model_gpu.py at 128x192).
(b) B=16 (the bench batch): `f16x3` and `bf16x6` against the HIP `f32` mode (exact fp32 products) on every output
    key: both sit within the same distance of it.
The oracle costs ~4 s per frame on the GPU box's host; the float64 pass about twice that.
"""
import copy

import pytest
import torch

import creste_public_amd
from creste_public_amd import MaxEntIRL, synth
from creste_public_amd.config import maxent_irl_cfg
from test_model_gpu import _rms, _stage, assert_within_fp32_noise_floor, calibrate_bn

pytestmark = pytest.mark.gpu

H, W = 608, 1216
MODES = ["f32", "bf16x6", "f16x3"]


@pytest.fixture(scope="module")
def oracle_full():
    from oracle.irl import MaxEntIRL as OracleIRL
    B = 2
    torch.manual_seed(4321)
    cfg = maxent_irl_cfg((H, W), solve_mdp=False)
    oracle = OracleIRL(cfg)
    rgbd, p2p = synth.make_frames(B, H, W, seed=77)
    calibrate_bn(oracle, lambda: oracle((rgbd, p2p)))
    with torch.no_grad():
        oracle.traversability_head.r.postpool[0].norm.weight.mul_(0.01)      # costmaps of O(1), as after training
        oracle.traversability_head.r.postpool[0].norm.bias.mul_(0.01)
        ref = oracle((rgbd, p2p))
        o64 = copy.deepcopy(oracle).double()
        o64.fov_mask = oracle.fov_mask
        ref64 = o64((rgbd.double(), p2p.double()))
    del o64
    # a populated BEV map (a random-init depth head would put every pixel at the mean bin value, outside the grid)
    occ = float((ref["bev_densities"] > 0).float().mean())
    return oracle, ref, ref64, (rgbd, p2p), occ


@pytest.fixture(scope="module", params=MODES)
def hip_full(request, oracle_full):
    oracle, ref, ref64, (rgbd, p2p), _ = oracle_full
    creste_public_amd.set_precision(request.param)
    model = MaxEntIRL(maxent_irl_cfg((H, W), solve_mdp=False))
    model.load_state_dict(oracle.state_dict(), strict=True)
    model = model.cuda().eval()
    with torch.no_grad():
        got = {k: v.clone() for k, v in model((rgbd.cuda(), p2p.cuda())).items()}
    torch.cuda.synchronize()
    yield request.param, model, got
    creste_public_amd.set_precision("f32")


def test_fullsize_outputs_and_encoder(hip_full, oracle_full):
    mode, _, got = hip_full
    _, ref, _, _, _ = oracle_full
    assert set(got) == {k for k in ref if not k.startswith("_")}
    assert got["depth_preds_feats"].shape[-2:] == (H // 4, W // 4) and got["traversability_preds_full"].shape[-2:] == (256, 256)
    for k in ("depth_preds_feats", "depth_preds_logits", "dino_pe_feats"):
        _stage(got[k], ref[k], 5e-5, f"{mode}:{k}")
    torch.testing.assert_close(got["depth_preds_metric"].cpu(), ref["depth_preds_metric"], rtol=1e-4, atol=2e-3)
    same = (got["depth_preds_bins"].cpu() == ref["depth_preds_bins"]).float().mean().item()
    assert same > 0.999, f"{mode}: argmax depth bins agree on {same:.5f} of the pixels"


def test_fullsize_stages_on_identical_inputs(hip_full, oracle_full):
    """Each stage fed with the ORACLE's tensors of the previous stage (no upstream round-off amplified): bit-exact
    voxel coordinates, fp32 costmap within 1e-4 -- at 608x1216 / 46,208 points per frame."""
    mode, model, _ = hip_full
    _, ref, _, (rgbd, p2p), occ = oracle_full
    B = rgbd.shape[0]
    cu = lambda t: t.detach().cuda().contiguous()
    assert occ > 0.02, f"BEV map of the test network is empty (occupied fraction {occ:.4f})"
    with torch.no_grad():
        sp = model.backbone.cam2map([cu(ref["depth_preds_metric"]).view(B, 1, H // 4, W // 4),
                                     cu(ref["depth_preds_feats"]).view(B, 1, 256, H // 4, W // 4), p2p.cuda()])
        assert torch.equal(sp["bev_coords"].cpu(), ref["bev_coords"])
        _stage(sp["bev_densities"], ref["bev_densities"], 1e-6, f"{mode}:bev_densities")
        _stage(sp["bev_features"], ref["bev_features"], 1e-5, f"{mode}:bev_features")
        heads = model.backbone.bevclassifier({"bev_features": cu(ref["bev_features"])})
        for k in ("inpainting_sam_preds", "inpainting_sam_dynamic_preds", "elevation_preds",
                  "inpainting_sam_features", "inpainting_sam_dynamic_features", "elevation_features"):
            _stage(heads[k], ref[k], 2e-5, f"{mode}:{k}")
        vin = model.traversability_head({k: cu(ref[k]) for k in model.traversability_head.reward_cfg["input_keys"]},
                                        None, False)
        assert torch.equal(vin["input_view"].cpu(), ref["input_view"].detach())
        torch.testing.assert_close(vin["traversability_preds"].cpu(), ref["traversability_preds"].detach(),
                                   rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(vin["traversability_preds_full"].cpu(), ref["traversability_preds_full"],
                                   rtol=1e-4, atol=1e-4)


def test_fullsize_end_to_end_within_fp32_noise_floor(hip_full, oracle_full):
    """Every float output against the float64 oracle, judged by the fp32 CPU oracle's own distance to it.
    Encoder-side keys (smooth in the inputs): rms and 99.9th percentile of |error| within 4x.  Keys behind the splat are
    DISCONTINUOUS in the point coordinates -- a point whose z sits within fp32 round-off of the +-range bound flips
    its range mask and moves its whole feature vector in or out of the map; the ResNet heads spread that over their
    receptive field.  One such flip among 92k points raised the rms of `bev_features` 9x and the 99.9th percentile of
    the head outputs 6x in ONE of the three modes (bf16x6, by chance: profiles/r02_fullsize_noise.md) while the other
    two, and the same mode on other data, sit at or below the fp32 reference's own error.  Those keys are therefore
    judged by the median and the 90th percentile of |error| (within 4x of the fp32 reference's; a local event cannot
    move them) plus a loose rms bound (16x)."""
    mode, _, got = hip_full
    _, ref, ref64, _, _ = oracle_full
    n = assert_within_fp32_noise_floor(got, ref, ref64, f"{mode}:")
    assert n >= 14
    flips = (got["bev_coords"].cpu().floor() != ref["bev_coords"].floor()).any(dim=-1).float().mean().item()
    assert flips < 2e-2, f"{mode}: end-to-end voxel-index mismatch rate {flips:.2e}"
    assert (got["traversability_preds"] >= 0).all()


def test_bench_batch_split_modes_against_exact_fp32_mode():
    """(b) batch 16 (the bench workload): every output key of the `f16x3` run and of the `bf16x6` run against the HIP
    `f32` run (exact fp32 products on the fp32 MFMA).  The 22-bit fp16 split must sit in the same noise class as the
    24-bit bf16 split: its distance to the exact-product run is within 3x of bf16x6's (+ an fp32-epsilon floor), and
    both are tiny in absolute terms."""
    B = 16
    torch.manual_seed(0)
    creste_public_amd.set_precision("f32")
    model = MaxEntIRL(maxent_irl_cfg((H, W), solve_mdp=False))
    synth.randomize_bn(model, seed=1)
    model = model.cuda().eval()
    rgbd, p2p = synth.make_frames(B, H, W, seed=1337)
    rgbd, p2p = rgbd.cuda(), p2p.cuda()
    synth.calibrate_bn_hip(model, rgbd[:2], p2p[:2])
    outs = {}
    try:
        for mode in MODES:
            creste_public_amd.set_precision(mode)
            with torch.no_grad():
                outs[mode] = {k: v.clone() for k, v in model((rgbd, p2p)).items()}
            torch.cuda.synchronize()
    finally:
        creste_public_amd.set_precision("f32")
    ref = outs["f32"]
    assert float((ref["bev_densities"] > 0).float().mean()) > 0.05
    for k, r in ref.items():
        if not r.is_floating_point():
            continue
        rr = r.double()
        d16, d6 = _rms(outs["f16x3"][k].double() - rr), _rms(outs["bf16x6"][k].double() - rr)
        scale = max(_rms(rr), 1e-12)
        assert d16 <= 3.0 * d6 + 2e-7 * scale, f"{k}: f16x3 {d16 / scale:.2e} vs bf16x6 {d6 / scale:.2e} (rel rms to f32)"
        # behind the splat the distance between two product roundings is a handful of flipped voxel indices, not arithmetic:
        # bf16x6 itself sits at 2.5e-2 of the costmap's rms from the exact-product run, f16x3 at 1.3e-2 .. 2.1e-2 depending on
        # the last bits of the BatchNorm statistics the calibration pass leaves (its channel sums changed their summation order
        # in round 6: 1.3e-2 -> 2.1e-2 with bf16x6 unchanged) -- the relative bound above is the test, this one a sanity limit
        lim = 4e-2 if k.startswith(("bev_", "inpainting", "elevation", "traversability", "input_view")) else 1e-4
        assert d16 <= lim * scale, f"{k}: f16x3 rel rms {d16 / scale:.2e}"
    for mode in ("bf16x6", "f16x3"):
        same = (outs[mode]["depth_preds_bins"] == ref["depth_preds_bins"]).float().mean().item()
        assert same > 0.999, (mode, same)
        flips = (outs[mode]["bev_coords"].floor() != ref["bev_coords"].floor()).any(dim=-1).float().mean().item()
        assert flips < 2e-2, (mode, flips)
