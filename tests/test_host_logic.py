"""CPU: host-side logic of the mirrored reference API -- checkpoint loading modes and key renaming
(reference terrainnet.py:111-261, distillation.py:95-127, depth.py:35-58, lfd.py:126-154), package
aliasing, config protocol, error conventions.  No GPU compute."""
import sys

import pytest
import torch

import creste_public_amd
from creste_public_amd import Cfg, HipLibraryError, maxent_irl_cfg, terrainnet_cfg
from creste_public_amd.creste.models.distillation import DistillationBackbone
from creste_public_amd.creste.models.lfd import MaxEntIRL
from creste_public_amd.creste.models.terrainnet import TerrainNet
from creste_public_amd.creste.utils.loss_utils import LossManager

IMG = (64, 96)


def _ckpt(tmp_path, sd, name="w.ckpt"):
    p = str(tmp_path / name)
    torch.save({"state_dict": sd, "epoch": 3}, p)
    return p


@pytest.fixture(scope="module")
def donor():
    torch.manual_seed(0)
    return TerrainNet(terrainnet_cfg(IMG))


def test_state_dict_names_match_the_reference_contract(donor):
    keys = set(donor.state_dict().keys())
    for k in ["depthcomp.depthcomp.vision_backbone.model.trunk._conv_stem.weight",
              "depthcomp.depthcomp.vision_backbone.model.trunk._blocks.0._depthwise_conv.weight",
              "depthcomp.depthcomp.vision_backbone.model.trunk._blocks.1._expand_conv.weight",
              "depthcomp.depthcomp.vision_backbone.model.trunk._blocks.15._se_reduce.bias",
              "depthcomp.depthcomp.vision_backbone.model.trunk._blocks.15._bn2.running_var",
              "depthcomp.depthcomp.vision_backbone.model.trunk._conv_head.weight",
              "depthcomp.depthcomp.vision_backbone.model.trunk._fc.bias",
              "depthcomp.depthcomp.vision_backbone.model.up3.conv.4.num_batches_tracked",
              "depthcomp.depthcomp.vision_backbone.model.conv.bias",
              "depthcomp.depthcomp.depth_head.model.1.running_mean",
              "depthcomp.dino_head.model.6.weight", "depthcomp.dino_head.model.7.bias",
              "cam2map.lidar2map", "cam2map.grid_size", "cam2map.z_proj.2.weight",
              "cam2map.vision_fusion.convs.1.running_var",
              "bevclassifier.conv1.weight", "bevclassifier.layer2.0.downsample.0.weight",
              "bevclassifier.layer3.1.bn2.weight", "bevclassifier.out_heads.2.up1.conv.3.weight",
              "bevclassifier.out_heads.0.up2.2.running_mean", "bevclassifier.out_heads.1.proj.bias"]:
        assert k in keys, k
    assert "depthcomp.depthcomp.vision_backbone.model.trunk._blocks.0._expand_conv.weight" not in keys
    assert sum(p.numel() for p in donor.parameters()) == 25560908            # SURVEY.md 2.4 (25.6 M)
    assert list(donor.cam2map.grid_size) == [256, 256, 1]
    irl = MaxEntIRL(maxent_irl_cfg(IMG))
    ik = set(irl.state_dict().keys())
    assert {"dynamics", "transition_probs", "traversability_head.w",
            "traversability_head.r.prepool.0.conv.weight", "traversability_head.r.trunk.2.running_mean",
            "traversability_head.r.trunk.4.conv.weight", "traversability_head.r.postpool.0.norm.bias"} <= ik
    assert all(("backbone." + k) in ik for k in keys)
    assert sum(p.numel() for p in irl.traversability_head.parameters()) == 102866
    assert all(not p.requires_grad for p in irl.backbone.parameters()) and not irl.backbone.training
    irl.train()
    assert not irl.backbone.training and irl.traversability_head.training    # frozen backbone stays in eval


def test_load_weights_key_renaming_and_settings(tmp_path, donor):
    sd = donor.state_dict()
    # a stage-1 style checkpoint: Lightning 'model.' prefix, single 'depthcomp.' level, top-level dino_head
    old = {}
    for k, v in sd.items():
        if k.startswith("depthcomp.depthcomp."):
            k2 = k.replace("depthcomp.depthcomp.", "depthcomp.", 1)
        elif k.startswith("depthcomp.dino_head."):
            k2 = k.replace("depthcomp.dino_head.", "dino_head.", 1)
        else:
            k2 = k
        old["model." + k2] = v.clone() + (0.25 if v.is_floating_point() else 0)
    old["model.loss.some_buffer"] = torch.zeros(1)
    path = _ckpt(tmp_path, old)
    for mode, trainable in [("strict", None), ("strict_freeze", lambda n: False),
                            ("strict_unfreezesplat", lambda n: "cam2map." in n),
                            ("ft_decoders_all", lambda n: "bevclassifier.out_heads" in n),
                            ("ft_decoders_partial",
                             lambda n: "bevclassifier.out_heads" in n and ("up2" in n or "proj" in n))]:
        cfg = terrainnet_cfg(IMG)
        cfg["load_setting"] = mode
        m = TerrainNet(cfg)
        m.load_weights(path)
        got = m.state_dict()
        for k, v in sd.items():
            dropped = (mode == "ft_decoders_all" and "bevclassifier.out_heads" in k) or \
                      (mode == "ft_decoders_partial" and "bevclassifier.out_heads" in k and ("up2" in k or "proj" in k))
            if v.is_floating_point() and not dropped:
                assert torch.equal(got[k], v + 0.25), (mode, k)
        if trainable is not None:
            for n, p in m.named_parameters():
                assert p.requires_grad == bool(trainable(n)), (mode, n)
    cfg = terrainnet_cfg(IMG)
    cfg["load_setting"] = "bogus"
    with pytest.raises(ValueError):
        TerrainNet(cfg).load_weights(path)
    # DistillationBackbone / DepthCompletion loaders (drop bevclassifier + cam2map, undo the renaming)
    new_style = {"model." + k: v for k, v in sd.items()}
    p2 = _ckpt(tmp_path, new_style, "new.ckpt")
    db = DistillationBackbone(terrainnet_cfg(IMG))
    db.load_weights(p2)
    assert torch.equal(db.state_dict()["depthcomp.depth_head.model.0.weight"],
                       sd["depthcomp.depthcomp.depth_head.model.0.weight"])
    db.depthcomp.load_weights(path)       # stage-1 layout: one 'depthcomp.' level (reference depth.py:41-58)
    assert torch.equal(db.depthcomp.state_dict()["depth_head.model.0.weight"],
                       sd["depthcomp.depthcomp.depth_head.model.0.weight"] + 0.25)
    db.unfreeze_backbone()
    assert all(p.requires_grad for p in db.depthcomp.parameters())
    # MaxEntIRL.load_weights: strict load + freezing
    irl = MaxEntIRL(maxent_irl_cfg(IMG))
    p3 = _ckpt(tmp_path, {"model." + k: v for k, v in irl.state_dict().items()}, "irl.ckpt")
    irl2 = MaxEntIRL(maxent_irl_cfg(IMG))
    irl2.freeze_head = True
    irl2.load_weights(p3)
    assert all(not p.requires_grad for p in irl2.parameters())
    irl2.train()
    assert not irl2.traversability_head.training


def test_install_as_creste_aliases_reference_import_paths():
    saved = {k: v for k, v in sys.modules.items() if k == "creste" or k.startswith("creste.")}
    try:
        creste_public_amd.install_as_creste()
        from creste.models.lfd import MaxEntIRL as A                     # noqa: the reference's own import lines
        from creste.models.terrainnet import TerrainNet as B
        from creste.models.blocks.splat_projection import Camera2MapMulti  # noqa: F401
        from creste.models.blocks.vin import VIN                         # noqa: F401
        from creste.utils.loss_utils import LossManager as L
        import creste.utils.train_utils as tu
        assert A is MaxEntIRL and B is TerrainNet and L is LossManager
        assert tu.create_trapezoidal_fov_mask(128, 128, 70, 70, 0, 100).dtype == torch.bool
    finally:
        for k in [k for k in sys.modules if k == "creste" or k.startswith("creste.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_error_conventions_and_cfg_protocol():
    cfg = terrainnet_cfg(IMG)
    assert cfg.vision_backbone.effnet_cfgs.image_size == list(IMG) and cfg.get("nope", 7) == 7
    assert isinstance(cfg["camera_projector"]["vision_fusion"], Cfg) and cfg.to_dict()["views"] == 1
    bad = terrainnet_cfg(IMG)
    bad["vision_backbone"]["class_name"] = "FoundationBackbone"
    with pytest.raises(NotImplementedError):
        TerrainNet(bad)
    bad = terrainnet_cfg(IMG)
    bad["bev_classifier"]["name"] = "Nope"
    with pytest.raises(NotImplementedError):
        TerrainNet(bad)
    bad = terrainnet_cfg(IMG)
    bad["use_temporal"] = True
    with pytest.raises(NotImplementedError):
        TerrainNet(bad)
    c = maxent_irl_cfg(IMG)
    c["loss"][0]["name"] = "VicregLoss"
    with pytest.raises(NotImplementedError):
        LossManager(c)
    c = maxent_irl_cfg(IMG)
    c["vision_backbone"]["project_name"] = "Other"
    with pytest.raises(ValueError):
        MaxEntIRL(c)
    m = TerrainNet(terrainnet_cfg(IMG)).eval()
    with pytest.raises(HipLibraryError):                                   # no CPU fallback
        m((torch.zeros(1, 1, 4, *IMG), torch.eye(4).view(1, 1, 4, 4)))
    if torch.cuda.is_available():
        with pytest.raises(NotImplementedError):                           # backbone training: later round
            m.cuda().train()((torch.zeros(1, 1, 4, *IMG).cuda(), torch.eye(4).view(1, 1, 4, 4).cuda()))


def test_reward_net_training_refuses_cpu():
    """the product reward network trains on the HIP kernels only: a CPU tensor must fail loudly, not fall back"""
    from creste_public_amd.creste.models.blocks.conv import MultiScaleFCN
    from creste_public_amd.ops import HipLibraryError
    cfg = maxent_irl_cfg()["traversability_head"]["net_kwargs"]["reward_cfg"]["net_kwargs"]
    net = MultiScaleFCN(cfg).train()
    with pytest.raises(HipLibraryError):
        net(torch.rand(1, 40, 16, 16, requires_grad=True))


def test_depth_label_binning_and_smoothl1_match_reference_golden():
    """host-side label logic of the distillation losses against vectors produced by the reference's own code
    (tests/golden/make_golden.py: depth_utils.bin_depths target bins; SmoothL1Depth through its LossManager)."""
    import os
    import numpy as np
    from creste_public_amd.creste.utils.loss_utils import SmoothL1Depth, _bin_depths_ud
    gold = os.path.join(os.path.dirname(__file__), "golden")
    u = np.load(os.path.join(gold, "utils.npz"))
    got = _bin_depths_ud(torch.from_numpy(u["depth_map"]), 300, 25600, 128)
    assert torch.equal(got, torch.from_numpy(u["bins_target"]))
    d = np.load(os.path.join(gold, "distill_losses.npz"))
    disc = dict(mode="UD", num_bins=128, depth_min=300, depth_max=25600)
    l1 = SmoothL1Depth(dict(name="SmoothL1Depth", weight=0.1, pred_key="outputs/depth_preds_bins",
                            lab_key="inputs/depth_label", beta=0.5, discretize=disc))
    out, _ = l1({"outputs/depth_preds_bins": torch.from_numpy(d["pred_bins"]),
                 "inputs/depth_label": torch.from_numpy(d["depth_label"])})
    w, v = out["depth/reg_loss"]
    assert abs(float(v) - float(d["loss/SmoothL1Depth/depth/reg_loss"])) < 1e-5 * float(v) and w == 0.1


def test_ssc_losses_match_reference_golden(tmp_path):
    """SupPixelConLoss / MultiPosConLoss, CrossEntropy (class weights, class_dim), SmoothL1 (relative elevation, nan/inf
    masked) and SmoothL1Depth on metric depth: the host-side mirrors against the reference's own LossManager
    (tests/golden/make_golden.py::gen_ssc_losses) -- losses, metric, weighted total and all four prediction gradients."""
    import os
    import numpy as np
    from creste_public_amd.creste.utils.loss_utils import LossManager
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "ssc_losses.npz"))
    t = lambda k: torch.from_numpy(d[k])          # noqa: E731
    wfile = tmp_path / "w6.txt"
    np.savetxt(wfile, d["class_freq"])
    disc = dict(mode="UD", num_bins=128, depth_min=300, depth_max=25600)
    lm = LossManager({"loss": [
        dict(name="SupPixelConLoss", views=1, weight=1.0, pred_key="outputs/inpainting_sam_preds",
             lab_key="inputs/3d_sam_label", ignore_index=0, temperature=0.1, task="joint", contrast_mode="batch_all"),
        dict(name="CrossEntropy", weight=2.0, pred_key="outputs/inpainting_sam_dynamic_preds",
             lab_key="inputs/3d_sam_dynamic_label", num_class=6, class_weights=str(wfile), class_dim=1, task="joint"),
        dict(name="SmoothL1Depth", weight=0.1, pred_key="outputs/depth_preds_metric", lab_key="inputs/depth_label",
             beta=0.5, discretize=disc),
        dict(name="SmoothL1", weight=3.0, beta=0.2, pred_key="outputs/elevation_preds", lab_key="inputs/elevation_label",
             absolute=False, task="joint")]})
    preds = {k: t(k).clone().requires_grad_(True) for k in ("sam_pred", "dyn_pred", "depth_pred", "elev_pred")}
    elev_label = t("elev_label").clone()
    td = {"outputs/inpainting_sam_preds": preds["sam_pred"], "inputs/3d_sam_label": t("sam_label"),
          "outputs/inpainting_sam_dynamic_preds": preds["dyn_pred"], "inputs/3d_sam_dynamic_label": t("dyn_label"),
          "inputs/fov_mask": t("fov"), "outputs/depth_preds_metric": preds["depth_pred"],
          "inputs/depth_label": t("depth_label"), "outputs/elevation_preds": preds["elev_pred"],
          "inputs/elevation_label": elev_label, "task": "joint"}
    torch.manual_seed(77)
    ld, meta = lm(td)
    total = sum(w * v for w, v in ld.values())
    total.backward()
    assert set(ld) == {k[5:] for k in d.files if k.startswith("loss/")}
    for k, (w, v) in ld.items():
        assert abs(float(v) - float(d[f"loss/{k}"])) < 1e-5 * abs(float(d[f"loss/{k}"])), k
        assert abs(float(w) - float(d[f"weight/{k}"])) < 1e-7, k
    assert abs(float(meta["CrossEntropy/joint/mIoU"]) - float(d["meta/CrossEntropy/joint/mIoU"])) < 1e-6
    assert abs(float(total) - float(d["total"])) < 1e-5 * float(d["total"])
    for k, gk in (("sam_pred", "g_sam"), ("dyn_pred", "g_dyn"), ("depth_pred", "g_depth"), ("elev_pred", "g_elev")):
        torch.testing.assert_close(preds[k].grad, t(gk), rtol=1e-4, atol=1e-7)
    nan_ok = torch.isnan(elev_label) == torch.isnan(t("elev_label"))       # the caller's label tensor is not rewritten
    assert nan_ok.all() and torch.equal(torch.nan_to_num(elev_label), torch.nan_to_num(t("elev_label")))


def test_value_iteration_sweep_count_contract_and_prefetch_key():
    """Host-side contracts that need no GPU: the sign convention of the asynchronous sweep count (ops.check_vi_sweeps) and
    the identity-based key of MaxEntIRL's prefetched frozen half (a key of addresses alone matched a NEW batch allocated in
    a dropped batch's block)."""
    import pytest
    import torch
    from creste_public_amd import ops
    from creste_public_amd._lib import HipLibraryError
    from creste_public_amd.creste.models.lfd import MaxEntIRL
    assert ops.check_vi_sweeps(torch.tensor([690], dtype=torch.int32)) == 690
    with pytest.raises(HipLibraryError, match="no convergence within 16 sweeps"):
        ops.check_vi_sweeps(torch.tensor([-16], dtype=torch.int32))
    with pytest.raises(HipLibraryError, match="resident"):
        ops.check_vi_sweeps(torch.tensor([-2 ** 31], dtype=torch.int32))
    a, p = torch.zeros(2, 3), torch.zeros(2, 4, 4)
    key = MaxEntIRL._input_key((a, p))
    assert MaxEntIRL._same_inputs(key, (a, p))
    b = torch.zeros(2, 3)                          # same shape / dtype / (possibly) address, another tensor
    assert not MaxEntIRL._same_inputs(key, (b, p))
    a.add_(1.0)                                    # same tensor, written since the prefetch
    assert not MaxEntIRL._same_inputs(key, (a, p))


def test_upsample_conv_as_phase_convolutions_identity_in_float64():
    """The algebra behind ops.upconv2x (reference DeconvHead.up2, inpainting.py:56-60: Upsample(x2, bilinear, align_corners=False) ->
    Conv2d(3, padding=1)), in float64 on the CPU, no kernels:
    (a) ops.phase_upconv_weights: the conv over the upsampled image == the 3x3 conv with the composed 4 x Cout kernels on the
        replicate-padded low-resolution image, phase (a, b) of pixel (y, x) being pixel (2y + a, 2x + b) -- exactly, away from the border;
    (b) on the outermost ring the two differ by the taps whose source lies outside the upsampled image (zero there, the replicate-padded
        interpolation in the phase form): subtracting w[ky, kx] * u~(source) for exactly those taps -- what
        creste_upconv2x_ring_fix_f32 does, u~ = the bilinear formula with an UNCLAMPED source coordinate on clamped indices -- restores
        equality everywhere, corners included."""
    import math
    import torch.nn.functional as F
    from creste_public_amd import ops
    torch.manual_seed(0)
    H, W, Cin, Cout = 5, 6, 2, 3
    x = torch.randn(1, Cin, H, W, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 3, 3, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False), w, padding=1)
    R = torch.tensor([[[0.75, 0.25, 0.0], [0.25, 0.75, 0.0], [0.0, 0.75, 0.25]],
                      [[0.25, 0.75, 0.0], [0.0, 0.75, 0.25], [0.0, 0.25, 0.75]]], dtype=torch.float64)
    wp = torch.einsum("ayp,bxq,oiyx->aboipq", R, R, w).reshape(4 * Cout, Cin, 3, 3)
    # the packer composes the same kernels (in float64, rounded once to fp32)
    assert torch.allclose(ops.phase_upconv_weights(w.float()).double(), wp, rtol=0, atol=2e-7 * float(wp.abs().max()))
    ph = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), wp)
    out = torch.zeros_like(ref)
    for a in range(2):
        for b in range(2):
            out[:, :, a::2, b::2] = ph[:, (2 * a + b) * Cout:(2 * a + b + 1) * Cout]
    d = (out - ref).abs()
    assert float(d[:, :, 1:-1, 1:-1].max()) < 1e-12 and float(d.max()) > 1e-3          # exact inside, wrong on the ring

    def u_tilde(Y, X):
        sy, sx = 0.5 * (Y + 0.5) - 0.5, 0.5 * (X + 0.5) - 0.5
        y0, x0 = math.floor(sy), math.floor(sx)
        fy, fx = sy - y0, sx - x0
        cl = lambda v, n: min(max(v, 0), n - 1)
        ya, yb, xa, xb = cl(y0, H), cl(y0 + 1, H), cl(x0, W), cl(x0 + 1, W)
        return (1 - fy) * ((1 - fx) * x[0, :, ya, xa] + fx * x[0, :, ya, xb]) + fy * ((1 - fx) * x[0, :, yb, xa] + fx * x[0, :, yb, xb])

    H2, W2 = 2 * H, 2 * W
    for oy in range(H2):
        for ox in range(W2):
            if oy in (0, H2 - 1) or ox in (0, W2 - 1):
                for ky in range(3):
                    for kx in range(3):
                        Y, X = oy + ky - 1, ox + kx - 1
                        if not (0 <= Y < H2 and 0 <= X < W2):
                            out[0, :, oy, ox] -= w[:, :, ky, kx] @ u_tilde(Y, X)
    assert float((out - ref).abs().max()) < 1e-12
