"""GPU: BEV-SSC training path -- Camera2MapMulti (pixel geometry + z-MLP + fusion conv + splat) as an autograd Function
and the whole TerrainNet.train() forward/backward, against float64 autograd of the oracle."""
import pytest
import torch
import torch.nn.functional as F

from creste_public_amd import synth
from creste_public_amd.config import terrainnet_cfg

pytestmark = pytest.mark.gpu


def _p95(a, b):
    a, b = a.double().cpu().flatten(), b.double().flatten()
    k = max(1, int(0.95 * a.numel()))
    return float((a - b).abs().kthvalue(k).values / b.abs().max().clamp_min(1e-30))


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def test_splat_stage_training():
    """depth, features -> bev_features, bev_densities and back: gradients w.r.t. features, DEPTH (through the tap
    weights and the densities) and the z-MLP / fusion parameters."""
    from oracle import perception as op
    from creste_public_amd.creste.models.blocks.splat_projection import Camera2MapMulti
    from creste_public_amd.train_terrain import SplatFn, SplatTrainEngine
    torch.manual_seed(2)
    cfg = terrainnet_cfg()["camera_projector"]
    ref = op.Camera2MapMulti(cfg)
    synth.randomize_bn(ref, seed=3)
    m = Camera2MapMulti(cfg)
    m.load_state_dict(ref.state_dict(), strict=True)
    ref = ref.double().train()
    m = m.cuda().train()
    B, Hs, Ws, Fd = 2, 24, 36, 256
    g = torch.Generator().manual_seed(5)
    depth = torch.rand(B, Hs, Ws, generator=g) * 12.0 + 0.5
    feats = torch.randn(B, Fd, Hs, Ws, generator=g)
    p2p = synth.make_p2p(B, Hs * 4, Ws * 4)
    dr, fr = depth.double().requires_grad_(True), feats.double().requires_grad_(True)
    out_r = ref([dr.view(B, 1, Hs, Ws), fr.view(B, 1, Fd, Hs, Ws), p2p.double()])
    wb = torch.randn(out_r["bev_features"].shape, generator=g)
    wd = torch.randn(out_r["bev_densities"].shape, generator=g) * 0.1
    ((out_r["bev_features"] * wb.double()).sum() + (out_r["bev_densities"] * wd.double()).sum()).backward()

    eng = SplatTrainEngine(m)
    dg, fg = depth.cuda().requires_grad_(True), feats.cuda().requires_grad_(True)
    bev, dens, coords = SplatFn.apply(eng, dg, fg, p2p.view(B, 4, 4).cuda(), None, *eng.params())
    ((bev * wb.cuda()).sum() + (dens * wd.cuda()).sum()).backward()
    torch.cuda.synchronize()
    assert _rel(coords, out_r["bev_coords"]) < 1e-5
    # a point within float32 round-off of a cell border lands in the neighbouring cell of the float64 oracle: compare
    # through percentiles / rms, the kernels themselves are held to 1e-4..1e-5 in the primitive tests
    assert _p95(bev, out_r["bev_features"]) < 1e-4 and _p95(dens, out_r["bev_densities"]) < 1e-4
    assert _p95(fg.grad, fr.grad) < 1e-3
    assert _p95(dg.grad, dr.grad) < 2e-3, _p95(dg.grad, dr.grad)
    ref_p = dict(ref.named_parameters())
    gscale = max(float(p.grad.abs().max()) for p in ref_p.values())
    for name, p in m.named_parameters():        # (the conv bias in front of the BatchNorm has an exactly zero gradient)
        assert p.grad is not None, name
        e = _p95(p.grad, ref_p[name].grad) * float(ref_p[name].grad.abs().max()) / gscale
        assert e < 5e-3, (name, e)


def _ssc_objective(out, depth_label, fimg, ws):
    """a BEV-SSC-shaped objective built from plain tensor ops (the SSC losses themselves are a later round):
    linear read-outs of the three BEV heads + depth classification + metric-depth regression + feature matching"""
    bins = ((depth_label - 300.0) / ((25600.0 - 300.0) / 128)).view(depth_label.shape[0], *depth_label.shape[-2:])
    bad = (bins < 0) | (bins > 128) | ~torch.isfinite(bins)
    bins = bins.masked_fill(bad, 128).long()
    valid = bins != 128
    loss = F.cross_entropy(out["depth_preds_logits"].permute(0, 2, 3, 1)[valid], bins[valid])
    loss = loss + 0.1 * F.smooth_l1_loss(out["depth_preds_metric"][valid], (depth_label.view_as(bins) / 1000.0)[valid], beta=0.5)
    loss = loss + F.mse_loss(out["dino_pe_feats"], fimg)
    for k, w in ws.items():
        loss = loss + (out[k] * w).mean()
    return loss


def test_movability_masked_splat_matches_reference_golden(golden):
    """Camera2MapMulti.train() with the immovable mask as 4th input against the REFERENCE's own run
    (tests/golden/splat_mv.npz): `_mv` keys, values, BatchNorm running statistics, gradients."""
    from conftest import analytic_cotangent
    from creste_public_amd.creste.models.blocks.splat_projection import Camera2MapMulti
    g = golden("splat_mv.npz")
    m = Camera2MapMulti(terrainnet_cfg()["camera_projector"])
    m.load_state_dict(g.sd(), strict=True)
    m = m.cuda().train()
    depth, feats = g.t("depth").cuda().requires_grad_(True), g.t("feats").cuda().requires_grad_(True)
    out = m([depth, feats, g.t("p2p").cuda(), g.t("mv_mask").cuda()])
    assert set(out) == {"bev_features_mv", "bev_densities_mv", "bev_coords_mv"}
    bf, dens = out["bev_features_mv"], out["bev_densities_mv"]
    assert torch.equal(out["bev_coords_mv"].cpu(), g.t("bev_coords"))
    torch.testing.assert_close(dens.cpu(), g.t("bev_densities"), rtol=0, atol=1e-6)
    idx = g.t("touched_idx")
    torch.testing.assert_close(bf.permute(0, 2, 3, 1).cpu()[idx[:, 0], idx[:, 1], idx[:, 2]], g.t("touched_feats"),
                               rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bf.abs().sum().cpu(), g.t("bev_features_abs_sum"), rtol=1e-5, atol=0)
    ((bf * analytic_cotangent(bf.shape, 0.0).cuda()).sum() + (dens * analytic_cotangent(dens.shape, 1.0).cuda()).sum()).backward()
    torch.cuda.synchronize()
    assert _rel(depth.grad, g.t("g_depth")) < 1e-3 and _p95(depth.grad, g.t("g_depth")) < 1e-3
    assert _rel(feats.grad, g.t("g_feats")) < 1e-4
    assert _rel(m.vision_fusion.convs[0].weight.grad, g.t("g_fuse_w")) < 1e-3
    assert _rel(m.z_proj[0].weight.grad, g.t("g_z0_w")) < 1e-3
    for k in ("running_mean", "running_var"):
        torch.testing.assert_close(getattr(m.vision_fusion.convs[1], k).cpu(), g.t(f"after/vision_fusion.convs.1.{k}"),
                                   rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("movability", [False, True])
def test_terrainnet_training_step(movability):
    import oracle.blocks as ob
    from oracle.perception import TerrainNet as OracleNet
    from creste_public_amd import train_backbone as TB
    from creste_public_amd.creste.models.terrainnet import TerrainNet
    H, W, B = 64, 96, 2
    torch.manual_seed(31)
    cfg = terrainnet_cfg((H, W))
    if movability:          # terrainnet.py:310-344: anchor splat + mask splat, two passes of the BEV heads
        cfg["use_movability"] = True
    ob.DROP_CONNECT, TB.DROP_CONNECT = 0.0, 0.0
    try:
        ref = OracleNet(cfg)
        synth.randomize_bn(ref, seed=2)
        model = TerrainNet(cfg)
        model.load_state_dict(ref.state_dict(), strict=True)
        ref = ref.double().train()
        model = model.cuda().train()
        rgbd, p2p = synth.make_frames(B, H, W, seed=3)
        rgbd[:, :, 3] /= 1000.0
        g = torch.Generator().manual_seed(4)
        Hs, Ws = H // 4, W // 4
        depth_label = torch.rand(B, 1, Hs, Ws, generator=g) * 30000.0 - 2000.0
        fimg = torch.randn(B, 1, 128, Hs, Ws, generator=g)
        keys = ["inpainting_sam_preds", "inpainting_sam_dynamic_preds", "elevation_preds"]
        mv = (torch.rand(B, 1, Hs, Ws, generator=g) > 0.25).float()
        extra_r, extra = ((mv.double(),), (mv.cuda(),)) if movability else ((), ())
        out_r = ref((rgbd.double(), p2p.double()) + extra_r)
        ws = {k: torch.randn(out_r[k].shape, generator=g) for k in keys}
        loss_r = _ssc_objective(out_r, depth_label.double(), fimg.double(), {k: v.double() for k, v in ws.items()})
        loss_r.backward()
        out = model((rgbd.cuda(), p2p.cuda()) + extra)
        loss = _ssc_objective(out, depth_label.cuda(), fimg.cuda(), {k: v.cuda() for k, v in ws.items()})
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ob.DROP_CONNECT, TB.DROP_CONNECT = 0.2, 0.2
    assert set(out.keys()) == set(out_r.keys())
    for k in ("depth_preds_logits", "depth_preds_metric", "dino_pe_feats"):
        assert _rel(out[k], out_r[k]) < 2e-4, k
    more = ("bev_features_mv", "bev_densities_mv", "inpainting_sam_mv_preds") if movability else ()
    assert all(k in out for k in more)
    for k in ("bev_features", "bev_densities") + tuple(keys) + more:
        assert _p95(out[k], out_r[k]) < 2e-3, (k, _p95(out[k], out_r[k]))
    if movability:          # both passes updated the BatchNorm running statistics, in the reference's order
        for name in ("bevclassifier.bn1", "cam2map.vision_fusion.convs.1"):
            a, b = model.get_submodule(name), ref.get_submodule(name)
            assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 2, name
            assert _rel(a.running_var, b.running_var) < 1e-3 and _rel(a.running_mean, b.running_mean) < 1e-3, name
    assert abs(float(loss) - float(loss_r)) < 2e-3 * abs(float(loss_r))
    ref_p = dict(ref.named_parameters())
    unused = ("_conv_head", "trunk._bn1", "_fc")
    gscale = max(float(p.grad.abs().max()) for p in ref_p.values() if p.grad is not None)
    bad = []
    for name, p in model.named_parameters():
        if any(u in name for u in unused):
            continue
        assert p.grad is not None, name
        r = ref_p[name].grad
        e = _p95(p.grad, r) * float(r.abs().max()) / gscale
        if e > 2e-2:
            bad.append((name, f"{e:.1e}"))
    assert not bad, (len(bad), bad[:30])


def _ssc_batch(B, H, W, seed=0, G=256):
    rgbd, p2p = synth.make_frames(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    Hs, Ws = H // 4, W // 4
    blocks = torch.randint(0, 5, (B, 1, G // 16, G // 16), generator=g)        # piecewise-constant SAM segments
    data = {"image": rgbd, "p2p": p2p,
            "depth_label": torch.rand(B, 1, Hs, Ws, generator=g) * 26000.0,
            "fimg_label": torch.randn(B, 1, 128, Hs, Ws, generator=g),
            "3d_sam_label": blocks.repeat_interleave(16, 2).repeat_interleave(16, 3),
            "3d_sam_dynamic_label": torch.stack([torch.zeros(B, G, G),
                                                 torch.randint(0, 6, (B, G // 8, G // 8), generator=g).float()
                                                 .repeat_interleave(8, 1).repeat_interleave(8, 2)], dim=1),
            "fov_mask": torch.rand(B, G, G, generator=g) > 0.5,
            "elevation_label": torch.randn(B, 2, G, G, generator=g)}
    return {"joint": {k: v.cuda() for k, v in data.items()}}


def test_ssc_trainer_steps_freeze_schedule_and_checkpoint(tmp_path):
    """row H (train_ssc.py): the full six-loss SSC objective drives Adam steps of TerrainNet on the HIP training path;
    the backbone is frozen for the first epoch and unfrozen afterwards; Lightning-layout checkpoint round trip."""
    from creste_public_amd import harness
    from creste_public_amd.creste.models.terrainnet import TerrainNet
    from creste_public_amd.creste.utils.loss_utils import LossManager
    H, W, B = 64, 96, 2
    harness.seed_everything(5)
    cfg = harness.ssc_cfg((H, W), class_weights=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05], freeze_backbone_epochs=1)
    model = TerrainNet(cfg).cuda()
    synth.randomize_bn(model, seed=2)
    tr = harness.SSCTrainer(model, LossManager(cfg).cuda(), cfg)
    batch = _ssc_batch(B, H, W)
    assert tr.backbone_frozen and not any(p.requires_grad for p in model.depthcomp.parameters())
    w0 = model.depthcomp.depthcomp.vision_backbone.model.up3.conv[0].weight.detach().clone()
    h0 = model.bevclassifier.conv1.weight.detach().clone()
    logs = [tr.training_step(batch) for _ in range(3)]
    assert all(torch.isfinite(l["train/loss"]) for l in logs)
    assert torch.equal(w0, model.depthcomp.depthcomp.vision_backbone.model.up3.conv[0].weight)     # frozen
    assert not torch.equal(h0, model.bevclassifier.conv1.weight)
    expect = {"train/SupPixelConLoss/joint/3d_sam_label/supcon/sem_loss", "train/CrossEntropy/joint/cls_loss",
              "train/MSELoss/loss", "train/CrossEntropyDepth/depth/cls_loss", "train/SmoothL1Depth/depth/reg_loss",
              "train/SmoothL1/val", "train/CrossEntropy/joint/mIoU", "train/loss"}
    assert expect <= set(logs[0])
    tr.on_train_epoch_end()
    tr.on_train_epoch_start()                                   # epoch 1 >= freeze_backbone_epochs -> unfreeze
    assert not tr.backbone_frozen
    losses = [float(tr.training_step(batch)["train/loss"]) for _ in range(4)]
    assert not torch.equal(w0, model.depthcomp.depthcomp.vision_backbone.model.up3.conv[0].weight)
    assert losses[-1] < float(logs[0]["train/loss"]), (losses, float(logs[0]["train/loss"]))
    path = tmp_path / "ssc.ckpt"
    tr.save_checkpoint(str(path))
    model2 = TerrainNet(cfg)
    tr2 = harness.SSCTrainer(model2, LossManager(cfg), cfg)
    tr2.load_checkpoint(str(path))
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a.cpu(), b), k


@pytest.mark.parametrize("N,D,ncls,weights", [(700, 32, 9, False), (2500, 32, 40, True), (300, 8, 3, True), (1030, 64, 1, False)])
def test_fused_multipos_contrastive_loss(N, D, ncls, weights):
    """csrc/losses.hip (no N x N tensors) against the tensor-code restatement of MultiPosConLoss evaluated in float64
    on the CPU: loss and gradient w.r.t. the un-normalised features (classes with a single member, i.e. rows without
    positives, included)."""
    from creste_public_amd.creste.utils.loss_utils import MultiPosConLoss
    g = torch.Generator().manual_seed(N)
    feats = torch.randn(N, D, generator=g)
    labels = torch.randint(0, ncls, (N,), generator=g)
    labels[0] = ncls + 5                                        # a singleton class: no positives for row 0
    cw = (torch.rand(ncls + 6, generator=g) + 0.5) if weights else None
    ref = MultiPosConLoss(0.1, cw.double() if weights else None)
    fr = feats.double().requires_grad_(True)
    lr = ref({"feats": fr, "labels": labels})["loss"]
    lr.backward()
    hip = MultiPosConLoss(0.1, cw.cuda() if weights else None)
    fg = feats.cuda().requires_grad_(True)
    out = hip({"feats": fg, "labels": labels.cuda()})
    (out["loss"] * 1.7).backward()
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(lr)) < 2e-6 * abs(float(lr))
    assert _rel(fg.grad, 1.7 * fr.grad) < 2e-5
    # the reference's mask caching: a second batch of the SAME size reuses the first batch's positives
    labels2 = torch.randint(0, ncls, (N,), generator=g)
    l2r = ref({"feats": fr.detach(), "labels": labels2})["loss"]
    l2h = hip({"feats": fg.detach(), "labels": labels2.cuda()})["loss"]
    assert abs(float(l2h) - float(l2r)) < 2e-6 * abs(float(l2r))


@pytest.mark.parametrize("N,M,off,D", [(200, 523, 200, 16), (333, 1000, 500, 32), (97, 97, 0, 64), (64, 4096, 1024, 32)])
def test_multipos_contrastive_kernels_with_gathered_columns(N, M, off, D):
    """the data-parallel shape of the loss: N local rows against M >= N gathered columns, the local rows sitting at
    `off` in the gathered list (self pairs excluded there), row weights -- forward and both gradients against float64
    tensor code (definition: supcon_loss.py:56-115); D = 16 / 32 / 64 run on the matrix cores (csrc/supcon_mfma.hip)."""
    from creste_public_amd.loss_ops import MultiPosConFn
    g = torch.Generator().manual_seed(N + M)
    T = 0.07
    a = torch.nn.functional.normalize(torch.randn(M, D, generator=g), dim=1)
    la = torch.randint(0, 7, (M,), generator=g)
    la[off] = 99                                                # local row 0: no positives
    w = torch.rand(N, generator=g) + 0.25
    ar = a.double().requires_grad_(True)
    fr = ar[off:off + N]
    z = fr @ ar.t() / T
    eye = torch.zeros(N, M, dtype=torch.bool)
    eye[torch.arange(N), torch.arange(N) + off] = True
    pos = (la[off:off + N, None] == la[None, :]) & ~eye
    lse = torch.logsumexp(z.masked_fill(eye, float("-inf")), dim=1)
    cnt = pos.sum(1)
    per = torch.where(cnt > 0, lse - (z * pos).sum(1) / cnt.clamp(min=1), torch.zeros_like(lse))
    loss_r = (per * w.double()).mean()
    loss_r.backward()
    fg = a[off:off + N].clone().cuda().requires_grad_(True)
    ag = a.clone().cuda().requires_grad_(True)
    loss = MultiPosConFn.apply(fg, ag, la[off:off + N].cuda(), la.cuda(), w.cuda(), off, T)
    (2.5 * loss).backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_r)) < 3e-6 * abs(float(loss_r))
    gt = ar.grad.clone()                                         # float64 autograd put both roles into one tensor
    got = ag.grad.cpu().double()
    got[off:off + N] += fg.grad.cpu().double()
    assert _rel(got, 2.5 * gt) < 3e-5, _rel(got, 2.5 * gt)


def test_ssc_losses_on_gpu_match_reference_golden(tmp_path):
    """the SSC objective as the GPU runs it (fused contrastive + depth-CE/MSE kernels where they apply) against the
    reference's own LossManager outputs."""
    import os
    import numpy as np
    from creste_public_amd.creste.utils.loss_utils import LossManager
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "ssc_losses.npz"))
    t = lambda k: torch.from_numpy(d[k]).cuda()          # noqa: E731
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
             absolute=False, task="joint")]}).cuda()
    preds = {k: t(k).clone().requires_grad_(True) for k in ("sam_pred", "dyn_pred", "depth_pred", "elev_pred")}
    td = {"outputs/inpainting_sam_preds": preds["sam_pred"], "inputs/3d_sam_label": t("sam_label"),
          "outputs/inpainting_sam_dynamic_preds": preds["dyn_pred"], "inputs/3d_sam_dynamic_label": t("dyn_label"),
          "inputs/fov_mask": t("fov"), "outputs/depth_preds_metric": preds["depth_pred"],
          "inputs/depth_label": t("depth_label"), "outputs/elevation_preds": preds["elev_pred"],
          "inputs/elevation_label": t("elev_label"), "task": "joint"}
    torch.manual_seed(77)
    ld, meta = lm(td)
    total = sum(w * v for w, v in ld.values())
    total.backward()
    for k, (w, v) in ld.items():
        assert abs(float(v) - float(d[f"loss/{k}"])) < 2e-5 * abs(float(d[f"loss/{k}"])), k
    assert abs(float(total) - float(d["total"])) < 2e-5 * float(d["total"])
    for k, gk in (("sam_pred", "g_sam"), ("dyn_pred", "g_dyn"), ("depth_pred", "g_depth"), ("elev_pred", "g_elev")):
        assert _rel(preds[k].grad, torch.from_numpy(d[gk])) < 5e-5, k


def test_device_label_ops_match_the_reference_loops():
    """csrc/labels.hip against the reference-style host loops kept in loss_utils (remap_labels_in_batch,
    extract_max_per_class + boolean-mask gathers): identical labels, identical picks for the same host RNG state."""
    from creste_public_amd import label_ops
    from creste_public_amd.creste.utils import loss_utils as lu
    g = torch.Generator().manual_seed(3)
    B, H, W = 3, 40, 56
    # sparse label sets per sample; sample 1 has no ignore label (the reference's offset quirk), sample 2 only ignore
    gt = torch.stack([torch.randint(0, 9, (H, W), generator=g) * 7 % 23,
                      torch.randint(1, 6, (H, W), generator=g) * 5,
                      torch.zeros(H, W, dtype=torch.long)])
    ref = lu.remap_labels_in_batch(gt.clone(), ignore_idx=0)
    got, nclass = label_ops.remap_labels_in_batch(gt.cuda(), ignore_idx=0)
    assert torch.equal(got.cpu(), ref)
    assert int(nclass.item()) == int(ref.max()) + 1
    fov = torch.rand(B, H, W, generator=g) > 0.3
    valid = (ref != 0) & fov
    lab_c = ref[valid]
    counts = torch.bincount(lab_c)
    nz = counts[counts.nonzero(as_tuple=True)].float()
    median = min(nz.median().int(), 1000)
    torch.manual_seed(5)
    sel = lu.extract_max_per_class(lab_c, median)
    flat_idx = valid.flatten().nonzero().flatten()[sel]
    torch.manual_seed(5)
    cell, sel_labels = label_ops.sample_cells_per_class(got, fov.cuda(), int(nclass.item()), 0)
    assert torch.equal(cell.cpu().long(), flat_idx)
    assert torch.equal(sel_labels.cpu(), lab_c[sel])
    # rows: gather + scatter against indexing
    Z = 32
    pred = torch.randn(B, Z, H, W, generator=g).cuda().requires_grad_(True)
    rows = label_ops.RowsFn.apply(pred, cell)
    ref_rows = pred.detach().permute(0, 2, 3, 1).reshape(-1, Z)[cell.long()]
    assert torch.equal(rows, ref_rows)
    w = torch.randn(rows.shape, generator=torch.Generator().manual_seed(1)).cuda()
    (rows * w).sum().backward()
    gref = torch.zeros(B * H * W, Z, device="cuda")
    gref[cell.long()] = w
    assert torch.equal(pred.grad, gref.view(B, H, W, Z).permute(0, 3, 1, 2))


@pytest.mark.parametrize("mode", ["sum", "max"])
def test_splat_backward_sum_and_max_modes(mode):
    """scatter_mode 'sum' / 'max' of Camera2MapMulti.splat_soft (reference splat_projection.py:334-344): forward +
    backward of the HIP op against float64 autograd of a torch restatement ('max' = torch_scatter's scatter-max per tap
    folded with torch.maximum against the zero volume).  Features are strictly positive so that no maximum ties with
    the zero initial volume -- the sub-gradient of exact ties is implementation-defined in the reference too."""
    from creste_public_amd import ops
    B, P, F, G = 2, 300, 16, 12
    g = torch.Generator().manual_seed(9)
    X = torch.rand(B, P, generator=g) * (G + 1) - 1.0
    Y = torch.rand(B, P, generator=g) * (G + 1) - 1.0
    xyz = torch.zeros(B, P, 3)
    xyz[..., 1] = -(X * 0.1 - 0.6)          # mx = -y + off_x ; X = mx / vox
    xyz[..., 0] = -(Y * 0.1 - 0.6)
    feats = torch.rand(B, P, F, generator=g) + 0.1
    off, vox = (0.6, 0.6), (0.1, 0.1)
    fa = ops.Act(feats.view(B, 1, P, F).cuda().contiguous(), F)
    coords, bev, dens = ops.bev_splat(xyz.cuda(), fa, off, vox, G, G, 1.0, mode)
    wb = torch.randn(B, G, G, F, generator=g)
    wd = torch.randn(B, G, G, generator=g) * 0.1
    g_feats, g_xyz = ops.bev_splat_bwd(coords, fa, ops.Act(wb.cuda().contiguous(), F), wd.cuda().contiguous(), bev, dens, vox,
                                       1.0, scatter_mode=mode)
    torch.cuda.synchronize()

    # float64 restatement with autograd
    xy = coords.cpu().double().requires_grad_(True)
    f64 = feats.double().requires_grad_(True)
    X0, Y0 = xy[..., 0].floor(), xy[..., 1].floor()
    rX, rY = xy[..., 0] - X0, xy[..., 1] - Y0
    vol = torch.zeros(B, G * G, F, dtype=torch.float64)
    den = torch.zeros(B, G * G, dtype=torch.float64)
    for xd in (0, 1):
        for yd in (0, 1):
            w = (rX if xd else 1 - rX) * (rY if yd else 1 - rY)
            xi, yi = X0.long() + xd, Y0.long() + yd
            ok = (xi >= 0) & (xi < G) & (yi >= 0) & (yi < G)
            idx = (yi * G + xi).clamp(0, G * G - 1)
            wv = w * ok
            den = den.scatter_add(1, idx, wv)
            src = wv.unsqueeze(-1) * f64
            ie = idx.unsqueeze(-1).expand(-1, -1, F)
            if mode == "sum":
                vol = vol.scatter_add(1, ie, src)
            else:
                tap = torch.zeros(B, G * G, F, dtype=torch.float64).scatter_reduce(1, ie, src, reduce="amax", include_self=True)
                vol = torch.maximum(tap, vol)
    ((vol.view(B, G, G, F) * wb.double()).sum() + (den.view(B, G, G) * wd.double()).sum()).backward()
    assert _rel(bev.nchw().permute(0, 2, 3, 1).cpu(), vol.view(B, G, G, F).detach()) < 1e-6
    assert _rel(g_feats.buf.view(B, P, F).cpu(), f64.grad) < 1e-5
    # d/dcoords -> d/dxyz: X = (-y + off)/vox, Y = (-x + off)/vox
    gx_ref = torch.stack([-xy.grad[..., 1] / vox[1], -xy.grad[..., 0] / vox[0], torch.zeros(B, P, dtype=torch.float64)], dim=-1)
    assert _p95(g_xyz.cpu(), gx_ref) < 1e-4


def test_bev_ce_rejects_out_of_range_labels():
    """A label outside [0, C) that is not ignore_index must neither index pred / class_weights out of bounds nor yield
    a silently finite loss (torch.nn.CrossEntropyLoss raises for it): the fused kernel skips the pixel in the
    gradient and returns a NaN loss; with the value declared as ignore_index the loss is the reference's."""
    from creste_public_amd.loss_ops import BevCEFn
    torch.manual_seed(0)
    B, C, H, W = 2, 6, 16, 16
    pred = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    gt = torch.zeros(B, 2, H, W, device="cuda")
    gt[:, 1] = torch.randint(0, C, (B, H, W), device="cuda").float()
    gt[0, 1, 3, 4] = 255.0                                  # "unlabelled" marker, far outside the class range
    gt[1, 1, 0, 0] = -3.0
    fov = torch.ones(B, H, W, dtype=torch.bool, device="cuda")
    cw = torch.rand(C, device="cuda") + 0.5
    loss, stats = BevCEFn.apply(pred, gt, fov, cw, 1, None, 1e-5)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isnan(loss)
    assert torch.isfinite(pred.grad).all() and float(pred.grad[0, :, 3, 4].abs().sum()) == 0.0
    # the same labels with 255 ignored and the negative one made valid: equal to torch's CrossEntropyLoss
    gt[1, 1, 0, 0] = 2.0
    pred2 = pred.detach().clone().requires_grad_(True)
    loss2, _ = BevCEFn.apply(pred2, gt, fov, cw, 1, 255, 1e-5)
    ref = torch.nn.functional.cross_entropy(pred.detach().permute(0, 2, 3, 1).reshape(-1, C), gt[:, 1].long().reshape(-1),
                                            weight=cw, ignore_index=255)
    torch.testing.assert_close(loss2, ref, rtol=1e-5, atol=1e-6)
