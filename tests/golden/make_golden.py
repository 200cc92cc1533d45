#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the REFERENCE's own Python.

Runs only in the build container (needs /root/reference); the fixtures (data: inputs, weights
and the reference's outputs) are committed, this script is committed, the reference source is
not.  Third-party packages the reference imports but that are absent here are replaced by empty
stub modules (recipe: SURVEY.md Appendix B); every stage generated below executes the
reference's hand-written code unmodified.

    python tests/golden/make_golden.py
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- import shims
class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def wrap(o):
    if isinstance(o, dict):
        return AttrDict({k: wrap(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return [wrap(v) for v in o]
    return o


def install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("torch_scatter")
    mod("cv2")
    mod("kornia"); mod("kornia.geometry"); mod("kornia.geometry.transform")
    mod("kornia.losses", focal_loss=None)
    mod("pytorch_lightning")
    tv = mod("torchvision")
    tvt = mod("torchvision.transforms", GaussianBlur=object)
    tvt.__path__ = []
    mod("torchvision.transforms.functional", InterpolationMode=object)
    mod("torchvision.models"); mod("torchvision.models.resnet")
    tv.transforms = tvt
    mod("efficientnet_pytorch", EfficientNet=object, utils=SimpleNamespace())
    oc = SimpleNamespace(create=wrap, to_object=lambda x: x)
    mod("omegaconf", DictConfig=AttrDict, OmegaConf=oc, open_dict=None)
    noop = lambda *a, **k: None
    sys.path.insert(0, REF)
    import creste.utils  # noqa: F401  (real package; then stub its visualization sub-module)
    mod("creste.utils.visualization", numpy_to_pcd=noop, show_bev_map=noop,
        visualize_bev_policy=noop, visualize_bev_label=noop, save_depth_color_image=noop,
        save_depth_image=noop, draw_sparse_depth_on_image=noop)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **out)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


def sd_arrays(module, prefix="sd/"):
    return {prefix + k: v for k, v in module.state_dict().items()}


def randomise_bn(module, gen):
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)


def make_p2p(B, hs, ws, full_w=612.0, tz=0.4):
    """pixel(u*d, v*d, d, 1) at feature resolution -> LiDAR xyz; pinhole + axis swap."""
    s = full_w / ws
    fx = fy = 730.0 * (612.0 / 1216.0) / s * 2.0
    cx, cy = ws / 2.0, hs / 2.0
    kinv = torch.tensor([[1 / fx, 0, -cx / fx, 0], [0, 1 / fy, -cy / fy, 0], [0, 0, 1, 0],
                         [0, 0, 0, 1]], dtype=torch.float32)
    c2l = torch.tensor([[0, 0, 1, 0.1], [-1, 0, 0, 0.05], [0, -1, 0, tz], [0, 0, 0, 1]],
                       dtype=torch.float32)
    return (c2l @ kinv).view(1, 1, 4, 4).repeat(B, 1, 1, 1)


# ----------------------------------------------------------------------------- generators
def gen_splat():
    from creste.models.blocks.splat_projection import Camera2MapMulti
    from creste_public_amd.config import terrainnet_cfg
    cfg = wrap(terrainnet_cfg().to_dict()["camera_projector"])
    g = torch.Generator().manual_seed(1337)
    torch.manual_seed(1337)
    m = Camera2MapMulti(cfg, mode="bilinear")
    randomise_bn(m, g)
    m.eval()
    for tag, (B, hs, ws) in {"small": (2, 16, 19), "wide": (1, 24, 40)}.items():
        depth = torch.rand(B, 1, hs, ws, generator=g) * 16.0 + 0.3
        depth[:, :, :2, :3] = 30.0          # far points -> out of range (masked, x > 12.8)
        depth[:, :, -1, :] = 0.05           # near the sensor
        feats = torch.randn(B, 1, 256, hs, ws, generator=g)
        p2p = make_p2p(B, hs, ws)
        p2p[:, :, :3, 3] += torch.randn(B, 1, 3, generator=g) * 0.05
        with torch.no_grad():
            out = m([depth, feats, p2p])
            xyz, mask, fused = m._prepare_features_and_coords([depth, feats, p2p])
        bf = out["bev_features"]
        dens = out["bev_densities"]
        touched = (dens[:, 0] != 0) | (bf != 0).any(dim=1)
        idx = touched.nonzero()
        vals = bf.permute(0, 2, 3, 1)[touched]
        arrs = dict(depth=depth, feats=feats, p2p=p2p, bev_coords=out["bev_coords"],
                    bev_densities=dens, touched_idx=idx, touched_feats=vals,
                    bev_features_abs_sum=bf.abs().sum(), xyz=xyz, mask=mask, fused=fused,
                    grid_size=m.grid_size)
        if tag == "small":
            arrs.update(sd_arrays(m))
        npz(f"splat_{tag}.npz", **arrs)


def gen_splat_modes():
    """Camera2MapMulti with scatter_mode 'sum' (splat_projection.py:334-352), several frames per batch element.
    Not generated: 'max' needs torch_scatter (absent here); num_cams > 1 raises inside the reference itself
    (`.view` of a permuted tensor at :228), so the multi-camera concatenation has no reference output to pin."""
    from creste.models.blocks.splat_projection import Camera2MapMulti
    from creste_public_amd.config import terrainnet_cfg
    base = terrainnet_cfg().to_dict()["camera_projector"]
    g = torch.Generator().manual_seed(4242)
    for tag, (nc, smode, B, N, hs, ws) in {"onecam_sum": (1, "sum", 2, 2, 14, 18)}.items():
        cfgd = dict(base); cfgd["num_cams"] = nc
        torch.manual_seed(7)
        m = Camera2MapMulti(wrap(cfgd), mode="bilinear", scatter_mode=smode)
        randomise_bn(m, g)
        m.eval()
        depth = torch.rand(B, N, hs, ws, generator=g) * 14.0 + 0.3
        depth[:, :, :1, :2] = 30.0
        feats = torch.randn(B, N, 256, hs, ws, generator=g)
        p2p = make_p2p(B, hs, ws).repeat(1, N, 1, 1)
        p2p[:, :, :3, 3] += torch.randn(B, N, 3, generator=g) * 0.3      # every view its own extrinsic
        with torch.no_grad():
            out = m([depth, feats, p2p])
        bf, dens = out["bev_features"], out["bev_densities"]
        touched = (dens[:, 0] != 0) | (bf != 0).any(dim=1)
        npz(f"splat_{tag}.npz", depth=depth, feats=feats, p2p=p2p, bev_coords=out["bev_coords"], bev_densities=dens,
            touched_idx=touched.nonzero(), touched_feats=bf.permute(0, 2, 3, 1)[touched],
            bev_features_abs_sum=bf.abs().sum(), num_cams=torch.tensor(nc), **sd_arrays(m))


def analytic_cotangent(shape, phase):
    """a fixed smooth [B,C,H,W] cotangent (float64 sine pattern rounded to fp32) -- cheap to rebuild in the tests"""
    B, C, H, W = shape
    b, c, y, x = np.meshgrid(np.arange(B), np.arange(C), np.arange(H), np.arange(W), indexing="ij")
    return torch.from_numpy(np.sin(0.37 * c + 0.11 * y + 0.05 * x + 1.3 * b + phase).astype(np.float32))


def gen_splat_mv():
    """Camera2MapMulti in TRAINING mode with the immovable-object mask as 4th input (splat_projection.py:214-219:
    `_mv` keys, mask multiplied into the range mask, BatchNorm on batch statistics) + the gradients of a fixed linear
    functional of the `_mv` features w.r.t. depth and features."""
    from creste.models.blocks.splat_projection import Camera2MapMulti
    from creste_public_amd.config import terrainnet_cfg
    cfgd = dict(terrainnet_cfg().to_dict()["camera_projector"]); cfgd["num_cams"] = 1
    g = torch.Generator().manual_seed(777)
    torch.manual_seed(11)
    m = Camera2MapMulti(wrap(cfgd), mode="bilinear")
    randomise_bn(m, g)
    sd0 = sd_arrays(m)
    sd0 = {k: v.clone() for k, v in sd0.items()}
    m.train()
    B, hs, ws = 2, 14, 20
    depth = (torch.rand(B, 1, hs, ws, generator=g) * 14.0 + 0.3).requires_grad_(True)
    with torch.no_grad():
        depth[:, :, :1, :3] = 30.0
    feats = torch.randn(B, 1, 256, hs, ws, generator=g).requires_grad_(True)
    p2p = make_p2p(B, hs, ws)
    p2p[:, :, :3, 3] += torch.randn(B, 1, 3, generator=g) * 0.05
    mv = (torch.rand(B, 1, hs, ws, generator=g) > 0.3).float()
    out = m([depth, feats, p2p, mv])
    assert set(out) == {"bev_features_mv", "bev_densities_mv", "bev_coords_mv"}
    bf, dens = out["bev_features_mv"], out["bev_densities_mv"]
    R, Rd = analytic_cotangent(bf.shape, 0.0), analytic_cotangent(dens.shape, 1.0)     # not stored: tests rebuild them
    ((bf * R).sum() + (dens * Rd).sum()).backward()
    touched = (dens[:, 0] != 0) | (bf != 0).any(dim=1)
    after = {"after/" + k: v for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
    npz("splat_mv.npz", depth=depth, feats=feats, p2p=p2p, mv_mask=mv, bev_coords=out["bev_coords_mv"],
        bev_densities=dens, touched_idx=touched.nonzero(), touched_feats=bf.permute(0, 2, 3, 1)[touched],
        bev_features_abs_sum=bf.abs().sum(), g_depth=depth.grad, g_feats=feats.grad,
        g_fuse_w=m.vision_fusion.convs[0].weight.grad, g_z0_w=m.z_proj[0].weight.grad, **sd0, **after)


def gen_vin_svf_loss():
    import creste.models.blocks.vin as vin_mod
    from creste.models.blocks.vin import VIN
    import creste.models.lfd as lfd
    import creste.utils.train_utils as tu
    from creste.utils.loss_utils import LossManager
    from creste_public_amd.config import maxent_irl_cfg
    cfgd = maxent_irl_cfg().to_dict()
    nk = cfgd["traversability_head"]["net_kwargs"]
    torch.manual_seed(7)
    g = torch.Generator().manual_seed(7)
    vin = VIN(wrap(nk["reward_cfg"]), wrap(nk["qvalue_cfg"]))
    randomise_bn(vin, g)
    vin.eval()
    npz("vin_w.npz", w=vin.w)

    calls = {"n": 0}
    real_conv = vin_mod.F.conv2d

    def counting_conv(*a, **k):
        calls["n"] += 1
        return real_conv(*a, **k)

    # -- value iteration
    vi = {}
    for tag, shape in {"a": (3, 1, 16, 32), "b": (2, 1, 64, 128)}.items():
        r = torch.rand(shape, generator=g)
        if tag == "a":
            r[0, 0, 4:9, 10:20] = 0.0       # an obstacle block of zero reward
            r[2] *= 0.25
        calls["n"] = 0
        vin_mod.F.conv2d = counting_conv
        try:
            with torch.no_grad():
                v, pol, q = vin.value_iteration_manual(r, None, threshold=0.001, discount=0.99)
        finally:
            vin_mod.F.conv2d = real_conv
        vi[tag] = (r, v, pol, q)
        npz(f"vi_{tag}.npz", r=r, v=v, policy=pol, q=q, sweeps=calls["n"] - 1, discount=0.99,
            threshold=0.001)

    # -- reward net + VIN.forward (solve_mdp False and True) on a small BEV dict
    fm = {"inpainting_sam_preds": torch.randn(2, 32, 32, 64, generator=g),
          "inpainting_sam_dynamic_preds": torch.randn(2, 6, 32, 64, generator=g),
          "elevation_preds": torch.randn(2, 2, 32, 64, generator=g)}
    S = torch.zeros(2, 50, 2, dtype=torch.long)
    out = vin(fm, S, solve_mdp=True)
    npz("vin_forward.npz", **{f"in/{k}": v for k, v in fm.items()},
        traversability_preds=out["traversability_preds"],
        traversability_preds_full=out["traversability_preds_full"],
        input_view=out["input_view"], policy=out["policy"], q_estimate=out["q_estimate"],
        value_estimate=out["value_estimate"], **sd_arrays(vin))
    vin.train()
    out_tr = vin(fm, S, solve_mdp=False)
    npz("vin_forward_train.npz", traversability_preds=out_tr["traversability_preds"],
        **{f"sd_after/{k}": v for k, v in vin.state_dict().items() if "running" in k})
    vin.eval()

    # -- SVF (MaxEntIRL.expected_state_visitation_frequency, unbound)
    H, W, T = 64, 128, 50
    fov = tu.create_trapezoidal_fov_mask(H * 2, W, 70, 70, 0, 100).view(1, 1, H * 2, W)[:, :, :H, :W]
    tp = torch.zeros(8, 1, 3, 3)
    center = [[2, 2], [2, 1], [2, 0], [1, 2], [1, 0], [0, 2], [0, 1], [0, 0]]
    for i in range(8):
        tp[i, :, center[i][0], center[i][1]] = 1.0
    dyn = torch.tensor([[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 1], [1, -1], [1, 0], [1, 1]])
    _, _, pol_b, _ = vi["b"]
    B = pol_b.shape[0]

    def pose_batch(xy):  # [B,T,2] -> [B,T,3,3]
        P = torch.eye(3).repeat(xy.shape[0], xy.shape[1], 1, 1)
        P[:, :, :2, 2] = xy
        return P

    tt = torch.linspace(0, 1, T).view(1, T, 1)
    # full-res (256 grid) coordinates; ds=2 -> grid 64x128 rows 0..63
    e0 = torch.tensor([[120.0, 128.0]]) + tt * torch.tensor([[-100.0, 30.0]])     # in fov, forward
    e1 = torch.tensor([[126.5, 10.0]]) + tt * torch.tensor([[0.0, 20.0]])         # never in fov
    experts = pose_batch(torch.cat([e0, e1], dim=0))
    for ztag, zts in {"": False, "_zts": True}.items():
        ns = SimpleNamespace(
            traversability_head_cfg=wrap(cfgd["traversability_head"]), fov_mask=fov,
            action_horizon=T, policy_cfg=wrap(cfgd["policy_kwargs"]), zero_terminal_state=zts,
            transition_probs=tp, dynamics=dyn, map_size=[H, W])
        ns._state_to_coord = lambda s, vectorized=False: lfd.MaxEntIRL._state_to_coord(ns, s, vectorized)
        ns._coord_to_state = lambda c, vectorized=False: lfd.MaxEntIRL._coord_to_state(ns, c, vectorized)
        with torch.no_grad():
            o = lfd.MaxEntIRL.expected_state_visitation_frequency(ns, pol_b.clone(), experts.clone())
        # policy input = vi_b.npz["policy"] (not stored twice)
        npz(f"svf{ztag}.npz", expert=experts, fov_mask=fov, exp_svf=o["exp_svf"],
            state_preds=o["state_preds"], state_preds_grid=o["state_preds_grid"],
            transition_probs=tp)
        if not zts:
            svf_out = o

    # -- MaxEntIRLLoss through LossManager (with and without counterfactuals)
    lcfg = wrap({"loss": cfgd["loss"]})
    lm = LossManager(lcfg)
    fov256 = tu.create_trapezoidal_fov_mask(256, 256, 70, 70, 0, 200).unsqueeze(0).repeat(B, 1, 1)
    vin.zero_grad()
    feat40 = torch.randn(B, 40, 64, 128, generator=g, requires_grad=True)
    r = vin.r(feat40)
    rng = np.random.RandomState(3)
    cf = [dict(trajectories=(np.array([[100.0, 128.0]]) +
                             np.linspace(0, 1, 20)[None, :, None] *
                             rng.uniform(-80, 80, size=(3, 1, 2))).astype(np.float32),
               rank=np.array([0, 1, 2])), None]
    for tag, cfl in {"cf": cf, "nocf": [None, None]}.items():
        td = {"outputs/exp_svf": svf_out["exp_svf"].clone(), "inputs/traversability_label": experts,
              "inputs/fov_mask": fov256, "outputs/traversability_preds": r,
              "outputs/input_view": feat40, "inputs/counterfactuals_label": cfl, "task": "x"}
        ld, md = lm(td)
        (wgt, val), = ld.values()
        vin.zero_grad()
        (wgt * val).backward(retain_graph=True)
        grads = {f"grad/{k}": p.grad for k, p in vin.r.named_parameters()}
        # inputs shared by both cases live in irl_loss_inputs.npz; exp_svf = svf.npz["exp_svf"]
        npz(f"irl_loss_{tag}.npz", loss_key=list(ld.keys())[0], loss_weight=wgt,
            loss=val, **{f"meta/{k}": v for k, v in md.items()}, **grads)
    npz("irl_loss_inputs.npz", expert=experts, fov_mask=fov256, input_view=feat40, reward=r,
        cf_traj=cf[0]["trajectories"], cf_rank=cf[0]["rank"], **sd_arrays(vin.r, "sd_r/"))
    pts, cnt = lm.losses[0].compute_expert_visitation(experts, 2, [64, 128])
    npz("expert_raster.npz", expert=experts, points=pts, counts=cnt)


def gen_policy_fc():
    """MaxEntIRL.iterative_policy_rollout (policy_method 'fc', lfd.py:279-312) on a seeded Q map."""
    import torch.nn as nn
    import creste.models.lfd as lfd
    g = torch.Generator().manual_seed(2024)
    B, lq, H, W, T = 3, 10, 24, 40, 50
    q = torch.randn(B, lq, H, W, generator=g)
    fc = nn.Linear(lq, 8, bias=False)
    with torch.no_grad():
        fc.weight.copy_(torch.randn(8, lq, generator=g))
    tt = torch.linspace(0, 1, T).view(1, T, 1)
    S = (torch.tensor([[[20.0, 20.0]], [[1.0, 2.0]], [[23.0, 39.0]]]) +
         tt * torch.tensor([[[-18.0, 15.0]], [[0.0, 36.0]], [[-6.0, -38.0]]])).long()
    S[:, :, 0].clamp_(0, H - 1)
    S[:, :, 1].clamp_(0, W - 1)
    dyn = torch.tensor([[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 1], [1, -1], [1, 0], [1, 1]], dtype=torch.long)
    ns = SimpleNamespace(fc=fc, sm=nn.Softmax(dim=1), dynamics=dyn)
    with torch.no_grad():
        o = lfd.MaxEntIRL.iterative_policy_rollout(ns, q, S, T)
    npz("policy_fc.npz", q=q, S=S, fc_weight=fc.weight.detach(), policy_fc=o["policy_fc"], state_preds=o["state_preds"])


def gen_blocks_and_utils():
    from creste.models.blocks.conv import MultiLayerConv, ConvEncoder, MultiScaleFCN
    from creste.models.blocks.effnet import Up
    from creste.models.blocks.inpainting import DeconvHead
    import creste.utils.depth_utils as du
    import creste.utils.train_utils as tu
    from creste_public_amd.config import terrainnet_cfg, maxent_irl_cfg
    g = torch.Generator().manual_seed(11)
    torch.manual_seed(11)
    arrs = {}

    def run(tag, m, *xs):
        randomise_bn(m, g)
        m.eval()
        with torch.no_grad():
            y = m(*xs)
        ys = y if isinstance(y, tuple) else (y,)
        for i, x in enumerate(xs):
            arrs[f"{tag}/in{i}"] = x
        for i, yy in enumerate(ys):
            arrs[f"{tag}/out{i}"] = yy
        arrs.update(sd_arrays(m, f"{tag}/sd/"))

    run("mlc", MultiLayerConv(wrap(dict(dims=[8, 12, 6], kernels=[3, 1], paddings=[1, 0],
                                        norm_type="batch_norm"))),
        torch.randn(2, 8, 9, 11, generator=g))
    run("enc", ConvEncoder(wrap(dict(dims=[10, 6], kernels=[1], paddings=[0],
                                     norm_type="batch_norm"))),
        torch.randn(2, 10, 5, 7, generator=g))
    run("up2", Up(8 + 4, 12, scale_factor=2), torch.randn(1, 8, 6, 7, generator=g),
        torch.randn(1, 4, 12, 14, generator=g))
    # the odd-width decoder step of the 512x612 config: 64x76 -> 128x153 (effnet.py:64-68)
    sf = (128 / 64, 153 / 76)
    run("upodd", Up(3 + 1, 2, scale_factor=sf), torch.randn(1, 3, 64, 76, generator=g),
        torch.randn(1, 1, 128, 153, generator=g))
    import torch.nn as nn
    run("deconv", DeconvHead(16 + 8, 5, nn.BatchNorm2d), torch.randn(1, 16, 4, 4, generator=g),
        torch.randn(1, 8, 16, 16, generator=g))
    nk = maxent_irl_cfg().to_dict()["traversability_head"]["net_kwargs"]
    run("msfcn", MultiScaleFCN(wrap(nk["reward_cfg"]["net_kwargs"])),
        torch.randn(2, 40, 16, 24, generator=g))
    npz("blocks.npz", **arrs)

    d = terrainnet_cfg().to_dict()["discretize"]
    logits = torch.relu(torch.randn(2, 128, 6, 7, generator=g) * 3)
    depth = du.convert_to_metric_depth_differentiable(logits, d["mode"], d["depth_min"],
                                                      d["depth_max"], d["num_bins"])
    dm = torch.rand(4, 9, generator=g) * 30000
    npz("utils.npz", logits=logits, metric_depth_mm=depth,
        depth_map=dm, bins=du.bin_depths(dm.clone(), "UD", 300, 25600, 128),
        bins_target=du.bin_depths(dm.clone(), "UD", 300, 25600, 128, target=True),
        fov_128=tu.create_trapezoidal_fov_mask(128, 128, 70, 70, 0, 100),
        fov_default=tu.create_trapezoidal_fov_mask(40, 60),
        rc_in=torch.arange(2 * 1 * 8 * 8).view(2, 1, 8, 8).byte(),
        rc_out=tu.resize_and_crop(torch.arange(2 * 1 * 8 * 8).view(2, 1, 8, 8).byte(), (4, 4),
                                  (0, 2, 0, 4)))


def gen_distill_losses():
    """CrossEntropyDepth / SmoothL1Depth / MSELoss of the stage-1 distillation config, run through the reference's
    own LossManager (loss_utils.py:63-91, 477-573, 606-647): losses, metrics and the gradients w.r.t. the
    predictions."""
    import creste.utils.loss_utils as lu
    g = torch.Generator().manual_seed(23)
    B, Hs, Ws, Z = 3, 10, 13, 32
    disc = dict(mode="UD", num_bins=128, depth_min=300, depth_max=25600)
    cfg = wrap(dict(loss=[
        dict(name="CrossEntropyDepth", weight=0.5, pred_key="outputs/depth_preds_logits",
             lab_key="inputs/depth_label", discretize=disc),
        dict(name="SmoothL1Depth", weight=0.1, pred_key="outputs/depth_preds_bins", lab_key="inputs/depth_label",
             beta=0.5, discretize=disc),
        dict(name="MSELoss", weight=1.0, pred_key="outputs/dino_pe_feats", lab_key="inputs/fimg_label",
             overlap_only=False)]))
    lm = lu.LossManager(cfg)
    logits = (torch.randn(B, 128, Hs, Ws, generator=g) * 3).requires_grad_(True)
    depth = torch.rand(B, 1, Hs, Ws, generator=g) * 30000.0 - 1500.0
    depth[0, 0, 0, :4] = float("nan")
    depth[1, 0, 2, 3] = 25600.0
    depth[2, 0, 5, 5] = 300.0
    feats = torch.randn(B, 1, Z, Hs, Ws, generator=g).requires_grad_(True)
    label = torch.randn(B, 1, Z, Hs, Ws, generator=g)
    label[0, 0, :, 1, 2] = float("inf")
    label[2, 0, 3, 4, 4] = float("-inf")
    bins = logits.detach().argmax(1)
    td = {"outputs/depth_preds_logits": logits, "outputs/depth_preds_bins": bins, "outputs/dino_pe_feats": feats,
          "inputs/depth_label": depth.clone(), "inputs/fimg_label": label.clone(), "task": None}
    with torch.enable_grad():
        ld, meta = lm(td)
        total = sum(w * v for w, v in ld.values())
        total.backward()
    out = dict(logits=logits.detach(), depth_label=depth, feats=feats.detach(), fimg_label=label, pred_bins=bins,
               total=total.detach(), g_logits=logits.grad, g_feats=feats.grad)
    for k, (w, v) in ld.items():
        out[f"loss/{k}"] = v.detach()
        out[f"weight/{k}"] = torch.tensor(float(w))
    for k, v in meta.items():
        out[f"meta/{k}"] = v.detach()
    npz("distill_losses.npz", **out)


def gen_ssc_losses():
    """SupPixelConLoss (+MultiPosConLoss), CrossEntropy (class_dim=1, class weights), SmoothL1 (relative elevation),
    SmoothL1Depth on metric depth -- through the reference's own LossManager (loss_utils.py:203-286, 379-474,
    530-603; models/losses/supcon_loss.py:56-115), in a 1-rank gloo group because MultiPosConLoss all-gathers."""
    import tempfile
    import torch.distributed as dist
    import creste.utils.loss_utils as lu
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"file://{tempfile.mkdtemp()}/pg", rank=0, world_size=1)
    g = torch.Generator().manual_seed(41)
    B, G, Z = 2, 24, 8
    freq = np.array([0.5, 0.2, 0.1, 0.1, 0.05, 0.05])
    wfile = os.path.join(tempfile.mkdtemp(), "w6.txt")
    np.savetxt(wfile, freq)
    disc = dict(mode="UD", num_bins=128, depth_min=300, depth_max=25600)
    cfg = wrap(dict(loss=[
        dict(name="SupPixelConLoss", views=1, weight=1.0, pred_key="outputs/inpainting_sam_preds",
             lab_key="inputs/3d_sam_label", ignore_index=0, temperature=0.1, task="joint", contrast_mode="batch_all"),
        dict(name="CrossEntropy", weight=2.0, pred_key="outputs/inpainting_sam_dynamic_preds",
             lab_key="inputs/3d_sam_dynamic_label", num_class=6, class_weights=wfile, class_dim=1, task="joint"),
        dict(name="SmoothL1Depth", weight=0.1, pred_key="outputs/depth_preds_metric", lab_key="inputs/depth_label",
             beta=0.5, discretize=disc),
        dict(name="SmoothL1", weight=3.0, beta=0.2, pred_key="outputs/elevation_preds", lab_key="inputs/elevation_label",
             absolute=False, task="joint")]))
    lm = lu.LossManager(cfg)
    sam_pred = torch.randn(B, Z, G, G, generator=g).requires_grad_(True)
    sam_label = torch.randint(0, 5, (B, 1, G, G), generator=g)              # 0 = ignore; remapped per sample
    dyn_pred = torch.randn(B, 6, G, G, generator=g).requires_grad_(True)
    dyn_label = torch.stack([torch.zeros(B, G, G), torch.randint(0, 6, (B, G, G), generator=g).float()], dim=1)
    fov = torch.rand(B, G, G, generator=g) > 0.3
    Hs, Ws = 7, 9
    depth_pred = (torch.rand(B, Hs, Ws, generator=g) * 20.0).requires_grad_(True)
    depth_label = torch.rand(B, 1, Hs, Ws, generator=g) * 30000.0 - 1500.0
    elev_pred = torch.randn(B, 2, G, G, generator=g).requires_grad_(True)
    elev_label = torch.randn(B, 2, G, G, generator=g)
    elev_label[0, :, 2, 3] = float("nan")
    elev_label[1, 1, 4, 4] = float("inf")
    td = {"outputs/inpainting_sam_preds": sam_pred, "inputs/3d_sam_label": sam_label.clone(),
          "outputs/inpainting_sam_dynamic_preds": dyn_pred, "inputs/3d_sam_dynamic_label": dyn_label.clone(),
          "inputs/fov_mask": fov, "outputs/depth_preds_metric": depth_pred, "inputs/depth_label": depth_label.clone(),
          "outputs/elevation_preds": elev_pred, "inputs/elevation_label": elev_label.clone(), "task": "joint"}
    torch.manual_seed(77)                                                     # extract_max_per_class draws randperm
    with torch.enable_grad():
        ld, meta = lm(td)
        total = sum(w * v for w, v in ld.values())
        total.backward()
    out = dict(sam_pred=sam_pred.detach(), sam_label=sam_label, dyn_pred=dyn_pred.detach(), dyn_label=dyn_label,
               fov=fov, depth_pred=depth_pred.detach(), depth_label=depth_label, elev_pred=elev_pred.detach(),
               elev_label=elev_label, class_freq=torch.from_numpy(freq), total=total.detach(),
               g_sam=sam_pred.grad, g_dyn=dyn_pred.grad, g_depth=depth_pred.grad, g_elev=elev_pred.grad)
    for k, (w, v) in ld.items():
        out[f"loss/{k}"] = v.detach()
        out[f"weight/{k}"] = torch.tensor(float(w))
    for k, v in meta.items():
        out[f"meta/{k}"] = v.detach()
    npz("ssc_losses.npz", **out)


def gen_projection():
    """projection.npz: the reference's `get_pixel2pts_transform` / `get_pts2pixel_transform`
    (creste/utils/projection.py:11-61) and `pixels_to_depth` (:64-155) run on two synthetic scans.

    `pixels_to_depth` calls `torch_scatter.scatter(src, index, dim=0, dim_size=H*W, reduce=...)` (:124-128); torch_scatter
    is absent from this image, so IN THIS GENERATOR ONLY that one call is backed by
    `torch.zeros(dim_size).scatter_reduce(0, index, src, 'amax'|'amin', include_self=False)` -- torch_scatter's documented
    semantics (reduce over equal indices, slots no index names hold 0).  Every other line executed is the reference's.

    Scan a: 128-beam scan, every 21st point, + an intensity column (the function slices [:, :3]); points behind the
    camera (z_cam < 0), on the image plane's horizon (z_cam == 0 -> inf / nan pixel), left / right / above / below the
    image; 32x64 image so that most pixels collect several returns (duplicate-pixel reduction, last-write-wins 'depth').
    Scan b: float32 points incl. z_cam ~ 1e-30 (|u|, |v| beyond the int32 range -> np.clip), rectifying rotation R != I,
    lidar2camrect given as a torch tensor (the function's own .cpu().numpy() branch)."""
    import creste.utils.projection as rp
    import torch_scatter

    def scatter(src, index, dim=0, dim_size=None, reduce="max"):
        assert dim == 0 and reduce in ("max", "min")
        out = torch.zeros(dim_size, dtype=src.dtype)
        return out.scatter_reduce(0, index, src, "amax" if reduce == "max" else "amin", include_self=False)

    torch_scatter.scatter = scatter
    from creste_public_amd import synth
    out = {}
    keys = ["image_pts", "image_depth", "depth", "pc_pts", "pc_mask"]
    for tag, (H, W, step, seed) in {"a": (32, 64, 7, 11), "b": (40, 64, 13, 12)}.items():
        g = torch.Generator().manual_seed(seed)
        pts = synth.lidar_scan(1, g)[0, ::step].numpy().astype(np.float64)
        K, T = synth.camera_matrices(H, W)
        K, T = K.numpy().astype(np.float64), T.numpy().astype(np.float64)       # T = cam -> lidar
        lidar2cam = np.linalg.inv(T)
        if tag == "a":
            R = np.eye(3)
            pts = np.hstack([pts, np.linspace(0, 1, pts.shape[0]).reshape(-1, 1)])      # intensity column
        else:
            a, b = 0.02, -0.015                                                      # small rectifying rotation
            Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
            Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
            R = Rx @ Ry
            pts = pts.astype(np.float32)
        P = np.hstack([K, np.array([[0.3], [-0.2], [0.0]])])
        calib = dict(lidar2cam=lidar2cam, R=R, P=P)
        p2p = rp.get_pixel2pts_transform(calib)
        pts2pix = rp.get_pts2pixel_transform(calib)
        l2cr = pts2pix.copy()
        # hand-placed cases, in camera coordinates (x right, y down, z forward) mapped back to the LiDAR frame
        back = np.linalg.inv(pts2pix)

        def from_pix(u, v, z):
            return (back @ np.array([u * z, v * z, z, 1.0]))[:3]
        special = [from_pix(5.5, 7.5, -4.0), from_pix(5.5, 7.5, 0.0), from_pix(-0.5, 3.0, 6.0), from_pix(W + 0.5, 3.0, 6.0),
                   from_pix(3.0, -0.25, 6.0), from_pix(3.0, H + 2.0, 6.0), from_pix(W - 0.001, H - 0.001, 9.0),
                   from_pix(0.0, 0.0, 2.0), from_pix(10.2, 11.7, 3.0), from_pix(10.9, 11.1, 8.0), from_pix(10.5, 11.5, 5.0)]
        if tag == "b":
            special += [from_pix(3.0e12, -2.0e12, 1.0), np.array(from_pix(1.0, 1.0, 1e-30)) + np.array([0.0, 1e-3, 0.0])]
        sp = np.zeros((len(special), pts.shape[1]), dtype=pts.dtype)
        sp[:, :3] = np.asarray(special)
        pts = np.vstack([pts[:pts.shape[0] // 2], sp, pts[pts.shape[0] // 2:]])
        out[f"{tag}/points"] = pts
        out[f"{tag}/lidar2cam"], out[f"{tag}/R"], out[f"{tag}/P"] = lidar2cam, R, P
        out[f"{tag}/lidar2camrect"] = l2cr
        out[f"{tag}/hw"] = np.array([H, W])
        out[f"{tag}/p2p"], out[f"{tag}/pts2pix"] = p2p, pts2pix
        for prio in ("max", "min"):
            cal = {"lidar2camrect": torch.from_numpy(l2cr) if tag == "b" else l2cr}
            with np.errstate(all="ignore"):
                vals = rp.pixels_to_depth(torch.from_numpy(pts) if tag == "b" else pts, cal, H, W, return_keys=keys,
                                          depth_priority=prio)
            for k, v in zip(keys, vals):
                out[f"{tag}/{prio}/{k}"] = v
        n_valid = int(out[f"{tag}/max/pc_mask"].sum())
        n_pix = out[f"{tag}/max/image_depth"].shape[0]
        print(f"  projection {tag}: {pts.shape[0]} points, {n_valid} in view, {n_pix} pixels hit "
              f"({n_valid / max(n_pix, 1):.1f} returns per pixel)")
    npz("projection.npz", **out)


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference tree not mounted: fixtures can only be made in the build container"
    sys.path.insert(0, os.path.abspath(os.path.join(OUT, "..", "..")))
    install_shims()
    gens = [gen_splat, gen_splat_modes, gen_splat_mv, gen_vin_svf_loss, gen_policy_fc, gen_blocks_and_utils,
            gen_distill_losses, gen_ssc_losses, gen_projection]
    only = set(sys.argv[1:])            # e.g. `make_golden.py gen_projection`; no arguments = everything
    for gen in gens:
        if not only or gen.__name__ in only:
            gen()
