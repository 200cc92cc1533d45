"""Oracle (TEST INFRASTRUCTURE) -- CPU/PyTorch restatement of the perception half of the path:
RGB-D encoder -> depth bins -> metric depth -> camera->BEV bilinear splat -> BEV heads.

Follows /root/reference/creste/models/{vision_encoder.py:11-49, depth.py:17-158,
distillation.py:19-207, terrainnet.py:24-350}, /root/reference/creste/models/blocks/
splat_projection.py:12-354 and /root/reference/creste/utils/depth_utils.py:300-313.
Restated: the configuration the shipped YAMLs select (single view, no temporal layer, no multiview
distillation, scatter_mode 'mean') plus the scatter modes 'sum' / 'max' and the training-only `use_movability` double
splat (terrainnet.py:310-344); anything else raises NotImplementedError.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .blocks import (ConvEncoder, EffNet, InpaintingResNet18MultiHead, MultiLayerConv, _get)


# ------------------------------------------------------------------------------ depth
def metric_depth_from_logits(logits, depth_min, depth_max, num_bins):
    """softmax-expectation over uniformly spaced bin values, in the unit of depth_min/max
    (depth_utils.py:300-313)."""
    p = F.softmax(logits, dim=1)
    bins = torch.linspace(depth_min, depth_max, num_bins, device=logits.device).view(1, -1, 1, 1)
    return torch.sum(p * bins, dim=1)


class VisionEncoder(nn.Module):
    def __init__(self, vcfg):
        super().__init__()
        self.input_type = vcfg["input_type"]
        if self.input_type not in ("rgb", "rgbd"):
            raise NotImplementedError(f"Input type {self.input_type} not supported")
        e = vcfg["effnet_cfgs"]
        self.model = EffNet(name=vcfg["name"], inC=e["in_channels"], outC=e["out_channels"],
                            image_size=e["image_size"], downsample=e["downsample"],
                            return_2nd_last_layer_output=False)

    def forward(self, img):
        if self.input_type == "rgb":
            img = img[:, :3]
        return self.model(img)


class DepthCompletion(nn.Module):
    """encoder -> depth_head logits -> (metres, argmax bins) (depth.py:102-133, :61-100)."""

    def __init__(self, cfg):
        super().__init__()
        self.disc = cfg["discretize"]
        self.return_feats = cfg["vision_backbone"]["return_feats"]
        self.vision_backbone = VisionEncoder(cfg["vision_backbone"])
        self.depth_head = MultiLayerConv(cfg["depth_head"])

    def forward(self, x):
        feats = self.vision_backbone(x)
        logits = self.depth_head(feats)
        d = self.disc
        out = {"depth_preds_logits": logits,
               "depth_preds_metric": metric_depth_from_logits(
                   logits, d["depth_min"], d["depth_max"], d["num_bins"]) / 1000,
               "depth_preds_bins": logits.argmax(dim=1)}
        if self.return_feats:
            out["depth_preds_feats"] = feats
        return out


class DistillationBackbone(nn.Module):
    """DepthCompletion + DINO feature head (distillation.py:145-207, single view, no PE map)."""

    def __init__(self, cfg):
        super().__init__()
        if _get(cfg, "multiview_distillation", False) or _get(cfg, "pe_map", None) is not None:
            raise NotImplementedError("oracle restates the shipped single-view config only")
        self.depthcomp = DepthCompletion(cfg)
        self.dino_head = MultiLayerConv(cfg["distillation_head"]["feature_head"])

    def forward(self, rgbd):
        B, V, C, H, W = rgbd.shape
        out = dict(self.depthcomp(rgbd.view(B * V, C, H, W)))
        feats = out["depth_preds_feats"]
        _, Z, Hs, Ws = feats.shape
        dino = self.dino_head(feats)
        out["dino_pe_feats"] = dino.view(B, 1, dino.shape[1], Hs, Ws)   # V forced to 1 (:172)
        return out


# ------------------------------------------------------------------------------ splat
class Camera2MapMulti(nn.Module):
    """pixel*depth -> LiDAR xyz -> z-MLP + 1x1 fuse -> range mask -> 4-tap bilinear scatter-add
    with mean normalisation (splat_projection.py:53-354)."""

    def __init__(self, cfg, scatter_mode="mean"):
        super().__init__()
        self.scatter_mode = scatter_mode
        pcr = torch.tensor(cfg["point_cloud_range"])
        self.register_buffer("point_cloud_range", pcr)
        self.register_buffer("max_bound", pcr[3:].reshape(1, -1))
        self.register_buffer("min_bound", pcr[:3].reshape(1, -1))
        self.register_buffer("voxel_size", torch.tensor(cfg["voxel_size"]))
        self.register_buffer("grid_size", ((pcr[3:] - pcr[:3]) / self.voxel_size).long())
        mb = self.min_bound
        self.register_buffer("lidar2map", torch.tensor(
            [[0, -1, 0, -mb[0, 0]], [-1, 0, 0, -mb[0, 1]], [0, 0, -1, -mb[0, 2]],
             [0, 0, 0, 1]]).float())
        self.min_weight = 1.0
        self.NC = _get(cfg, "num_cams", 2)
        if cfg["z_embed_mode"] != "mlp":
            raise Exception("Unknown z_embed_mode:", cfg["z_embed_mode"])
        zd = cfg["z_embed_dim"]
        self.z_proj = nn.Sequential(nn.Linear(1, zd * 2), nn.ReLU(), nn.Linear(zd * 2, zd),
                                    nn.ReLU())
        self.vision_fusion = ConvEncoder(cfg["vision_fusion"])

    @staticmethod
    def pixels_to_lidar(depth, p2p):
        """[B,N,H,W] metres x [B,N,4,4] -> xyz [B,N,3,H,W]; c=[u*d, v*d, d, 1], xyz=P@c (:19-51)."""
        B, N, H, W = depth.shape
        d = depth.reshape(B * N, 1, H, W)
        u, v = torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy")
        uv1 = torch.stack([u, v, torch.ones_like(u)], 0).unsqueeze(0).to(d.device)
        c = torch.cat([uv1 * d, torch.ones_like(d)], dim=1)
        xyz = torch.bmm(p2p.reshape(B * N, 4, 4), c.flatten(start_dim=2))
        return xyz.view(B, N, 4, H, W)[:, :, :3]

    def fuse(self, depth, feats, p2p):
        """(:131-173) -> xyz [B,N,3,H,W], mask [B,N,1,H,W] bool, fused feats [B,N,C,H,W]."""
        B, N, Fd, H, W = feats.shape
        xyz = self.pixels_to_lidar(depth, p2p)
        z = xyz[:, :, 2].reshape(B * N * H * W, 1)
        zf = self.z_proj(z).view(B, N, H, W, -1).permute(0, 1, 4, 2, 3)
        f = self.vision_fusion(torch.cat([feats, zf], dim=2).view(B * N, -1, H, W))
        f = f.view(B, N, f.shape[1], H, W)
        pts = xyz.permute(0, 1, 3, 4, 2).reshape(B * N, H * W, 3)
        ok = torch.all((pts < self.max_bound) & (pts >= self.min_bound), dim=2, keepdim=True)
        return xyz, ok.view(B, N, H, W, 1).permute(0, 1, 4, 2, 3), f

    def to_voxel_coords(self, pts):
        """[B,P,3] LiDAR -> un-floored map coords [B,P,2] = (lidar2map @ [p;1])[:2] / voxel (:175-189)."""
        h = torch.cat([pts, torch.ones_like(pts[:, :, :1])], dim=2)
        m = (self.lidar2map @ h.permute(0, 2, 1)).permute(0, 2, 1)
        return m[:, :, :2] / self.voxel_size[:2]

    def splat_mean(self, xy, feats, grid_hw):
        """xy [B,P,2] (X=col, Y=row), feats [B,F,P] -> ([B,F,G], [B,G,1]) (:262-354).
        Taps visited in the reference's order (0,0),(0,1),(1,0),(1,1) as (xdiff,ydiff)."""
        H, W = int(grid_hw[0]), int(grid_hw[1])
        G = H * W
        B, Fd, P = feats.shape
        XY = xy.floor().long()
        r = xy - XY.type_as(xy)
        X, Y = XY[..., 0:1], XY[..., 1:2]
        rX, rY = r[..., 0:1], r[..., 1:2]
        dens = feats.new_zeros(B, G, 1)
        vol = feats.new_zeros(B, Fd, G)
        taps = []
        for xd in (0, 1):
            wX = (1 - xd) + (2 * xd - 1) * rX
            for yd in (0, 1):
                wY = (1 - yd) + (2 * yd - 1) * rY
                X_, Y_ = X + xd, Y + yd
                valid = ((0 <= X_) & (X_ < W) & (0 <= Y_) & (Y_ < H))
                idx = torch.where(valid, Y_ * W + X_, torch.zeros_like(X_))
                w = (wX * wY) * valid.type_as(rX)
                dens.scatter_add_(1, idx, w)
                idx_f = idx.view(B, 1, P).expand(B, Fd, P)
                if self.scatter_mode in ("mean", "sum"):
                    vol.scatter_add_(2, idx_f, w.view(B, 1, P) * feats)
                elif self.scatter_mode == "max":
                    # torch_scatter.scatter(src, idx, dim=2, reduce='max', dim_size=G) (third-party, absent here:
                    # PARITY UNPINNED for this mode): per-cell max of src, 0 where no point lands; then
                    # torch.maximum with the running volume (:340-344)
                    tap = torch.full_like(vol, float("-inf")).scatter_reduce_(
                        2, idx_f, w.view(B, 1, P) * feats, reduce="amax", include_self=True)
                    tap = torch.where(torch.isinf(tap), torch.zeros_like(tap), tap)
                    vol = torch.maximum(tap, vol)
                else:
                    raise Exception("Unknown splat scatter mode:", self.scatter_mode)
                taps.append((Y_ * W + X_).squeeze(-1))
        if self.scatter_mode == "mean":
            vol = vol / dens.view(B, 1, G).clamp(self.min_weight)
        return vol, dens, torch.stack(taps, dim=1)

    def forward(self, x):
        depth, feats, p2p = x[:3]
        xyz, mask, f = self.fuse(depth, feats, p2p)
        suffix = ""
        if self.training and len(x) == 4:                 # immovable-object mask [B,N,H,W] (:214-219)
            mask = mask * x[3].unsqueeze(2)
            suffix = "_mv"
        f = f * mask
        B, N, Fd, H, W = f.shape
        assert N % self.NC == 0
        NS = N // self.NC
        pts = xyz.permute(0, 1, 3, 4, 2).reshape(B * NS, self.NC * H * W, 3)
        fl = f.permute(0, 1, 3, 4, 2).reshape(B, NS, self.NC * H * W, Fd).permute(0, 1, 3, 2)
        fl = fl.reshape(B * NS, Fd, self.NC * H * W)
        xy = self.to_voxel_coords(pts)
        vol, dens, taps = self.splat_mean(xy, fl, self.grid_size[:2])
        gh, gw = int(self.grid_size[0]), int(self.grid_size[1])
        return {f"bev_features{suffix}": vol.view(B * NS, Fd, gh, gw),
                f"bev_densities{suffix}": dens.view(B * NS, gh, gw, 1).permute(0, 3, 1, 2),
                f"bev_coords{suffix}": xy,
                "_tap_indices": taps}  # oracle-only extra: int64 [B,4,P] linear idx per tap


# ------------------------------------------------------------------------------ TerrainNet
class TerrainNet(nn.Module):
    """(rgbd [B,N,4,H,W], p2p [B,N,4,4]) -> dict (terrainnet.py:272-350)."""

    def __init__(self, cfg):
        super().__init__()
        if _get(cfg, "use_temporal", False):
            raise NotImplementedError("the temporal ConvGRU branch is out of scope")
        self.use_movability = _get(cfg, "use_movability", False)
        name = _get(cfg["vision_backbone"], "class_name", None) or "DistillationBackbone"
        if name != "DistillationBackbone":
            raise NotImplementedError(f"Vision backbone {name} not implemented")
        self.views = _get(cfg, "views", 1)
        self.depthcomp = DistillationBackbone(cfg)
        self.cam2map = Camera2MapMulti(cfg["camera_projector"])
        self.splat_key = _get(cfg["camera_projector"], "splat_key", "depth_preds_feats")
        bc = _get(cfg, "bev_classifier", None)
        self.bevclassifier = None
        if bc is not None:
            if bc["name"] != "InpaintingResNet18MultiHead":
                raise NotImplementedError(f"Bev classifier {bc['name']} not implemented")
            self.bevclassifier = InpaintingResNet18MultiHead(**bc["net_kwargs"])

    def forward(self, x, keep_taps=False):
        rgbd, p2p = x[:2]
        B = rgbd.shape[0]
        out = dict(self.depthcomp(rgbd))
        assert self.splat_key in out
        Z, Hs, Ws = out[self.splat_key].shape[-3:]
        N = self.views
        depth = out["depth_preds_metric"].view(B, N, Hs, Ws)
        feats = out[self.splat_key].view(B, N, Z, Hs, Ws)
        if self.training and self.use_movability:
            # anchor view, then every view with the immovable mask (:310-322); the second head pass overwrites the
            # un-suffixed `inpainting_sam_dynamic_*` / `elevation_*` entries -- only the `inpainting_sam` prefix takes
            # the suffix (inpainting.py:41-44, terrainnet.py:342-344)
            sp = self.cam2map([depth[:, 0:1], feats[:, 0:1], p2p[:, 0:1]])
            sp.pop("_tap_indices")
            out.update(sp)
            if len(x) > 2 and x[2] is not None:
                self.cam2map.NC = N
                sp = self.cam2map([depth, feats, p2p, x[2]])
                self.cam2map.NC = 1
                sp.pop("_tap_indices")
                out.update(sp)
            if self.bevclassifier is not None:
                out.update(self.bevclassifier(out))
                out.update(self.bevclassifier(out, key_suffix="_mv"))
            return out
        sp = self.cam2map([depth, feats, p2p])
        if not keep_taps:
            sp.pop("_tap_indices")
        out.update(sp)
        if self.bevclassifier is not None:
            out.update(self.bevclassifier(out))
        return out
