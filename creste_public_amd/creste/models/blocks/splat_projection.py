"""Depth-guided camera->BEV splat on HIP kernels.

Mirrors /root/reference/creste/models/blocks/splat_projection.py: Camera2World (:12-51) and
Camera2MapMulti (:53-354) -- same buffers/parameters (point_cloud_range, max_bound, min_bound,
voxel_size, grid_size, lidar2map, z_proj.{0,2}, vision_fusion.convs.{0,1}) and the same output dict
(`bev_features [B,F,GH,GW]`, `bev_densities [B,1,GH,GW]`, `bev_coords [B,P,2]`).

Pipeline (mode 'bilinear'; scatter_mode 'mean' | 'sum' | 'max'; NC cameras of a frame are concatenated into
one point set before the splat, reference :227-234 -- TerrainNet configures one camera and 'mean',
terrainnet.py:74-77):
  creste_pixel_geometry_f32  xyz (bit-exact fma chain), range mask, z-MLP features -> channels
                             [F, F+Z) of the fusion conv's input buffer (no concat copy)
  creste_conv2d_nhwc         1x1 fuse conv + folded BN + ReLU, range mask applied per row in the epilogue
  creste_bev_splat_f32       bin / scan / fill / sort / gather (no float atomics, BEV written once)
"""
import torch
from torch import nn

from .... import ops
from ....hipnn import Act, Cached, require_hip
from ....hipnn import get_precision as hipnn_precision
from .conv import ConvEncoder, _cfg_get


class Camera2World(nn.Module):
    """pixel (u*d, v*d, d, 1) -> LiDAR xyz (reference splat_projection.py:12-51)."""

    def forward(self, x):
        depth, p2p = x
        require_hip(depth, "Camera2World")
        B, N, H, W = depth.shape
        dummy = torch.zeros(1, device=depth.device)
        one = torch.zeros(1, 1, device=depth.device)
        zbuf = Act.empty(B * N, H, W, 1, depth.device)
        bounds = torch.tensor([-3e38] * 3 + [3e38] * 3, device=depth.device)
        xyz, _ = ops.pixel_geometry(depth.reshape(B * N, H, W).contiguous().float(),
                                    p2p.reshape(B * N, 4, 4).contiguous().float(), bounds,
                                    dummy, dummy, one, dummy, zbuf)
        return xyz.view(B, N, H, W, 3).permute(0, 1, 4, 2, 3)


class Camera2MapMulti(nn.Module):
    def __init__(self, model_cfg, mode="bilinear", scatter_mode="mean"):
        super().__init__()
        from ....hipnn import hook_invalidate
        hook_invalidate(self)      # load_state_dict drops the packed / BN-folded weight caches (hipnn.invalidate_caches)
        self.model_cfg = model_cfg
        pcr = torch.tensor(model_cfg["point_cloud_range"])
        self.register_buffer("point_cloud_range", pcr)
        self.register_buffer("max_bound", pcr[3:].reshape(1, -1))
        self.register_buffer("min_bound", pcr[:3].reshape(1, -1))
        self.register_buffer("voxel_size", torch.tensor(model_cfg["voxel_size"]))
        self.register_buffer("grid_size", ((pcr[3:] - pcr[:3]) / self.voxel_size).long())
        mb = self.min_bound
        self.register_buffer("lidar2map", torch.tensor(
            [[0, -1, 0, -mb[0, 0]], [-1, 0, 0, -mb[0, 1]], [0, 0, -1, -mb[0, 2]], [0, 0, 0, 1]]).float())
        self.mode, self.scatter_mode, self.min_weight = mode, scatter_mode, 1.0
        self.NC = _cfg_get(model_cfg, "num_cams", 2)
        self.cam2world = Camera2World()
        if model_cfg["z_embed_mode"] != "mlp":
            raise Exception("Unknown z_embed_mode:", model_cfg["z_embed_mode"])
        zd = model_cfg["z_embed_dim"]
        self.z_proj = nn.Sequential(nn.Linear(1, zd * 2, bias=True), nn.ReLU(),
                                    nn.Linear(zd * 2, zd, bias=True), nn.ReLU())
        self.vision_fusion = ConvEncoder(model_cfg["vision_fusion"])
        self.z_dim = zd
        self._geo = Cached(
            lambda: [self.z_proj[0].weight, self.z_proj[0].bias, self.z_proj[2].weight,
                     self.z_proj[2].bias, self.min_bound, self.max_bound, self.lidar2map, self.voxel_size],
            self._pack_geo)

    def _pack_geo(self):
        l2m = self.lidar2map.detach().cpu()
        expect = torch.tensor([[0., -1., 0.], [-1., 0., 0.]])
        if not torch.equal(l2m[:2, :3], expect):
            raise NotImplementedError("bev_splat kernel assumes the reference's axis-swap lidar2map")
        vs = self.voxel_size.detach().cpu()
        return dict(
            bounds=torch.cat([self.min_bound.view(-1), self.max_bound.view(-1)]).float().contiguous(),
            w1=self.z_proj[0].weight.detach().reshape(-1).contiguous(),
            b1=self.z_proj[0].bias.detach().contiguous(),
            w2=self.z_proj[2].weight.detach().contiguous(), b2=self.z_proj[2].bias.detach().contiguous(),
            off=(float(l2m[0, 3]), float(l2m[1, 3])), vox=(float(vs[0]), float(vs[1])),
            grid=(int(self.grid_size[0]), int(self.grid_size[1])))

    def fusion_buffer(self, N, H, W, F, device) -> Act:
        """[N,H,W,F+Z] buffer whose first F channels receive the splat features (written in place by
        the encoder's last conv) and whose last Z channels receive the z-MLP features."""
        return Act.empty(N, H, W, F + self.z_dim, device)

    def forward_act(self, depth: torch.Tensor, fbuf: Act, p2p: torch.Tensor, feats_amax=None):
        """depth [B,Hs,Ws] metres, fbuf = fusion_buffer with features in [0,F), p2p [B,4,4]; feats_amax: device
        float >= max|features| when the producer tracked it (then only the 32 z channels are scanned for the fusion
        conv's operand bound instead of all 288)."""
        if self.mode != "bilinear":
            raise Exception("Unknown splat mode:", self.mode)
        g = self._geo.get()
        F = fbuf.cs - self.z_dim
        gh, gw = g["grid"]
        plan = None
        if self.NC == 1:
            # one camera per frame: the points of a frame ARE the pixels of its view, and the binning plan's first kernel
            # (voxel coordinates, base-cell keys) runs inside the geometry kernel
            xyz, mask, plan = ops.pixel_geometry_plan(depth, p2p, g["bounds"], g["w1"], g["b1"], g["w2"], g["b2"],
                                                      fbuf.slice(F, self.z_dim), g["off"], g["vox"], gh, gw)
        else:
            xyz, mask = ops.pixel_geometry(depth, p2p, g["bounds"], g["w1"], g["b1"], g["w2"], g["b2"],
                                           fbuf.slice(F, self.z_dim))
        # NC cameras per frame: views (b, s, c) are consecutive, so the reference's concatenation of the cameras'
        # points ([B*NS, NC*H*W, .], :227-234) is a reshape of the per-view buffers
        BN = xyz.shape[0]
        assert BN % self.NC == 0, f"Number of frames must be divisible by {self.NC}"
        if self.scatter_mode not in ops.SPLAT_MODES:
            raise Exception("Unknown splat scatter mode:", self.scatter_mode)
        # the binning plan needs the points only: enqueued here, ahead of the fusion conv whose output the gather reads
        if plan is None:
            with ops.shared_rows():
                plan = ops.bev_splat_plan(xyz.reshape(BN // self.NC, -1, 3), g["off"], g["vox"], gh, gw)
        whole = Act(fbuf.buf, fbuf.cs, 0)
        if feats_amax is not None and (ops.TRACK_AMAX or hipnn_precision() == "f16x3"):
            whole.amax = ops.max2(feats_amax, ops.absmax(fbuf.slice(F, self.z_dim)))
        fused = self.vision_fusion.forward_act(whole, row_mask=mask)
        fl = Act(fused.buf.view(BN // self.NC, self.NC * fused.H, fused.W, fused.cs), fused.C, fused.co, fused.amax)
        with ops.shared_rows():
            bev, dens = ops.bev_splat_gather(plan, fl, self.min_weight, self.scatter_mode)
        coords = plan.coords
        return dict(bev=bev, dens=dens, coords=coords, xyz=xyz, mask=mask, fused=fused, fused_in=whole)

    def forward(self, x):
        assert len(x) >= 3, "Input must contain depth, features and camera projection matrix."
        depth, feats, p2p = x[:3]
        require_hip(depth, "Camera2MapMulti")
        B, N, F, H, W = feats.shape
        assert N % self.NC == 0, f"Number of frames must be divisible by {self.NC}"
        if self.training:
            # BatchNorm on batch statistics, autograd-connected; with the immovable-object mask as 4th input the keys
            # take the `_mv` suffix and the mask multiplies the range mask (reference :214-219)
            from ....train_terrain import splat_forward_train
            mv, sfx = (x[3].reshape(B * N, H, W), "_mv") if len(x) == 4 else (None, "")
            bev, dens, coords = splat_forward_train(self, depth.reshape(B * N, H, W), feats.reshape(B * N, F, H, W),
                                                    p2p.reshape(B * N, 4, 4), mv, "_train_engine" + sfx)
            return {f"bev_features{sfx}": bev, f"bev_densities{sfx}": dens, f"bev_coords{sfx}": coords}
        fbuf = self.fusion_buffer(B * N, H, W, F, feats.device)
        ops.nchw_to_nhwc(feats.reshape(B * N, F, H, W).contiguous().float(), out=fbuf.slice(0, F))
        r = self.forward_act(depth.reshape(B * N, H, W).contiguous().float(), fbuf,
                             p2p.reshape(B * N, 4, 4).contiguous().float())
        return {"bev_features": r["bev"].nchw(), "bev_densities": r["dens"].unsqueeze(1),
                "bev_coords": r["coords"]}
