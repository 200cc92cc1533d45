"""TerrainNet: RGB-D backbone -> depth-guided BEV splat -> BEV completion heads, on HIP kernels.

Mirrors /root/reference/creste/models/terrainnet.py (TerrainNet :24-350): constructor config, the six
`load_setting` modes of `load_weights` (:111-261, incl. the `depthcomp.` -> `depthcomp.depthcomp.`
key renaming :125-149), forward input tuple and output-dict keys.
Forward is one NHWC pipeline with no layout round trips between the sub-modules: the encoder's last
conv writes the 256 splat features straight into the fusion conv's input buffer, the z-MLP fills the
remaining 32 channels, and the three heads' 1x1 projections write one 40-channel buffer (the reward
network's input).  Output tensors are zero-copy [N,C,H,W]-shaped views of the NHWC buffers.
"""
import os

import torch
from torch import nn

from ... import hipnn as _hipnn
from ... import ops
from ...hipnn import Act, require_hip
from .blocks.conv import _cfg_get
from .blocks.inpainting import InpaintingResNet18MultiHead
from .blocks.splat_projection import Camera2MapMulti
from .distillation import DistillationBackbone

_BACKBONES = {"DistillationBackbone": DistillationBackbone}
_BEV_HEADS = {"InpaintingResNet18MultiHead": InpaintingResNet18MultiHead}


def _plain(cfg):
    return cfg.to_dict() if hasattr(cfg, "to_dict") else cfg


class TerrainNet(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        # checkpoint loaders that write through `.data` do not bump tensor versions: drop derived-weight caches
        self.register_load_state_dict_post_hook(lambda m, keys: _hipnn.invalidate_caches())
        self.model_cfg = model_cfg
        self.views = _cfg_get(model_cfg, "views", 1)
        self.vision_cfg = model_cfg["vision_backbone"]
        self.camproj_cfg = model_cfg["camera_projector"]
        self.depth_cfg = model_cfg["depth_head"]
        self.discretize_cfg = model_cfg["discretize"]
        self.ckpt_path = _cfg_get(model_cfg, "ckpt_path", "")
        self.weights_path = _cfg_get(model_cfg, "weights_path", "")
        self.freeze_weights = _cfg_get(model_cfg, "freeze_weights", False)
        self.use_temporal = _cfg_get(model_cfg, "use_temporal", False)
        self.use_movability = _cfg_get(model_cfg, "use_movability", False)
        self.load_setting = _cfg_get(model_cfg, "load_setting", "strict")
        if self.use_temporal:
            raise NotImplementedError("the temporal ConvGRU is disabled in the shipped configs and out of this "
                                      "tier's scope")
        self.bev_classifer_cfg = _cfg_get(model_cfg, "bev_classifier", None)
        self.bev_semantic_head_cfg = _cfg_get(model_cfg, "bev_semantic_head", None)
        if self.bev_semantic_head_cfg is not None:
            raise NotImplementedError("bev_semantic_head is not used by the shipped configs")

        name = _cfg_get(self.vision_cfg, "class_name", None) or "DistillationBackbone"
        try:
            self.depthcomp = _BACKBONES[name](model_cfg)
        except KeyError:
            raise NotImplementedError(f"Vision backbone {name} not implemented")
        self.cam2map = Camera2MapMulti(self.camproj_cfg, mode="bilinear")
        self.splat_key = _cfg_get(self.camproj_cfg, "splat_key", "depth_preds_feats")
        if self.splat_key != "depth_preds_feats":
            raise NotImplementedError("HIP pipeline splats `depth_preds_feats` (the shipped splat_key)")
        self.bevclassifier = None
        if self.bev_classifer_cfg is not None:
            try:
                self.bevclassifier = _BEV_HEADS[self.bev_classifer_cfg["name"]](
                    **_plain(self.bev_classifer_cfg["net_kwargs"]))
            except KeyError:
                raise NotImplementedError(f"Bev classifier {self.bev_classifer_cfg['name']} not implemented")
        if self.weights_path and os.path.isfile(self.weights_path) and not os.path.isfile(self.ckpt_path or ""):
            self.load_weights(self.weights_path)

    # ------------------------------------------------------------------ checkpoints
    def load_weights(self, weights_path):
        sd = torch.load(weights_path, weights_only=False)["state_dict"]
        sd = {(k.replace("model.", "", 1) if k.startswith("model.") else k): v for k, v in sd.items()}
        n0, renamed = len(sd), {}
        for k, v in sd.items():                           # terrainnet.py:125-139
            if k.startswith("depthcomp.") and not k.startswith(("depthcomp.depthcomp.", "depthcomp.dino_head.")):
                k = k.replace("depthcomp.", "depthcomp.depthcomp.", 1)
            elif k.startswith("dino_head."):
                k = k.replace("dino_head.", "depthcomp.dino_head.", 1)
            renamed[k] = v
        assert len(renamed) == n0, f"Number of keys changed after filtering. Before: {n0}, After: {len(renamed)}"
        sd = renamed
        no_loss = {k: v for k, v in sd.items() if not k.startswith("loss.")}

        def only_trainable(pred):
            for n, p in self.named_parameters():
                p.requires_grad = bool(pred(n))

        mode = self.load_setting
        if mode == "ft_semantic_head":
            self.load_state_dict(sd, strict=False)
            only_trainable(lambda n: "bev_semantic_head" in n)
            for head in self.bevclassifier.out_heads:
                if head.proj.out_channels == 1:
                    for p in head.parameters():
                        p.requires_grad = True
        elif mode == "ft_decoders_all":
            self.load_state_dict({k: v for k, v in sd.items() if "bevclassifier.out_heads" not in k}, strict=False)
            only_trainable(lambda n: "bevclassifier.out_heads" in n)
        elif mode == "ft_decoders_partial":
            tail = lambda n: "bevclassifier.out_heads" in n and ("up2" in n or "proj" in n)
            self.load_state_dict({k: v for k, v in sd.items() if not tail(k)}, strict=False)
            only_trainable(tail)
        elif mode == "strict_freeze":
            self.load_state_dict(no_loss, strict=True)
            only_trainable(lambda n: False)
        elif mode == "strict":
            self.load_state_dict(no_loss, strict=True)
        elif mode == "strict_unfreezesplat":
            self.load_state_dict(no_loss, strict=False)
            only_trainable(lambda n: "cam2map." in n)
        else:
            raise ValueError(f"Invalid load_setting {mode}")

    # ------------------------------------------------------------------ forward
    def forward_act(self, rgbd: torch.Tensor, p2p: torch.Tensor, preds_buf_channels=None):
        """rgbd [B,N,4,H,W], p2p [B,N,4,4] -> dict of internal results (Acts / tensors)."""
        B, N, C, H, W = rgbd.shape
        if N != 1 or self.views != 1:
            raise NotImplementedError("HIP pipeline: one view per sample (views=1, the shipped config)")
        x = ops.nchw_to_nhwc(rgbd.reshape(B * N, C, H, W).contiguous().float())
        ds = self.vision_cfg["effnet_cfgs"]["downsample"]
        F = self.vision_cfg["effnet_cfgs"]["out_channels"]
        assert H % ds == 0 and W % ds == 0, "image size must be a multiple of the encoder downsample"
        with ops.shared_rows():                  # outputs (`depth_preds_feats`, the heads' predictions): see ops.PartContext
            fbuf = self.cam2map.fusion_buffer(B * N, H // ds, W // ds, F, rgbd.device)
        fslice = fbuf.slice(0, F)                # the encoder's last conv leaves max|features| on this slice
        r = self.depthcomp.forward_act(x, feats_out=fslice)
        sp = self.cam2map.forward_act(r["depth"], fbuf, p2p.reshape(B * N, 4, 4).contiguous().float(),
                                      feats_amax=fslice.amax)
        r.update(sp)
        if self.bevclassifier is not None:
            nc = self.bevclassifier.num_classes
            with ops.shared_rows():
                pb = Act.empty(B, sp["bev"].H, sp["bev"].W, sum(nc), rgbd.device) \
                    if preds_buf_channels is None else preds_buf_channels
            r["heads"] = self.bevclassifier.forward_act(sp["bev"], preds_buf=pb)
            r["preds_buf"] = pb
        return r

    def pack_outputs(self, r, B):
        out = self.depthcomp.pack_outputs(r, B)
        out["bev_features"] = r["bev"].nchw()
        out["bev_densities"] = r["dens"].unsqueeze(1)
        out["bev_coords"] = r["coords"]
        if self.bevclassifier is not None:
            named = [dict(preds=h["preds"].nchw(), features=h["features"].nchw()) for h in r["heads"]]
            out.update(self.bevclassifier._wrap(named))
        return out

    def forward(self, x):
        """x = (rgbd [B,N,4,H,W], p2p [B,N,4,4][, mv_mask]) -> dict (reference terrainnet.py:272-350)."""
        rgbd, p2p = x[:2]
        require_hip(rgbd, "TerrainNet")
        if self.training:
            from ...train_terrain import terrainnet_forward_train
            mv = x[2] if self.use_movability and len(x) > 2 else None     # immovable mask [B,N,Hs,Ws] (:317-320)
            return terrainnet_forward_train(self, rgbd, p2p, mv)
        parts = ops.parts_for(rgbd.shape[0], rgbd.device, self.inference_parts, self.inference_part_rows)
        if parts > 1:                                  # pipelined like MaxEntIRL's frozen half (lfd.py: inference_parts)
            ctx, res = ops.forward_in_parts(lambda a, b: self.pack_outputs(self.forward_act(a, b), a.shape[0]),
                                            (rgbd, p2p), parts, owner=self)
            return ops.whole_outputs(ctx, res)
        return self.pack_outputs(self.forward_act(rgbd, p2p), rgbd.shape[0])

    inference_parts = 2            # batches of >= 2 x inference_part_rows frames run as two forwards on two streams
    inference_part_rows = 6
