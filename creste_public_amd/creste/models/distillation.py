"""RGB-D backbone with the DINOv2-distilled feature head -- mirrors
/root/reference/creste/models/distillation.py (DistillationBackbone :19-207): `depthcomp`
(DepthCompletion) + `dino_head` (3 x [1x1 conv + BN + ReLU]); single-view, no learnable PE map, no
multiview splat (what the shipped configs select)."""
import os

import torch
from torch import nn

from ... import hipnn as _hipnn
from ... import ops
from ...hipnn import Act, require_hip
from .blocks.conv import MultiLayerConv, _cfg_get
from .depth import DepthCompletion


class DistillationBackbone(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        # checkpoint loaders that write through `.data` do not bump tensor versions: drop derived-weight caches
        self.register_load_state_dict_post_hook(lambda m, keys: _hipnn.invalidate_caches())
        self.model_cfg = model_cfg
        self.vision_cfg = model_cfg["vision_backbone"]
        self.depth_cfg = model_cfg["depth_head"]
        self.distillation_cfg = model_cfg["distillation_head"]
        self.input_image_shape = self.vision_cfg["effnet_cfgs"]["image_size"]
        self.ckpt_path = _cfg_get(model_cfg, "ckpt_path", "") or _cfg_get(self.vision_cfg, "ckpt_path", "")
        self.weights_path = _cfg_get(model_cfg, "weights_path", "") or _cfg_get(self.vision_cfg, "weights_path", "")
        self.multiview_distillation = _cfg_get(model_cfg, "multiview_distillation", False)
        self.freeze_weights = _cfg_get(model_cfg, "freeze_weights", False)
        self.pe_map_cfg = _cfg_get(model_cfg, "pe_map", None)
        if self.multiview_distillation or self.pe_map_cfg is not None:
            raise NotImplementedError("multiview distillation / learnable PE map are not configured by "
                                      "the shipped models and are not on the HIP path")
        trunk = _cfg_get(self.vision_cfg, "depth_trunk", "DepthCompletion")
        if trunk != "DepthCompletion":
            raise NotImplementedError(f"depth trunk {trunk}")
        self.depthcomp = DepthCompletion(model_cfg)
        head = self.distillation_cfg["feature_head"]
        if head["name"] != "MultiLayerConv":
            raise NotImplementedError(f"feature head {head['name']}")
        self.dino_head = MultiLayerConv(head)
        for p in (self.ckpt_path, self.weights_path):
            if p and os.path.isfile(p):
                self.load_weights(p)

    def load_weights(self, weights_path):
        """reference distillation.py:95-127 key surgery, strict load, optional freeze."""
        sd = torch.load(weights_path, weights_only=False)["state_dict"]
        sd = {k.replace("model.", "", 1): v for k, v in sd.items() if k.startswith("model.")}
        sd = {k.replace("depthcomp.depthcomp.", "depthcomp.", 1): v for k, v in sd.items()}
        sd = {k.replace("depthcomp.dino_head.", "dino_head.", 1): v for k, v in sd.items()}
        sd = {k: v for k, v in sd.items() if "bevclassifier" not in k and "cam2map" not in k}
        self.load_state_dict(sd, strict=True)
        if self.freeze_weights:
            for name, p in self.named_parameters():
                p.requires_grad = name not in sd

    def unfreeze_backbone(self):
        for p in self.depthcomp.parameters():
            p.requires_grad = True

    def forward_act(self, x: Act, feats_out: Act = None):
        r = self.depthcomp.forward_act(x, feats_out=feats_out)
        with ops.shared_rows():
            r["dino"] = self.dino_head.forward_act(r["feats"])
        return r

    def pack_outputs(self, r, B):
        out = self.depthcomp._pack_outputs(r)
        d = r["dino"]
        out["dino_pe_feats"] = d.nchw().unsqueeze(1)     # [B,V=1,D,Hs,Ws] (distillation.py:172,206)
        return out

    def forward(self, x):
        rgbd = x
        require_hip(rgbd, "DistillationBackbone")
        if self.training:
            from ...train_backbone import backbone_forward_train
            return backbone_forward_train(self, rgbd)
        B, V, C, H, W = rgbd.shape
        r = self.forward_act(ops.nchw_to_nhwc(rgbd.reshape(B * V, C, H, W).contiguous().float()))
        return self.pack_outputs(r, B)
