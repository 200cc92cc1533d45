"""Depth completion head -- mirrors /root/reference/creste/models/depth.py (DepthCompletion :17-158).

encoder -> 3x3 depth head (+BN+ReLU, MFMA implicit GEMM) -> softmax-expectation over the bin values
and argmax in one kernel (reference depth.py:61-100 + utils/depth_utils.py:300-313)."""
import os

import torch
from torch import nn

from ... import ops
from ...hipnn import Act, Cached, require_hip
from .blocks.conv import MultiLayerConv
from .vision_encoder import VisionEncoder


class DepthCompletion(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        from ...hipnn import hook_invalidate
        hook_invalidate(self)      # load_state_dict drops the packed / BN-folded weight caches (hipnn.invalidate_caches)
        self.vision_cfg = model_cfg["vision_backbone"]
        self.depth_cfg = model_cfg["depth_head"]
        self.discretize_cfg = model_cfg["discretize"]
        self.return_feats = self.vision_cfg["return_feats"]
        self.vision_backbone = VisionEncoder(self.vision_cfg)
        self.depth_head = MultiLayerConv(self.depth_cfg)
        self._bins = None
        wp = self.vision_cfg["weights_path"]
        if wp and os.path.isfile(wp):
            self.load_weights(wp)

    def load_weights(self, weights_path):
        """Lightning checkpoint -> this module (reference depth.py:35-58: strip `model.` and a leading
        `depthcomp.`, drop keys this module does not own, then load strictly)."""
        sd = torch.load(weights_path, weights_only=False)["state_dict"]
        sd = {k.replace("model.", "", 1): v for k, v in sd.items() if k.startswith("model.")}
        sd = {(k.replace("depthcomp.", "", 1) if k.startswith("depthcomp.") else k): v for k, v in sd.items()}
        own = set(self.state_dict().keys())
        self.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=True)

    def _bin_values(self, device):
        d = self.discretize_cfg
        if d["mode"] != "UD":
            raise NotImplementedError("softmax-expectation depth uses uniform bins (mode 'UD')")
        if self._bins is None or self._bins.device != device:
            ops.note_cache_build()
            self._bins = torch.linspace(d["depth_min"], d["depth_max"], d["num_bins"], device=device)
        return self._bins

    def forward_act(self, x: Act, feats_out: Act = None):
        feats = self.vision_backbone.forward_act(x, out=feats_out)
        with ops.shared_rows():
            logits = self.depth_head.forward_act(feats)
            depth, bins = ops.depth_expectation(logits, self._bin_values(x.buf.device))
        return dict(logits=logits, depth=depth, bins=bins, feats=feats)

    def _pack_outputs(self, r):
        out = {"depth_preds_logits": r["logits"].nchw(), "depth_preds_metric": r["depth"],
               "depth_preds_bins": r["bins"]}
        if self.return_feats:
            out["depth_preds_feats"] = r["feats"].nchw()
        return out

    def forward(self, x):
        require_hip(x, "DepthCompletion")
        if self.training:
            raise NotImplementedError("the encoder's HIP training engine covers DistillationBackbone / TerrainNet as a whole "
                                      "(creste_public_amd/train_backbone.py); call those in train() mode, or this "
                                      "module in eval() mode")
        return self._pack_outputs(self.forward_act(ops.nchw_to_nhwc(x.contiguous().float())))
