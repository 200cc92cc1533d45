"""RGB(-D) encoder selector -- mirrors /root/reference/creste/models/vision_encoder.py:11-49."""
from torch import nn

from ... import ops
from ...hipnn import Act, require_hip
from .blocks.effnet import EffNet


class VisionEncoder(nn.Module):
    def __init__(self, vision_cfg):
        super().__init__()
        from ...hipnn import hook_invalidate
        hook_invalidate(self)      # load_state_dict drops the packed / BN-folded weight caches (hipnn.invalidate_caches)
        self.vision_cfg = vision_cfg
        self.input_type = vision_cfg["input_type"]
        self.name = vision_cfg["name"]
        if self.input_type in ("rgb", "rgbd"):
            if "efficientnet" in self.name:
                e = vision_cfg["effnet_cfgs"]
                self.model = EffNet(name=self.name, inC=e["in_channels"], outC=e["out_channels"],
                                    image_size=e["image_size"], downsample=e["downsample"],
                                    return_2nd_last_layer_output=False)
            else:
                raise NotImplementedError(f"encoder {self.name} is not on the HIP path")
        else:
            raise NotImplementedError(f"Input type {self.input_type} not supported")

    def forward_act(self, img: Act, out: Act = None) -> Act:
        if self.input_type == "rgb":
            img = img.slice(0, 3)
        return self.model.forward_act(img, out=out)[0]

    def forward(self, img):
        require_hip(img, "VisionEncoder")
        return self.forward_act(ops.nchw_to_nhwc(img.contiguous().float())).nchw()
