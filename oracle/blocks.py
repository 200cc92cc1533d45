"""Oracle (TEST INFRASTRUCTURE) -- CPU/PyTorch restatement of the dense blocks.

Restates, op for op:
  * conv stacks   : /root/reference/creste/models/blocks/conv.py:5-32 (MultiLayerConv),
                    :37-58 (ConvEncoder), :63-85 (ConvLayer), :88-161 (MultiScaleFCN)
  * U-Net decoder : /root/reference/creste/models/blocks/effnet.py:8-28 (Up), :31-97 (EffNet)
  * EfficientNet-B0 MBConv trunk: third-party `efficientnet_pytorch` (un-vendored; call
    sites effnet.py:37-45,83) -- published architecture, see oracle/__init__.py
  * ResNet-18 BasicBlock trunk + DeconvHead: /root/reference/creste/models/blocks/inpainting.py:52-109
    (+ third-party torchvision `resnet18`, call site inpainting.py:80-90)

Module/parameter names equal the reference's state_dict keys.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


def _get(cfg, key, default=None):
    try:
        return cfg[key]
    except (KeyError, TypeError, IndexError):
        return getattr(cfg, key, default)


# --------------------------------------------------------------------------- conv stacks
class MultiLayerConv(nn.Module):
    """conv(+bias) -> [BN] -> ReLU, repeated; always ends in ReLU (conv.py:21-29)."""

    def __init__(self, cfg):
        super().__init__()
        dims, ks, ps = cfg["dims"], cfg["kernels"], cfg["paddings"]
        strides = _get(cfg, "stride", None) or [1] * len(ks)
        layers = []
        for i, k in enumerate(ks):
            layers.append(nn.Conv2d(dims[i], dims[i + 1], k, padding=ps[i], stride=strides[i]))
            if cfg["norm_type"] == "batch_norm":
                layers.append(nn.BatchNorm2d(dims[i + 1]))
            layers.append(nn.ReLU())
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class ConvEncoder(nn.Module):
    """Same stack, attribute name `convs` (conv.py:48-55)."""

    def __init__(self, cfg):
        super().__init__()
        dims, ks, ps = cfg["dims"], cfg["kernels"], cfg["paddings"]
        assert len(ks) == len(ps)
        layers = []
        for i, k in enumerate(ks):
            layers.append(nn.Conv2d(dims[i], dims[i + 1], kernel_size=k, padding=ps[i]))
            if cfg["norm_type"] == "batch_norm":
                layers.append(nn.BatchNorm2d(dims[i + 1]))
            layers.append(nn.ReLU())
        self.convs = nn.Sequential(*layers)

    def forward(self, x):
        return self.convs(x)


class ConvLayer(nn.Sequential):
    """conv(k, pad=k//2, no bias by default) [-> norm] [-> ReLU] (conv.py:63-85)."""

    def __init__(self, cin, cout, kernel=3, stride=1, bn=False, norm_type="batch_norm",
                 relu=True, bias=False):
        super().__init__()
        self.add_module("conv", nn.Conv2d(cin, cout, kernel_size=kernel, stride=stride,
                                          padding=kernel // 2, bias=bias))
        if bn:
            if norm_type == "batch_norm":
                self.add_module("norm", nn.BatchNorm2d(cout))
            elif norm_type == "group_norm":
                self.add_module("norm", nn.GroupNorm(num_groups=2, num_channels=cout))
            else:
                raise Exception("Unknown norm type:", norm_type)
        if relu:
            self.add_module("relu", nn.ReLU(inplace=True))


class MultiScaleFCN(nn.Module):
    """Reward / costmap network (conv.py:88-161).

    prepool -> {skip, trunk(maxpool2, [conv,ReLU,BN,ReLU]*, bilinear x2)} -> cat[trunk, skip]
    -> postpool.  Note the trunk's ConvLayer has relu=True and bn=False, and a separate
    BatchNorm2d + ReLU follow it (conv.py:118-128): conv -> ReLU -> BN -> ReLU.
    """

    def __init__(self, cfg):
        super().__init__()
        def stack(c):
            return nn.Sequential(*[
                ConvLayer(c["dims"][i], c["dims"][i + 1], kernel=c["kernels"][i],
                          stride=c["stride"][i], bn=True, norm_type=c["norm_type"],
                          relu=True, bias=False)
                for i in range(len(c["kernels"]))])

        self.prepool = stack(cfg["prepool"])
        self.skip = stack(cfg["skip"])
        tc = cfg["trunk"]
        trunk = [nn.MaxPool2d(kernel_size=2, stride=2)]
        for i in range(len(tc["kernels"])):
            trunk.append(ConvLayer(tc["dims"][i], tc["dims"][i + 1], kernel=tc["kernels"][i]))
            if tc["norm_type"] == "batch_norm":
                trunk.append(nn.BatchNorm2d(tc["dims"][i + 1]))
            trunk.append(nn.ReLU(inplace=True))
        trunk.append(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
        self.trunk = nn.Sequential(*trunk)
        self.postpool = stack(cfg["postpool"])
        for m in self.modules():  # conv.py:141-146
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.prepool(x)
        skip = self.skip(x)
        x = self.trunk(x)
        return self.postpool(torch.cat([x, skip], dim=1))


# --------------------------------------------------------------------------- U-Net decoder
class Up(nn.Module):
    """bilinear up(x1) ; cat([x2, up]) ; 2 x (3x3 conv no-bias + BN + ReLU) (effnet.py:8-28)."""

    def __init__(self, cin, cout, scale_factor=2, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.up = nn.Upsample(scale_factor=scale_factor, mode="bilinear", align_corners=False)
        self.conv = nn.Sequential(
            nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False), norm_layer(cout),
            nn.ReLU(inplace=True),
            nn.Conv2d(cout, cout, kernel_size=3, padding=1, bias=False), norm_layer(cout),
            nn.ReLU(inplace=True))

    def forward(self, x1, x2):
        return self.conv(torch.cat([x2, self.up(x1)], dim=1))


# ------------------------------------------------------------------ EfficientNet-B0 trunk
# (kernel, stride, expand, in, out, repeats) -- published B0 block table.
B0_STAGES = [(3, 1, 1, 32, 16, 1), (3, 2, 6, 16, 24, 2), (5, 2, 6, 24, 40, 2),
             (3, 2, 6, 40, 80, 3), (5, 1, 6, 80, 112, 3), (5, 2, 6, 112, 192, 4),
             (3, 1, 6, 192, 320, 1)]
BN_EPS, BN_MOM = 1e-3, 0.01
DROP_CONNECT = 0.2


def same_pad(size, k, s):
    """TF-'same' static pad for one axis of a `size`-long input: (before, after)."""
    out = math.ceil(size / s)
    total = max((out - 1) * s + (k - 1) + 1 - size, 0)
    return total // 2, total - total // 2


def b0_block_specs():
    """Per-block (k, s, expand, cin, cout, pad(before,after)) with the static padding the
    pretrained constructor computes for a 224x224 image (efficientnet_pytorch builds every
    MBConv's depthwise conv as Conv2dStaticSamePadding with the running 224-based size)."""
    specs, size = [], 112  # 224 after the stride-2 stem
    for (k, s, e, cin, cout, reps) in B0_STAGES:
        for r in range(reps):
            st = s if r == 0 else 1
            ci = cin if r == 0 else cout
            specs.append(dict(k=k, s=st, e=e, cin=ci, cout=cout, pad=same_pad(size, k, st)))
            size = math.ceil(size / st)
    return specs


class _StaticSameConv(nn.Conv2d):
    """Conv2d with a fixed asymmetric zero pad (left/top = before, right/bottom = after).
    Registers the (parameter-free) `static_padding` child the third-party class has, so
    module trees print alike; it contributes no state_dict keys."""

    def __init__(self, cin, cout, k, stride, pad_h, pad_w, groups=1, bias=False):
        super().__init__(cin, cout, k, stride=stride, groups=groups, bias=bias)
        self.static_padding = nn.ZeroPad2d((pad_w[0], pad_w[1], pad_h[0], pad_h[1]))

    def forward(self, x):
        return F.conv2d(self.static_padding(x), self.weight, self.bias, self.stride, 0,
                        self.dilation, self.groups)


class MBConv(nn.Module):
    def __init__(self, spec):
        super().__init__()
        k, s, e, cin, cout, pad = (spec[n] for n in ("k", "s", "e", "cin", "cout", "pad"))
        mid = cin * e
        self.expand = e != 1
        self.stride, self.cin, self.cout = s, cin, cout
        if self.expand:
            self._expand_conv = _StaticSameConv(cin, mid, 1, 1, (0, 0), (0, 0))
            self._bn0 = nn.BatchNorm2d(mid, momentum=BN_MOM, eps=BN_EPS)
        self._depthwise_conv = _StaticSameConv(mid, mid, k, s, pad, pad, groups=mid)
        self._bn1 = nn.BatchNorm2d(mid, momentum=BN_MOM, eps=BN_EPS)
        sq = max(1, int(cin * 0.25))
        self._se_reduce = _StaticSameConv(mid, sq, 1, 1, (0, 0), (0, 0), bias=True)
        self._se_expand = _StaticSameConv(sq, mid, 1, 1, (0, 0), (0, 0), bias=True)
        self._project_conv = _StaticSameConv(mid, cout, 1, 1, (0, 0), (0, 0))
        self._bn2 = nn.BatchNorm2d(cout, momentum=BN_MOM, eps=BN_EPS)

    @staticmethod
    def swish(x):
        return x * torch.sigmoid(x)

    def forward(self, inputs, drop_connect_rate=None):
        x = inputs
        if self.expand:
            x = self.swish(self._bn0(self._expand_conv(x)))
        x = self.swish(self._bn1(self._depthwise_conv(x)))
        sq = F.adaptive_avg_pool2d(x, 1)
        sq = self._se_expand(self.swish(self._se_reduce(sq)))
        x = torch.sigmoid(sq) * x
        x = self._bn2(self._project_conv(x))
        if self.stride == 1 and self.cin == self.cout:
            if drop_connect_rate and self.training:
                keep = 1 - drop_connect_rate
                mask = torch.floor(keep + torch.rand([x.shape[0], 1, 1, 1], dtype=x.dtype,
                                                     device=x.device))
                x = x / keep * mask
            x = x + inputs
        return x


class EfficientNetB0Trunk(nn.Module):
    """EfficientNet-B0 feature trunk; `_conv_head/_bn1/_fc` exist only so the state_dict has
    the reference's keys (the reference computes the 1280-ch head and discards it,
    SURVEY.md K1; the oracle skips computing it -- it is not an output)."""

    def __init__(self, in_ch, image_size):
        super().__init__()
        ph, pw = same_pad(image_size[0], 3, 2), same_pad(image_size[1], 3, 2)
        self._conv_stem = _StaticSameConv(in_ch, 32, 3, 2, ph, pw)      # effnet.py:41-44
        self._bn0 = nn.BatchNorm2d(32, momentum=BN_MOM, eps=BN_EPS)
        self._blocks = nn.ModuleList([MBConv(s) for s in b0_block_specs()])
        self._conv_head = _StaticSameConv(320, 1280, 1, 1, (0, 0), (0, 0))
        self._bn1 = nn.BatchNorm2d(1280, momentum=BN_MOM, eps=BN_EPS)
        self._fc = nn.Linear(1280, 1000)

    def extract_endpoints(self, x):
        eps = {}
        x = MBConv.swish(self._bn0(self._conv_stem(x)))
        prev = x
        n = len(self._blocks)
        for i, blk in enumerate(self._blocks):
            x = blk(x, drop_connect_rate=DROP_CONNECT * float(i) / n)
            if prev.size(2) > x.size(2):
                eps[f"reduction_{len(eps) + 1}"] = prev
            elif i == n - 1:
                eps[f"reduction_{len(eps) + 1}"] = x
            prev = x
        return eps


class EffNet(nn.Module):
    """EfficientNet-B0 + `Up` decoder to 1/downsample resolution + 1x1 conv (effnet.py:31-97)."""

    def __init__(self, name, inC, outC, image_size, downsample, return_2nd_last_layer_output=True):
        super().__init__()
        if name != "efficientnet-b0":
            raise NotImplementedError
        self.trunk = EfficientNetB0Trunk(inC, image_size)
        channels = [320, 112, 40, 24, 16, inC]
        scaled = [tuple(image_size)]
        for _ in range(5):
            scaled.insert(0, (scaled[0][0] // 2, scaled[0][1] // 2))
        scale, i, C = 32 // downsample, 0, channels[0]
        while scale > 1:
            if not (scaled[i + 1][0] % 2 or scaled[i + 1][1] % 2):
                sf = 2
            else:
                sf = (scaled[i + 1][0] / scaled[i][0], scaled[i + 1][1] / scaled[i][1])
            scale //= 2
            i += 1
            C += channels[i]
            setattr(self, f"up{i}", Up(C, C, sf))
        self.n_ups = i
        self.conv = nn.Conv2d(C, outC, kernel_size=1, padding=0)
        self.return_2nd = return_2nd_last_layer_output

    def forward(self, x):
        eps = self.trunk.extract_endpoints(x)
        y = eps["reduction_5"]
        for i in range(1, self.n_ups + 1):
            y = getattr(self, f"up{i}")(y, eps[f"reduction_{5 - i}"])
        out = self.conv(y)
        return (out, y) if self.return_2nd else out


# ------------------------------------------------------------------ BEV ResNet-18 heads
class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        nn.init.constant_(self.bn2.weight, 0)  # zero_init_residual=True (inpainting.py:80)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class DeconvHead(nn.Module):
    """Up(x4) -> [bilinear x2, 3x3 256->128, BN, ReLU] -> 1x1 proj (inpainting.py:52-68)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.up1 = Up(cin, 256, scale_factor=4)
        self.up2 = nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False),
                                 nn.Conv2d(256, 128, kernel_size=3, padding=1, bias=False),
                                 nn.BatchNorm2d(128), nn.ReLU(inplace=True))
        self.proj = nn.Conv2d(128, cout, kernel_size=1, padding=0)

    def forward(self, x1, x2):
        x = self.up2(self.up1(x1, x2))
        return self.proj(x), x


class InpaintingResNet18MultiHead(nn.Module):
    """7x7/2 stem + resnet18 layer1..3 + N DeconvHeads, dict-in/dict-out with key prefixing
    (inpainting.py:9-50,70-109; prefix_dict train_utils.py:560-564)."""

    def __init__(self, num_input_features, num_classes, input_key=None, output_prefix=None,
                 norm_layer="batch_norm"):
        super().__init__()
        if norm_layer != "batch_norm":
            raise Exception("Unsupported norm layer:", norm_layer)
        self.input_key = input_key or "merged_bev_features"
        self.output_prefix = output_prefix or "inpainting"
        self.conv1 = nn.Conv2d(num_input_features, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(BasicBlock(64, 64, 1), BasicBlock(64, 64, 1))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128, 1))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256, 1))
        self.out_heads = nn.ModuleList([DeconvHead(64 + 256, n) for n in num_classes])

    def forward(self, tensor_dict, key_suffix=""):
        x = self.relu(self.bn1(self.conv1(tensor_dict[f"{self.input_key}{key_suffix}"])))
        x1 = self.layer1(x)
        x = self.layer3(self.layer2(x1))
        ret = {}
        for prefix, head in zip(self.output_prefix, self.out_heads):
            pred, fea = head(x, x1)
            if prefix == "inpainting_sam":
                prefix = f"{prefix}{key_suffix}"
            ret[f"{prefix}_preds"] = pred
            ret[f"{prefix}_features"] = fea
        return ret
