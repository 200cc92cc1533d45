"""RGB-D U-Net encoder: EfficientNet-B0 trunk + `Up` decoder, on HIP kernels.

Mirrors /root/reference/creste/models/blocks/effnet.py (Up :8-28, EffNet :31-97).  The reference
builds its trunk from the un-vendored `efficientnet_pytorch` package (call sites effnet.py:37-45,83);
`EfficientNetB0Trunk` below is a parameter container with that package's state_dict key names
(`_conv_stem`, `_bn0`, `_blocks.N._expand_conv/_bn0/_depthwise_conv/_bn1/_se_reduce/_se_expand/
_project_conv/_bn2`, `_conv_head`, `_bn1`, `_fc`) and the published B0 architecture; its arithmetic
runs as: 1x1 expand / project convs = MFMA implicit GEMM (BN folded, swish / residual fused, the
squeeze-excite gate applied to the project conv's A operand), depthwise conv + BN + swish = one
NHWC kernel, squeeze-excite = deterministic two-stage mean + tiny FC kernel.
No ImageNet download (effnet.py:37 `from_pretrained`): weights come from the checkpoint.
The 1280-channel `_conv_head` is kept for checkpoint compatibility but never computed -- the
reference computes and discards it.
"""
import math

import torch
from torch import nn

from .... import ops
from ....hipnn import (ACT_NONE, ACT_RELU, ACT_SWISH, Act, Cached, ConvUnit, require_hip, up_out_size,
                       up_scales)

# (kernel, stride, expand, in, out, repeats): EfficientNet-B0
_B0 = [(3, 1, 1, 32, 16, 1), (3, 2, 6, 16, 24, 2), (5, 2, 6, 24, 40, 2), (3, 2, 6, 40, 80, 3),
       (5, 1, 6, 80, 112, 3), (5, 2, 6, 112, 192, 4), (3, 1, 6, 192, 320, 1)]
_BN = dict(momentum=0.01, eps=1e-3)


def _same_pad(size, k, s):
    out = math.ceil(size / s)
    tot = max((out - 1) * s + k - size, 0)
    return tot // 2, tot - tot // 2


class _PadConv2d(nn.Conv2d):
    """nn.Conv2d parameter container that remembers its static 'same' padding
    (top, bottom, left, right); the pad is applied by the HIP kernels' bounds checks."""

    def __init__(self, cin, cout, k, stride=1, groups=1, bias=False, pad=(0, 0, 0, 0)):
        super().__init__(cin, cout, k, stride=stride, groups=groups, bias=bias)
        self.static_pad = pad


# expand 1x1 + depthwise conv of the thin-input MBConv blocks as one kernel (csrc/mbconv.hip); off = the two-kernel path
FUSE_MBCONV = True


class MBConvBlock(nn.Module):
    def __init__(self, k, s, e, cin, cout, pad):
        super().__init__()
        mid = cin * e
        self.k, self.s, self.cin, self.cout, self.mid, self.has_expand = k, s, cin, cout, mid, e != 1
        if self.has_expand:
            self._expand_conv = _PadConv2d(cin, mid, 1)
            self._bn0 = nn.BatchNorm2d(mid, **_BN)
        self._depthwise_conv = _PadConv2d(mid, mid, k, stride=s, groups=mid, pad=pad)
        self._bn1 = nn.BatchNorm2d(mid, **_BN)
        sq = max(1, int(cin * 0.25))
        self._se_reduce = _PadConv2d(mid, sq, 1, bias=True)
        self._se_expand = _PadConv2d(sq, mid, 1, bias=True)
        self._project_conv = _PadConv2d(mid, cout, 1)
        self._bn2 = nn.BatchNorm2d(cout, **_BN)
        self._plan = None

    def _build(self):
        dw, b1 = self._depthwise_conv, self._bn1

        def dw_pack():
            if b1.training:
                raise NotImplementedError("training-mode BatchNorm is not on the HIP path")
            sc = b1.weight.detach() / torch.sqrt(b1.running_var + b1.eps)
            w = (dw.weight.detach()[:, 0] * sc.view(-1, 1, 1)).reshape(self.mid, -1).t().contiguous()
            return w.float(), (b1.bias.detach() - b1.running_mean * sc).float().contiguous()

        def se_pack():
            r, x = self._se_reduce, self._se_expand
            return (r.weight.detach().reshape(r.out_channels, -1).contiguous(), r.bias.detach().contiguous(),
                    x.weight.detach().reshape(x.out_channels, -1).contiguous(), x.bias.detach().contiguous())

        def expand_pack():                     # [Cin][Cexp] BN-folded weights + bias of the fused expand + depthwise kernel
            ex, b0 = self._expand_conv, self._bn0
            if b0.training:
                raise NotImplementedError("training-mode BatchNorm is not on the HIP path")
            sc = b0.weight.detach() / torch.sqrt(b0.running_var + b0.eps)
            w = (ex.weight.detach().reshape(self.mid, self.cin) * sc.view(-1, 1)).t().contiguous()
            return w.float(), (b0.bias.detach() - b0.running_mean * sc).float().contiguous()

        return dict(
            expand_fused=Cached(lambda: [self._expand_conv.weight, self._bn0.weight, self._bn0.bias,
                                         self._bn0.running_mean, self._bn0.running_var], expand_pack)
            if self.has_expand else None,
            expand=ConvUnit(self._expand_conv, self._bn0, ACT_SWISH) if self.has_expand else None,
            dw=Cached(lambda: [dw.weight, b1.weight, b1.bias, b1.running_mean, b1.running_var], dw_pack),
            se=Cached(lambda: [self._se_reduce.weight, self._se_reduce.bias, self._se_expand.weight,
                               self._se_expand.bias], se_pack),
            project=ConvUnit(self._project_conv, self._bn2, ACT_NONE))

    def forward_act(self, x: Act) -> Act:
        if self._plan is None:
            self._plan = self._build()
        p = self._plan
        w, b = p["dw"].get()
        pad = self._depthwise_conv.static_pad
        Ho, Wo = (x.H + pad[0] + pad[1] - self.k) // self.s + 1, (x.W + pad[2] + pad[3] - self.k) // self.s + 1
        if FUSE_MBCONV and p["expand"] is not None and ops.mbconv_fusable(x, self.mid, self.k, self.s, Ho, Wo):
            # thin-input blocks: the 6x expanded tensor stays in LDS between the expand and the depthwise conv
            we, be = p["expand_fused"].get()
            h, gate = ops.mbconv_expand_dw_se(x, we, be, w, b, self.k, self.s, pad, *p["se"].get())
        else:
            h = p["expand"](x) if p["expand"] is not None else x
            h, gate = ops.dwconv2d_se(h, w, b, self.k, self.s, pad, ACT_SWISH, *p["se"].get())
        return self.project_act(x, h, gate)

    def project_act(self, x: Act, h: Act, gate) -> Act:
        """SE-gated project conv + BN (+ identity skip) on the depthwise output h."""
        if self._plan is None:
            self._plan = self._build()
        res = x if (self.s == 1 and self.cin == self.cout) else None   # id_skip (drop-connect is train-only)
        return self._plan["project"](h, res=res, a_scale=gate)


class EfficientNetB0Trunk(nn.Module):
    def __init__(self, in_ch, image_size):
        super().__init__()
        ph, pw = _same_pad(image_size[0], 3, 2), _same_pad(image_size[1], 3, 2)
        self._conv_stem = _PadConv2d(in_ch, 32, 3, stride=2, pad=(*ph, *pw))      # effnet.py:41-44
        self._bn0 = nn.BatchNorm2d(32, **_BN)
        blocks, size = [], 112          # every other conv keeps the pads computed for 224x224 (see oracle)
        for (k, s, e, cin, cout, reps) in _B0:
            for r in range(reps):
                st, ci = (s, cin) if r == 0 else (1, cout)
                p = _same_pad(size, k, st)
                blocks.append(MBConvBlock(k, st, e, ci, cout, (*p, *p)))
                size = math.ceil(size / st)
        self._blocks = nn.ModuleList(blocks)
        self._conv_head = _PadConv2d(320, 1280, 1)
        self._bn1 = nn.BatchNorm2d(1280, **_BN)
        self._fc = nn.Linear(1280, 1000)
        self._stem = None

    def _stem_pack(self):
        """[(ky*3+kx)*4+ci][32] BN-folded stem weights + bias of the fused stem + depthwise kernel"""
        cv, b0 = self._conv_stem, self._bn0
        if b0.training:
            raise NotImplementedError("training-mode BatchNorm is not on the HIP path")
        sc = b0.weight.detach() / torch.sqrt(b0.running_var + b0.eps)
        w = (cv.weight.detach() * sc.view(-1, 1, 1, 1)).permute(2, 3, 1, 0).reshape(-1, cv.out_channels).contiguous()
        return w.float(), (b0.bias.detach() - b0.running_mean * sc).float().contiguous()

    def extract_endpoints_act(self, x: Act):
        if self._stem is None:
            self._stem = ConvUnit(self._conv_stem, self._bn0, ACT_SWISH, pad=self._conv_stem.static_pad)
            self._stem_fused = Cached(lambda: [self._conv_stem.weight, self._bn0.weight, self._bn0.bias,
                                               self._bn0.running_mean, self._bn0.running_var], self._stem_pack)
        b0, pad = self._blocks[0], self._conv_stem.static_pad
        H1, W1 = (x.H + pad[0] + pad[1] - 3) // 2 + 1, (x.W + pad[2] + pad[3] - 3) // 2 + 1
        skip0 = (FUSE_MBCONV and not b0.has_expand and b0.k == 3 and b0.s == 1 and self._conv_stem.in_channels == 4
                 and ops.stem_dw_fusable(x, 32, H1, W1))
        if skip0:
            # stem conv -> block 0's depthwise conv in one kernel: the 32-channel stem output stays in LDS
            if b0._plan is None:
                b0._plan = b0._build()
            ws, bs = self._stem_fused.get()
            wd, bd = b0._plan["dw"].get()
            h, gate = ops.stem_dw_se(x, ws, bs, pad, wd, bd, b0._depthwise_conv.static_pad, *b0._plan["se"].get())
            x = b0.project_act(None, h, gate)
        else:
            x = self._stem(x)
        eps, prev, n = {}, x, len(self._blocks)
        for i, blk in enumerate(self._blocks):
            if not (skip0 and i == 0):
                x = blk.forward_act(x)
            if prev.H > x.H:
                eps[f"reduction_{len(eps) + 1}"] = prev
            elif i == n - 1:
                eps[f"reduction_{len(eps) + 1}"] = x
            prev = x
        return eps


class Up(nn.Module):
    """bilinear up(x1); cat([x2, up]); 2 x (3x3 conv no-bias + BN + ReLU) (reference effnet.py:8-28).
    The upsample and the concat are ONE kernel pass writing the conv's input; both convs are MFMA
    implicit GEMMs with BN folded and ReLU fused."""

    def __init__(self, inC, outC, scale_factor=2, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.up = nn.Upsample(scale_factor=scale_factor, mode="bilinear", align_corners=False)
        self.conv = nn.Sequential(
            nn.Conv2d(inC, outC, kernel_size=3, padding=1, bias=False), norm_layer(outC),
            nn.ReLU(inplace=True),
            nn.Conv2d(outC, outC, kernel_size=3, padding=1, bias=False), norm_layer(outC),
            nn.ReLU(inplace=True))
        self._units = None

    def _u(self):
        if self._units is None:
            self._units = [ConvUnit(self.conv[0], self.conv[1], ACT_RELU),
                           ConvUnit(self.conv[3], self.conv[4], ACT_RELU)]
        return self._units

    def _geom(self, x1: Act, x2: Act):
        sf, (rh, rw) = up_scales(self.up.scale_factor)
        Ho, Wo = up_out_size(x1.H, x1.W, sf)
        assert (Ho, Wo) == (x2.H, x2.W), f"Up: skip is {x2.H}x{x2.W}, upsampled map is {Ho}x{Wo}"
        return Ho, Wo, rh, rw

    def concat_act(self, x1: Act, x2: Act) -> Act:
        Ho, Wo, rh, rw = self._geom(x1, x2)
        # not formed yet: the F(4x4,3x3) conv that consumes it builds it inside its input transform (ops.LazyUpCat)
        return ops.upsample_concat_lazy(x1, x2, Ho, Wo, rh, rw)

    def convs_act(self, cat: Act, out: Act = None) -> Act:
        u = self._u()
        return ops.conv2d_pair(cat, u[0].packed(), u[1].packed(), out=out)

    def forward_act(self, x1: Act, x2: Act, out: Act = None) -> Act:
        return self.convs_act(self.concat_act(x1, x2), out=out)

    def forward(self, x1, x2):
        require_hip(x1, "Up")
        return self.forward_act(ops.nchw_to_nhwc(x1.contiguous()), ops.nchw_to_nhwc(x2.contiguous())).nchw()


class EffNet(nn.Module):
    def __init__(self, name, inC, outC, image_size, downsample, return_2nd_last_layer_output=True,
                 apply_final_batch_norm=False):
        super().__init__()
        from ....hipnn import hook_invalidate
        hook_invalidate(self)      # load_state_dict drops the packed / BN-folded weight caches (hipnn.invalidate_caches)
        if name != "efficientnet-b0":
            raise NotImplementedError
        self.trunk = EfficientNetB0Trunk(inC, image_size)
        channels = [320, 112, 40, 24, 16, inC]
        scaled = [tuple(image_size)]
        for _ in range(5):
            scaled.insert(0, (scaled[0][0] // 2, scaled[0][1] // 2))
        scale, i, C = 32 // downsample, 0, channels[0]
        while scale > 1:                                                     # effnet.py:59-73
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
        if apply_final_batch_norm:
            self.bn = nn.BatchNorm2d(outC)
        self.apply_final_batch_norm = apply_final_batch_norm
        self.return_2nd_last_layer_output = return_2nd_last_layer_output
        self._final = None

    def forward_act(self, x: Act, out: Act = None):
        """x [N,H,W,inC] -> (y [N,H/ds,W/ds,outC] written into `out` if given, 2nd-last features)."""
        eps = self.trunk.extract_endpoints_act(x)
        h = eps["reduction_5"]
        for i in range(1, self.n_ups + 1):
            h = getattr(self, f"up{i}").forward_act(h, eps[f"reduction_{5 - i}"])
        if self._final is None:
            bn = self.bn if self.apply_final_batch_norm else None
            self._final = ConvUnit(self.conv, bn, ACT_RELU if bn is not None else ACT_NONE)
        return self._final(h, out=out), h

    def forward(self, x):
        require_hip(x, "EffNet")
        y, h = self.forward_act(ops.nchw_to_nhwc(x.contiguous()))
        return (y.nchw(), h.nchw()) if self.return_2nd_last_layer_output else y.nchw()
