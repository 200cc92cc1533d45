"""Conv stacks of the hot path on the HIP conv engine.

Mirrors /root/reference/creste/models/blocks/conv.py: MultiLayerConv (:5-32), ConvEncoder (:37-58),
ConvLayer (:63-85), MultiScaleFCN (:88-161) -- same constructor configs and state_dict keys.
Eval-mode BatchNorm is folded into the packed GEMM weights; every conv runs as an MFMA implicit GEMM
with bias/activation fused in the epilogue.
"""
import torch
from torch import nn

from ....hipnn import ACT_RELU, Act, Cached, ConvUnit, bn_affine, require_hip
from .... import ops


def _cfg_get(cfg, key, default=None):
    try:
        return cfg[key]
    except (KeyError, TypeError, IndexError):
        return getattr(cfg, key, default)


def _units_from_sequential(seq):
    """[conv, (bn), relu, conv, (bn), relu ...] -> ConvUnits with folded BN + fused ReLU."""
    mods, units, i = list(seq), [], 0
    while i < len(mods):
        conv = mods[i]
        assert isinstance(conv, nn.Conv2d)
        bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d) else None
        j = i + (2 if bn is not None else 1)
        act = ACT_RELU if j < len(mods) and isinstance(mods[j], nn.ReLU) else 0
        units.append(ConvUnit(conv, bn, act))
        i = j + (1 if act else 0)
    return units


class _ConvStack(nn.Module):
    _attr = "model"

    def _build(self, dims, kernels, paddings, strides, norm_type):
        layers = []
        for i, k in enumerate(kernels):
            layers.append(nn.Conv2d(dims[i], dims[i + 1], k, padding=paddings[i], stride=strides[i]))
            if norm_type == "batch_norm":
                layers.append(nn.BatchNorm2d(dims[i + 1]))
            layers.append(nn.ReLU())
        setattr(self, self._attr, nn.Sequential(*layers))
        self._units = None

    def _chain(self):
        """the fused form of a stack of three 1x1 conv + BN + ReLU layers of 128 channels (the distillation head,
        reference distillation.py:179): ops.conv1x1_chain3, or None when the stack / operand mode is not that shape"""
        from .... import hipnn
        us = self._units
        prec = ops.conv_precision(hipnn._precision, 1, 1, us[0].conv.in_channels) if us else None
        ok = (len(us) == 3 and all(u.conv.kernel_size == (1, 1) and u.conv.stride == (1, 1) and u.pad == (0, 0, 0, 0) and u.act == ACT_RELU
                                   and (u.bn is None or not u.bn.training) for u in us)
              and ops.conv1x1_chain3_supported(prec, [us[0].conv.in_channels] + [u.conv.out_channels for u in us]))
        if not ok:
            return None
        key = (tuple(hipnn._sig(u._tensors()) for u in us), prec)
        if getattr(self, "_chain_pk", None) is None or self._chain_key != key:
            layers = [(u.conv.weight, u.conv.bias, None if u.bn is None else (u.bn.weight, u.bn.bias, u.bn.running_mean, u.bn.running_var, u.bn.eps))
                      for u in us]
            self._chain_pk, self._chain_key = ops.pack_conv1x1_chain3(layers, prec), key
        return self._chain_pk

    def forward_act(self, x: Act, out: Act = None, row_mask=None) -> Act:
        if self._units is None:
            self._units = _units_from_sequential(getattr(self, self._attr))
        if row_mask is None and isinstance(x, Act) and x.co % 4 == 0:
            pk = self._chain()
            if pk is not None:
                return ops.conv1x1_chain3(x, pk, out=out)
        for i, u in enumerate(self._units):
            last = i == len(self._units) - 1
            x = u(x, out=out if last else None, row_mask=row_mask if last else None)
        return x

    def forward(self, x):
        require_hip(x, type(self).__name__)
        return self.forward_act(ops.nchw_to_nhwc(x.contiguous())).nchw()


class MultiLayerConv(_ConvStack):
    """conv(+bias) -> [BN] -> ReLU stack; always ends in ReLU (reference conv.py:21-29)."""
    _attr = "model"

    def __init__(self, model_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        ks = model_cfg["kernels"]
        self._build(model_cfg["dims"], ks, model_cfg["paddings"],
                    _cfg_get(model_cfg, "stride", None) or [1] * len(ks), model_cfg["norm_type"])


class ConvEncoder(_ConvStack):
    """Same stack under the attribute name `convs` (reference conv.py:37-58)."""
    _attr = "convs"

    def __init__(self, model_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        ks = model_cfg["kernels"]
        assert len(ks) == len(model_cfg["paddings"])
        self._build(model_cfg["dims"], ks, model_cfg["paddings"], [1] * len(ks), model_cfg["norm_type"])


class ConvLayer(nn.Sequential):
    """conv(k, pad k//2, bias off by default) [-> norm] [-> ReLU] (reference conv.py:63-85)."""

    def __init__(self, in_channels, out_channels, kernel=3, stride=1, dropout=0.1, bn=False,
                 norm_type="batch_norm", relu=True, bias=False):
        super().__init__()
        self.add_module("conv", nn.Conv2d(in_channels, out_channels, kernel_size=kernel, stride=stride,
                                          padding=kernel // 2, bias=bias))
        if bn:
            if norm_type == "batch_norm":
                self.add_module("norm", nn.BatchNorm2d(out_channels))
            elif norm_type == "group_norm":
                self.add_module("norm", nn.GroupNorm(num_groups=2, num_channels=out_channels))
            else:
                raise Exception("Unknown norm type:", norm_type)
        if relu:
            self.add_module("relu", nn.ReLU(inplace=True))

    def unit(self):
        norm = getattr(self, "norm", None)
        if norm is not None and not isinstance(norm, nn.BatchNorm2d):
            raise NotImplementedError("group_norm ConvLayer is not on the HIP path")
        return ConvUnit(self.conv, norm, ACT_RELU if hasattr(self, "relu") else 0)


class MultiScaleFCN(nn.Module):
    """Reward / costmap network (reference conv.py:88-161).

    eval(): HIP conv engine with folded BatchNorm (inference costmap).  train(): the HIP training engine
    (creste_public_amd/train_ops.py): batch-statistics BatchNorm, fp32 conv forward / dgrad / wgrad, and the
    gradient penalty's second-order term (reference loss_utils.py:1207-1217) as a tangent forward plus a
    joint backward -- behind one autograd Function, so `torch.autograd.grad(create_graph=True)` and
    `.backward()` in the loss code work unchanged.
    """

    def __init__(self, model_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        self.prepool_cfg, self.postpool_cfg = model_cfg["prepool"], model_cfg["postpool"]
        self.skip_cfg, self.trunk_cfg = model_cfg["skip"], model_cfg["trunk"]

        def stack(c):
            return nn.Sequential(*[
                ConvLayer(c["dims"][i], c["dims"][i + 1], kernel=c["kernels"][i], stride=c["stride"][i],
                          bn=True, norm_type=c["norm_type"], relu=True, bias=False)
                for i in range(len(c["kernels"]))])

        self.prepool, self.skip = stack(self.prepool_cfg), stack(self.skip_cfg)
        tc = self.trunk_cfg
        trunk = [nn.MaxPool2d(kernel_size=2, stride=2)]
        for i in range(len(tc["kernels"])):
            trunk.append(ConvLayer(tc["dims"][i], tc["dims"][i + 1], kernel=tc["kernels"][i]))
            if tc["norm_type"] == "batch_norm":
                trunk.append(nn.BatchNorm2d(tc["dims"][i + 1]))
            trunk.append(nn.ReLU(inplace=True))
        trunk.append(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
        self.trunk = nn.Sequential(*trunk)
        self.postpool = stack(self.postpool_cfg)
        self.initialize_weights_with_xavier()
        self._plan = None

    def initialize_weights_with_xavier(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    # ---- HIP inference path
    def _build_plan(self):
        plan = dict(prepool=[l.unit() for l in self.prepool], skip=[l.unit() for l in self.skip],
                    postpool=[l.unit() for l in self.postpool], trunk=[])
        mods = list(self.trunk)
        i = 1
        while i < len(mods) - 1:                       # (ConvLayer, [BN], ReLU) groups
            layer = mods[i]
            bn = mods[i + 1] if isinstance(mods[i + 1], nn.BatchNorm2d) else None
            aff = None
            if bn is not None:
                aff = Cached(lambda b=bn: [b.weight, b.bias, b.running_mean, b.running_var],
                             lambda b=bn: bn_affine(b))
            plan["trunk"].append((layer.unit(), aff))
            i += 3 if bn is not None else 2
        return plan

    def forward_act(self, x: Act) -> Act:
        if self._plan is None:
            self._plan = self._build_plan()
        p = self._plan
        for u in p["prepool"]:
            x = u(x)
        cat_c = self.postpool_cfg["dims"][0]
        cat = Act.empty(x.N, x.H, x.W, cat_c, x.buf.device)
        t = ops.maxpool2(x)
        for u, aff in p["trunk"]:
            t = u(t)                                    # conv -> ReLU
            if aff is not None:
                sc, sh = aff.get()
                t = ops.affine_act(t, sc, sh, ACT_RELU)  # BN -> ReLU (conv.py:118-128)
        ops.upsample_concat(t, None, x.H, x.W, 0.5, 0.5, out=cat.slice(0, t.C))
        s = x
        for i, u in enumerate(p["skip"]):
            last = i == len(p["skip"]) - 1
            s = u(s, out=cat.slice(t.C, cat_c - t.C) if last else None)
        y = cat
        for u in p["postpool"]:
            y = u(y)
        return y

    def forward(self, x):
        """Expects input of shape [B, C, H, W]."""
        require_hip(x, "MultiScaleFCN")
        if self.training:
            from ....train_ops import reward_forward_train
            return reward_forward_train(self, x)
        return self.forward_act(ops.nchw_to_nhwc(x.contiguous())).nchw()
