"""Glue between torch parameter containers and the HIP kernels.

The host-side mirror modules (creste_public_amd/creste/...) keep their weights in ordinary
nn.Conv2d / nn.BatchNorm2d / nn.Linear containers so that the reference's checkpoints load by key
name.  A `ConvUnit` ties one conv (+ optional eval-mode BatchNorm + activation) to its packed,
BN-folded GEMM weight on the device; it re-packs when a parameter is replaced or updated in place
(`tensor._version`), so `load_state_dict` / optimiser steps are picked up automatically.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_SWISH, Act, HipLibraryError  # noqa: F401


# (Round 1 built a 3x3 loader that formed cat([skip, bilinear_up(x1)]) in place; measured on MI355X it lost to the
# separate upsample_concat pass in every mode -- bf16x6 222.6 -> 218.5 frames/s, bf16 484 -> 388: the 4-tap gather sits
# on the loader's critical path and the three BEV heads no longer share one concat -- and was removed in round 3.)

PRECISIONS = {"f32": ops.PREC_F32, "bf16": ops.PREC_BF16, "bf16x3": ops.PREC_BF16X3, "bf16x6": ops.PREC_BF16X6,
              "f16x3": ops.PREC_F16X3}
_precision = ops.PREC_F32


def set_precision(name: str):
    """Operand precision of the dense convs on the matrix cores:
    'f32'    v_mfma_f32_32x32x2_f32, exact fp32 products (parity mode, every conv shape);
    'bf16x3' fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate
             (fp32-grade: ~2^-16 product error) -- stride-1 1x1/3x3 convs, the rest stays 'f32';
    'bf16x6' fp32 operands split into three bf16 pieces, 6 MFMAs per product: fp32-equivalent products
             (dropped terms <= 2^-24) at 2.7x the fp32 MFMA rate -- same coverage as bf16x3;
    'f16x3'  fp32 operands rescaled by exact powers of two (activations: per tensor, from the running |max|
             the producing conv leaves behind; weights: per output channel) and split into fp16 hi+lo = 22
             significand bits, 3 MFMAs per product: product error <= 2^-21, below the fp32 accumulation
             round-off of the dot products it feeds -- fp32-grade at HALF the MFMA count of bf16x6;
    'bf16'   one bf16 MFMA per product (throughput mode), same coverage."""
    global _precision
    _precision = PRECISIONS[name]
    ops.TRACK_AMAX = name == "f16x3"


def get_precision() -> str:
    return {v: k for k, v in PRECISIONS.items()}[_precision]


def invalidate_caches():
    """Forget every packed weight / folded-BN tensor derived from module parameters.

    The caches are keyed on (data_ptr, tensor._version): `load_state_dict`, optimiser steps and any in-place op on
    the parameter bump the version and re-pack automatically.  Writes THROUGH `.data` (`p.data.copy_(..)`,
    `p.data.mul_(..)`, EMA updates, BN running statistics edited via `.data`) do not -- call this after such an
    update.  What IS hooked: every mirror module that can be used stand-alone (MaxEntIRL, TerrainNet,
    DistillationBackbone, DepthCompletion, VisionEncoder / EffNet, Camera2MapMulti, Inpainting heads, VIN) registers a
    `load_state_dict` post-hook that calls this (`hook_invalidate`), so checkpoint loaders that assign through
    `param.data.copy_` inside `load_state_dict` are covered too.  `train()` / `eval()` need no hook: the folded
    BatchNorm caches are only read in eval mode and keyed on the running statistics' versions.  The epoch is global:
    one call re-packs every model of the process on its next forward."""
    ops.CACHE_EPOCH += 1


def hook_invalidate(module: nn.Module):
    """register the `load_state_dict` post-hook described in `invalidate_caches` on `module`"""
    module.register_load_state_dict_post_hook(lambda m, keys: invalidate_caches())


def _sig(tensors):
    return (ops.CACHE_EPOCH,) + tuple((t.data_ptr(), t._version, t.device.index) for t in tensors if t is not None)


def require_hip(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise HipLibraryError(
            f"{what}: the CREStE hot path runs on hand-written HIP kernels only; got a "
            f"{t.device} tensor (there is no CPU fallback -- move model and inputs to the GPU)")


def bn_affine(bn: nn.BatchNorm2d):
    """eval-mode BatchNorm as (scale, shift) fp32 device tensors."""
    if bn.training:
        raise NotImplementedError(
            "BatchNorm in training mode (batch statistics) is not on the HIP path; call .eval() "
            "on the frozen perception backbone")
    scale = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
    shift = (bn.bias.detach() - bn.running_mean * scale).float().contiguous()
    return scale, shift


class ConvUnit:
    """conv (+ folded BN) (+ activation) with a cached packed weight."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d | None, act: int, pad=None):
        self.conv, self.bn, self.act = conv, bn, act
        k = conv.kernel_size
        assert k[0] == k[1] and conv.stride[0] == conv.stride[1] and conv.groups == 1
        if pad is None:
            p = conv.padding
            pad = (p[0], p[0], p[1], p[1])
        self.pad = pad                    # (top, bottom, left, right)
        self._packed = None
        self._key = None

    def _tensors(self):
        ts = [self.conv.weight, self.conv.bias]
        if self.bn is not None:
            ts += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        return ts

    def packed(self) -> ops.PackedConv:
        k = self.conv.kernel_size[0]
        prec = ops.conv_precision(_precision, k, self.conv.stride[0], self.conv.in_channels)
        key = (_sig(self._tensors()), prec)
        if self._packed is None or key != self._key:
            require_hip(self.conv.weight, "conv weight")
            bn = None
            if self.bn is not None:
                if self.bn.training:
                    raise NotImplementedError(
                        "BatchNorm in training mode is not on the HIP path (eval-mode folding only)")
                b = self.bn
                bn = (b.weight, b.bias, b.running_mean, b.running_var, b.eps)
            self._packed = ops.pack_conv(self.conv.weight, self.conv.bias, bn, self.conv.stride[0],
                                         self.pad, self.act, prec)
            self._key = key
        return self._packed

    def __call__(self, x: Act, out: Act | None = None, res: Act | None = None, a_scale=None,
                 row_mask=None) -> Act:
        return ops.conv2d(x, self.packed(), out=out, res=res, a_scale=a_scale, row_mask=row_mask)


class UpConvUnit(ConvUnit):
    """nn.Upsample(x2, bilinear) -> conv3x3 (+ folded BN) (+ activation) (reference DeconvHead.up2, inpainting.py:56-60): where the
    operand mode has the F(4x4,3x3) engine, four phase convolutions on the LOW-resolution map (ops.upconv2x: a transformed input a
    quarter the size, the 256-cout GEMM tile, no upsampled tensor); otherwise the conv over the lazily upsampled map."""

    def __init__(self, conv, bn, act):
        super().__init__(conv, bn, act)
        self._packed_up, self._key_up = None, None

    def phase_ok(self, x: Act, sf, rh, rw) -> bool:
        prec = ops.conv_precision(_precision, 3, 1, self.conv.in_channels)
        return (self.conv.kernel_size == (3, 3) and self.conv.stride == (1, 1) and self.pad == (1, 1, 1, 1) and tuple(sf) == (2.0, 2.0)
                and float(rh) == 0.5 and float(rw) == 0.5 and x.H >= 2 and x.W >= 2 and x.C % 4 == 0 and x.co % 4 == 0
                and ops.upconv2x_supported(prec, self.conv.in_channels, self.conv.out_channels))

    def packed_up(self) -> ops.PackedUpConv:
        prec = ops.conv_precision(_precision, 3, 1, self.conv.in_channels)
        key = (_sig(self._tensors()), prec)
        if self._packed_up is None or key != self._key_up:
            require_hip(self.conv.weight, "conv weight")
            bn = None
            if self.bn is not None:
                if self.bn.training:
                    raise NotImplementedError("BatchNorm in training mode is not on the HIP path (eval-mode folding only)")
                b = self.bn
                bn = (b.weight, b.bias, b.running_mean, b.running_var, b.eps)
            self._packed_up = ops.pack_upconv2x(self.conv.weight, self.conv.bias, bn, self.act, prec)
            self._key_up = key
        return self._packed_up

    def up(self, x: Act, sf, rh, rw, out: Act | None = None) -> Act:
        """act(bn(conv(upsample(x))))"""
        if self.phase_ok(x, sf, rh, rw):
            return ops.upconv2x(x, self.packed_up(), out=out)
        Ho, Wo = up_out_size(x.H, x.W, sf)
        return self(ops.upsample_concat_lazy(x, None, Ho, Wo, rh, rw), out=out)


class Cached:
    """Small cache of derived device tensors keyed on the source tensors' identity/version."""

    def __init__(self, sources, build):
        self.sources, self.build = sources, build
        self._val, self._key = None, None

    def get(self):
        src = self.sources()
        key = _sig(src)
        if self._val is None or key != self._key:
            for t in src:
                if t is not None:
                    require_hip(t, "parameter")
            ops.note_cache_build()
            self._val, self._key = self.build(), key
        return self._val


def up_scales(scale_factor):
    """nn.Upsample(scale_factor=..) -> (sf_h, sf_w) and PyTorch's source-index scales (float32 of the
    double reciprocal, ATen area_pixel_compute_scale with an explicit scale factor)."""
    import numpy as np
    sfh, sfw = scale_factor if isinstance(scale_factor, (tuple, list)) else (scale_factor, scale_factor)
    return (float(sfh), float(sfw)), (np.float32(1.0 / float(sfh)), np.float32(1.0 / float(sfw)))


def up_out_size(h, w, sf):
    import math
    return int(math.floor(h * sf[0])), int(math.floor(w * sf[1]))
