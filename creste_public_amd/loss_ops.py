"""Fused training objectives of stage-1 distillation on HIP kernels (csrc/train_backbone.hip): each is one
autograd Function whose forward also produces the gradient w.r.t. its prediction (the loss is a scalar, so the
backward is a scaling by the incoming cotangent)."""
from __future__ import annotations

import torch

from . import _lib
from .ops import HipLibraryError, _stream
from .train_ops import as_act


def _work(dev):
    return torch.empty(_lib.load().creste_loss_workspace_bytes(), dtype=torch.uint8, device=dev)


class DepthCEFn(torch.autograd.Function):
    """CrossEntropyDepth (reference loss_utils.py:477-527): logits [N,128,H,W], label depth [N,H,W] in mm."""

    @staticmethod
    def forward(ctx, logits, gt_mm, num_bins, depth_min, depth_max):
        if not logits.is_cuda:
            raise HipLibraryError("depth cross-entropy runs on the HIP kernels only")
        la = as_act(logits)
        gt = gt_mm.detach().float().contiguous()
        dev = logits.device
        g = torch.empty_like(la.buf)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().creste_depth_ce_loss_f32(la.ptr, la.cs, gt.data_ptr(), la.N * la.H * la.W, int(num_bins),
                                                        float(depth_min), float(depth_max), 1.0, g.data_ptr(),
                                                        g.shape[3], out3.data_ptr(), _work(dev).data_ptr(), _stream()),
                   "depth_ce_loss")
        ctx.g = g.permute(0, 3, 1, 2)                     # [N,C,H,W]-shaped view of the NHWC gradient
        ctx.mark_non_differentiable(out3)
        return out3[0].clone(), out3

    @staticmethod
    def backward(ctx, gl, _):
        return ctx.g * gl, None, None, None, None


class MSEFn(torch.autograd.Function):
    """MSELoss over [P,Z] rows with +-inf labels masked (reference loss_utils.py:606-647)."""

    @staticmethod
    def forward(ctx, pred_nchw, gt_nchw):
        if not pred_nchw.is_cuda:
            raise HipLibraryError("feature MSE runs on the HIP kernels only")
        pa, ga = as_act(pred_nchw), as_act(gt_nchw)
        dev = pred_nchw.device
        g = torch.empty((pa.N, pa.H, pa.W, pa.C), dtype=torch.float32, device=dev)
        out2 = torch.empty(2, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().creste_mse_loss_f32(pa.ptr, pa.cs, ga.ptr, ga.cs, pa.N * pa.H * pa.W, pa.C, 1.0,
                                                   g.data_ptr(), pa.C, out2.data_ptr(), _work(dev).data_ptr(), _stream()),
                   "mse_loss")
        ctx.g = g.permute(0, 3, 1, 2)
        return out2[0].clone()

    @staticmethod
    def backward(ctx, gl):
        return ctx.g * gl, None
