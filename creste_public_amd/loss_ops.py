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


class MultiPosConFn(torch.autograd.Function):
    """fused multi-positive contrastive loss (csrc/losses.hip): feats [N,D] and all_feats [M,D] are the L2-normalised
    local / all-gathered features; mask_labels [N] / all_labels [M] decide the positives, row_weights [N] (or None)
    scale the rows.  No N x M tensor is formed."""

    @staticmethod
    def forward(ctx, feats, all_feats, mask_labels, all_labels, row_weights, self_offset, temperature):
        lib = _lib.load()
        f, a = feats.detach().float().contiguous(), all_feats.detach().float().contiguous()
        lf, la = mask_labels.detach().long().contiguous(), all_labels.detach().long().contiguous()
        rw = row_weights.detach().float().contiguous() if row_weights is not None else None
        N, D = f.shape
        M = a.shape[0]
        dev = f.device
        work = torch.empty(lib.creste_multipos_con_workspace_bytes(N, M, D), dtype=torch.uint8, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(lib.creste_multipos_con_forward_f32(f.data_ptr(), a.data_ptr(), lf.data_ptr(), la.data_ptr(),
                                                       rw.data_ptr() if rw is not None else None, N, M, D,
                                                       int(self_offset), float(temperature), loss.data_ptr(),
                                                       work.data_ptr(), _stream()), "multipos_con_forward")
        ctx.saved = (f, a, lf, la, rw, work, int(self_offset), float(temperature))
        return loss[0]

    @staticmethod
    def backward(ctx, gl):
        f, a, lf, la, rw, work, off, T = ctx.saved
        lib = _lib.load()
        gf, ga = torch.empty_like(f), torch.empty_like(a)
        _lib.check(lib.creste_multipos_con_backward_f32(f.data_ptr(), a.data_ptr(), lf.data_ptr(), la.data_ptr(),
                                                        rw.data_ptr() if rw is not None else None, f.shape[0], a.shape[0],
                                                        f.shape[1], off, T, 1.0, work.data_ptr(), gf.data_ptr(),
                                                        ga.data_ptr(), _stream()), "multipos_con_backward")
        return gf * gl, ga * gl, None, None, None, None, None
