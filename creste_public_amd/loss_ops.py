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


class BevCEFn(torch.autograd.Function):
    """CrossEntropy over BEV cells (reference loss_utils.py:379-474): pred [B,C,H,W], gt [B,Cg,H,W] (class index in
    channel `class_dim`, or a distribution when class_dim < 0), fov [B,H,W] bool -> (loss, stats[4] = loss, accuracy,
    weight sum, labelled count); one fused HIP op (csrc/losses.hip) that also leaves dloss/dpred."""

    @staticmethod
    def forward(ctx, pred, gt, fov, class_weights, class_dim, ignore_index, eps):
        if not pred.is_cuda:
            raise HipLibraryError("BEV cross-entropy runs on the HIP kernels only")
        pa = as_act(pred)
        B, C, H, W = pred.shape
        g_ = gt.detach().float().contiguous()
        f_ = fov.detach().to(torch.uint8).contiguous()
        cw = class_weights.detach().float().contiguous() if class_weights is not None else None
        dev = pred.device
        g = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
        out4 = torch.empty(4, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().creste_bev_ce_loss_f32(
            pa.ptr, pa.cs, C, g_.data_ptr(), g_.shape[1], H * W, B * H * W, f_.data_ptr(),
            cw.data_ptr() if cw is not None else None, int(class_dim), -1000000 if ignore_index is None else int(ignore_index),
            float(eps), 1.0, g.data_ptr(), C, out4.data_ptr(), _work(dev).data_ptr(), _stream()), "bev_ce_loss")
        ctx.g = g.permute(0, 3, 1, 2)
        ctx.mark_non_differentiable(out4)
        return out4[0].clone(), out4

    @staticmethod
    def backward(ctx, gl, _):
        return ctx.g * gl, None, None, None, None, None, None


class SmoothL1Fn(torch.autograd.Function):
    """kind 0: elevation smooth-L1 (reference loss_utils.py:576-603; pred [B,2,H,W], gt [B,2,H,W]);
    kind 1: metric-depth smooth-L1 (:530-573; pred [N,H,W] m, gt [N,H,W] mm).  Fused HIP op with gradient."""

    @staticmethod
    def forward(ctx, kind, pred, gt, absolute, beta, depth_min, depth_max, num_bins):
        if not pred.is_cuda:
            raise HipLibraryError("smooth-L1 runs on the HIP kernels only")
        dev = pred.device
        gt_ = gt.detach().float().contiguous()
        out2 = torch.empty(2, dtype=torch.float32, device=dev)
        lib = _lib.load()
        if kind == 0:
            pa = as_act(pred)
            B, C, H, W = pred.shape
            if C != 2 or tuple(gt_.shape) != (B, 2, H, W):
                raise HipLibraryError("SmoothL1 (elevation): [B,2,H,W] prediction and label expected")
            g = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
            _lib.check(lib.creste_smooth_l1_loss_f32(0, pa.ptr, pa.cs, gt_.data_ptr(), H * W, B * H * W, int(bool(absolute)),
                                                     float(beta), 0.0, 1.0, 0, 1.0, g.data_ptr(), 2, out2.data_ptr(),
                                                     _work(dev).data_ptr(), _stream()), "smooth_l1_loss")
            ctx.g = g.permute(0, 3, 1, 2)
        else:
            p_ = pred.detach().float().contiguous()
            if p_.shape != gt_.shape:
                raise HipLibraryError("SmoothL1Depth: prediction and label shapes differ")
            g = torch.empty_like(p_)
            _lib.check(lib.creste_smooth_l1_loss_f32(1, p_.data_ptr(), 1, gt_.data_ptr(), 1, p_.numel(), 0, float(beta),
                                                     float(depth_min), float(depth_max), int(num_bins), 1.0, g.data_ptr(),
                                                     1, out2.data_ptr(), _work(dev).data_ptr(), _stream()), "smooth_l1_loss")
            ctx.g = g
        return out2[0].clone()

    @staticmethod
    def backward(ctx, gl):
        return None, ctx.g * gl, None, None, None, None, None, None
