"""Training-mode execution of the IRL reward network on the HIP kernels (csrc/train.hip + the fp32 conv engine).

The reference trains `MultiScaleFCN` (creste/models/blocks/conv.py:88-161) through the MaxEnt / counterfactual
IRL objective (creste/utils/loss_utils.py:1118-1259), which differentiates the reward TWICE: the gradient
penalty is a function of g = d(sum r)/d(input_view) (`torch.autograd.grad(..., create_graph=True)`, :1207-1217).
Here the network is one `torch.autograd.Function` pair, so the loss code stays the reference's torch code:

  RewardFn        forward  = primal pass (training-mode BatchNorm: batch statistics, running stats updated)
                  backward = cotangent gr on r  ->  parameter gradients (wgrad / BN reductions) and the input
                             gradient g, the latter through InputGradFn so that it stays differentiable
  InputGradFn     forward  = g = J_x^T gr      (dgrad convs, BN backward, pool/upsample transposes)
                  backward = cotangent u on g  ->  d<u, J_x^T gr>/d theta = d<gr, J_x u>/d theta:
                             one TANGENT forward with xd = u, then one backward through the (primal, tangent)
                             pair with cotangent gr on the tangent output.  No generic double backward.

Every op below implements fwd / tan / bwd on NHWC `Act`s; `bwd(gy, gyd)` takes the cotangents of the primal
and tangent outputs (either may be None) and accumulates parameter gradients into `grads[param]`.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib, ops
from .ops import Act, HipLibraryError, _stream


def _lib_():
    return _lib.load()


def _new(like: Act, Cn=None, H=None, W=None) -> Act:
    return Act.empty(like.N, H or like.H, W or like.W, Cn or like.C, like.buf.device)


def _px(a: Act) -> int:
    return a.N * a.H * a.W


def _amax_slot(dev):
    """A zeroed device float for a producer to raise to max|out| -- only in the f16x3 operand mode, where the convs
    consuming the tensor need the bound (otherwise None: the kernels skip the reduction)."""
    from . import hipnn
    return ops._AmaxPool.slot(dev) if (hipnn._precision == ops.PREC_F16X3 or ops.TRACK_AMAX) else None


def as_act(t: torch.Tensor, pad_to4=False) -> Act:
    """[B,C,H,W] tensor (any strides) -> NHWC Act, zero-copy when it already is a view of an NHWC buffer."""
    if not t.is_cuda:
        raise HipLibraryError("reward-network training runs on the HIP kernels only (got a CPU tensor)")
    t = t.detach().float()
    B, Cc, H, W = t.shape
    if pad_to4 and Cc % 4:
        buf = torch.zeros((B, H, W, (Cc + 3) // 4 * 4), dtype=torch.float32, device=t.device)
        ops.nchw_to_nhwc(t.contiguous(), out=Act(buf, Cc, 0))
        return Act(buf, buf.shape[3], 0)
    p = t.permute(0, 2, 3, 1)
    if p.is_contiguous():
        return Act(p, Cc, 0)
    return ops.nchw_to_nhwc(t.contiguous())


def grad_slot(grads: dict, p: torch.Tensor):
    """-> (tensor the gradient of parameter p is written into, accumulate flag).  `grads` is a plain dict keyed
    by id(param) or a dist_utils.GradArena (then the tensor is a slice of the flat all-reduce buffer)."""
    acc = id(p) in grads
    if not acc:
        if hasattr(grads, "view"):
            grads.view(p)
        else:
            grads[id(p)] = torch.empty_like(p, memory_format=torch.contiguous_format)
    return grads[id(p)], int(acc)


def pointwise2(op: int, a: Act, b: Act | None, out: Act | None = None) -> Act:
    out = out or _new(a)
    out.amax = None                            # rewritten without tracking: a cached bound would be stale
    _lib.check(_lib_().creste_pointwise2_f32(op, a.ptr, a.cs, b.ptr if b is not None else None,
                                             b.cs if b is not None else 0, out.ptr, out.cs, _px(a), a.C, _stream()),
               "pointwise2")
    return out


# True: in the f16x3 operand mode (fp32 operands as fp16 hi+lo, product error <= 2^-21) the reward network's convs -- forward,
# tangent, both input gradients and the weight gradients -- run on the f16 matrix cores like the backbone's (parity tests
# pass).  Off by default: measured, the IRL step does not get faster (reference config 31.2 -> 31.9 ms, 256x256 MDP grid
# 56.9 -> 56.1 ms: the reward network is launch- and BatchNorm-bound, not MFMA-bound), so it keeps exact fp32 products.
REWARD_FOLLOWS_F16X3 = False
# bf16x6 (fp32-equivalent products, no operand bounds to track): the reward network's forward / tangent / input-gradient
# convs follow the backbone's mode -- its 5x5 convs run on the three-piece row kernel since round 3
REWARD_FOLLOWS_BF16X6 = True


class ConvT:
    """stride-1 'same' conv without bias: y = W * x."""

    def __init__(self, conv: nn.Conv2d):
        k = conv.kernel_size[0]
        if conv.bias is not None or conv.stride != (1, 1) or conv.kernel_size != (k, k) or \
                conv.padding != (k // 2, k // 2) or conv.groups != 1 or conv.dilation != (1, 1):
            raise NotImplementedError("HIP training path: stride-1 'same' bias-free convs (reward network) only")
        self.conv, self.K = conv, k
        self._fw = self._bw = None
        self._key = None

    def params(self):
        return [self.conv.weight]

    def _prec(self, cin):
        from . import hipnn
        if REWARD_FOLLOWS_F16X3 and hipnn._precision == ops.PREC_F16X3:
            return ops.conv_precision(ops.PREC_F16X3, self.K, 1, cin)
        if REWARD_FOLLOWS_BF16X6 and hipnn._precision == ops.PREC_BF16X6:
            return ops.conv_precision(ops.PREC_BF16X6, self.K, 1, cin)
        return ops.PREC_F32

    def _packed(self):
        w = self.conv.weight
        Cout, Cin = w.shape[:2]
        pf, pb = self._prec(Cin), self._prec((Cout + 3) // 4 * 4)
        key = (w.data_ptr(), w._version, ops.CACHE_EPOCH, pf, pb)
        if key != self._key:
            self._fw = ops.pack_conv(w, None, None, 1, self.K // 2, ops.ACT_NONE, pf)
            cpad = (Cout + 3) // 4 * 4
            wt = torch.empty((Cin, cpad, self.K, self.K), dtype=torch.float32, device=w.device)
            _lib.check(_lib_().creste_conv_flip_weight_f32(w.detach().contiguous().data_ptr(), wt.data_ptr(), Cout,
                                                           Cin, self.K, cpad, _stream()), "conv_flip_weight")
            self._bw = ops.pack_conv(wt, None, None, 1, self.K // 2, ops.ACT_NONE, pb)
            self._key = key
        return self._fw, self._bw

    def fwd(self, x: Act, out=None) -> Act:
        self.x = x
        return ops.conv2d(x, self._packed()[0], out=out)

    def tan(self, xd: Act, out=None) -> Act:
        self.xd = xd
        return ops.conv2d(xd, self._packed()[0], out=out)

    def _dgrad(self, gy: Act) -> Act:
        bw = self._packed()[1]
        if gy.C != bw.Cin:                       # Cout not a multiple of 4: zero-padded channel copy
            buf = torch.zeros((gy.N, gy.H, gy.W, bw.Cin), dtype=torch.float32, device=gy.buf.device)
            pointwise2(2, gy, Act(buf, gy.C, 0), out=Act(buf, gy.C, 0))
            gy = Act(buf, bw.Cin, 0)
        return ops.conv2d(gy, bw)

    def _wgrad(self, x: Act, gy: Act, grads):
        w = self.conv.weight
        Cout, Cin = w.shape[:2]
        lib = _lib_()
        _, acc = grad_slot(grads, w)
        if (self._prec(Cin) == ops.PREC_F16X3 and Cin % 4 == 0 and Cout % 4 == 0 and Cin >= 8 and Cout >= 8
                and x.cs % 4 == 0 and gy.cs % 4 == 0 and x.co % 4 == 0 and gy.co % 4 == 0):
            work = torch.empty(lib.creste_conv_wgrad_strided_workspace_bytes(x.N, x.H, x.W, Cin, Cout, self.K),
                               dtype=torch.uint8, device=w.device)
            _lib.check(lib.creste_conv_wgrad_f16x3(x.ptr, x.cs, gy.ptr, gy.cs, grads[id(w)].data_ptr(),
                                                   ops.absmax(x).data_ptr(), ops.absmax(gy).data_ptr(), x.N, x.H, x.W,
                                                   x.H, x.W, Cin, Cout, self.K, 1, self.K // 2, self.K // 2, int(acc),
                                                   work.data_ptr(), _stream()), "conv_wgrad_f16x3")
            return
        work = torch.empty(lib.creste_conv_wgrad_workspace_bytes(x.N, x.H, x.W, Cin, Cout, self.K),
                           dtype=torch.uint8, device=w.device)
        _lib.check(lib.creste_conv_wgrad_f32(x.ptr, x.cs, gy.ptr, gy.cs, grads[id(w)].data_ptr(), x.N, x.H, x.W, Cin,
                                             Cout, self.K, self.K // 2, int(acc), work.data_ptr(), _stream()),
                   "conv_wgrad")

    def bwd(self, gy, gyd, grads, need_input=True):
        if grads is not None:
            if gy is not None:
                self._wgrad(self.x, gy, grads)
            if gyd is not None:
                self._wgrad(self.xd, gyd, grads)
        if not need_input:
            return None, None
        return (self._dgrad(gy) if gy is not None else None), (self._dgrad(gyd) if gyd is not None else None)


class BNT:
    """training-mode BatchNorm2d (+ fused ReLU on the primal output)."""

    def __init__(self, bn: nn.BatchNorm2d, relu: bool):
        if not isinstance(bn, nn.BatchNorm2d) or bn.momentum is None or not bn.affine or not bn.track_running_stats:
            raise NotImplementedError("HIP training path: affine BatchNorm2d with momentum and running stats only")
        # relu: False / True, or the activation code of creste_bn_train_forward_f32 (2 = swish, first order only)
        self.bn, self.act = bn, int(relu)
        self.relu = self.act == 1

    def params(self):
        return [self.bn.weight, self.bn.bias]

    def _work(self, dev):
        return torch.empty(_lib_().creste_bn_workspace_bytes(self.bn.num_features), dtype=torch.uint8, device=dev)

    def fwd(self, x: Act, out=None) -> Act:
        bn, dev = self.bn, x.buf.device
        Cn = bn.num_features
        self.x = x
        self.mean = torch.empty(Cn, device=dev)
        self.invstd = torch.empty(Cn, device=dev)
        var = torch.empty(Cn, device=dev)
        y = out or _new(x)
        y.amax = _amax_slot(dev)               # max|y| for an f16x3 conv consuming it (None outside that mode)
        st = getattr(x, "stats", None)
        if st is not None and st[0].shape[2] == Cn and x.co == 0:
            # the producing conv left per-workgroup channel sums of x: no statistics pass over the tensor
            _lib.check(_lib_().creste_bn_train_forward_stats_f32(
                x.ptr, x.cs, _px(x), Cn, bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), float(bn.momentum),
                bn.running_mean.data_ptr(), bn.running_var.data_ptr(), self.mean.data_ptr(), self.invstd.data_ptr(),
                var.data_ptr(), y.ptr, y.cs, self.act, y.amax.data_ptr() if y.amax is not None else None,
                st[0].data_ptr(), st[1], _stream()), "bn_train_forward_stats")
            bn.num_batches_tracked += 1
            self.y = y
            return y
        _lib.check(_lib_().creste_bn_train_forward_f32(
            x.ptr, x.cs, _px(x), Cn, bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), float(bn.momentum),
            bn.running_mean.data_ptr(), bn.running_var.data_ptr(), self.mean.data_ptr(), self.invstd.data_ptr(),
            var.data_ptr(), y.ptr, y.cs, self.act, y.amax.data_ptr() if y.amax is not None else None,
            self._work(dev).data_ptr(), _stream()), "bn_train_forward")
        bn.num_batches_tracked += 1
        self.y = y
        return y

    def tan(self, xd: Act, out=None) -> Act:
        if self.act == 2:
            raise NotImplementedError("BatchNorm + swish is built for first-order training (the backbone) only")
        bn, dev = self.bn, xd.buf.device
        self.xd = xd
        self.mom_t = torch.empty((2, bn.num_features), device=dev)
        yd = _new(xd) if (out is None or self.relu) else out
        _lib.check(_lib_().creste_bn_train_tangent_f32(
            self.x.ptr, self.x.cs, xd.ptr, xd.cs, _px(xd), bn.num_features, bn.weight.data_ptr(),
            self.mean.data_ptr(), self.invstd.data_ptr(), self.mom_t.data_ptr(), yd.ptr, yd.cs,
            self._work(dev).data_ptr(), _stream()), "bn_train_tangent")
        if self.relu:
            yd = pointwise2(1, self.y, yd, out=out)
        return yd

    def bwd(self, gy, gyd, grads, need_input=True):
        bn = self.bn
        dev = self.x.buf.device
        ref = gy if gy is not None else gyd
        gx = _new(ref)
        gx.amax = _amax_slot(dev)
        gxd = _new(ref) if gyd is not None else None
        mom_b = torch.empty((5, bn.num_features), device=dev)
        gg = gb = None
        acc = 0
        if grads is not None:
            gg, acc = grad_slot(grads, bn.weight)
            gb, _ = grad_slot(grads, bn.bias)
        has_t = gyd is not None
        if self.act == 2:                      # BatchNorm + swish: act'(z) from the recomputed z, no stored z
            if has_t:
                raise NotImplementedError("BatchNorm + swish is built for first-order training (the backbone) only")
            _lib.check(_lib_().creste_bn_act_train_backward_f32(
                2, self.x.ptr, self.x.cs, gy.ptr, gy.cs, _px(self.x), bn.num_features, bn.weight.data_ptr(),
                bn.bias.data_ptr(), self.mean.data_ptr(), self.invstd.data_ptr(), mom_b.data_ptr(), gx.ptr, gx.cs,
                gg.data_ptr() if gg is not None else None, gb.data_ptr() if gb is not None else None, acc,
                gx.amax.data_ptr() if gx.amax is not None else None, self._work(dev).data_ptr(), _stream()),
                "bn_act_train_backward")
            return gx, None
        # with the fused ReLU the kernels mask the cotangents themselves (the mask is recomputed from x with the forward's
        # expression): no separate pass over y and gy
        fn = _lib_().creste_bn_relu_train_backward_f32 if self.relu else _lib_().creste_bn_train_backward_f32
        extra = (bn.bias.data_ptr(),) if self.relu else ()
        _lib.check(fn(
            self.x.ptr, self.x.cs, self.xd.ptr if has_t else None, self.xd.cs if has_t else 0,
            gy.ptr if gy is not None else None, gy.cs if gy is not None else 0,
            gyd.ptr if has_t else None, gyd.cs if has_t else 0, _px(self.x), bn.num_features,
            bn.weight.data_ptr(), *extra, self.mean.data_ptr(), self.invstd.data_ptr(),
            self.mom_t.data_ptr() if has_t else None, mom_b.data_ptr(), gx.ptr, gx.cs,
            gxd.ptr if has_t else None, gxd.cs if has_t else 0,
            gg.data_ptr() if gg is not None else None, gb.data_ptr() if gb is not None else None, acc,
            gx.amax.data_ptr() if gx.amax is not None else None, self._work(dev).data_ptr(), _stream()),
            "bn_train_backward")
        return gx, gxd


class ReLUT:
    def params(self):
        return []

    def fwd(self, x, out=None):
        self.y = pointwise2(0, x, None, out=out)
        return self.y

    def tan(self, xd, out=None):
        return pointwise2(1, self.y, xd, out=out)

    def bwd(self, gy, gyd, grads, need_input=True):
        return (pointwise2(1, self.y, gy) if gy is not None else None,
                pointwise2(1, self.y, gyd) if gyd is not None else None)


class PoolT:
    """nn.MaxPool2d(2, 2) with the argmax kept for the tangent and the backward."""

    def params(self):
        return []

    def fwd(self, x, out=None):
        self.shape = (x.N, x.H, x.W, x.C)
        y = out or _new(x, H=x.H // 2, W=x.W // 2)
        y.amax = None                              # (re)written without tracking: a cached bound would be stale
        self.idx = torch.empty((x.N, x.H // 2, x.W // 2, x.C), dtype=torch.uint8, device=x.buf.device)
        _lib.check(_lib_().creste_maxpool2_idx_f32(x.ptr, x.cs, x.N, x.H, x.W, x.C, y.ptr, y.cs, self.idx.data_ptr(),
                                                   _stream()), "maxpool2_idx")
        return y

    def _route(self, backward, t: Act, out=None):
        N, H, W, Cn = self.shape
        o = out or (Act.empty(N, H, W, Cn, t.buf.device) if backward else Act.empty(N, H // 2, W // 2, Cn, t.buf.device))
        o.amax = None
        _lib.check(_lib_().creste_maxpool2_route_f32(int(backward), t.ptr, t.cs, self.idx.data_ptr(), o.ptr, o.cs, N,
                                                     H, W, Cn, _stream()), "maxpool2_route")
        return o

    def tan(self, xd, out=None):
        return self._route(False, xd, out)

    def bwd(self, gy, gyd, grads, need_input=True):
        return (self._route(True, gy) if gy is not None else None,
                self._route(True, gyd) if gyd is not None else None)


class UpT:
    """nn.Upsample(scale_factor, bilinear, align_corners=False)."""

    def __init__(self, up: nn.Upsample):
        if up.mode != "bilinear" or up.align_corners:
            raise NotImplementedError("HIP training path: bilinear align_corners=False upsampling only")
        from .hipnn import up_out_size, up_scales
        self.sf, self.r = up_scales(up.scale_factor)
        self._out_size = up_out_size

    def params(self):
        return []

    def _run(self, x, out):
        Ho, Wo = self._out_size(x.H, x.W, self.sf)
        self.in_hw = (x.H, x.W)
        return ops.upsample_concat(x, None, Ho, Wo, self.r[0], self.r[1], out=out)

    def fwd(self, x, out=None):
        return self._run(x, out)

    def tan(self, xd, out=None):
        return self._run(xd, out)

    def _t(self, gy: Act):
        H1, W1 = self.in_hw
        gx = Act.empty(gy.N, H1, W1, gy.C, gy.buf.device)
        _lib.check(_lib_().creste_upsample_bwd_nhwc_f32(gy.ptr, gy.cs, gy.H, gy.W, gx.ptr, gx.cs, gy.N, H1, W1, gy.C,
                                                        float(self.r[0]), float(self.r[1]), _stream()), "upsample_bwd")
        return gx

    def bwd(self, gy, gyd, grads, need_input=True):
        return (self._t(gy) if gy is not None else None, self._t(gyd) if gyd is not None else None)


class Chain:
    def __init__(self, op_list):
        self.ops = op_list

    def params(self):
        return [p for o in self.ops for p in o.params()]

    def fwd(self, x, out=None):
        for i, o in enumerate(self.ops):
            x = o.fwd(x, out=out if i == len(self.ops) - 1 else None)
        return x

    def tan(self, xd, out=None):
        for i, o in enumerate(self.ops):
            xd = o.tan(xd, out=out if i == len(self.ops) - 1 else None)
        return xd

    def bwd(self, gy, gyd, grads, need_input=True):
        for i in range(len(self.ops) - 1, -1, -1):
            gy, gyd = self.ops[i].bwd(gy, gyd, grads, need_input=need_input or i > 0)
        return gy, gyd


def _ops_of(mods) -> list:
    """nn modules of one MultiScaleFCN branch -> training ops (BatchNorm followed by ReLU is one op)."""
    flat = []
    for m in mods:
        flat += list(m) if isinstance(m, nn.Sequential) else [m]
    out, i = [], 0
    while i < len(flat):
        m = flat[i]
        nxt = flat[i + 1] if i + 1 < len(flat) else None
        if isinstance(m, nn.Conv2d):
            out.append(ConvT(m))
        elif isinstance(m, nn.BatchNorm2d):
            fuse = isinstance(nxt, nn.ReLU)
            out.append(BNT(m, fuse))
            i += int(fuse)
        elif isinstance(m, nn.ReLU):
            out.append(ReLUT())
        elif isinstance(m, nn.MaxPool2d):
            if m.kernel_size not in (2, (2, 2)) or m.stride not in (2, (2, 2)):
                raise NotImplementedError("HIP training path: 2x2/2 max-pool only")
            out.append(PoolT())
        elif isinstance(m, nn.Upsample):
            out.append(UpT(m))
        else:
            raise NotImplementedError(f"HIP training path: no training op for {type(m).__name__}")
        i += 1
    return out


class RewardTrainEngine:
    """prepool -> {trunk, skip} -> concat -> postpool of a MultiScaleFCN, in training mode."""

    def __init__(self, net):
        self.prepool = Chain(_ops_of(net.prepool))
        self.trunk = Chain(_ops_of(net.trunk))
        self.skip = Chain(_ops_of(net.skip))
        self.postpool = Chain(_ops_of(net.postpool))
        self.ct = net.trunk_cfg["dims"][-1]
        self.ccat = net.postpool_cfg["dims"][0]
        self.gen = 0          # forward counter: a backward must belong to the latest forward (activations live here)
        self.use_graphs = False
        self.phases = _Phases(self)

    def params(self):
        return self.prepool.params() + self.trunk.params() + self.skip.params() + self.postpool.params()

    def check_gen(self, gen):
        if gen != self.gen:
            raise RuntimeError("reward network (HIP training path): backward of a stale forward -- the saved "
                               "activations belong to a later forward of the same module; run forward/backward in pairs")

    def forward(self, x: Act) -> Act:
        h = self.prepool.fwd(x)
        cat = Act.empty(h.N, h.H, h.W, self.ccat, h.buf.device)
        self.trunk.fwd(h, out=cat.slice(0, self.ct))
        self.skip.fwd(h, out=cat.slice(self.ct, self.ccat - self.ct))
        return self.postpool.fwd(cat)

    def tangent(self, xd: Act) -> Act:
        hd = self.prepool.tan(xd)
        catd = Act.empty(hd.N, hd.H, hd.W, self.ccat, hd.buf.device)
        self.trunk.tan(hd, out=catd.slice(0, self.ct))
        self.skip.tan(hd, out=catd.slice(self.ct, self.ccat - self.ct))
        return self.postpool.tan(catd)

    def backward(self, gr, grd, grads, need_input=True):
        """cotangents of (r, rd) -> (gx, gxd); parameter gradients accumulate into `grads` (None: skip them)."""
        gc, gcd = self.postpool.bwd(gr, grd, grads)
        sl = lambda a, lo, n: a.slice(lo, n) if a is not None else None     # noqa: E731
        gt, gtd = self.trunk.bwd(sl(gc, 0, self.ct), sl(gcd, 0, self.ct), grads)
        gs, gsd = self.skip.bwd(sl(gc, self.ct, self.ccat - self.ct), sl(gcd, self.ct, self.ccat - self.ct), grads)
        gh = pointwise2(2, gt, gs) if gt is not None else None
        ghd = pointwise2(2, gtd, gsd) if gtd is not None else None
        return self.prepool.bwd(gh, ghd, grads, need_input=need_input)


def _grad_list(params, grads):
    return tuple(grads.get(id(p)) for p in params)


class _Phases:
    """The four launch sequences of one training step as tensor -> tensor functions, each optionally captured
    into a hipGraph (torch.cuda.CUDAGraph) on its second call and replayed afterwards.

    The reward network is ~280 kernels of a few microseconds each per step: eager, the step is bound by the
    Python launch path (~15 ms); replayed from four graphs it is bound by the GPU (~1 ms).  Capture needs
    static shapes (one graph set per input shape) and turns the outputs into static buffers that the next
    replay overwrites -- the usual CUDA-graph contract, hence opt-in (`MultiScaleFCN.train_graphs = True`,
    `IRLTrainer(graphs=True)`); results are bit-identical to the eager path (same kernels, same order)."""

    def __init__(self, eng):
        self.eng = eng
        self.graphs = {}

    # ---- the phases (eager bodies)
    def forward(self, x):
        eng = self.eng
        if eng.use_graphs:                       # the weights changed since the last step: re-pack inside the graph
            for ch in (eng.prepool, eng.trunk, eng.skip, eng.postpool):
                for o in ch.ops:
                    if isinstance(o, ConvT):
                        o._key = None
        return (eng.forward(as_act(x)).nchw(),)

    def input_grad(self, gr):
        gra = as_act(gr, pad_to4=True)
        gx, _ = self.eng.backward(Act(gra.buf, gr.shape[1], 0), None, None)
        return (gx.nchw(),)

    def second_order(self, u, gr):
        eng = self.eng
        rd = eng.tangent(as_act(u))
        gra = as_act(gr, pad_to4=True)
        grads = {}
        eng.backward(None, Act(gra.buf, gr.shape[1], 0), grads, need_input=False)
        return (rd.nchw(), *_grad_list(eng.params(), grads))

    def backward(self, gr):
        gra = as_act(gr, pad_to4=True)
        grads = {}
        self.eng.backward(Act(gra.buf, gr.shape[1], 0), None, grads, need_input=False)
        return _grad_list(self.eng.params(), grads)

    def backward_with_input(self, gr):
        gra = as_act(gr, pad_to4=True)
        grads = {}
        gx, _ = self.eng.backward(Act(gra.buf, gr.shape[1], 0), None, grads, need_input=True)
        return (gx.nchw(), *_grad_list(self.eng.params(), grads))

    # ---- dispatch
    def run(self, name, *ins):
        fn = getattr(self, name)
        if not self.eng.use_graphs:
            return fn(*ins)
        key = (name, tuple((tuple(t.shape), t.dtype) for t in ins))
        ent = self.graphs.get(key)
        if ent is None:                          # first call: eager (lazy kernel loads, function attributes)
            self.graphs[key] = "warm"
            return fn(*ins)
        if ent == "warm":
            static_in = [t.detach().clone() for t in ins]
            torch.cuda.synchronize()
            ops.reset_amax_pool()                # |max| slots taken inside the graph are zero-filled BY the graph ...
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = fn(*static_in)
            ops.reset_amax_pool()                # ... and nobody outside it shares their block
            ent = self.graphs[key] = (g, static_in, outs)
        g, static_in, outs = ent
        for st, t in zip(static_in, ins):
            st.copy_(t)
        g.replay()
        return outs


class InputGradFn(torch.autograd.Function):
    """g = J_x^T gr, differentiable w.r.t. the parameters (and gr) through the tangent pass."""

    @staticmethod
    def forward(ctx, eng, gr, *params):
        ctx.eng, ctx.gr, ctx.gen = eng, gr.detach(), eng.gen
        return eng.phases.run("input_grad", gr.detach())[0]

    @staticmethod
    def backward(ctx, u):
        eng = ctx.eng
        eng.check_gen(ctx.gen)
        rd, *pg = eng.phases.run("second_order", u.detach(), ctx.gr)
        return (None, rd, *pg)


def _params_wanted(params) -> bool:
    """True unless the running backward pass provably stops short of every parameter's accumulator."""
    probe = getattr(torch._C, "_will_engine_execute_node", None)
    if probe is None:
        return True
    try:
        for p in params:
            if p.requires_grad:
                node = p.view_as(p).grad_fn.next_functions[0][0]        # the AccumulateGrad node of the leaf
                if probe(node):
                    return True
        return False
    except Exception:
        return True


class RewardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, *params):
        ctx.eng = eng
        ctx.params = params
        eng.gen += 1
        ctx.gen = eng.gen
        return eng.phases.run("forward", x.detach())[0]

    @staticmethod
    def backward(ctx, gr):
        eng = ctx.eng
        eng.check_gen(ctx.gen)
        if torch.is_grad_enabled():
            # create_graph=True (the gradient penalty): the input gradient must remain a function of the
            # parameters; the parameter gradients of THIS call are first order (not differentiated again)
            # and are only computed when this backward pass will actually deliver them somewhere
            # (`autograd.grad(inputs=[input_view])` does not).
            gx = InputGradFn.apply(eng, gr, *ctx.params)
            pg = eng.phases.run("backward", gr.detach()) if _params_wanted(ctx.params) else (None,) * len(ctx.params)
        elif ctx.needs_input_grad[1]:
            gx, *pg = eng.phases.run("backward_with_input", gr.detach())
        else:
            gx, pg = None, eng.phases.run("backward", gr.detach())
        return (None, gx, *pg)


def reward_forward_train(net, x: torch.Tensor) -> torch.Tensor:
    """MultiScaleFCN.forward in training mode on the HIP path: x [B,C,H,W] -> r [B,1,H,W] (autograd-aware)."""
    eng = getattr(net, "_train_engine", None)
    if eng is None:
        eng = net._train_engine = RewardTrainEngine(net)
    eng.use_graphs = bool(getattr(net, "train_graphs", False))
    return RewardFn.apply(eng, x, *eng.params())
