"""Training-mode execution of the RGB-D backbone (stage-1 distillation) on the HIP kernels.

Reference: train_pefree.py:71-99 (`DistillationModel.training_step`) -> creste/models/distillation.py:145-207
-> depth.py:102-133 -> blocks/effnet.py:82-97 (+ the third-party efficientnet_pytorch MBConv blocks) with the
objectives of configs/model/distillation/effnet_ds2_dinov2_128.yaml:72-88 (CrossEntropyDepth, SmoothL1Depth,
MSELoss; loss_utils.py:477-573, 606-647).

`DistillationBackbone.train()` forward = `BackboneFn` (one autograd Function): the primal pass runs the layer
ops below and keeps what their backward needs; `backward` receives the cotangents of `depth_preds_logits`,
`depth_preds_feats` and `dino_pe_feats` and walks the ops in reverse, returning one gradient per parameter.
Convolutions use the forward conv engines (dgrad = the same engine on flipped, channel-transposed weights),
the MFMA wgrad kernels, training-mode BatchNorm (csrc/train.hip) and the depthwise / squeeze-excite /
swish kernels of csrc/train_backbone.hip.  Nothing here computes with torch ops (torch = memory, RNG for
drop-connect, autograd bookkeeping).
"""
from __future__ import annotations

import contextlib

import torch
from torch import nn

from . import _lib, hipnn, ops
from .ops import Act, HipLibraryError, _stream
from .train_ops import BNT, UpT, _amax_slot, _new, _px, as_act, pointwise2
from .train_ops import grad_slot as _acc

DROP_CONNECT = 0.2          # efficientnet_pytorch global_params.drop_connect_rate of efficientnet-b0


WGRAD_WINOGRAD, WGRAD_WINOGRAD_MIN_C = True, 256      # weight gradients of the wide 3x3 convs through F(4x4,3x3) (bf16x6 grade)


def _lib_():
    return _lib.load()


def tpoint(op: int, a: Act, b: Act | None = None, gate: torch.Tensor | None = None, r: torch.Tensor | None = None,
           out: Act | None = None, hw: int | None = None) -> Act:
    """creste_train_pointwise_f32 (swish / swish' / per-sample gate ops); hw=1 turns the per-sample gate into a
    per-pixel one (the splat's range mask)."""
    out = out or _new(a)
    HW = hw or a.H * a.W
    out.amax = _amax_slot(a.buf.device)
    _lib.check(_lib_().creste_train_pointwise_f32(
        op, a.ptr, a.cs, b.ptr if b is not None else None, b.cs if b is not None else 0,
        gate.data_ptr() if gate is not None else None, gate.shape[1] if gate is not None else 0,
        r.data_ptr() if r is not None else None, out.ptr, out.cs, HW, _px(a), a.C,
        out.amax.data_ptr() if out.amax is not None else None, _stream()), "train_pointwise")
    return out


def sample_reduce(a: Act, b: Act | None, scale: float, per_sample=True) -> torch.Tensor:
    """[N,C] (or [1,C] over the whole batch) channel sums of a (* b)."""
    lib = _lib_()
    N, HW = (a.N, a.H * a.W) if per_sample else (1, _px(a))
    out = torch.empty((N, a.C), dtype=torch.float32, device=a.buf.device)
    work = torch.empty(lib.creste_sample_reduce_workspace_bytes(N, a.C), dtype=torch.uint8, device=a.buf.device)
    _lib.check(lib.creste_sample_reduce_f32(a.ptr, a.cs, b.ptr if b is not None else None,
                                            b.cs if b is not None else 0, out.data_ptr(), N, HW, a.C, float(scale),
                                            work.data_ptr(), _stream()), "sample_reduce")
    return out


def add(a: Act | None, b: Act | None) -> Act | None:
    if a is None:
        return b
    if b is None:
        return a
    return pointwise2(2, a, b)


WGRAD_STREAM_MIN = 1 << 22          # pixels x channels below which a weight gradient stays on the backward's stream


class ConvG:
    """dense conv, any stride / static padding, optional bias (forward: conv engine; dgrad: stride 1 only)."""

    def __init__(self, conv: nn.Conv2d, pad=None):
        k = conv.kernel_size[0]
        if conv.kernel_size != (k, k) or conv.stride[0] != conv.stride[1] or conv.groups != 1 or conv.dilation != (1, 1):
            raise NotImplementedError("HIP training path: square dense convs only")
        if pad is None:
            pad = getattr(conv, "static_pad", None) or (conv.padding[0], conv.padding[0], conv.padding[1], conv.padding[1])
        self.conv, self.K, self.s, self.pad = conv, k, conv.stride[0], tuple(pad)
        self._fw = self._bw = self._key = None

    def params(self):
        return [self.conv.weight] + ([self.conv.bias] if self.conv.bias is not None else [])

    def _prec(self):
        p = hipnn._precision
        return ops.conv_precision(p, self.K, self.s, self.conv.in_channels)

    def _packed(self, want_bw):
        w, b = self.conv.weight, self.conv.bias
        key = (w.data_ptr(), w._version, b._version if b is not None else None, self._prec(), ops.CACHE_EPOCH)
        if key != self._key:
            self._fw = ops.pack_conv(w, b, None, self.s, self.pad, ops.ACT_NONE, self._prec())
            self._bw, self._key = None, key
        if want_bw and self._bw is None:
            Cout, Cin = w.shape[:2]
            cpad = (Cout + 3) // 4 * 4                   # the conv engine wants Cin % 4 == 0: zero input channels
            wt = torch.empty((Cin, cpad, self.K, self.K), dtype=torch.float32, device=w.device)
            _lib.check(_lib_().creste_conv_flip_weight_f32(w.detach().contiguous().data_ptr(), wt.data_ptr(), Cout, Cin,
                                                           self.K, cpad, _stream()), "conv_flip_weight")
            t, b_, l, r = self.pad
            k1 = self.K - 1
            prec = ops.conv_precision(hipnn._precision, self.K, 1, cpad)
            # stride s: the same stride-1 conv, applied to the zero-inserted cotangent; the far-side pads then depend
            # on the input extent (how many trailing rows the strided conv never reached) and are set per call
            self._bw = ops.pack_conv(wt, None, None, 1, (k1 - t, k1 - b_, k1 - l, k1 - r), ops.ACT_NONE, prec)
        return self._fw, self._bw

    def fwd(self, x: Act, out=None, want_stats=False) -> Act:
        self.x = x
        return ops.conv2d(x, self._packed(False)[0], out=out, want_stats=want_stats)

    def bwd(self, gy: Act, grads, need_input=True):
        lib = _lib_()
        w = self.conv.weight
        Cout, Cin = w.shape[:2]
        x = self.x
        if grads is not None:
            gw, acc = _acc(grads, w)
            side = ops.wgrad_stream(w.device) if x.N * x.H * x.W * max(Cin, Cout) >= WGRAD_STREAM_MIN else None
            f16_wgrad = (hipnn._precision == ops.PREC_F16X3 and Cin % 4 == 0 and Cout % 4 == 0 and Cin >= 8 and Cout >= 8
                         and x.cs % 4 == 0 and gy.cs % 4 == 0 and x.co % 4 == 0 and gy.co % 4 == 0)
            if side is not None and f16_wgrad:
                # the |max| bounds are CACHED on the Acts and read again by this stream's input-gradient conv: they are
                # made here, on the backward's stream (the side stream waits for it below), never on the side stream
                # (the Winograd-domain weight gradient below does not read them: the cost is one small pass)
                ops.absmax(x), ops.absmax(gy)
            if side is not None:
                # the weight gradient reads gy and the saved x and nothing of this backward reads IT: on the side stream,
                # behind everything issued so far, while this stream goes on with the input gradient
                side[0].wait_stream(torch.cuda.current_stream(w.device))
                side[1] = True
                for t in (gy.buf, x.buf, gw):
                    t.record_stream(side[0])
            with (torch.cuda.stream(side[0]) if side is not None else contextlib.nullcontext()):
                quads = (Cin % 4 == 0 and Cout % 4 == 0 and x.cs % 4 == 0 and gy.cs % 4 == 0 and x.co % 4 == 0 and gy.co % 4 == 0)
                if (WGRAD_WINOGRAD and hipnn._precision in (ops.PREC_BF16X6, ops.PREC_F16X3) and quads and self.pad[0] == 1 and self.pad[2] == 1
                        and min(Cin, Cout) >= WGRAD_WINOGRAD_MIN_C
                        and lib.creste_conv_wgrad_wino4_supported(self.K, self.s, x.H, x.W, gy.H, gy.W, Cin, Cout)):
                    # wide 3x3 (bf16x6, and f16x3 -- at the wider bf16x6 grade): through the F(4x4,3x3) transform, 4x fewer
                    # matrix products (csrc/conv_wino4.hip)
                    work = torch.empty(lib.creste_conv_wgrad_wino4_workspace_bytes(x.N, x.H, x.W, Cin, Cout), dtype=torch.uint8,
                                       device=w.device)
                    _lib.check(lib.creste_conv_wgrad_wino4(x.ptr, x.cs, gy.ptr, gy.cs, gw.data_ptr(), x.N, x.H, x.W, Cin, Cout,
                                                           self.pad[0], self.pad[2], acc, work.data_ptr(), _stream()),
                               "conv_wgrad_wino4")
                else:
                    work = torch.empty(lib.creste_conv_wgrad_strided_workspace_bytes(gy.N, gy.H, gy.W, Cin, Cout, self.K),
                                       dtype=torch.uint8, device=w.device)
                    if f16_wgrad:
                        # f16x3 wgrad: fp16 hi+lo operands from the tensors' |max| bounds, fp32 accumulation
                        _lib.check(lib.creste_conv_wgrad_f16x3(x.ptr, x.cs, gy.ptr, gy.cs, gw.data_ptr(),
                                                               ops.absmax(x).data_ptr(), ops.absmax(gy).data_ptr(), x.N, x.H,
                                                               x.W, gy.H, gy.W, Cin, Cout, self.K, self.s, self.pad[0],
                                                               self.pad[2], acc, work.data_ptr(), _stream()), "conv_wgrad_f16x3")
                    else:
                        # bf16x6: the wide 3x3 convs at the forward's operand grade (six bf16 piece products), the rest exact fp32
                        fn = lib.creste_conv_wgrad_bf16x6 if hipnn._precision == ops.PREC_BF16X6 else lib.creste_conv_wgrad_strided_f32
                        _lib.check(fn(x.ptr, x.cs, gy.ptr, gy.cs, gw.data_ptr(), x.N, x.H, x.W,
                                      gy.H, gy.W, Cin, Cout, self.K, self.s, self.pad[0],
                                      self.pad[2], acc, work.data_ptr(), _stream()),
                                   "conv_wgrad_strided")
            if self.conv.bias is not None:               # (a small reduction: stays on the backward's stream)
                gb, accb = _acc(grads, self.conv.bias)
                s = sample_reduce(gy, None, 1.0, per_sample=False).view(-1)
                if accb:
                    gb += s          # noqa (bias gradients are accumulated at most once per step)
                else:
                    gb.copy_(s)
        if not need_input:
            return None
        bw = self._packed(True)[1]
        if gy.C != bw.Cin:                               # Cout % 4 != 0 (e.g. the 2- and 6-class projections)
            buf = torch.zeros((gy.N, gy.H, gy.W, bw.Cin), dtype=torch.float32, device=gy.buf.device)
            pointwise2(2, gy, Act(buf, gy.C, 0), out=Act(buf, gy.C, 0))
            gy = Act(buf, bw.Cin, 0)
        if self.s == 1:
            return ops.conv2d(gy, bw)
        gz = Act.empty(gy.N, (gy.H - 1) * self.s + 1, (gy.W - 1) * self.s + 1, gy.C, gy.buf.device)
        _lib.check(lib.creste_zero_insert_nhwc_f32(gy.ptr, gy.cs, gz.ptr, gy.N, gy.H, gy.W, gy.C, self.s, _stream()),
                   "zero_insert")
        k1 = self.K - 1
        import dataclasses
        pt, pl = k1 - self.pad[0], k1 - self.pad[2]
        bw = dataclasses.replace(bw, pad_t=pt, pad_l=pl, pad_b=x.H - gz.H - pt + k1, pad_r=x.W - gz.W - pl + k1)
        return ops.conv2d(gz, bw)


# BatchNorm -> swish pairs as ONE forward and ONE backward op (the activation inside the BatchNorm kernels, its derivative
# recomputed from x in the backward): z is never stored and two streaming passes per pair go away
FUSE_BN_SWISH = True


def _swish():
    return [] if FUSE_BN_SWISH else [SwishT()]


class SwishT:
    def params(self):
        return []

    def fwd(self, x: Act, out=None) -> Act:
        self.x = x
        return tpoint(0, x, out=out)

    def bwd(self, gy: Act, grads, need_input=True):
        return tpoint(1, self.x, gy)


class BN:
    """training-mode BatchNorm (train_ops.BNT) with a forward/backward-only interface."""

    def __init__(self, bn: nn.BatchNorm2d, relu=False, swish=False):
        self.op = BNT(bn, 2 if swish else relu)

    def params(self):
        return self.op.params()

    def fwd(self, x, out=None):
        return self.op.fwd(x, out=out)

    def bwd(self, gy, grads, need_input=True):
        return self.op.bwd(gy, None, grads)[0]


class DwConvT:
    """depthwise KxK conv with efficientnet's static 'same' padding."""

    def __init__(self, conv):
        self.conv, self.K, self.s = conv, conv.kernel_size[0], conv.stride[0]
        self.pad = tuple(conv.static_pad)
        self._w = self._key = None

    def params(self):
        return [self.conv.weight]

    def _taps(self):
        w = self.conv.weight
        key = (w.data_ptr(), w._version, ops.CACHE_EPOCH)
        if key != self._key:
            Cn = w.shape[0]
            self._w = w.detach()[:, 0].reshape(Cn, -1).t().contiguous()          # [K*K, C] tap-major
            self._wflip = self._w.flip(0).contiguous()                           # taps of the stride-1 input gradient
            self._zero = torch.zeros(Cn, dtype=torch.float32, device=w.device)
            self._key = key
        return self._w

    def fwd(self, x: Act, out=None) -> Act:
        self.x = x
        w = self._taps()
        return ops.dwconv2d(x, w, self._zero, self.K, self.s, self.pad, ops.ACT_NONE)

    def bwd(self, gy: Act, grads, need_input=True):
        lib, x, w = _lib_(), self.x, self.conv.weight
        Cn = x.C
        if grads is not None:
            gt = torch.empty((self.K * self.K, Cn), dtype=torch.float32, device=w.device)
            work = torch.empty(lib.creste_dwconv_wgrad_workspace_bytes(Cn, self.K), dtype=torch.uint8, device=w.device)
            _lib.check(lib.creste_dwconv_wgrad_f32(x.ptr, gy.ptr, gt.data_ptr(), x.N, x.H, x.W, Cn, gy.H, gy.W, self.K,
                                                   self.s, self.pad[0], self.pad[2], 0, work.data_ptr(), _stream()),
                       "dwconv_wgrad")
            dst, acc = _acc(grads, w)
            if acc:
                raise RuntimeError("depthwise weight gradient written twice in one backward")
            dst.copy_(gt.t().reshape(w.shape))                                  # layout change back to [C,1,K,K]
        if not need_input:
            return None
        if self.s == 1 and ops.DW_TILE and Cn >= ops.DW_TILE_MIN_C and gy.cs == gy.C and gy.co == 0 \
                and (gy.H, gy.W) == (x.H, x.W):
            # stride 1: the input gradient is the depthwise conv of gy with the flipped taps and pad' = K - 1 - pad
            self._taps()
            k1 = self.K - 1
            return ops.dwconv2d(gy, self._wflip, None, self.K, 1, (k1 - self.pad[0], k1 - self.pad[1], k1 - self.pad[2],
                                                                    k1 - self.pad[3]), ops.ACT_NONE)
        gx = _new(x)
        _lib.check(lib.creste_dwconv_dgrad_f32(gy.ptr, self._taps().data_ptr(), gx.ptr, x.N, x.H, x.W, Cn, gy.H, gy.W,
                                               self.K, self.s, self.pad[0], self.pad[2], _stream()), "dwconv_dgrad")
        return gx


class SET:
    """squeeze-excite: y = x * sigmoid(W2 swish(W1 mean_hw(x) + b1) + b2)."""

    def __init__(self, reduce: nn.Conv2d, expand: nn.Conv2d):
        self.r, self.e = reduce, expand
        self.Cse = reduce.out_channels

    def params(self):
        return [self.r.weight, self.r.bias, self.e.weight, self.e.bias]

    def fwd(self, x: Act, out=None) -> Act:
        lib, dev = _lib_(), x.buf.device
        self.x = x
        self.s = sample_reduce(x, None, 1.0 / (x.H * x.W))
        self.hpre = torch.empty((x.N, self.Cse), device=dev)
        self.hact = torch.empty((x.N, self.Cse), device=dev)
        self.gate = torch.empty((x.N, x.C), device=dev)
        _lib.check(lib.creste_se_fc_forward_f32(self.s.data_ptr(), self.r.weight.data_ptr(), self.r.bias.data_ptr(),
                                                self.e.weight.data_ptr(), self.e.bias.data_ptr(), self.hpre.data_ptr(),
                                                self.hact.data_ptr(), self.gate.data_ptr(), x.N, x.C, self.Cse,
                                                _stream()), "se_fc_forward")
        return tpoint(2, x, gate=self.gate, out=out)

    def bwd(self, gy: Act, grads, need_input=True):
        lib, x, dev = _lib_(), self.x, self.x.buf.device
        gg = sample_reduce(gy, x, 1.0)                                    # d loss / d gate
        gz = torch.empty_like(self.gate)
        ghp = torch.empty_like(self.hpre)
        gs = torch.empty_like(self.s)
        _lib.check(lib.creste_se_fc_backward_f32(gg.data_ptr(), self.gate.data_ptr(), self.hpre.data_ptr(),
                                                 self.r.weight.data_ptr(), self.e.weight.data_ptr(), gz.data_ptr(),
                                                 ghp.data_ptr(), gs.data_ptr(), x.N, x.C, self.Cse, _stream()),
                   "se_fc_backward")
        if grads is not None:
            gw2, a2 = _acc(grads, self.e.weight)
            gb2, _ = _acc(grads, self.e.bias)
            _lib.check(lib.creste_fc_wgrad_f32(gz.data_ptr(), self.hact.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), x.N,
                                               x.C, self.Cse, a2, _stream()), "fc_wgrad")
            gw1, a1 = _acc(grads, self.r.weight)
            gb1, _ = _acc(grads, self.r.bias)
            _lib.check(lib.creste_fc_wgrad_f32(ghp.data_ptr(), self.s.data_ptr(), gw1.data_ptr(), gb1.data_ptr(), x.N,
                                               self.Cse, x.C, a1, _stream()), "fc_wgrad")
        return tpoint(4, x, gy, gate=self.gate, r=gs)


class Seq:
    def __init__(self, op_list):
        self.ops = op_list

    def params(self):
        return [p for o in self.ops for p in o.params()]

    def fwd(self, x, out=None):
        for i, o in enumerate(self.ops):
            last = i == len(self.ops) - 1
            if isinstance(o, ConvG) and not last and isinstance(self.ops[i + 1], BN):
                # the BatchNorm that follows takes its batch statistics from this conv's epilogue (ops.conv2d want_stats)
                x = o.fwd(x, want_stats=True)
            else:
                x = o.fwd(x, out=out if last else None)
        return x

    def bwd(self, gy, grads, need_input=True):
        for i in range(len(self.ops) - 1, -1, -1):
            gy = self.ops[i].bwd(gy, grads, need_input=need_input or i > 0)
            if hasattr(grads, "done"):
                grads.done(self.ops[i].params())
        return gy

    def backward_order(self):
        return [p for o in reversed(self.ops) for p in o.params()]


class MBConvT:
    """expand 1x1 + BN + swish -> depthwise + BN + swish -> squeeze-excite -> project 1x1 + BN (+ drop-connect, skip)."""

    def __init__(self, blk, rate: float):
        body = []
        if blk.has_expand:
            body += [ConvG(blk._expand_conv, pad=(0, 0, 0, 0)), BN(blk._bn0, swish=FUSE_BN_SWISH), *_swish()]
        body += [DwConvT(blk._depthwise_conv), BN(blk._bn1, swish=FUSE_BN_SWISH), *_swish(), SET(blk._se_reduce, blk._se_expand),
                 ConvG(blk._project_conv, pad=(0, 0, 0, 0)), BN(blk._bn2)]
        self.body = Seq(body)
        self.skip = blk.s == 1 and blk.cin == blk.cout
        self.rate = rate if self.skip else 0.0

    def params(self):
        return self.body.params()

    def fwd(self, x: Act) -> Act:
        y = self.body.fwd(x)
        if not self.skip:
            return y
        if self.rate:
            keep = 1.0 - self.rate
            u = torch.rand([x.N, 1, 1, 1], dtype=torch.float32)              # host RNG, as the reference's CPU path
            self.mask = (torch.floor(keep + u) / keep).view(x.N, 1).to(x.buf.device, non_blocking=True)
        else:
            self.mask = torch.ones((x.N, 1), dtype=torch.float32, device=x.buf.device)
        return tpoint(3, y, x, gate=self.mask)                                # x + y * mask[n]

    def bwd(self, g: Act, grads, need_input=True):
        if not self.skip:
            return self.body.bwd(g, grads, need_input)
        gb = tpoint(2, g, gate=self.mask)
        gin = self.body.bwd(gb, grads, True)
        return add(g, gin)


class UpBlockT:
    """cat([skip, bilinear_up(x1)]) -> 2 x (3x3 conv + BN + ReLU)  (reference effnet.py:8-28)."""

    def __init__(self, up):
        self.up = UpT(up.up)
        self.convs = Seq([ConvG(up.conv[0]), BN(up.conv[1], relu=True), ConvG(up.conv[3]), BN(up.conv[4], relu=True)])

    def params(self):
        return self.convs.params()

    def fwd(self, x1: Act, skip: Act) -> Act:
        Ho, Wo = self.up._out_size(x1.H, x1.W, self.up.sf)
        assert (Ho, Wo) == (skip.H, skip.W)
        self.up.in_hw = (x1.H, x1.W)
        self.c1, self.c2 = x1.C, skip.C
        cat = ops.upsample_concat(x1, skip, Ho, Wo, self.up.r[0], self.up.r[1])
        return self.convs.fwd(cat)

    def bwd(self, gy: Act, grads):
        gcat = self.convs.bwd(gy, grads, True)
        g_skip = gcat.slice(0, self.c2)
        g_x1 = self.up._t(gcat.slice(self.c2, self.c1))
        return g_x1, g_skip


def _stack_ops(seq: nn.Sequential):
    """[Conv2d, (BatchNorm2d), ReLU]* (MultiLayerConv heads)."""
    mods, out, i = list(seq), [], 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            out.append(ConvG(m))
        elif isinstance(m, nn.BatchNorm2d):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            out.append(BN(m, relu=fuse))
            i += int(fuse)
        elif isinstance(m, nn.ReLU):
            from .train_ops import ReLUT

            class _R(ReLUT):
                def bwd(self, gy, grads, need_input=True):
                    return ReLUT.bwd(self, gy, None, grads)[0]
            out.append(_R())
        else:
            raise NotImplementedError(type(m).__name__)
        i += 1
    return out


class BackboneTrainEngine:
    """DistillationBackbone (depthcomp.vision_backbone EffNet + depth head + dino head) in training mode."""

    def __init__(self, model):
        dc = model.depthcomp
        if dc.vision_backbone.input_type != "rgbd":
            raise NotImplementedError("HIP training path: rgbd encoder input")
        eff = dc.vision_backbone.model
        if eff.apply_final_batch_norm:
            raise NotImplementedError("apply_final_batch_norm")
        tr = eff.trunk
        self.stem = Seq([ConvG(tr._conv_stem), BN(tr._bn0, swish=FUSE_BN_SWISH), *_swish()])
        n = len(tr._blocks)
        self.blocks = [MBConvT(b, DROP_CONNECT * float(i) / n) for i, b in enumerate(tr._blocks)]
        self.ups = [UpBlockT(getattr(eff, f"up{i}")) for i in range(1, eff.n_ups + 1)]
        self.final = ConvG(eff.conv)
        self.depth_head = Seq(_stack_ops(dc.depth_head.model))
        self.dino_head = Seq(_stack_ops(model.dino_head.model))
        self.model = model
        self.gen = 0
        self.arena = False            # True: gradients live in one flat buffer, all-reduced bucket by bucket during backward
        self.bucket_bytes = 32 << 20

    def params(self):
        ps = self.stem.params()
        for b in self.blocks:
            ps += b.params()
        for u in self.ups:
            ps += u.params()
        return ps + self.final.params() + self.depth_head.params() + self.dino_head.params()

    def backward_order(self):
        """parameters in the order the backward finishes them (layout of dist_utils.GradArena)."""
        order = self.depth_head.backward_order() + self.dino_head.backward_order() + list(self.final.params())
        for u in reversed(self.ups):
            order += u.convs.backward_order()
        for b in reversed(self.blocks):
            order += b.body.backward_order()
        return order + self.stem.backward_order()

    def new_grad_store(self):
        if self.arena:
            from .dist_utils import GradArena
            return GradArena(self.backward_order(), self.bucket_bytes)
        return {}

    def forward(self, x: Act):
        h = self.stem.fwd(x)
        outs = [h]                                       # outs[i + 1] = output of block i
        for b in self.blocks:
            h = b.fwd(h)
            outs.append(h)
        # endpoints as efficientnet_pytorch.extract_endpoints: the last map before each resolution drop + the last
        ends, n = [], len(self.blocks)
        for i in range(1, n + 1):
            if outs[i - 1].H > outs[i].H:
                ends.append(i - 1)
            elif i == n:
                ends.append(i)
        self.ends, self.n_out = ends, len(outs)
        h = outs[ends[4]]
        for j, u in enumerate(self.ups):
            h = u.fwd(h, outs[ends[3 - j]])
        feats = self.final.fwd(h)
        logits = self.depth_head.fwd(feats)
        dino = self.dino_head.fwd(feats)
        return logits, feats, dino

    def backward(self, g_logits, g_feats, g_dino, grads):
        gf = g_feats
        if g_logits is not None:
            gf = add(gf, self.depth_head.bwd(g_logits, grads, True))
        if g_dino is not None:
            gf = add(gf, self.dino_head.bwd(g_dino, grads, True))
        if gf is None:
            return
        g = self.final.bwd(gf, grads, True)
        if hasattr(grads, "done"):
            grads.done(self.final.params())
        pending = {}                                      # output index -> cotangent from a decoder skip
        for j in range(len(self.ups) - 1, -1, -1):
            g, g_skip = self.ups[j].bwd(g, grads)
            pending[self.ends[3 - j]] = g_skip
        # g is now the cotangent of outs[ends[4]] (= the last block's output)
        cur, idx = g, self.ends[4]
        for i in range(len(self.blocks) - 1, -1, -1):     # block i maps outs[i] -> outs[i + 1]
            if i + 1 > idx:
                continue
            if i + 1 in pending and i + 1 != idx:
                cur = add(cur, pending.pop(i + 1))
            cur = self.blocks[i].bwd(cur, grads, True)
        if 0 in pending:
            cur = add(cur, pending.pop(0))
        self.stem.bwd(cur, grads, need_input=False)


class BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, rgbd, *params):
        x = ops.nchw_to_nhwc(rgbd.detach().contiguous().float())
        eng.gen += 1
        ctx.eng, ctx.gen = eng, eng.gen
        logits, feats, dino = eng.forward(x)
        depth, bins = ops.depth_expectation(logits, eng.model.depthcomp._bin_values(x.buf.device))
        ctx.mark_non_differentiable(bins)
        ctx.logits = logits
        return logits.nchw(), feats.nchw(), dino.nchw(), depth, bins

    @staticmethod
    def backward(ctx, g_logits, g_feats, g_dino, _gd, _gb):   # _gd: cotangent of depth_preds_metric
        eng = ctx.eng
        if ctx.gen != eng.gen:
            raise RuntimeError("backbone (HIP training path): backward of a stale forward; run forward/backward in pairs")
        grads = eng.new_grad_store()
        a = lambda t: as_act(t) if t is not None else None       # noqa: E731
        gl = a(g_logits)
        if _gd is not None:                                       # metric depth = softmax expectation of the logits
            if gl is not None and (gl.cs != gl.C or not gl.buf.is_contiguous()):
                gl = Act(gl.nchw().permute(0, 2, 3, 1).contiguous(), gl.C, 0)
            elif gl is not None:
                gl = Act(gl.buf.clone(), gl.C, 0)                 # accumulated in place below: do not alias autograd's
            gl = ops.depth_expectation_bwd(ctx.logits, eng.model.depthcomp._bin_values(ctx.logits.buf.device),
                                           _gd.detach().float().contiguous(), g_logits=gl)
        eng.backward(gl, a(g_feats), a(g_dino), grads)
        ops.wgrad_join(ctx.logits.buf.device)
        if hasattr(grads, "finish"):
            grads.finish()                                        # tail bucket, wait, average over ranks
        return (None, None, *(grads.get(id(p)) for p in eng.params()))


def backbone_forward_train(model, rgbd: torch.Tensor) -> dict:
    """DistillationBackbone.forward in training mode: rgbd [B,V=1,4,H,W] -> the reference's output dict."""
    if not rgbd.is_cuda:
        raise HipLibraryError("backbone training runs on the HIP kernels only (got a CPU tensor)")
    B, V, Cc, H, W = rgbd.shape
    eng = getattr(model, "_train_engine", None)
    if eng is None:
        eng = model._train_engine = BackboneTrainEngine(model)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d) and not m.training:
            raise NotImplementedError("mixed train/eval BatchNorm inside the backbone is not on the HIP training path")
    logits, feats, dino, depth, bins = BackboneFn.apply(eng, rgbd.reshape(B * V, Cc, H, W), *eng.params())
    out = {"depth_preds_logits": logits, "depth_preds_metric": depth, "depth_preds_bins": bins}
    if model.depthcomp.return_feats:
        out["depth_preds_feats"] = feats
    out["dino_pe_feats"] = dino.unsqueeze(1)
    return out
