"""Training-mode execution of the BEV completion network (InpaintingResNet18MultiHead, reference
creste/models/blocks/inpainting.py:9-109 with torchvision resnet18 layers) on the HIP kernels: 7x7/2 stem,
BasicBlocks with training-mode BatchNorm and residuals, three DeconvHeads (bilinear x4 + concat + 2 convs,
bilinear x2 + conv, 1x1 projection).  Forward/backward ops come from train_backbone.py / train_ops.py; strided
convs take their input gradient through the zero-inserted cotangent."""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .ops import Act
from .train_backbone import BN, ConvG, Seq, UpBlockT, add
from .train_ops import UpT, as_act, pointwise2


class ReLUAddT:
    """y = relu(a + b); backward: g * (y > 0) to both."""

    def fwd(self, a: Act, b: Act) -> Act:
        self.y = pointwise2(0, pointwise2(2, a, b), None)
        return self.y

    def bwd(self, gy: Act) -> Act:
        return pointwise2(1, self.y, gy)


class BasicBlockT:
    def __init__(self, blk):
        self.main = Seq([ConvG(blk.conv1), BN(blk.bn1, relu=True), ConvG(blk.conv2), BN(blk.bn2)])
        self.down = Seq([ConvG(blk.downsample[0]), BN(blk.downsample[1])]) if blk.downsample is not None else None
        self.out = ReLUAddT()

    def params(self):
        return self.main.params() + (self.down.params() if self.down else [])

    def backward_order(self):
        return self.main.backward_order() + (self.down.backward_order() if self.down else [])

    def fwd(self, x: Act) -> Act:
        idt = self.down.fwd(x) if self.down else x
        return self.out.fwd(self.main.fwd(x), idt)

    def bwd(self, gy: Act, grads) -> Act:
        g = self.out.bwd(gy)
        gx = self.main.bwd(g, grads, True)
        return add(gx, self.down.bwd(g, grads, True) if self.down else g)


class UpOnlyT:
    """nn.Upsample as a Seq element."""

    def __init__(self, up: nn.Upsample):
        self.up = UpT(up)

    def params(self):
        return []

    def fwd(self, x, out=None):
        return self.up.fwd(x, out=out)

    def bwd(self, gy, grads, need_input=True):
        return self.up._t(gy)


class DeconvHeadT:
    def __init__(self, head):
        self.up1 = UpBlockT(head.up1)
        self.up2 = Seq([UpOnlyT(head.up2[0]), ConvG(head.up2[1]), BN(head.up2[2], relu=True)])
        self.proj = ConvG(head.proj)

    def params(self):
        return self.up1.params() + self.up2.params() + self.proj.params()

    def backward_order(self):
        return list(self.proj.params()) + self.up2.backward_order() + self.up1.convs.backward_order()

    def fwd(self, x: Act, x1: Act):
        fea = self.up2.fwd(self.up1.fwd(x, x1))
        return self.proj.fwd(fea), fea

    def bwd(self, g_pred, g_fea, grads):
        g = add(g_fea, self.proj.bwd(g_pred, grads, True) if g_pred is not None else None)
        if hasattr(grads, "done"):
            grads.done(self.proj.params())
        if g is None:
            return None, None
        return self.up1.bwd(self.up2.bwd(g, grads, True), grads)        # (g_x, g_x1)


class BevHeadTrainEngine:
    def __init__(self, net):
        self.stem = Seq([ConvG(net.conv1), BN(net.bn1, relu=True)])
        self.l1 = [BasicBlockT(b) for b in net.layer1]
        self.l23 = [BasicBlockT(b) for b in list(net.layer2) + list(net.layer3)]
        self.heads = [DeconvHeadT(h) for h in net.out_heads]
        self.gen = 0
        self.arena, self.bucket_bytes = False, 32 << 20

    def params(self):
        ps = self.stem.params()
        for b in self.l1 + self.l23:
            ps += b.params()
        for h in self.heads:
            ps += h.params()
        return ps

    def backward_order(self):
        order = []
        for h in reversed(self.heads):
            order += h.backward_order()
        for b in reversed(self.l1 + self.l23):
            order += b.backward_order()
        return order + self.stem.backward_order()

    def new_grad_store(self):
        if self.arena:
            from .dist_utils import GradArena
            return GradArena(self.backward_order(), self.bucket_bytes)
        return {}

    def forward(self, bev: Act):
        x = self.stem.fwd(bev)
        for b in self.l1:
            x = b.fwd(x)
        x1 = x
        for b in self.l23:
            x = b.fwd(x)
        return [h.fwd(x, x1) for h in self.heads]                  # [(pred, features)] per head

    def backward(self, g_outs, grads, need_input=True):
        """g_outs: [(g_pred | None, g_features | None)] per head -> cotangent of the BEV input."""
        gx = gx1 = None
        for h, (gp, gf) in zip(reversed(self.heads), reversed(g_outs)):
            a, b = h.bwd(gp, gf, grads)
            gx, gx1 = add(gx, a), add(gx1, b)
        if gx is None:
            return None
        for b in reversed(self.l23):
            gx = b.bwd(gx, grads)
        g = add(gx, gx1)
        for b in reversed(self.l1):
            g = b.bwd(g, grads)
        return self.stem.bwd(g, grads, need_input=need_input)


class BevHeadFn(torch.autograd.Function):
    """bev_features [B,F,GH,GW] -> (pred_0, fea_0, pred_1, fea_1, ...)."""

    @staticmethod
    def forward(ctx, eng, bev, *params):
        eng.gen += 1
        ctx.eng, ctx.gen = eng, eng.gen
        ctx.set_materialize_grads(False)          # a head no loss consumes costs no backward work
        outs = eng.forward(as_act(bev))
        return tuple(t.nchw() for pair in outs for t in pair)

    @staticmethod
    def backward(ctx, *gouts):
        eng = ctx.eng
        if ctx.gen != eng.gen:
            raise RuntimeError("BEV heads (HIP training path): backward of a stale forward")
        a = lambda t: as_act(t) if t is not None else None           # noqa: E731
        pairs = [(a(gouts[2 * i]), a(gouts[2 * i + 1])) for i in range(len(eng.heads))]
        grads = eng.new_grad_store()
        g_in = eng.backward(pairs, grads, need_input=ctx.needs_input_grad[1])
        ops.wgrad_join()                                  # weight gradients issued on the side stream (train_backbone.ConvG.bwd)
        if hasattr(grads, "finish"):
            grads.finish()
        return (None, g_in.nchw() if g_in is not None else None, *(grads.get(id(p)) for p in eng.params()))


def bev_heads_forward_train(net, bev: torch.Tensor, slot: str = "_train_engine") -> list:
    """InpaintingResNet18MultiHead in training mode: [B,F,GH,GW] -> [(preds, features)] per head (autograd-aware).
    `slot`: attribute holding the engine -- a second pass before the first one's backward (`_mv`) needs its own."""
    eng = getattr(net, slot, None)
    if eng is None:
        eng = BevHeadTrainEngine(net)
        setattr(net, slot, eng)
    flat = BevHeadFn.apply(eng, bev, *eng.params())
    return [(flat[2 * i], flat[2 * i + 1]) for i in range(len(eng.heads))]
