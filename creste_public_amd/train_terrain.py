"""TerrainNet in training mode (BEV-SSC step, reference train_ssc.py:92-129 -> creste/models/terrainnet.py:272-350):
three autograd Functions chained by ordinary tensors --

    BackboneFn (train_backbone.py)   rgbd -> depth logits / metric depth / features / DINO features
    SplatFn    (this file)           (metric depth, features, p2p) -> bev_features, bev_densities   [Camera2MapMulti]
    BevHeadFn  (train_bev.py)        bev_features -> 3 x (preds, features)                           [ResNet-18 heads]

so gradients of any loss on the BEV predictions flow through the splat into BOTH the features and, via the
bilinear tap weights and the softmax-expectation depth, the depth logits -- as autograd does in the reference
(`splat is differentiable in feats AND in xy->depth`).
"""
from __future__ import annotations

import torch

from . import _lib, ops
from .ops import Act, HipLibraryError, _stream
from .train_backbone import Seq, _stack_ops, backbone_forward_train, sample_reduce, tpoint
from .train_bev import bev_heads_forward_train
from .train_ops import as_act, grad_slot


def _lib_():
    return _lib.load()


class SplatTrainEngine:
    """Camera2MapMulti: pixel geometry + z-MLP -> 1x1 fusion conv + BatchNorm(train) + ReLU -> range mask -> splat."""

    def __init__(self, cam2map):
        self.m = cam2map
        self.fusion = Seq(_stack_ops(cam2map.vision_fusion.convs))
        self.gen = 0

    def params(self):
        z = self.m.z_proj
        return [z[0].weight, z[0].bias, z[2].weight, z[2].bias] + self.fusion.params()

    def forward(self, depth: torch.Tensor, feats: Act, p2p: torch.Tensor, mv_mask: torch.Tensor = None):
        """mv_mask [B*Hs*Ws] (or None): the immovable-object mask of the reference's `_mv` pass, multiplied into the
        range mask (splat_projection.py:214-219)."""
        m = self.m
        if m.NC != 1 or m.scatter_mode != "mean" or m.mode != "bilinear":
            raise NotImplementedError("HIP splat: single camera, bilinear, mean (the shipped config)")
        g = m._geo.get()
        F = feats.C
        fbuf = m.fusion_buffer(feats.N, feats.H, feats.W, F, depth.device)
        fbuf.buf[..., :F].copy_(feats.buf[..., feats.co:feats.co + F])          # memory move into the concat buffer
        xyz, mask = ops.pixel_geometry(depth, p2p, g["bounds"], g["w1"], g["b1"], g["w2"], g["b2"], fbuf.slice(F, m.z_dim))
        fused = self.fusion.fwd(Act(fbuf.buf, fbuf.cs, 0))
        self.mask = mask.view(-1, 1)
        if mv_mask is not None:
            self.mask = self.mask * mv_mask.reshape(-1, 1).to(self.mask.dtype)
        masked = tpoint(2, fused, gate=self.mask, hw=1)                          # per-pixel range mask
        gh, gw = g["grid"]
        coords, bev, dens = ops.bev_splat(xyz, masked, g["off"], g["vox"], gh, gw, m.min_weight)
        self.saved = dict(depth=depth, p2p=p2p, masked=masked, coords=coords, bev=bev, dens=dens, F=F, g=g)
        return bev, dens, coords

    def backward(self, g_bev: Act, g_dens, grads):
        s, m = self.saved, self.m
        g = s["g"]
        g_masked, g_xyz = ops.bev_splat_bwd(s["coords"], s["masked"], g_bev, g_dens, s["bev"], s["dens"], g["vox"],
                                            m.min_weight)
        g_fused = tpoint(2, g_masked, gate=self.mask, hw=1)
        g_fbuf = self.fusion.bwd(g_fused, grads, True)
        F, Z = s["F"], m.z_dim
        g_zf = g_fbuf.slice(F, Z)
        depth, p2p = s["depth"], s["p2p"]
        B, Hs, Ws = depth.shape
        P, dev = B * Hs * Ws, depth.device
        zhid = g["w1"].numel()
        g_depth = torch.empty_like(depth)
        gq = torch.empty((P, Z), device=dev)
        ghp = torch.empty((P, zhid), device=dev)
        hbuf = torch.empty((P, zhid), device=dev)
        zbuf = torch.empty((P, 1), device=dev)
        lib = _lib_()
        _lib.check(lib.creste_pixel_geometry_bwd_f32(depth.data_ptr(), p2p.data_ptr(), B, Hs, Ws, g["w1"].data_ptr(),
                                                     g["b1"].data_ptr(), g["w2"].data_ptr(), g["b2"].data_ptr(), zhid, Z,
                                                     g_xyz.data_ptr(), g_zf.ptr, g_zf.cs, g_depth.data_ptr(),
                                                     gq.data_ptr(), ghp.data_ptr(), hbuf.data_ptr(), zbuf.data_ptr(),
                                                     _stream()), "pixel_geometry_bwd")
        if grads is not None:
            z = m.z_proj

            def fc_wgrad(xbuf, gybuf, w):                    # a 1x1 "conv" over the pixel list: gw[o][i] = sum_p gy[p][o] x[p][i]
                O, I = w.shape
                gw, acc = grad_slot(grads, w)
                work = torch.empty(lib.creste_conv_wgrad_strided_workspace_bytes(1, 1, P, I, O, 1), dtype=torch.uint8,
                                   device=dev)
                _lib.check(lib.creste_conv_wgrad_strided_f32(xbuf.data_ptr(), I, gybuf.data_ptr(), O, gw.data_ptr(), 1, 1,
                                                             P, 1, P, I, O, 1, 1, 0, 0, acc, work.data_ptr(), _stream()),
                           "fc_wgrad")
            fc_wgrad(hbuf, gq, z[2].weight)
            fc_wgrad(zbuf, ghp, z[0].weight)
            for buf, b in ((gq, z[2].bias), (ghp, z[0].bias)):
                gb, _ = grad_slot(grads, b)
                gb.copy_(sample_reduce(Act(buf.view(1, 1, P, buf.shape[1]), buf.shape[1]), None, 1.0,
                                       per_sample=False).view(-1))
        return g_depth, g_fbuf.slice(0, F)


class SplatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, depth, feats, p2p, mv_mask, *params):
        eng.gen += 1
        ctx.eng, ctx.gen = eng, eng.gen
        ctx.set_materialize_grads(False)
        bev, dens, coords = eng.forward(depth.detach().float().contiguous(), as_act(feats),
                                        p2p.detach().float().contiguous(),
                                        mv_mask.detach().float().contiguous() if mv_mask is not None else None)
        ctx.mark_non_differentiable(coords)
        return bev.nchw(), dens.unsqueeze(1), coords

    @staticmethod
    def backward(ctx, g_bev, g_dens, _):
        eng = ctx.eng
        if ctx.gen != eng.gen:
            raise RuntimeError("splat (HIP training path): backward of a stale forward")
        if g_bev is None:
            g_bev = torch.zeros_like(eng.saved["bev"].nchw())
        gb = as_act(g_bev)
        if gb.cs != gb.C or gb.co != 0:
            gb = Act(gb.buf[..., gb.co:gb.co + gb.C].contiguous(), gb.C, 0)
        gd = g_dens.detach().float().reshape(eng.saved["dens"].shape).contiguous() if g_dens is not None else None
        grads = {}
        g_depth, g_feats = eng.backward(gb, gd, grads)
        ops.wgrad_join()
        return (None, g_depth, g_feats.nchw(), None, None, *(grads.get(id(p)) for p in eng.params()))


def splat_forward_train(cam2map, depth, feats, p2p, mv_mask=None, slot="_train_engine"):
    """Camera2MapMulti in training mode (BatchNorm on batch statistics) as ONE autograd node; `slot` names the engine
    instance -- a second forward whose backward is still pending (the `_mv` pass) needs its own saved state."""
    eng = getattr(cam2map, slot, None)
    if eng is None:
        eng = SplatTrainEngine(cam2map)
        setattr(cam2map, slot, eng)
    return SplatFn.apply(eng, depth, feats, p2p, mv_mask, *eng.params())


def terrainnet_forward_train(model, rgbd: torch.Tensor, p2p: torch.Tensor, mv_mask: torch.Tensor = None) -> dict:
    """TerrainNet.forward in training mode -> the reference's output dict, every float output autograd-connected.
    `use_movability` (terrainnet.py:310-344): the anchor splat, then a second splat with the immovable-object mask
    (`bev_*_mv`) and a second pass of the BEV heads on it -- whose un-suffixed `inpainting_sam_dynamic_*` /
    `elevation_*` entries REPLACE the first pass's (only the `inpainting_sam` prefix takes the suffix,
    inpainting.py:41-44)."""
    if not rgbd.is_cuda:
        raise HipLibraryError("TerrainNet training runs on the HIP kernels only (got a CPU tensor)")
    B, N, Cc, H, W = rgbd.shape
    if N != 1 or model.views != 1:
        raise NotImplementedError("HIP pipeline: one view per sample (views=1, the shipped config)")
    out = backbone_forward_train(model.depthcomp, rgbd)
    p2p_ = p2p.reshape(B * N, 4, 4)
    bev, dens, coords = splat_forward_train(model.cam2map, out["depth_preds_metric"], out["depth_preds_feats"], p2p_)
    out.update({"bev_features": bev, "bev_densities": dens, "bev_coords": coords})
    bev_mv = None
    if model.use_movability and mv_mask is not None:
        bev_mv, dens_mv, coords_mv = splat_forward_train(model.cam2map, out["depth_preds_metric"],
                                                         out["depth_preds_feats"], p2p_, mv_mask, "_train_engine_mv")
        out.update({"bev_features_mv": bev_mv, "bev_densities_mv": dens_mv, "bev_coords_mv": coords_mv})
    if model.bevclassifier is not None:
        heads = bev_heads_forward_train(model.bevclassifier, bev)
        out.update(model.bevclassifier._wrap([dict(preds=p, features=f) for p, f in heads]))
        if model.use_movability:
            if bev_mv is None:
                raise KeyError("bev_features_mv: use_movability needs the immovable mask as third input")
            heads = bev_heads_forward_train(model.bevclassifier, bev_mv, slot="_train_engine_mv")
            out.update(model.bevclassifier._wrap([dict(preds=p, features=f) for p, f in heads], "_mv"))
    return out
