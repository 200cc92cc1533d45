"""Device-side label bookkeeping of SupPixelConLoss (csrc/labels.hip): per-sample label remapping, the stable
class-wise grouping of the valid BEV cells and the row gather / scatter of the sampled embeddings.

reference: creste/utils/utils.py:59-77 (`remap_labels_in_batch`), creste/utils/train_utils.py:324-352
(`extract_max_per_class`), creste/utils/loss_utils.py:203-286.  The random choice inside an over-full class stays a
host-side `torch.randperm` drawn in class order from the default CPU generator, exactly as the reference draws it;
only (class, rank) pairs travel to the device.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .ops import HipLibraryError, _stream, fill_
from .train_ops import as_act


def _minmax(labels: torch.Tensor):
    out = torch.empty(2, dtype=torch.int64, device=labels.device)
    _lib.check(_lib.load().creste_label_minmax_i64(labels.data_ptr(), labels.numel(), out.data_ptr(), _stream()), "label_minmax")
    lo, hi = out.tolist()                                     # host sync: sizes the presence table
    return lo, hi


def remap_labels_in_batch(gt: torch.Tensor, ignore_idx: int = 0):
    """gt [B,...] int64 CUDA -> (remapped labels, K = largest new label + 1)."""
    if not gt.is_cuda or gt.dtype != torch.int64:
        raise HipLibraryError("remap_labels_in_batch (HIP): CUDA int64 labels expected")
    g = gt.contiguous()
    B, HW = g.shape[0], g[0].numel()
    lo, hi = _minmax(g)
    if lo < 0:
        raise HipLibraryError(f"remap_labels_in_batch: negative label {lo}")
    L = max(hi, ignore_idx) + 1
    table = torch.empty(B * (L + 2), dtype=torch.int32, device=g.device)
    out = torch.empty_like(g)
    nclass = torch.empty(1, dtype=torch.int32, device=g.device)
    _lib.check(_lib.load().creste_remap_labels_i64(g.data_ptr(), B, HW, int(ignore_idx), L, table.data_ptr(), out.data_ptr(),
                                                   nclass.data_ptr(), _stream()), "remap_labels")
    return out, nclass


def sample_cells_per_class(labels: torch.Tensor, fov: torch.Tensor | None, K: int, ignore_idx: int, cap: int = 1000):
    """labels [B,H,W] int64 (classes in [0,K)), fov [B,H,W] bool or None -> (cell [S] int32 flat (b,y,x) indices,
    sel_labels [S] int64): for every present class in ascending order its valid cells in row-major order, at most
    m = min(int(median of the non-zero class counts), cap) of them chosen by torch.randperm(count)[:m] on the host --
    loss_utils.py:262-266 + train_utils.py:324-352."""
    lib = _lib.load()
    dev = labels.device
    lab = labels.contiguous()
    n = lab.numel()
    f8 = fov.to(torch.uint8).contiguous() if fov is not None else None
    counts = torch.empty(K, dtype=torch.int32, device=dev)
    offsets = torch.empty(K + 1, dtype=torch.int32, device=dev)
    class_list = torch.empty(n, dtype=torch.int32, device=dev)
    work = torch.empty(max(1, lib.creste_group_by_class_workspace_bytes(n, K)), dtype=torch.uint8, device=dev)
    _lib.check(lib.creste_group_by_class_i64(lab.data_ptr(), f8.data_ptr() if f8 is not None else None, n, K, int(ignore_idx),
                                             counts.data_ptr(), offsets.data_ptr(), class_list.data_ptr(), work.data_ptr(),
                                             _stream()), "group_by_class")
    cnt = counts.cpu().numpy()                                # host sync: the reference needs the counts on the host too
    present = np.nonzero(cnt)[0]
    if present.size == 0:
        raise HipLibraryError("SupPixelConLoss: no valid labelled cell in the batch")
    nz = np.sort(cnt[present].astype(np.float32))
    m = min(int(nz[(nz.size - 1) // 2]), cap)                 # torch.median: the lower of the two middle values
    sel_cls, sel_rank = [], []
    for c in present:                                         # ascending classes, the reference's loop order
        k = int(cnt[c])
        r = torch.randperm(k)[:m].numpy() if k > m else np.arange(k)
        sel_rank.append(r.astype(np.int32))
        sel_cls.append(np.full(r.shape[0], c, dtype=np.int32))
    sel_cls, sel_rank = np.concatenate(sel_cls), np.concatenate(sel_rank)
    S = int(sel_cls.shape[0])
    d_cls = torch.from_numpy(sel_cls).to(dev)
    d_rank = torch.from_numpy(sel_rank).to(dev)
    cell = torch.empty(S, dtype=torch.int32, device=dev)
    _lib.check(lib.creste_pick_cells_i32(class_list.data_ptr(), offsets.data_ptr(), d_cls.data_ptr(), d_rank.data_ptr(), S,
                                         cell.data_ptr(), _stream()), "pick_cells")
    return cell, d_cls.long()


class RowsFn(torch.autograd.Function):
    """embeddings [B,Z,H,W] (an NHWC-strided prediction) -> rows [S,Z] at flat cell indices; the backward scatters the
    row cotangents into a zero-filled map (the picked cells are distinct)."""

    @staticmethod
    def forward(ctx, pred, cell):
        pa = as_act(pred)
        S, Z = cell.numel(), pa.C
        rows = torch.empty((S, Z), dtype=torch.float32, device=pred.device)
        _lib.check(_lib.load().creste_gather_rows_f32(pa.ptr, pa.cs, Z, cell.data_ptr(), S, rows.data_ptr(), _stream()),
                   "gather_rows")
        ctx.cell, ctx.shape = cell, (pa.N, pa.H, pa.W, Z)
        return rows

    @staticmethod
    def backward(ctx, g_rows):
        N, H, W, Z = ctx.shape
        g = fill_(torch.empty((N, H, W, Z), dtype=torch.float32, device=g_rows.device), 0.0)
        gr = g_rows.detach().float().contiguous()
        _lib.check(_lib.load().creste_scatter_rows_f32(gr.data_ptr(), ctx.cell.data_ptr(), ctx.cell.numel(), Z, g.data_ptr(), Z,
                                                       _stream()), "scatter_rows")
        return g.permute(0, 3, 1, 2), None
