"""One-process-per-GPU helpers (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo" on
CPU for tests).

* Inference shards FRAMES: every rank runs independent replicas, no data-path collective
  (`shard_range`, `max_over_ranks` for the bench's timing contract).
* The IRL training step exchanges only the reward network's gradients: 102,866 fp32 = 0.41 MB
  (reference: Lightning DDP, train_traversability.py:400).  At that size the
  all-reduce is latency-bound, so all gradients travel in ONE flat buffer / one collective instead of
  DDP's per-bucket calls.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def shard_range(total: int, rank: int, world: int):
    """Contiguous [lo, hi) share of `total` independent frames for `rank` (sizes differ by <= 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, device=None) -> float:
    if not is_dist():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not is_dist():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


@torch.no_grad()
def allreduce_mean_grads(params) -> int:
    """Average the .grad of `params` across ranks with a single flat all-reduce; parameters without a
    gradient on this rank contribute zeros (DDP find_unused_parameters semantics).  Returns the number
    of elements exchanged."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    dev, dt = params[0].device, params[0].dtype
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dt)
                      for p in params])
    if is_dist():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= dist.get_world_size()
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return int(flat.numel())
