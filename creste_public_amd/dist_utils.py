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

import os

import torch
import torch.distributed as dist


_collectives_enabled = True
STATS = {"bytes": 0, "calls": 0}      # payload bytes / calls of the gradient + feature collectives since reset_stats()


def init_rccl(device, **kw):
    """dist.init_process_group("nccl") with the collectives on a HIGH-PRIORITY stream.  Not for their latency: torch takes
    the process group's stream from its pool of default-priority streams otherwise, HIP maps that pool onto the same few
    hardware queues as everybody else's streams, and a collective that shares a queue with the backward's stream sits in
    FRONT of the backward's later kernels while it waits for the weight-gradient stream (measured on one rank: 12-22 ms per
    distillation / BEV-SSC step, depending only on how many streams had been created before).  High-priority streams have
    hardware queues of their own."""
    opts = None
    try:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    except Exception:                                      # (a torch without the option: default stream choice)
        opts = None
    return dist.init_process_group("nccl", device_id=device, **({"pg_options": opts} if opts is not None else {}), **kw)


def is_dist() -> bool:
    return _collectives_enabled and dist.is_available() and dist.is_initialized()


def reset_stats():
    STATS["bytes"], STATS["calls"] = 0, 0


def _count(t: torch.Tensor):
    STATS["bytes"] += t.numel() * t.element_size()
    STATS["calls"] += 1


class collectives:
    """`with collectives(False):` -- every data-path collective of this module (gradient all-reduces, the contrastive
    loss's all-gather) becomes the one-process no-op, under an initialised process group.  bench.py times a training step
    with and without its exchange this way (the difference = the exposed, i.e. not overlapped, communication time); the
    ranks' parameters diverge under it, so it is a measurement device only."""

    def __init__(self, enabled: bool):
        self.enabled = enabled

    def __enter__(self):
        global _collectives_enabled
        self.prev, _collectives_enabled = _collectives_enabled, self.enabled
        return self

    def __exit__(self, *exc):
        global _collectives_enabled
        _collectives_enabled = self.prev
        return False


def shard_range(total: int, rank: int, world: int):
    """Contiguous [lo, hi) share of `total` independent frames for `rank` (sizes differ by <= 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _pg() -> bool:       # bench bookkeeping reduces over ranks whether or not the data-path collectives are switched off
    return dist.is_available() and dist.is_initialized()


def max_over_ranks(value: float, device=None) -> float:
    if not _pg():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not _pg():
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
    if not is_dist():                 # one rank: nothing to average (and no flat copy / 300 per-parameter copies)
        return sum(p.numel() for p in params)
    dev, dt = params[0].device, params[0].dtype
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dt)
                      for p in params])
    _count(flat)
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


def _send_slice(arena, upto):
    """Hand flat[sent:upto] to the collective.  The conv weight gradients of the slice are written on the weight-gradient
    side stream (ops.wgrad_stream), which runs BEHIND the backward's stream, and torch's process group orders a collective
    behind the stream that is current when it is issued.  The collective is therefore issued from a third stream that holds
    nothing but two waits -- for what the backward's stream and for what the weight-gradient stream have issued so far -- so
    that neither of the two waits for the other.  Measured on one rank (distillation / BEV-SSC step, 2 / 5 buckets): exposed
    0.3 / 1.0 ms this way; 16 / 21-25 ms when the backward's stream waits for the side stream at every bucket (the side
    stream is ~15 ms behind, and the input-gradient chain is the critical path) -- and the same when the collective is
    issued from the side stream itself or one bucket late.  (CRESTE_COLL_ISSUE=join: the waiting form.)"""
    if upto > arena.sent and is_dist():
        from . import ops
        _count(arena.flat[arena.sent:upto])
        ctx = None
        if arena.flat.is_cuda:
            dev = arena.flat.device
            main = torch.cuda.current_stream(dev)
            s = ops._wgrad_streams.get((dev.index, main.cuda_stream))
            if s and s[1]:
                # the third stream is PROBED like every side stream: on the hardware queue of the backward's stream its wait
                # for the weight-gradient stream would sit in front of the backward's later kernels (measured: 5 / 13 ms)
                c = None if os.environ.get("CRESTE_COLL_ISSUE", "third") == "join" else ops.concurrent_stream(dev, "issue")
                if c is None:
                    ops.wgrad_join(dev)
                else:
                    c.wait_stream(main)
                    c.wait_stream(s[0])
                    ctx = torch.cuda.stream(c)
        if ctx is None:
            arena.handles.append(dist.all_reduce(arena.flat[arena.sent:upto], op=dist.ReduceOp.SUM, async_op=True))
        else:
            with ctx:
                arena.handles.append(dist.all_reduce(arena.flat[arena.sent:upto], op=dist.ReduceOp.SUM, async_op=True))
    arena.sent = max(arena.sent, upto)


class GradArena(dict):
    """Gradient store of one backward pass: every parameter's gradient is a view into ONE flat fp32 buffer laid
    out in the order in which the backward finishes them, so data-parallel averaging needs no gather/scatter
    copies and can start while the backward is still running.

    `order`: parameters in backward-completion order.  The backward calls `done(params)` after the op owning
    them has run; whenever the completed prefix crosses a bucket boundary the bucket goes out as an async
    all-reduce (RCCL on its own stream; ring all-reduce over xGMI is per-link bound, so a few 32 MB buckets --
    not hundreds of per-tensor calls -- keep the links busy while the remaining wgrad kernels run).
    `finish()` sends the tail, waits, and divides by the world size.  Without an initialised process group it
    is just the arena.  Keys are id(param) (what the training engines use)."""

    def __init__(self, order, bucket_bytes: int = 32 << 20, device=None):
        super().__init__()
        self.order = [p for p in order]
        dev = device or self.order[0].device
        self.offsets, n = {}, 0
        for p in self.order:
            self.offsets[id(p)] = (n, p.numel(), p.shape)
            n += p.numel()
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.views = {k: self.flat[o:o + c].view(shape) for k, (o, c, shape) in self.offsets.items()}
        self.pos = {id(p): i for i, p in enumerate(self.order)}
        self.finished = [False] * len(self.order)
        self.prefix = 0                      # number of leading parameters that are final
        self.sent = 0                        # elements already handed to the collective
        self.bucket = max(1, bucket_bytes // 4)
        self.handles = []
        self.touched = set()

    def view(self, p):
        """the arena slice of parameter p (registers it as written)."""
        self.touched.add(id(p))
        v = self.views[id(p)]
        self[id(p)] = v
        return v

    def _send(self, upto, last=False):
        _send_slice(self, upto)

    def done(self, params):
        for p in params:
            i = self.pos.get(id(p))
            if i is not None:
                if id(p) not in self.touched:            # no gradient reached it on this rank: contributes zeros
                    self.view(p).zero_()
                self.finished[i] = True
        while self.prefix < len(self.order) and self.finished[self.prefix]:
            self.prefix += 1
        end = self.offsets[id(self.order[self.prefix - 1])] if self.prefix else (0, 0, None)
        upto = end[0] + end[1]
        if upto - self.sent >= self.bucket:
            self._send(upto)

    def finish(self):
        self.done([p for p in self.order if not self.finished[self.pos[id(p)]]])
        self._send(self.flat.numel(), last=True)
        for h in self.handles:
            h.wait()
        if is_dist():
            self.flat /= dist.get_world_size()
        return self


class HookedArena:
    """GradArena for a backward that torch autograd drives (the BEV-SSC step: BackboneFn <- SplatFn <- BevHeadFn chained
    by ordinary tensors): every trainable parameter's `.grad` is a VIEW into one flat fp32 buffer laid out in reverse
    registration order (~ the order in which the backward finishes them: heads first, encoder stem last), and a
    post-accumulate-grad hook per parameter marks it final.  Whenever the finished PREFIX of the buffer has grown by a
    bucket, that slice goes out as an async all-reduce -- the collective is enqueued behind the kernels that produced
    the slice and runs on the communication stream while the rest of the backward (the encoder: ~60 ms of the step)
    is still computing.  `finish()` sends the tail, waits, and divides by the world size; parameters that received no
    gradient on this rank contribute zeros (the buffer is cleared by ONE memset per step).

    Ring all-reduce over point-to-point xGMI is per-link bound: a few large buckets (default 32 MB) keep all seven
    links busy; per-tensor calls (330 of them here) would be latency-bound.  A reduce-scatter + all-gather split
    (what a sharded optimiser wants) moves the same bytes per link as the ring all-reduce RCCL runs for these sizes,
    so with a replicated Adam state it buys nothing -- one collective per bucket it is."""

    def __init__(self, params, bucket_bytes: int = 32 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.order = list(reversed(self.params))
        dev = self.order[0].device
        n, self.span = 0, {}
        for p in self.order:
            self.span[id(p)] = (n, n + p.numel())
            n += p.numel()
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.pos = {id(p): i for i, p in enumerate(self.order)}
        self.bucket = max(1, bucket_bytes // 4)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.order]
        self.launched = 0                         # async collectives issued before finish() in the last step
        self.begin()

    def begin(self):
        """start of a step: clear the buffer (one memset) and re-point every .grad at its slice"""
        self.flat.zero_()
        for p in self.order:
            lo, hi = self.span[id(p)]
            p.grad = self.flat[lo:hi].view_as(p)
        self.finished = [False] * len(self.order)
        self.prefix = self.sent = 0
        self.handles = []

    def _send(self, upto, last=False):
        _send_slice(self, upto)

    @torch.no_grad()
    def _on_grad(self, p):
        lo, hi = self.span[id(p)]
        if p.grad.data_ptr() != self.flat.data_ptr() + 4 * lo:      # autograd replaced the view (first accumulation)
            self.flat[lo:hi].view_as(p).copy_(p.grad)
            p.grad = self.flat[lo:hi].view_as(p)
        self.finished[self.pos[id(p)]] = True
        while self.prefix < len(self.order) and self.finished[self.prefix]:
            self.prefix += 1
        upto = self.span[id(self.order[self.prefix - 1])][1] if self.prefix else 0
        if upto - self.sent >= self.bucket:
            self._send(upto)

    def finish(self):
        self.launched = len(self.handles)
        self._send(self.flat.numel(), last=True)
        for h in self.handles:
            h.wait()
        if is_dist():
            self.flat /= dist.get_world_size()
        else:
            # one process: a parameter no gradient reached keeps `.grad = None`, as plain autograd leaves it, so the
            # optimiser skips it (under a process group it contributes zeros to the average: DDP semantics)
            for p, fin in zip(self.order, self.finished):
                if not fin:
                    p.grad = None
        return self

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def gather_varlen(feats: torch.Tensor, labels: torch.Tensor):
    """All-gather rows whose COUNT differs between ranks (the per-class sample of the contrastive loss depends on the
    rank's data; the reference all-gathers the raw tensors, supcon_loss.py:85-86, which only works while every rank
    happens to draw the same count): counts are exchanged first, every rank pads to the maximum, the padded blocks
    are gathered -- features WITH gradient -- and the padding is dropped.  -> (all_feats [sum n, D], all_labels
    [sum n], offset of this rank's rows).  Without a process group: the inputs, offset 0."""
    if not is_dist():
        return feats, labels, 0
    from torch.distributed.nn import all_gather as all_gather_with_grad
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([feats.shape[0]], dtype=torch.int64, device=feats.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    pad = n_max - feats.shape[0]
    f = torch.nn.functional.pad(feats, (0, 0, 0, pad)) if pad else feats
    l = torch.nn.functional.pad(labels, (0, pad), value=-1) if pad else labels
    _count(f); _count(l)
    parts = all_gather_with_grad(f.contiguous())
    lab_parts = [torch.empty_like(l) for _ in range(world)]
    dist.all_gather(lab_parts, l.contiguous())
    all_feats = torch.cat([p_[:c] for p_, c in zip(parts, counts)], dim=0)
    all_labels = torch.cat([p_[:c] for p_, c in zip(lab_parts, counts)], dim=0)
    return all_feats, all_labels, sum(counts[:rank])


def measure_dp_step(step_fn, steps: int, frames_per_rank: int, device=None, warmup: int = 1) -> dict:
    """Bookkeeping of bench.py's data-parallel training legs: `step_fn()` = one synchronous-SGD step of THIS rank (its
    own micro-batch; the gradient exchange inside).  Every rank times `steps` steps with the exchange and `steps` without
    (`collectives(False)`), each as the median of per-step wall times bracketed by a device synchronisation; the job's
    step time is the MAX over ranks.  -> step_ms, step_ms_no_collective, allreduce_exposed_ms (their difference: what the
    overlap did not hide), allreduce_bytes / collective_calls per step (payload handed to the collectives by this rank),
    frames_per_s = world x frames_per_rank / step_ms."""
    import time
    world = dist.get_world_size() if _pg() else 1

    def sync():
        if device is not None and torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)

    def timed(n):
        ts = []
        for _ in range(n):
            if _pg():
                dist.barrier()
            sync()
            t0 = time.perf_counter()
            step_fn()
            sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    for _ in range(warmup):
        step_fn()
    reset_stats()
    with_ms = max_over_ranks(timed(steps), device)
    by, calls = STATS["bytes"] // max(steps, 1), STATS["calls"] / max(steps, 1)
    with collectives(False):
        for _ in range(warmup):
            step_fn()
        without_ms = max_over_ranks(timed(steps), device)
    return {"world": world, "step_ms": round(with_ms, 3), "step_ms_no_collective": round(without_ms, 3),
            "allreduce_exposed_ms": round(with_ms - without_ms, 3), "allreduce_bytes": int(by),
            "collective_calls": round(calls, 2), "frames_per_rank": frames_per_rank,
            "frames_per_s": round(world * frames_per_rank / with_ms * 1e3, 2)}
