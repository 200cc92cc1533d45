"""Tensor-level wrappers over the C ABI (torch is used for device memory and streams only).

Activations travel as `Act`: a view onto an NHWC fp32 buffer `[N,H,W,cs]` that names a channel
slice `[co, co+C)` of it -- the zero-copy concat the kernels understand (pixel stride `cs`).
Every wrapper validates device/dtype/contiguity and raises; nothing here computes with torch.
"""
from __future__ import annotations

import os
import threading

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import (ACT_NONE, ACT_RELU, ACT_SWISH, PREC_BF16, PREC_BF16X3, PREC_BF16X6, PREC_F16X3, PREC_F32,  # noqa: F401
                   ConvDesc, HipLibraryError)

# When set, every conv leaves the running max |out| of a freshly allocated output in `Act.amax` (one
# device float): the F16X3 engine of the NEXT conv reads it as its operand bound instead of making a
# pass over the tensor.  hipnn.set_precision('f16x3') turns it on.
TRACK_AMAX = False

# Bumped by hipnn.invalidate_caches(): part of the key of every cache of tensors derived from module parameters
# (packed / flipped weights, folded BatchNorm), for updates that do not bump `tensor._version` (writes through `.data`).
CACHE_EPOCH = 0

# Counts the (re)builds of such caches (packed weights, folded BatchNorm, geometry constants ...).  A cache is filled by
# launches on the stream of whoever misses first; a pipelined forward (MaxEntIRL._frozen_parts) whose part 0 built anything
# makes the later parts' streams wait for part 0 before they read it.
CACHE_BUILDS = 0


# ---- side streams that REALLY run beside the caller's.  HIP maps a process's streams onto a few hardware queues
# (GPU_MAX_HW_QUEUES, default 4) in creation order; two streams that land on one queue run their kernels in issue order --
# and worse than that: the pipelined inference step measured 38.2 ms on a concurrent side stream, 43.0 ms (= two half-batch
# forwards back to back) or 57 ms on a side stream that shared the caller's queue, and which of the two a process gets
# depends on how many streams anything (torch's pool, RCCL, a copy stream) created before.  Two queues of one dispatch PIPE
# are no better for full-chip kernels: the second kernel waits until the first has handed out its whole grid (two tiny
# kernels run side by side there, so the probe must use a large grid).  So a side stream is PROBED (_runs_beside).
_side_streams: dict = {}
_probe_log: list = []


def _runs_beside(main: torch.cuda.Stream, cand: torch.cuda.Stream) -> bool:
    """Does a kernel on `main` get onto the device while `cand` is still handing out the workgroups of a large grid (and
    the other way round)?  A 60000-workgroup do-nothing kernel (~3 waves of residents of 40 us) on one stream, a
    one-workgroup 10 us kernel on the other, issued right behind it: beside each other the small one is done in tens of
    microseconds; on one hardware queue, or on two queues of one dispatch pipe, only after the large one (~340 us; threshold 150)."""
    lib = _lib.load()
    dev = main.device
    worst = 0.0
    for big, small in ((cand, main), (main, cand)):
        best = 1e9
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record(big)
            small.wait_event(e0)
            _lib.check(lib.creste_spin_us(40, 60000, big.cuda_stream), "spin_us")
            _lib.check(lib.creste_spin_us(10, 1, small.cuda_stream), "spin_us")
            e1.record(small)
            torch.cuda.synchronize(dev)
            best = min(best, e0.elapsed_time(e1))
        worst = max(worst, best)
    _probe_log.append(round(worst * 1e3, 1))           # microseconds (diagnostics: scripts/host_issue_time.py)
    return worst < 0.150


def concurrent_stream(device, role: str, tries: int = 12):
    """A stream of `device` for `role` ('parts', 'wgrad', 'prefetch': one stream each, cached) that was measured to run
    beside the CURRENT stream and beside the other roles' streams; None when no such stream can be had (the caller then stays
    on one stream).  CRESTE_SIDE_STREAMS=0 turns every side stream off."""
    if os.environ.get("CRESTE_SIDE_STREAMS", "1") == "0":
        return None
    main = torch.cuda.current_stream(device)
    key = (device.index, role, main.cuda_stream)
    if key in _side_streams:
        return _side_streams[key]
    if _lib._recorder is not None or torch.cuda.is_current_stream_capturing():
        return None                                   # (never probe inside a plan trace / graph capture)
    others = [s for (d, r, m), s in _side_streams.items() if d == device.index and s is not None]
    found, rejected = None, []
    for _ in range(tries):
        cand = torch.cuda.Stream(device=device)
        if any(cand.cuda_stream == s.cuda_stream for s in others + rejected + [main]):
            continue
        if _runs_beside(main, cand) and all(_runs_beside(o, cand) for o in others):
            found = cand
            break
        rejected.append(cand)                         # (kept alive: torch hands its pool out round-robin)
    _side_streams[key] = found
    return found


# ---- the generic half of a pipelined forward (MaxEntIRL._frozen_parts, TerrainNet / DistillationBackbone.forward in eval mode)
def parts_for(B: int, device, want: int = 2, min_rows: int = 6) -> int:
    """How many parts an eval forward of B frames runs in: `want` when pipelining is allowed here (no autograd, no plan
    trace, no stream capture, B divisible, parts of >= min_rows frames, a probed side stream available), else 1."""
    n = int(want or 1)
    if (n < 2 or torch.is_grad_enabled() or (_lib._recorder is not None and not _lib._recorder.pipelined) or B % n or B // n < min_rows
            or os.environ.get("CRESTE_INFER_PARTS", "") in ("0", "1") or _PART.ctx is not None
            or torch.cuda.is_current_stream_capturing()):
        return 1
    if n == 2 and concurrent_stream(device, "parts") is None:
        return 1
    return n


def mark_stream(obj, stream, _seen=None):
    """record_stream(stream) on every CUDA tensor reachable from `obj` (tensors, Acts, dicts, lists / tuples): memory that
    was allocated on one stream's pool and is about to be used on `stream`."""
    seen = set() if _seen is None else _seen
    if torch.is_tensor(obj):
        if obj.is_cuda and obj.data_ptr() not in seen:
            seen.add(obj.data_ptr())
            obj.record_stream(stream)
    elif hasattr(obj, "buf"):
        mark_stream(obj.buf, stream, seen)
        mark_stream(getattr(obj, "amax", None), stream, seen)
    elif isinstance(obj, dict):
        for v in obj.values():
            mark_stream(v, stream, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            mark_stream(v, stream, seen)


def forward_in_parts(fn, batch_tensors, parts: int, owner=None):
    """fn(*slices) for each of `parts` equal slices of the batch-leading `batch_tensors`, part 0 on the caller's stream, the
    others on side streams (two parts: the probed 'parts' stream), joined before returning -> (PartContext, [results]).
    Inside, `shared_rows()` allocations are rows of shared whole-batch buffers (PartContext.whole gives the whole-batch view of
    part 0's result).  `owner`: any object; its `_parts_warm` attribute records that its lazily built caches exist (a part
    that (re)builds caches makes the later parts start behind it), its `_parts_shapes` the shared buffers of earlier calls.

    Ordering of the shared buffers: a buffer that another stream writes must not be a block that kernels still
    queued on the caller's stream are using.  The shared buffers an owner asked for the last time it ran these input shapes
    are therefore allocated BEFORE the fork event (everything queued on the caller's stream before it is complete when a
    side stream starts); a buffer allocated by part 0 in mid-forward (first call, changed shapes) carries an event recorded
    right behind the allocation, and the other parts wait for it before they touch the buffer (rows_empty)."""
    dev = batch_tensors[0].device
    main = torch.cuda.current_stream(dev)
    streams = [concurrent_stream(dev, "parts")] if parts == 2 else [torch.cuda.Stream(device=dev) for _ in range(parts - 1)]
    if any(s is None for s in streams):
        raise HipLibraryError("forward_in_parts: no side stream (ask parts_for first)")
    # The host never runs more than MAX_INFLIGHT pipelined forwards ahead of the device.  A buffer that two streams used goes
    # back to the caching allocator only when the other stream's event has COMPLETED (record_stream): a host that enqueues
    # forward after forward without looking back finds none of them reusable and grows the pool by a whole set of outputs
    # (4.6 GB at batch 16) per forward -- hipMalloc stalls in the middle of the steps (measured: single runs of 42-85 ms per
    # step where the same steps read 37.5 once the pool had grown).  Two forwards in flight keep the device fed.
    q = _inflight.setdefault((dev.index, main.cuda_stream), [])
    if _lib._recorder is None and not torch.cuda.is_current_stream_capturing():
        while len(q) >= MAX_INFLIGHT:
            q.pop(0).synchronize()
    n = batch_tensors[0].shape[0] // parts
    ctx, res = PartContext(parts), []
    skey = (parts, dev.index) + tuple((tuple(t.shape), t.dtype) for t in batch_tensors)
    known = getattr(owner, "_parts_shapes", None)
    if PREALLOCATE_SHARED and isinstance(known, dict) and skey in known:
        ctx.prealloc = [torch.empty(shape, dtype=dtype, device=dev) for shape, dtype in known[skey]]
    fork = torch.cuda.Event()
    _lib.event_record(fork, main)                      # the inputs (and the preallocated buffers) are ready once a stream gets here
    prev, _PART.ctx = _PART.ctx, ctx
    builds = CACHE_BUILDS
    try:
        for i in range(parts):
            st = main if i == 0 else streams[i - 1]
            ctx.begin(i)
            if i:
                _lib.event_wait(st, fork)
            with torch.cuda.stream(st):
                res.append(fn(*(t[i * n:(i + 1) * n] for t in batch_tensors)))
            if i == 0 and (CACHE_BUILDS != builds or not getattr(owner, "_parts_warm", False)):
                # part 0 (re)built caches -- packed weights, folded BatchNorm, constants -- by launches on ITS stream (always
                # assumed of an owner's first pipelined forward): the other parts read them only behind part 0
                fork = torch.cuda.Event()
                _lib.event_record(fork, main)
                if owner is not None:
                    owner._parts_warm = True
        ctx.begin(parts)                               # (checks that the last part took every shared buffer)
        if owner is not None:
            if not isinstance(known, dict):
                known = owner._parts_shapes = {}
            known[skey] = [(tuple(t.shape), t.dtype) for t in ctx.log]
    finally:
        _PART.ctx = prev
        for st in streams:                             # (also when a part raised: nothing stays un-joined)
            _lib.stream_wait_stream(main, st)
    # what a later part returned outside the shared buffers lives in ITS stream's pool and is read on the caller's from here on
    mark_stream(res[1:], main)
    if _lib._recorder is None and not torch.cuda.is_current_stream_capturing():
        done = torch.cuda.Event()
        done.record(main)
        q.append(done)
    return ctx, res


PREALLOCATE_SHARED = True       # (tests switch it off to exercise the event-ordered form of every call)
MAX_INFLIGHT = 2                # pipelined forwards the host may have enqueued and not yet seen finished
_inflight: dict = {}


def whole_outputs(ctx: PartContext, dicts) -> dict:
    """the whole-batch output dict of a pipelined forward from its parts' dicts: views of the shared buffers, a concatenation
    only for a tensor no shared buffer holds"""
    main = torch.cuda.current_stream()
    out = {}
    for k, v in dicts[0].items():
        w = ctx.whole(v)
        if w is None:
            ts = [d[k] for d in dicts]
            for t in ts[1:]:
                t.record_stream(main)
            w = torch.cat(ts)
        out[k] = w
    return out


# ---- weight gradients on a side stream (train_backbone.ConvG.bwd): a conv's weight gradient needs the layer's output gradient
# and its saved input but nothing downstream needs IT before the optimiser (or the gradient exchange): it leaves the
# backward's critical path and its matrix-bound kernels overlap the bandwidth-bound BatchNorm / transform kernels of the
# input-gradient chain.  `wgrad_join()` = "every weight gradient issued so far is ordered before what this stream does next".
WGRAD_STREAM = os.environ.get("CRESTE_WGRAD_STREAM", "1") != "0"
_wgrad_streams: dict = {}


def wgrad_stream(device):
    """[stream, used-since-the-last-join] -- or None: off, or no stream that runs beside the current one"""
    if not WGRAD_STREAM:
        return None
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    s = _wgrad_streams.get(key)
    if s is None:
        st = concurrent_stream(device, "wgrad")
        s = _wgrad_streams[key] = [st, False] if st is not None else False
    return s or None


def wgrad_join(device=None):
    for (idx, _), s in _wgrad_streams.items():
        if s and s[1] and (device is None or device.index == idx):
            torch.cuda.current_stream(s[0].device).wait_stream(s[0])
            s[1] = False


def note_cache_build():
    global CACHE_BUILDS
    CACHE_BUILDS += 1


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype=torch.float32, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HipLibraryError(f"{name}: expected a CUDA/HIP tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise HipLibraryError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise HipLibraryError(f"{name}: expected a contiguous tensor")
    return t


class _AmaxPool:
    """Zero-initialised device floats handed out one at a time (a fresh 4 KiB block every 1024 slots; a
    block lives as long as a slot view of it does)."""
    _blocks: dict = {}

    @classmethod
    def slot(cls, device) -> torch.Tensor:
        # one block per (device, STREAM): a block is zero-filled on the stream that creates it, and its slots are raised by
        # atomics of kernels on the stream that takes them -- a slot handed to another stream could be raised before the
        # fill has run there (the frozen half of the IRL step runs on a side stream), and the block's memory would belong
        # to the other stream's allocator pool
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        blk = cls._blocks.get(key)
        if blk is None or blk[1] >= blk[0].numel():
            blk = [fill_(torch.empty(1024, dtype=torch.float32, device=device), 0.0), 0]
            cls._blocks[key] = blk
        blk[1] += 1
        return blk[0][blk[1] - 1:blk[1]]


def fill_(t: torch.Tensor, value: float = 0.0) -> torch.Tensor:
    """t[...] = value for a contiguous 4-byte-element tensor, as a launch of THIS library (torch.zeros / fill_ would be
    a launch the plan recorder of deploy.export_plan cannot see)."""
    import struct
    if not t.is_cuda or not t.is_contiguous() or t.element_size() != 4:
        raise HipLibraryError("fill_: contiguous CUDA tensor of 4-byte elements expected")
    bits = struct.unpack("<I", struct.pack("<f", float(value)))[0] if t.dtype.is_floating_point else int(value) & 0xffffffff
    _lib.check(_lib.load().creste_fill_u32(t.data_ptr(), bits, t.numel(), _stream()), "fill_u32")
    return t


def max2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """device float max(a, b) -> a fresh |max| slot (the operand bound of a concatenation from its parts' bounds)."""
    out = _AmaxPool.slot(a.device)
    _lib.check(_lib.load().creste_max2_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), _stream()), "max2")
    return out


def reset_amax_pool():
    """Drop the current slot blocks: the next slot comes from a freshly zero-filled block.  Call right before a
    hipGraph capture so that the fill is part of the graph and every replay starts from zeroed |max| slots."""
    _AmaxPool._blocks.clear()


# ---- pipelined inference: one batch as PARTS forwards on PARTS streams (creste/models/lfd.py: MaxEntIRL._frozen_parts)
# Inside `shared_rows()` regions of such a forward, every buffer with the batch as its leading dimension is rows
# [i * n, (i + 1) * n) of ONE buffer of parts * n rows that all parts write: the k-th such allocation of part i is the k-th of
# part 0 (same code path, same shapes), so the whole-batch outputs exist without a concatenation (4.6 GB at batch 16).
class PartContext:
    def __init__(self, parts: int):
        self.parts, self.index, self.pos, self.depth = int(parts), 0, 0, 0
        self.log = []                       # the shared buffers in allocation order (allocated by part 0 on ITS stream)
        self.events = []                    # per buffer: None (allocated before the fork) or the event behind its allocation
        self.prealloc = []                  # buffers of the shapes the owner's last call asked for, allocated before the fork
        self.storages = set()

    def begin(self, index: int):
        if index and self.pos not in (0, len(self.log)):
            raise HipLibraryError("pipelined forward: a part made fewer shared allocations than part 0")
        self.index, self.pos = index, 0

    def whole(self, t: torch.Tensor):
        """part 0's view [n, ...] of a shared buffer -> the view of all parts * n rows (None if `t` is not such a view)."""
        if (not torch.is_tensor(t) or not t.is_cuda or t.dim() == 0
                or t.untyped_storage().data_ptr() not in self.storages or t.stride(0) == 0):
            return None
        return torch.as_strided(t, (t.shape[0] * self.parts,) + tuple(t.shape[1:]), t.stride(), t.storage_offset())

    def whole_act(self, a):
        b = self.whole(a.buf)
        return None if b is None else Act(b, a.C, a.co)


class _PartState(threading.local):
    ctx = None


_PART = _PartState()


class shared_rows:
    """`with shared_rows():` -- batch-leading buffers allocated inside come from the parts' shared whole-batch buffers when a
    pipelined forward is running (no effect otherwise).  Put around the producers of a forward's OUTPUTS."""

    def __enter__(self):
        if _PART.ctx is not None:
            _PART.ctx.depth += 1

    def __exit__(self, *exc):
        if _PART.ctx is not None:
            _PART.ctx.depth -= 1
        return False


def rows_empty(shape, dtype, device) -> torch.Tensor:
    """torch.empty(shape) whose dim 0 is the batch; inside a `shared_rows()` region of a pipelined forward: this part's rows
    of the shared whole-batch buffer."""
    c = _PART.ctx
    if c is None or c.depth == 0:
        return torch.empty(shape, dtype=dtype, device=device)
    n = int(shape[0])
    full_shape = (n * c.parts,) + tuple(int(v) for v in shape[1:])
    if c.index == 0:
        k = len(c.log)
        if k < len(c.prealloc) and tuple(c.prealloc[k].shape) == full_shape and c.prealloc[k].dtype == dtype:
            full, ev = c.prealloc[k], None
        else:
            # a block that part 0's stream freed a moment ago may still be in use by kernels queued there: the other parts'
            # streams (which waited for the fork only) may write it once THIS point of part 0's stream has been reached
            c.prealloc = c.prealloc[:k]
            full = torch.empty(full_shape, dtype=dtype, device=device)
            ev = torch.cuda.Event()
            _lib.event_record(ev, torch.cuda.current_stream(full.device))
        c.log.append(full)
        c.events.append(ev)
        c.storages.add(full.untyped_storage().data_ptr())
    else:
        if c.pos >= len(c.log):
            raise HipLibraryError("pipelined forward: a part made more shared allocations than part 0")
        full = c.log[c.pos]
        if tuple(full.shape) != full_shape or full.dtype != dtype:
            raise HipLibraryError(f"pipelined forward: shared allocation {c.pos} is {tuple(full.shape)} {full.dtype} in part 0 "
                                  f"and {full_shape} {dtype} in part {c.index}")
        here = torch.cuda.current_stream(full.device)
        full.record_stream(here)                       # allocated on part 0's stream, written on this one
        if c.events[c.pos] is not None:
            _lib.event_wait(here, c.events[c.pos])
    c.pos += 1
    return full[c.index * n:(c.index + 1) * n]


@dataclass
class Act:
    buf: torch.Tensor      # [N,H,W,cs] contiguous fp32
    C: int                 # channels in the slice
    co: int = 0            # channel offset of the slice
    amax: torch.Tensor | None = None   # device float >= max|slice| when known (see TRACK_AMAX)
    stats: tuple | None = None         # (per-workgroup channel sums [rows, 2, C], rows) left by the producing conv (conv2d want_stats)

    @property
    def N(self): return self.buf.shape[0]
    @property
    def H(self): return self.buf.shape[1]
    @property
    def W(self): return self.buf.shape[2]
    @property
    def cs(self): return self.buf.shape[3]
    @property
    def ptr(self): return self.buf.data_ptr() + 4 * self.co

    @staticmethod
    def empty(N, H, W, C, device, cs=None):
        return Act(rows_empty((N, H, W, cs or C), torch.float32, device), C, 0)

    def slice(self, co, C):
        assert co + C <= self.cs
        return Act(self.buf, C, co)

    def nchw(self) -> torch.Tensor:
        """[N,C,H,W]-shaped (channels-last strided) zero-copy view of the slice."""
        return self.buf[..., self.co:self.co + self.C].permute(0, 3, 1, 2)


def absmax(x: Act) -> torch.Tensor:
    """Device float max|x| of the slice (one streaming read); cached on the Act."""
    if x.amax is None:
        lib = _lib.load()
        slot = _AmaxPool.slot(x.buf.device)
        _lib.check(lib.creste_absmax_nhwc_f32(x.ptr, x.N * x.H * x.W, x.C, x.cs, slot.data_ptr(), _stream()),
                   "absmax")
        x.amax = slot
    return x.amax


FUSE_UPSAMPLE = True       # conv2d forms cat([skip, up2x(x1)]) inside the F(4x4) input transform instead of materialising it


class LazyUpCat:
    """cat([skip, bilinear_upsample(x1)], C) that has not been formed yet (reference Up.forward, effnet.py:16-23;
    DeconvHead.up2, inpainting.py:81).  `conv2d` consumes it directly when the conv runs on the F(4x4,3x3) path and the
    upsample is the exact 2x one (csrc/conv_wino4.hip forms the tensor inside its input transform, bit-identically);
    everything else calls `materialise()` (cached: the BEV heads share one x4 concat)."""

    def __init__(self, x1: Act, skip, Ho, Wo, rh, rw):
        self.x1, self.skip, self.Ho, self.Wo, self.rh, self.rw = x1, skip, Ho, Wo, rh, rw
        self._mat = None
        self._w4 = None            # (key, workspace) of the last F(4x4) conv over the materialised tensor

    N = property(lambda self: self.x1.N)
    H = property(lambda self: self.Ho)
    W = property(lambda self: self.Wo)
    C = property(lambda self: self.x1.C + (self.skip.C if self.skip is not None else 0))

    @property
    def exact2x(self):
        return (self.Ho == 2 * self.x1.H and self.Wo == 2 * self.x1.W and float(self.rh) == 0.5 and float(self.rw) == 0.5
                and self.x1.C % 4 == 0 and self.x1.co % 4 == 0
                and (self.skip is None or ((self.skip.H, self.skip.W) == (self.Ho, self.Wo) and self.skip.co % 4 == 0)))

    def materialise(self) -> Act:
        if self._mat is None:
            self._mat = upsample_concat(self.x1, self.skip, self.Ho, self.Wo, self.rh, self.rw)
        return self._mat


def upsample_concat_lazy(x1: Act, skip, Ho, Wo, rh, rw) -> LazyUpCat:
    return LazyUpCat(x1, skip, Ho, Wo, rh, rw)


@dataclass
class PackedConv:
    wpk: torch.Tensor          # packed GEMM weights (uint8 storage)
    bias: torch.Tensor | None  # [Cout] fp32 (conv bias and folded BN shift)
    Cin: int
    Cout: int
    KH: int
    KW: int
    stride: int
    pad_t: int
    pad_l: int
    pad_b: int
    pad_r: int
    act: int
    prec: int = PREC_F32
    w_unscale: torch.Tensor | None = None   # F16X3: [Cout] inverse power-of-two weight scale
    algo: int = 0                           # ALGO_DIRECT | ALGO_WINOGRAD: how `wpk` is packed

    def out_hw(self, H, W):
        return ((H + self.pad_t + self.pad_b - self.KH) // self.stride + 1,
                (W + self.pad_l + self.pad_r - self.KW) // self.stride + 1)


def conv_supported(prec: int, k: int, stride: int) -> bool:
    return bool(_lib.load().creste_conv_supported(prec, k, k, stride))


def conv_precision(prec: int, k: int, stride: int, cin: int) -> int:
    """The engine a conv of this shape runs on under the requested operand mode: `prec` where the split-operand
    engine is built for it, the exact-fp32 engine otherwise -- and for strided convs with fewer than 16 input channels
    (the 4 -> 32 encoder stem: one mostly-empty 16-channel chunk per 90 KB halo patch; measured 0.74 vs 0.42 ms)."""
    if stride > 1 and cin < 16:
        return PREC_F32
    if not conv_supported(prec, k, stride):
        # the narrower bf16 modes are built for the stride-1 1x1 / 3x3 kernels only: their strided / 5x5 / 7x7 convs take the
        # bf16x6 row kernels (more operand bits than asked for, on the bf16 matrix instruction) rather than the exact-fp32 engine
        # on the 16x slower fp32 one (cf-IRL step at configs[4]'s per-GPU shape: 7.4 of 35 ms of kernels were those convs)
        if prec in (PREC_BF16, PREC_BF16X3) and conv_supported(PREC_BF16X6, k, stride):
            return PREC_BF16X6
        return PREC_F32
    return prec


ALGO_DIRECT, ALGO_WINOGRAD, ALGO_WINOGRAD4 = 0, 1, 2
# Winograd F(2x2,3x3) policy: the stride-1 3x3 convs of the fp32-equivalent `bf16x6` mode with at least WINOGRAD_MIN_C
# input AND output channels.  2.25x fewer matrix-core products, paid for with the transforms and an fp32 round trip of
# the 16 per-position products: a win for wide layers only -- measured at batch 16 (scripts/wino_micro.py): 496->496
# @152x304 14.05 -> 11.17 ms, 472->472 @76x152 3.76 -> 2.63, 432->432 @38x76 1.13 -> 0.64, 320->256 @128x128 1.84 -> 1.39,
# 256->256 1.54 -> 1.20; 256->128 @256x256 2.80 -> 2.92 (loses: the 128-cout tile amortises the loader over half the products).
WINOGRAD = True
WINOGRAD_MIN_C = 256
# F(4x4,3x3) (csrc/conv_wino4.hip): 1.78x fewer products again and a GEMM loop that is LDS-DMA + MFMA only, for the price of
# the transformed input / the products crossing HBM once each; preferred over F(2x2) where both apply
WINOGRAD4 = True
WINOGRAD4_MIN_C, WINOGRAD4_MIN_CC = 128, 128 * 256     # both sides >= 128 channels, Cin * Cout >= 128 * 256 (128 -> 128 breaks even:
# 1.48 vs 1.53 ms @256x256x16; 256 -> 128: 2.35 vs 2.71; 128 -> 256: 2.19 vs 2.81)


def conv_algo(prec: int, k: int, stride: int, pad, cin: int, cout: int) -> int:
    if WINOGRAD4 and prec in (PREC_BF16X6, PREC_BF16X3) and k == 3 and stride == 1 and tuple(pad) == (1, 1, 1, 1) \
            and min(cin, cout) >= WINOGRAD4_MIN_C and cin * cout >= WINOGRAD4_MIN_CC \
            and _lib.load().creste_conv_wino4_supported(prec, k, k, stride, cin, cout):
        return ALGO_WINOGRAD4
    if WINOGRAD and prec == PREC_BF16X6 and k == 3 and stride == 1 and tuple(pad) == (1, 1, 1, 1) \
            and min(cin, cout) >= WINOGRAD_MIN_C and _lib.load().creste_conv_wino_supported(prec, k, k, stride, cin, cout):
        return ALGO_WINOGRAD
    return ALGO_DIRECT


def pack_conv(weight: torch.Tensor, bias, bn, stride, pad, act, prec=PREC_F32, algo=None) -> PackedConv:
    """weight OIHW (CUDA fp32); bn = None or (gamma, beta, mean, var, eps) -> folded eval-mode BN.
    pad = int | (pad_t, pad_b, pad_l, pad_r).  algo: None = the policy of `conv_algo`, or ALGO_DIRECT / ALGO_WINOGRAD."""
    lib = _lib.load()
    note_cache_build()
    w = _chk(weight.detach().contiguous(), name="conv weight")
    Cout, Cin, KH, KW = w.shape
    scale = None
    b = None if bias is None else bias.detach().float()
    if bn is not None:
        gamma, beta, mean, var, eps = bn
        scale = (gamma.detach() / torch.sqrt(var.detach() + eps)).float().contiguous()
        shift = beta.detach() - mean.detach() * scale
        b = shift if b is None else b * scale + shift
    if b is not None:
        b = b.contiguous()
    if isinstance(pad, int):
        pad = (pad, pad, pad, pad)
    if algo is None:
        algo = conv_algo(prec, KH, stride, pad, Cin, Cout) if KH == KW else ALGO_DIRECT
    sp = scale.data_ptr() if scale is not None else None
    if algo in (ALGO_WINOGRAD, ALGO_WINOGRAD4):
        f4 = algo == ALGO_WINOGRAD4
        nbytes = (lib.creste_conv_wino4_weight_bytes if f4 else lib.creste_conv_wino_weight_bytes)(Cout, Cin, prec)
        if nbytes <= 0 or (KH, KW, stride) != (3, 3, 1):
            raise HipLibraryError("pack_conv: the Winograd path is not built for this shape / precision")
        wpk = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        _lib.check((lib.creste_conv_wino4_pack_weight if f4 else lib.creste_conv_wino_pack_weight)(
            w.data_ptr(), sp, wpk.data_ptr(), Cout, Cin, prec, _stream()), "conv_wino_pack_weight")
        return PackedConv(wpk, b, Cin, Cout, KH, KW, stride, pad[0], pad[2], pad[1], pad[3], act, prec, None, algo)
    nbytes = lib.creste_conv_packed_weight_bytes(Cout, Cin, KH, KW, prec)
    if nbytes <= 0:
        raise HipLibraryError("conv_packed_weight_bytes: unsupported shape/precision")
    wpk = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    unscale = None
    if prec == PREC_F16X3:
        unscale = torch.empty(Cout, dtype=torch.float32, device=w.device)
        _lib.check(lib.creste_conv_pack_weight_f16(w.data_ptr(), sp, wpk.data_ptr(), unscale.data_ptr(), Cout,
                                                   Cin, KH, KW, _stream()), "conv_pack_weight_f16")
    else:
        _lib.check(lib.creste_conv_pack_weight(w.data_ptr(), sp, wpk.data_ptr(), Cout, Cin, KH, KW, prec,
                                               _stream()), "conv_pack_weight")
    return PackedConv(wpk, b, Cin, Cout, KH, KW, stride, pad[0], pad[2], pad[1], pad[3], act, prec, unscale)


# conv3x3 -> conv3x3 pairs (Up.conv): conv 1's output transform writes conv 2's transformed input (CRESTE_FUSE_PAIRS=0: off)
FUSE_CONV_PAIRS = os.environ.get("CRESTE_FUSE_PAIRS", "1") != "0"
FUSE_PAIR_MIN_FILL = 0.5   # ... where the map fills the fused kernel's 8 x 16-tile blocks at least this well (38 x 76: 0.37, slower)


class _TransformedInput:
    """The input of a F(4x4,3x3) conv that only exists as its transformed image V, at the start of the conv's workspace
    (written by the previous conv's fused output -> input transform, CRESTE_CONV_EMIT_NEXT_V)."""
    __slots__ = ("work", "N", "H", "W", "C", "device")

    def __init__(self, work, N, H, W, C, device):
        self.work, self.N, self.H, self.W, self.C, self.device = work, N, H, W, C, device


def conv_pair_fusable(pc1: PackedConv, pc2: PackedConv) -> bool:
    # (the fused kernel writes conv 2's transformed input as fp32: with the bf16-piece form forced -- CRESTE_W4_F32V=0, the
    # bit-exact twin the tests compare against -- the second GEMM would read it as pieces: never fuse there)
    return (FUSE_CONV_PAIRS and not TRACK_AMAX and os.environ.get("CRESTE_W4_F32V", "1") != "0"
            and pc1.algo == ALGO_WINOGRAD4 and pc2.algo == ALGO_WINOGRAD4
            and pc1.prec == PREC_BF16X6 and pc2.prec == PREC_BF16X6 and pc1.Cout == pc2.Cin
            and (pc1.pad_t, pc1.pad_l, pc2.pad_t, pc2.pad_l) == (1, 1, 1, 1) and pc1.out_hw(8, 8) == (8, 8)
            and pc2.out_hw(8, 8) == (8, 8))


def conv2d_pair(x, pc1: PackedConv, pc2: PackedConv, out: Act | None = None) -> Act:
    """conv2(conv1(x)) of a conv3x3 (+BN+ReLU) pair (reference Up.conv, effnet.py:15-28).  Where both run as F(4x4,3x3) in
    bf16x6, conv 1's output never crosses HBM as a tensor: its output transform writes conv 2's transformed input directly
    (same values, bit for bit, as the two separate calls)."""
    ty, tx = (x.H + 3) // 4, (x.W + 3) // 4
    fill = ty * tx / float(((ty + 7) // 8 * 8) * ((tx + 15) // 16 * 16))
    if not conv_pair_fusable(pc1, pc2) or fill < FUSE_PAIR_MIN_FILL:
        return conv2d(conv2d(x, pc1), pc2, out=out)
    return conv2d(conv2d(x, pc1, _emit_next=pc2), pc2, out=out)


CONV_STATS = os.environ.get("CRESTE_CONV_STATS", "1") != "0"     # training: BatchNorm statistics from the producing conv's epilogue


def conv2d(x: Act, pc: PackedConv, out: Act | None = None, res: Act | None = None,
           a_scale: torch.Tensor | None = None, row_mask: torch.Tensor | None = None, _emit_next: PackedConv | None = None,
           want_stats: bool = False) -> Act:
    """want_stats: where the kernel this conv dispatches to keeps them (creste_conv_stat_rows), the returned Act carries
    `.stats = (partial [rows, 2, Cout], rows)`: per-workgroup sums of the output, the training-mode BatchNorm's batch
    statistics without another pass (creste_bn_train_forward_stats_f32); otherwise `.stats` stays None."""
    lib = _lib.load()
    if isinstance(x, _TransformedInput):
        if res is not None or a_scale is not None or row_mask is not None or want_stats or _emit_next is not None:
            raise HipLibraryError("conv2d: the second conv of a fused pair takes no residual / gate / row mask / statistics request")
        return _conv2d_from_v(x, pc, out)
    up, shared = None, None
    if isinstance(x, LazyUpCat):
        if FUSE_UPSAMPLE and pc.algo == ALGO_WINOGRAD4 and x.exact2x and a_scale is None:
            up, x = x, (x.skip if x.skip is not None else x.x1)       # x: the tensor that owns device / batch below
        else:
            shared, x = x, x.materialise()        # several convs read this tensor: they share its transformed image
    N, H, W, cin, dev = x.N, x.H, x.W, x.C, x.buf.device
    if up is not None:
        H, W, cin = up.H, up.W, up.C
    if pc.algo == ALGO_WINOGRAD and N > 1 and N * H * W * x.cs >= (1 << 30):
        # the Winograd loader addresses its input with 32-bit byte offsets: batches of more than 4 GiB go in slices
        Ho, Wo = pc.out_hw(H, W)
        if out is None:
            out = Act.empty(N, Ho, Wo, pc.Cout, dev)
        per = max(1, ((1 << 30) - 1) // (H * W * x.cs))
        amax = None
        for n0 in range(0, N, per):
            n1 = min(N, n0 + per)
            o = conv2d(Act(x.buf[n0:n1], x.C, x.co, x.amax), pc, out=Act(out.buf[n0:n1], out.C, out.co),
                       res=None if res is None else Act(res.buf[n0:n1], res.C, res.co),
                       a_scale=None if a_scale is None else a_scale[n0:n1].contiguous(),
                       row_mask=None if row_mask is None else row_mask.reshape(N, -1)[n0:n1].contiguous())
            if o.amax is not None:
                amax = o.amax if amax is None else max2(amax, o.amax)
        out.amax = amax
        return out
    _chk(x.buf, name="conv input")
    if cin != pc.Cin:
        raise HipLibraryError(f"conv2d: input has {cin} channels, weights expect {pc.Cin}")
    Ho, Wo = pc.out_hw(H, W)
    nxt = None
    if _emit_next is not None:
        if out is not None or res is not None or row_mask is not None or a_scale is not None:
            raise HipLibraryError("conv2d: the fused conv pair takes no output slice / residual / mask / gate on its first conv")
        nxt = _TransformedInput(torch.empty(lib.creste_conv_wino4_workspace_bytes(N, Ho, Wo, _emit_next.Cin, _emit_next.Cout,
                                                                                   _emit_next.prec), dtype=torch.uint8, device=dev),
                                N, Ho, Wo, pc.Cout, dev)
    elif out is None:
        out = Act.empty(N, Ho, Wo, pc.Cout, dev)
    if nxt is None and (out.N, out.H, out.W, out.C) != (N, Ho, Wo, pc.Cout):
        raise HipLibraryError(f"conv2d: output slice {(out.N, out.H, out.W, out.C)} != "
                              f"{(N, Ho, Wo, pc.Cout)}")
    d = ConvDesc()
    d.in_ = x.ptr
    in_cs = x.cs
    if up is not None:
        _chk(up.x1.buf, name="conv upsample source")
        d.in_, in_cs = (up.skip.ptr, up.skip.cs) if up.skip is not None else (None, 0)
        d.up_src, d.up_H, d.up_W, d.up_C, d.up_cs = up.x1.ptr, up.x1.H, up.x1.W, up.x1.C, up.x1.cs
    d.wpk, d.out = pc.wpk.data_ptr(), (out.buf.data_ptr() if nxt is None else nxt.work.data_ptr())
    d.bias = pc.bias.data_ptr() if pc.bias is not None else None
    if res is not None:
        if (res.N, res.H, res.W, res.C) != (out.N, out.H, out.W, out.C):
            raise HipLibraryError("conv2d: residual shape mismatch")
        d.res, d.res_cs = res.ptr, res.cs
    else:
        d.res, d.res_cs = None, 0
    if a_scale is not None:
        _chk(a_scale, name="a_scale")
        if tuple(a_scale.shape) != (N, pc.Cin):
            raise HipLibraryError("conv2d: a_scale must be [N,Cin]")
        d.a_scale = a_scale.data_ptr()
    if row_mask is not None:
        _chk(row_mask, name="row_mask")
        if row_mask.numel() != N * Ho * Wo:
            raise HipLibraryError("conv2d: row_mask must have N*Ho*Wo elements")
        d.row_mask = row_mask.data_ptr()
    d.N, d.H, d.W, d.Cin, d.in_cs = N, H, W, pc.Cin, in_cs
    d.Ho, d.Wo, d.Cout, d.out_cs, d.out_co = (Ho, Wo, pc.Cout, out.cs, out.co) if nxt is None else (Ho, Wo, pc.Cout, pc.Cout, 0)
    d.KH, d.KW, d.stride, d.pad_t, d.pad_l = pc.KH, pc.KW, pc.stride, pc.pad_t, pc.pad_l
    d.act, d.prec, d.algo = pc.act, pc.prec, pc.algo
    work = None
    if pc.algo == ALGO_WINOGRAD:       # fp32 products of the 16 transform positions (stream-ordered reuse by the allocator)
        work = torch.empty(lib.creste_conv_wino_workspace_bytes(N, Ho, Wo, pc.Cout), dtype=torch.uint8, device=dev)
        d.work = work.data_ptr()
    elif pc.algo == ALGO_WINOGRAD4:    # transformed input (bf16 pieces) + fp32 products of the 36 positions
        need = lib.creste_conv_wino4_workspace_bytes(N, Ho, Wo, pc.Cin, pc.Cout, pc.prec)
        key = (pc.Cin, pc.prec, pc.pad_t, pc.pad_l, x.ptr, x.cs, x.buf._version)
        if shared is not None and shared._w4 is not None and shared._w4[0] == key and shared._w4[1].numel() >= need:
            work = shared._w4[1]              # the input transform of this tensor is already there (same stream)
            d.flags = 1                       # CRESTE_CONV_V_VALID
        else:
            work = torch.empty(need, dtype=torch.uint8, device=dev)
            if shared is not None:
                shared._w4 = (key, work)
        d.work = work.data_ptr()
    stats = None
    if want_stats and CONV_STATS and nxt is None and res is None and row_mask is None:
        rows = lib.creste_conv_stat_rows(C.byref(d))
        if rows > 0:
            stats = (torch.empty((rows, 2, pc.Cout), dtype=torch.float32, device=dev), rows)
            d.out_stats = stats[0].data_ptr()
    if nxt is not None:
        d.flags = d.flags | 2                  # CRESTE_CONV_EMIT_NEXT_V
        _lib.check(lib.creste_conv2d_nhwc(C.byref(d), _stream()), "conv2d_nhwc")
        return nxt
    if pc.prec == PREC_F16X3:
        d.a_amax, d.w_unscale = absmax(x).data_ptr(), pc.w_unscale.data_ptr()
    if TRACK_AMAX or pc.prec == PREC_F16X3:     # also into a caller's slice: the bound of THAT slice (its Act object)
        out.amax = _AmaxPool.slot(dev)
        d.out_amax = out.amax.data_ptr()
    _lib.check(lib.creste_conv2d_nhwc(C.byref(d), _stream()), "conv2d_nhwc")
    out.stats = stats
    return out


def _conv2d_from_v(x: "_TransformedInput", pc: PackedConv, out: Act | None) -> Act:
    """the second conv of a fused pair: its workspace already holds the transformed input (CRESTE_CONV_V_VALID)"""
    lib = _lib.load()
    N, H, W = x.N, x.H, x.W
    if pc.algo != ALGO_WINOGRAD4 or x.C != pc.Cin:
        raise HipLibraryError("conv2d: a transformed input feeds the F(4x4,3x3) conv it was made for")
    Ho, Wo = pc.out_hw(H, W)
    if out is None:
        out = Act.empty(N, Ho, Wo, pc.Cout, x.device)
    if (out.N, out.H, out.W, out.C) != (N, Ho, Wo, pc.Cout):
        raise HipLibraryError(f"conv2d: output slice {(out.N, out.H, out.W, out.C)} != {(N, Ho, Wo, pc.Cout)}")
    d = ConvDesc()
    d.in_, d.wpk, d.out = x.work.data_ptr(), pc.wpk.data_ptr(), out.buf.data_ptr()      # `in` is not read
    d.bias = pc.bias.data_ptr() if pc.bias is not None else None
    d.res, d.res_cs = None, 0
    d.N, d.H, d.W, d.Cin, d.in_cs = N, H, W, pc.Cin, pc.Cin
    d.Ho, d.Wo, d.Cout, d.out_cs, d.out_co = Ho, Wo, pc.Cout, out.cs, out.co
    d.KH, d.KW, d.stride, d.pad_t, d.pad_l = pc.KH, pc.KW, pc.stride, pc.pad_t, pc.pad_l
    d.act, d.prec, d.algo = pc.act, pc.prec, pc.algo
    d.work, d.flags = x.work.data_ptr(), 1
    _lib.check(lib.creste_conv2d_nhwc(C.byref(d), _stream()), "conv2d_nhwc")
    return out


# three 1x1 conv + BN + ReLU layers of 128 channels as one kernel (the distillation head; CRESTE_CHAIN_1X1=0: three conv launches)
CHAIN_1X1 = os.environ.get("CRESTE_CHAIN_1X1", "1") != "0"


@dataclass
class PackedChain3:
    wimg: torch.Tensor         # MFMA operand tiles of the three weight matrices (bf16 pieces, uint8 storage)
    bias: torch.Tensor         # [3][128] fp32
    Cin: int
    prec: int


def conv1x1_chain3_supported(prec: int, dims) -> bool:
    return (CHAIN_1X1 and not TRACK_AMAX and prec in (PREC_BF16X6, PREC_BF16X3) and len(dims) == 4 and tuple(dims[1:]) == (128, 128, 128)
            and dims[0] % 32 == 0 and dims[0] <= 1024)


def pack_conv1x1_chain3(layers, prec) -> PackedChain3:
    """layers: three (weight [128, C, 1, 1], bias or None, bn = None | (gamma, beta, mean, var, eps)) of 1x1 convs each followed by ReLU
    (reference MultiLayerConv, blocks/conv.py:5-32) -> the operand tiles of creste_conv1x1_chain3_f32: BatchNorm folded as pack_conv
    does, every fp32 weight split into its bf16 pieces (round to nearest, the remainder again), layer 1 in channel order, layers 2 / 3
    in the order of the accumulator layout (MFMA step 2t + j, k-octet h: channels 32t + 16j + {4h + e, 8 + 4h + e}, e < 4)."""
    note_cache_build()
    split = 3 if prec == PREC_BF16X6 else 2
    dev = layers[0][0].device
    tiles, biases = [], []
    for li, (weight, bias, bn) in enumerate(layers):
        w = _chk(weight.detach().contiguous(), name="conv weight").reshape(weight.shape[0], weight.shape[1]).float()
        b = torch.zeros(128, dtype=torch.float32, device=dev) if bias is None else bias.detach().float()
        if bn is not None:
            gamma, beta, mean, var, eps = bn
            scale = (gamma.detach() / torch.sqrt(var.detach() + eps)).float()
            w = w * scale[:, None]
            b = b * scale + (beta.detach() - mean.detach() * scale)
        K = w.shape[1]
        if li == 0:
            idx = torch.arange(K, device=dev).reshape(K // 16, 2, 8)                          # [step][k-octet][e]
        else:
            t, j, h, e = torch.meshgrid(torch.arange(4, device=dev), torch.arange(2, device=dev), torch.arange(2, device=dev),
                                        torch.arange(8, device=dev), indexing="ij")
            idx = (32 * t + 16 * j + torch.where(e < 4, 4 * h + e, 8 + 4 * h + e - 4)).reshape(8, 2, 8)
        g = w[:, idx].permute(1, 2, 0, 3).contiguous()                                       # [step][k-octet][128 couts][8]
        pieces, r = [], g
        for _ in range(split):
            pc = r.to(torch.bfloat16)
            pieces.append(pc)
            r = r - pc.float()
        tiles.append(torch.stack(pieces, dim=1))                                             # [step][piece][k-octet][128][8]
        biases.append(b)
    wimg = torch.cat([t.reshape(-1) for t in tiles]).contiguous().view(torch.uint8)
    need = _lib.load().creste_conv1x1_chain3_weight_bytes(layers[0][0].shape[1], prec)
    if wimg.numel() != need:
        raise HipLibraryError(f"pack_conv1x1_chain3: {wimg.numel()} bytes of operand tiles, the kernel expects {need}")
    return PackedChain3(wimg, torch.stack(biases).contiguous(), int(layers[0][0].shape[1]), prec)


def conv1x1_chain3(x: Act, pk: PackedChain3, out: Act | None = None) -> Act:
    """relu(W3 relu(W2 relu(W1 x + b1) + b2) + b3) per pixel, 128 channels out; the hidden layers never leave the registers"""
    _chk(x.buf, name="conv input")
    if x.C != pk.Cin:
        raise HipLibraryError(f"conv1x1_chain3: input has {x.C} channels, weights expect {pk.Cin}")
    if out is None:
        out = Act.empty(x.N, x.H, x.W, 128, x.buf.device)
    if (out.N, out.H, out.W, out.C) != (x.N, x.H, x.W, 128):
        raise HipLibraryError(f"conv1x1_chain3: output slice {(out.N, out.H, out.W, out.C)} != {(x.N, x.H, x.W, 128)}")
    _lib.check(_lib.load().creste_conv1x1_chain3_f32(x.ptr, x.cs, x.N * x.H * x.W, pk.Cin, pk.wimg.data_ptr(), pk.bias.data_ptr(), pk.prec,
                                                     out.buf.data_ptr(), out.cs, out.co, _stream()), "conv1x1_chain3")
    out.amax, out.stats = None, None
    return out


# `Upsample(x2, bilinear) -> conv3x3` as four phase convolutions on the low-resolution map (CRESTE_PHASE_UPCONV=0: off; then the
# upsample is formed inside the F(4x4) input transform of a conv over the high-resolution map, LazyUpCat)
PHASE_UPCONV = os.environ.get("CRESTE_PHASE_UPCONV", "1") != "0"


def phase_upconv_weights(weight: torch.Tensor) -> torch.Tensor:
    """OIHW 3x3 kernel of a conv that reads the exact 2x bilinear upsample (align_corners=False) of x -> the [4 * O, I, 3, 3] kernel
    of the equivalent conv on x itself: output channel (2a + b) * O + o at low-resolution pixel (y, x) = output channel o at
    high-resolution pixel (2y + a, 2x + b).  High-resolution row 2y + a - 1 + ky is 0.75 / 0.25 of two of the rows y - 1 .. y + 1
    (R[a][ky][ky']), so w'[a, b] = R[a]^T w R[b] per (o, i); composed in float64, rounded once."""
    R = torch.tensor([[[0.75, 0.25, 0.0], [0.25, 0.75, 0.0], [0.0, 0.75, 0.25]],
                      [[0.25, 0.75, 0.0], [0.0, 0.75, 0.25], [0.0, 0.25, 0.75]]], dtype=torch.float64, device=weight.device)
    w = weight.detach().double()
    O, I = w.shape[:2]
    wp = torch.einsum("ayp,bxq,oiyx->aboipq", R, R, w)                 # [2, 2, O, I, 3, 3]
    return wp.reshape(4 * O, I, 3, 3).float().contiguous()


@dataclass
class PackedUpConv:
    phase: PackedConv          # the composed 4 x Cout kernels, F(4x4,3x3)-packed, BatchNorm folded, bias replicated
    w_ring: torch.Tensor       # [3][3][Cin][Cout] fp32: the ORIGINAL kernel (BatchNorm scale folded) for the border ring
    Cout: int


def pack_upconv2x(weight: torch.Tensor, bias, bn, act, prec) -> PackedUpConv:
    """`nn.Upsample(scale_factor=2, mode='bilinear') -> nn.Conv2d(k=3, padding=1) (+ eval BatchNorm) (+ act)` (reference
    DeconvHead.up2, inpainting.py:56-60) packed for `upconv2x`."""
    Cout, Cin = weight.shape[:2]
    rep = lambda t: None if t is None else t.detach().repeat(4)
    bn4 = None if bn is None else (rep(bn[0]), rep(bn[1]), rep(bn[2]), rep(bn[3]), bn[4])
    pc = pack_conv(phase_upconv_weights(weight), rep(bias), bn4, 1, 1, act, prec, algo=ALGO_WINOGRAD4)
    w = weight.detach().float()
    if bn is not None:
        w = w * (bn[0].detach() / torch.sqrt(bn[3].detach() + bn[4])).float()[:, None, None, None]
    return PackedUpConv(pc, w.permute(2, 3, 1, 0).contiguous(), Cout)


def upconv2x_supported(prec: int, cin: int, cout: int) -> bool:
    return (PHASE_UPCONV and prec in (PREC_BF16X6, PREC_BF16X3) and cin % 64 == 0 and cout % 4 == 0 and not TRACK_AMAX
            and os.environ.get("CRESTE_W4_F32V", "1") != "0"
            and bool(_lib.load().creste_conv_wino4_supported(prec, 3, 3, 1, cin, 4 * cout)))


def upconv2x(x: Act, pu: PackedUpConv, out: Act | None = None) -> Act:
    """act(bn(conv3x3(bilinear_up2x(x)))) [N, 2H, 2W, Cout] without the upsampled tensor: the phase convolution on x with
    replicate padding (CRESTE_CONV_REPLICATE_PAD | CRESTE_CONV_PHASE2X), then the border ring put right
    (creste_upconv2x_ring_fix_f32: the high-resolution conv pads the UPSAMPLED image with zeros)."""
    lib = _lib.load()
    pc = pu.phase
    N, H, W, dev = x.N, x.H, x.W, x.buf.device
    _chk(x.buf, name="upconv2x input")
    if x.C != pc.Cin:
        raise HipLibraryError(f"upconv2x: input has {x.C} channels, weights expect {pc.Cin}")
    if out is None:
        out = Act.empty(N, 2 * H, 2 * W, pu.Cout, dev)
    if (out.N, out.H, out.W, out.C) != (N, 2 * H, 2 * W, pu.Cout):
        raise HipLibraryError(f"upconv2x: output slice {(out.N, out.H, out.W, out.C)} != {(N, 2 * H, 2 * W, pu.Cout)}")
    d = ConvDesc()
    d.in_, d.wpk, d.out = x.ptr, pc.wpk.data_ptr(), out.buf.data_ptr()
    d.bias = pc.bias.data_ptr() if pc.bias is not None else None
    d.res, d.res_cs = None, 0
    d.N, d.H, d.W, d.Cin, d.in_cs = N, H, W, pc.Cin, x.cs
    d.Ho, d.Wo, d.Cout, d.out_cs, d.out_co = H, W, pc.Cout, out.cs, out.co
    d.KH, d.KW, d.stride, d.pad_t, d.pad_l = 3, 3, 1, 1, 1
    d.act, d.prec, d.algo = pc.act, pc.prec, ALGO_WINOGRAD4
    work = torch.empty(lib.creste_conv_wino4_workspace_bytes(N, H, W, pc.Cin, pc.Cout, pc.prec), dtype=torch.uint8, device=dev)
    d.work, d.flags = work.data_ptr(), 4 | 8            # CRESTE_CONV_REPLICATE_PAD | CRESTE_CONV_PHASE2X
    _lib.check(lib.creste_conv2d_nhwc(C.byref(d), _stream()), "conv2d_nhwc (phase upconv)")
    _lib.check(lib.creste_upconv2x_ring_fix_f32(x.ptr, x.cs, N, H, W, pc.Cin, pu.w_ring.data_ptr(), pu.Cout, pc.act,
                                                out.buf.data_ptr(), out.cs, out.co, _stream()), "upconv2x_ring_fix")
    out.amax, out.stats = None, None
    return out


DW_TILE, DW_TILE_MIN_C = True, 16       # policy of dwconv2d / dwconv2d_se: LDS-tile kernel for channel counts >= DW_TILE_MIN_C


def dwconv2d(x: Act, w_taps: torch.Tensor, bias: torch.Tensor, K, stride, pad, act) -> Act:
    """w_taps [K*K, C]; pad = (pad_t, pad_b, pad_l, pad_r)."""
    lib = _lib.load()
    assert x.co == 0 and x.cs == x.C, "dwconv expects a dense NHWC tensor"
    Ho = (x.H + pad[0] + pad[1] - K) // stride + 1
    Wo = (x.W + pad[2] + pad[3] - K) // stride + 1
    out = Act.empty(x.N, Ho, Wo, x.C, x.buf.device)
    if DW_TILE and act in (ACT_NONE, ACT_SWISH) and x.C >= DW_TILE_MIN_C and (stride == 1 or K == 5) \
            and lib.creste_dwconv_se_tile_partial_count(Ho, Wo, x.C, K, stride) > 0:
        _lib.check(lib.creste_dwconv_tile_f32(x.ptr, _chk(w_taps).data_ptr(), _chk(bias).data_ptr() if bias is not None else None,
                                              out.ptr, x.N, x.H, x.W, x.C, Ho, Wo, K, stride, pad[0], pad[2], act, _stream()),
                   "dwconv_tile")
        return out
    if bias is None:
        bias = torch.zeros(x.C, dtype=torch.float32, device=x.buf.device)
    _lib.check(lib.creste_dwconv2d_nhwc_f32(x.ptr, _chk(w_taps).data_ptr(), _chk(bias).data_ptr(),
                                            out.ptr, x.N, x.H, x.W, x.C, Ho, Wo, K, stride, pad[0],
                                            pad[2], act, _stream()), "dwconv2d")
    return out


def dwconv2d_se(x: Act, w_taps, bias, K, stride, pad, act, se_w1, se_b1, se_w2, se_b2):
    """depthwise conv + activation and the squeeze-excite gate of its output (one read of the tensor)."""
    lib = _lib.load()
    assert x.co == 0 and x.cs == x.C, "dwconv expects a dense NHWC tensor"
    Ho = (x.H + pad[0] + pad[1] - K) // stride + 1
    Wo = (x.W + pad[2] + pad[3] - K) // stride + 1
    dev = x.buf.device
    out = Act.empty(x.N, Ho, Wo, x.C, dev)
    gate = torch.empty((x.N, x.C), dtype=torch.float32, device=dev)
    if TRACK_AMAX:
        out.amax = _AmaxPool.slot(dev)
    # measured at batch 16 (scripts/dwconv_micro.py): the tile kernel wins at stride 1 (1152 ch k5 19x38: 140 -> 83 us, 672 ch k5
    # 38x76: 182 -> 106, 240 ch k5 76x152: 217 -> 183) and at k5 / stride 2 (76 -> 70); k3 / stride 2 stays register-blocked
    use_tile = DW_TILE and act == ACT_SWISH and x.C >= DW_TILE_MIN_C and (stride == 1 or K == 5)
    ntile = lib.creste_dwconv_se_tile_partial_count(Ho, Wo, x.C, K, stride) if use_tile else -1
    if ntile > 0:          # many channels on a small map: the LDS-tile kernel (csrc/mbconv.hip)
        partial = torch.empty((x.N, ntile, x.C), dtype=torch.float32, device=dev)
        _lib.check(lib.creste_dwconv_se_tile_f32(x.ptr, _chk(w_taps).data_ptr(), _chk(bias).data_ptr(), out.ptr,
                                                 partial.data_ptr(), out.amax.data_ptr() if TRACK_AMAX else None,
                                                 x.N, x.H, x.W, x.C, Ho, Wo, K, stride, pad[0], pad[2], _stream()),
                   "dwconv_se_tile")
        _lib.check(lib.creste_se_gate_partial_f32(partial.data_ptr(), ntile, _chk(se_w1).data_ptr(), _chk(se_b1).data_ptr(),
                                                  _chk(se_w2).data_ptr(), _chk(se_b2).data_ptr(), gate.data_ptr(), x.N,
                                                  Ho * Wo, x.C, se_w1.shape[0], _stream()), "se_gate_partial")
        return out, gate
    partial = torch.empty((x.N, lib.creste_se_partial_count(Ho * Wo, x.C), x.C), dtype=torch.float32, device=dev)
    _lib.check(lib.creste_dwconv_se_nhwc_f32(x.ptr, _chk(w_taps).data_ptr(), _chk(bias).data_ptr(), out.ptr,
                                             partial.data_ptr(), out.amax.data_ptr() if TRACK_AMAX else None,
                                             x.N, x.H, x.W, x.C, Ho, Wo, K, stride, pad[0], pad[2], act,
                                             _stream()), "dwconv_se")
    _lib.check(lib.creste_se_gate_f32(None, partial.data_ptr(), _chk(se_w1).data_ptr(), _chk(se_b1).data_ptr(),
                                      _chk(se_w2).data_ptr(), _chk(se_b2).data_ptr(), gate.data_ptr(), x.N,
                                      Ho * Wo, x.C, se_w1.shape[0], _stream()), "se_gate")
    return out, gate


def mbconv_fusable(x: Act, Cexp: int, K: int, stride: int, Ho: int, Wo: int) -> bool:
    """is the fused expand + depthwise kernel built for this MBConv block AND a win?  Measured at batch 16 of 608x1216
    (scripts/mbconv_micro.py, whole block, f16x3): 16->96 k3/s2 950 -> 544 us, 24->144 k3/s1 712 -> 534, 24->144 k5/s2
    511 -> 370; 40->240 k5/s1 355 -> 595 and k3/s2 168 -> 184 LOSE (the exact-fp32 expand is VALU work that grows with
    Cin while the HBM round trip it saves shrinks with the map) -- so: up to 24 input channels."""
    return x.C <= 24 and _lib.load().creste_mbconv_partial_count(x.N, Ho, Wo, x.C, Cexp, K, stride) > 0


def mbconv_expand_dw_se(x: Act, w_expand, b_expand, w_taps, b_dw, K, stride, pad, se_w1, se_b1, se_w2, se_b2):
    """MBConv front half in one pass: expand 1x1 + BN + swish -> depthwise KxK + BN + swish -> squeeze-excite gate; the
    expanded tensor stays in LDS (csrc/mbconv.hip).  w_expand [Cin][Cexp], w_taps [K*K][Cexp] (BN-folded)."""
    lib = _lib.load()
    Cexp = w_expand.shape[1]
    Ho = (x.H + pad[0] + pad[1] - K) // stride + 1
    Wo = (x.W + pad[2] + pad[3] - K) // stride + 1
    dev = x.buf.device
    nchunk = lib.creste_mbconv_partial_count(x.N, Ho, Wo, x.C, Cexp, K, stride)
    if nchunk <= 0:
        raise HipLibraryError(f"mbconv_expand_dw is not built for Cin={x.C} Cexp={Cexp} K={K} stride={stride}")
    out = Act.empty(x.N, Ho, Wo, Cexp, dev)
    partial = torch.empty((x.N, nchunk, Cexp), dtype=torch.float32, device=dev)
    gate = torch.empty((x.N, Cexp), dtype=torch.float32, device=dev)
    if TRACK_AMAX:
        out.amax = _AmaxPool.slot(dev)
    _lib.check(lib.creste_mbconv_expand_dw_f32(
        x.ptr, x.N, x.H, x.W, x.C, x.cs, _chk(w_expand).data_ptr(), _chk(b_expand).data_ptr(), _chk(w_taps).data_ptr(),
        _chk(b_dw).data_ptr(), out.ptr, partial.data_ptr(), out.amax.data_ptr() if TRACK_AMAX else None, Cexp, Ho, Wo,
        K, stride, pad[0], pad[2], _stream()), "mbconv_expand_dw")
    _lib.check(lib.creste_se_gate_partial_f32(partial.data_ptr(), nchunk, _chk(se_w1).data_ptr(), _chk(se_b1).data_ptr(),
                                              _chk(se_w2).data_ptr(), _chk(se_b2).data_ptr(), gate.data_ptr(), x.N,
                                              Ho * Wo, Cexp, se_w1.shape[0], _stream()), "se_gate_partial")
    return out, gate


def stem_dw_fusable(x: Act, C1: int, H1: int, W1: int) -> bool:
    return x.C == 4 and x.cs == 4 and x.co == 0 and _lib.load().creste_stem_dw_partial_count(x.N, H1, W1, C1) > 0


def stem_dw_se(x: Act, w_stem, b_stem, pad, w_taps, b_dw, dpad, se_w1, se_b1, se_w2, se_b2):
    """encoder stem (3x3/2 conv of the 4-channel image + BN + swish) -> block 0's depthwise 3x3 + BN + swish -> SE gate
    in one pass (csrc/mbconv.hip); w_stem [36][C1] in (ky, kx, ci) order, w_taps [9][C1]."""
    lib = _lib.load()
    C1 = w_stem.shape[1]
    H1 = (x.H + pad[0] + pad[1] - 3) // 2 + 1
    W1 = (x.W + pad[2] + pad[3] - 3) // 2 + 1
    assert (H1 + dpad[0] + dpad[1] - 3 + 1, W1 + dpad[2] + dpad[3] - 3 + 1) == (H1, W1), "depthwise conv must keep the size"
    dev = x.buf.device
    nchunk = lib.creste_stem_dw_partial_count(x.N, H1, W1, C1)
    if nchunk <= 0 or x.C != 4 or x.cs != 4 or x.co != 0:
        raise HipLibraryError("stem_dw is built for a dense 4-channel NHWC image")
    out = Act.empty(x.N, H1, W1, C1, dev)
    partial = torch.empty((x.N, nchunk, C1), dtype=torch.float32, device=dev)
    gate = torch.empty((x.N, C1), dtype=torch.float32, device=dev)
    if TRACK_AMAX:
        out.amax = _AmaxPool.slot(dev)
    _lib.check(lib.creste_stem_dw_f32(x.ptr, x.N, x.H, x.W, _chk(w_stem).data_ptr(), _chk(b_stem).data_ptr(), pad[0], pad[2],
                                      _chk(w_taps).data_ptr(), _chk(b_dw).data_ptr(), dpad[0], dpad[2], out.ptr,
                                      partial.data_ptr(), out.amax.data_ptr() if TRACK_AMAX else None, C1, H1, W1,
                                      _stream()), "stem_dw")
    _lib.check(lib.creste_se_gate_partial_f32(partial.data_ptr(), nchunk, _chk(se_w1).data_ptr(), _chk(se_b1).data_ptr(),
                                              _chk(se_w2).data_ptr(), _chk(se_b2).data_ptr(), gate.data_ptr(), x.N,
                                              H1 * W1, C1, se_w1.shape[0], _stream()), "se_gate_partial")
    return out, gate


def se_gate(x: Act, w1, b1, w2, b2) -> torch.Tensor:
    lib = _lib.load()
    assert x.co == 0 and x.cs == x.C
    HW = x.H * x.W
    Cse = w1.shape[0]
    partial = torch.empty((x.N, lib.creste_se_partial_count(HW, x.C), x.C), dtype=torch.float32,
                          device=x.buf.device)
    gate = torch.empty((x.N, x.C), dtype=torch.float32, device=x.buf.device)
    _lib.check(lib.creste_se_gate_f32(x.ptr, partial.data_ptr(), _chk(w1).data_ptr(), _chk(b1).data_ptr(),
                                      _chk(w2).data_ptr(), _chk(b2).data_ptr(), gate.data_ptr(), x.N, HW,
                                      x.C, Cse, _stream()), "se_gate")
    return gate


def upsample_concat(x1: Act, skip: Act | None, Ho, Wo, rh, rw, out: Act | None = None) -> Act:
    lib = _lib.load()
    C2 = skip.C if skip is not None else 0
    fresh = out is None
    if out is None:
        out = Act.empty(x1.N, Ho, Wo, x1.C + C2, x1.buf.device)
    assert out.C == x1.C + C2 and (out.H, out.W) == (Ho, Wo)
    # max|out| without a reduction over the output: bilinear interpolation is a convex combination, so
    # max|out| <= max(max|x1|, max|skip|) -- and the sources' bounds are usually already known (the kernel's own
    # |max| update costs its 65k small workgroups a barrier pair and a contended L2 access each: ~9 % of its time)
    out.amax = None
    if fresh and TRACK_AMAX and x1.amax is not None and (skip is None or skip.amax is not None):
        out.amax = x1.amax if skip is None else max2(x1.amax, skip.amax)
    track = fresh and TRACK_AMAX and out.amax is None
    if track:
        out.amax = _AmaxPool.slot(x1.buf.device)
    _lib.check(lib.creste_upsample_concat_nhwc_f32(
        x1.ptr, x1.N, x1.H, x1.W, x1.C, x1.cs, skip.ptr if skip is not None else None, C2,
        skip.cs if skip is not None else 0, out.buf.data_ptr(), Ho, Wo, out.cs, out.co, float(rh),
        float(rw), out.amax.data_ptr() if track else None, _stream()), "upsample_concat")
    return out


def maxpool2(x: Act, Ho=None, Wo=None, ds: int = 2) -> Act:
    """F.max_pool2d(x, ds, ds) restricted to the first Ho x Wo pooled pixels (ds in {1, 2, 4}; ds = 1 is a crop)."""
    lib = _lib.load()
    Ho = Ho if Ho is not None else x.H // ds
    Wo = Wo if Wo is not None else x.W // ds
    out = Act.empty(x.N, Ho, Wo, x.C, x.buf.device)
    if TRACK_AMAX:
        out.amax = _AmaxPool.slot(x.buf.device)
    _lib.check(lib.creste_maxpool_nhwc_f32(x.ptr, x.N, x.H, x.W, x.C, x.cs, out.ptr, Ho, Wo, out.cs, int(ds),
                                           out.amax.data_ptr() if TRACK_AMAX else None, _stream()), "maxpool")
    return out


def affine_act(x: Act, scale, shift, act) -> Act:
    lib = _lib.load()
    out = Act.empty(x.N, x.H, x.W, x.C, x.buf.device)
    _lib.check(lib.creste_affine_act_nhwc_f32(x.ptr, x.cs, _chk(scale).data_ptr(), _chk(shift).data_ptr(),
                                              out.ptr, out.cs, x.N * x.H * x.W, x.C, act, _stream()),
               "affine_act")
    return out


def resize_plane(x: torch.Tensor, Ho, Wo, Hd, rh, rw, out: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    N, H, W = x.shape
    _lib.check(lib.creste_resize_plane_f32(_chk(x).data_ptr(), N, H, W, _chk(out).data_ptr(), Ho, Wo, Hd,
                                           float(rh), float(rw), _stream()), "resize_plane")
    return out


def nchw_to_nhwc(x: torch.Tensor, out: Act | None = None) -> Act:
    lib = _lib.load()
    N, Cc, H, W = x.shape
    if out is None:
        out = Act.empty(N, H, W, Cc, x.device)
    assert (out.N, out.H, out.W, out.C) == (N, H, W, Cc)
    _lib.check(lib.creste_nchw_to_nhwc_f32(_chk(x).data_ptr(), out.ptr, out.cs, N, Cc, H, W, _stream()),
               "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: Act) -> torch.Tensor:
    lib = _lib.load()
    out = torch.empty((x.N, x.C, x.H, x.W), dtype=torch.float32, device=x.buf.device)
    _lib.check(lib.creste_nhwc_to_nchw_f32(x.ptr, x.cs, out.data_ptr(), x.N, x.C, x.H, x.W, _stream()),
               "nhwc_to_nchw")
    return out


def lidar_depth_image(points: torch.Tensor, lidar2cam: torch.Tensor, H: int, W: int, out: torch.Tensor,
                      scale: float = 1.0, reduce: str = "max") -> torch.Tensor:
    """points [B,NP,>=3] fp32, lidar2cam [B,3|4,4] float64 -> `out` [B,H,W] (may be a strided batch of
    contiguous HxW planes, e.g. rgbd[:, 0, 3]) filled with the per-pixel max/min camera depth * scale."""
    lib = _lib.load()
    _chk(points, name="points")
    _chk(lidar2cam, torch.float64, name="lidar2cam")
    B, NP, ps = points.shape
    if not out.is_cuda or out.dtype != torch.float32 or tuple(out.shape) != (B, H, W) or \
            out.stride(2) != 1 or out.stride(1) != W:
        raise HipLibraryError("lidar_depth_image: out must be CUDA fp32 [B,H,W] with contiguous HxW planes")
    _lib.check(lib.creste_lidar_depth_image_f32(points.data_ptr(), ps, lidar2cam.data_ptr(),
                                                lidar2cam.shape[1] * 4, B, NP, H, W, int(reduce == "min"),
                                                float(scale), out.data_ptr(), out.stride(0) if B > 1 else H * W,
                                                _stream()), "lidar_depth_image")
    return out


def lidar_pixels_to_depth(points: torch.Tensor, lidar2cam: torch.Tensor, H: int, W: int, reduce: str = "max"):
    """One scan through `creste_lidar_pixels_to_depth_f64`: points [NP,>=3] fp32 | float64 (CUDA), lidar2cam [3|4,4]
    float64 -> (uv int32 [NP,2], mask bool [NP], reduced float64 [H,W], last_write fp32 [H,W])."""
    lib = _lib.load()
    if reduce not in ("max", "min"):
        raise ValueError(f"Invalid depth_priority {reduce}")
    if not points.is_cuda or points.dim() != 2 or points.shape[1] < 3 or points.dtype not in (torch.float32, torch.float64) \
            or not points.is_contiguous():
        raise HipLibraryError("lidar_pixels_to_depth: points must be a contiguous CUDA fp32 / float64 [NP,>=3] tensor")
    _chk(lidar2cam, torch.float64, name="lidar2cam")
    NP, dev = points.shape[0], points.device
    uv = torch.empty((NP, 2), dtype=torch.int32, device=dev)
    mask = torch.empty((NP,), dtype=torch.uint8, device=dev)
    reduced = torch.empty((H, W), dtype=torch.float64, device=dev)
    last = torch.empty((H, W), dtype=torch.float32, device=dev)
    work = torch.empty((2 * H * W,), dtype=torch.int64, device=dev)
    _lib.check(lib.creste_lidar_pixels_to_depth_f64(points.data_ptr(), int(points.dtype == torch.float64), points.shape[1],
                                                    lidar2cam.data_ptr(), NP, H, W, int(reduce == "min"), uv.data_ptr(),
                                                    mask.data_ptr(), reduced.data_ptr(), last.data_ptr(), work.data_ptr(),
                                                    _stream()), "lidar_pixels_to_depth")
    return uv, mask.bool(), reduced, last


def depth_expectation(logits: Act, bin_values: torch.Tensor):
    lib = _lib.load()
    P = logits.N * logits.H * logits.W
    depth = rows_empty((logits.N, logits.H, logits.W), torch.float32, logits.buf.device)
    bins = rows_empty((logits.N, logits.H, logits.W), torch.int64, logits.buf.device)
    _lib.check(lib.creste_depth_expectation_f32(logits.ptr, logits.cs, P, logits.C,
                                                _chk(bin_values).data_ptr(), depth.data_ptr(),
                                                bins.data_ptr(), _stream()), "depth_expectation")
    return depth, bins


def pixel_geometry(depth, p2p, bounds6, w1, b1, w2, b2, zfeat: Act):
    """depth [B,Hs,Ws] m, p2p [B,4,4] -> xyz [B,P,3], mask [B,P]; z features into `zfeat` slice."""
    lib = _lib.load()
    B, Hs, Ws = depth.shape
    xyz = torch.empty((B, Hs * Ws, 3), dtype=torch.float32, device=depth.device)
    mask = torch.empty((B, Hs * Ws), dtype=torch.float32, device=depth.device)
    zhid, zdim = w1.shape[0], w2.shape[0]
    assert zfeat.C == zdim
    _lib.check(lib.creste_pixel_geometry_f32(
        _chk(depth).data_ptr(), _chk(p2p).data_ptr(), B, Hs, Ws, _chk(bounds6).data_ptr(),
        _chk(w1).data_ptr(), _chk(b1).data_ptr(), _chk(w2).data_ptr(), _chk(b2).data_ptr(), zhid, zdim,
        xyz.data_ptr(), mask.data_ptr(), zfeat.buf.data_ptr(), zfeat.cs, zfeat.co, _stream()),
        "pixel_geometry")
    return xyz, mask


SPLAT_MODES = {"mean": 0, "sum": 1, "max": 2}
KEYED_GEOMETRY = True     # the splat plan's key kernel inside the pixel geometry (one launch, one read of xyz less)


def pixel_geometry_plan(depth, p2p, bounds6, w1, b1, w2, b2, zfeat: Act, off_xy, vox_xy, GH, GW):
    """pixel_geometry + bev_splat_plan of the same points (one camera per frame) -> xyz [B,P,3], mask [B,P], SplatPlan.  With
    the shipped 1 -> 64 -> 32 z-MLP the plan's first kernel (voxel coordinates + base-cell keys, reference
    splat_projection.py:185-187) runs inside the geometry kernel; any other z-MLP takes the two calls."""
    zhid, zdim = w1.shape[0], w2.shape[0]
    if not (KEYED_GEOMETRY and zhid == 64 and zdim == 32 and zfeat.cs % 4 == 0 and zfeat.co % 4 == 0):
        xyz, mask = pixel_geometry(depth, p2p, bounds6, w1, b1, w2, b2, zfeat)
        with shared_rows():
            return xyz, mask, bev_splat_plan(xyz, off_xy, vox_xy, GH, GW)
    lib = _lib.load()
    B, Hs, Ws = depth.shape
    P, dev = Hs * Ws, depth.device
    xyz = torch.empty((B, P, 3), dtype=torch.float32, device=dev)
    mask = torch.empty((B, P), dtype=torch.float32, device=dev)
    with shared_rows():
        coords = rows_empty((B, P, 2), torch.float32, dev)
    work = torch.empty(lib.creste_bev_splat_workspace_bytes(B, P, GH, GW), dtype=torch.uint8, device=dev)
    assert zfeat.C == zdim
    _lib.check(lib.creste_pixel_geometry_keyed_f32(
        _chk(depth).data_ptr(), _chk(p2p).data_ptr(), B, Hs, Ws, _chk(bounds6).data_ptr(),
        _chk(w1).data_ptr(), _chk(b1).data_ptr(), _chk(w2).data_ptr(), _chk(b2).data_ptr(), zhid, zdim,
        xyz.data_ptr(), mask.data_ptr(), zfeat.buf.data_ptr(), zfeat.cs, zfeat.co, float(off_xy[0]), float(off_xy[1]),
        float(vox_xy[0]), float(vox_xy[1]), GH, GW, coords.data_ptr(), work.data_ptr(), _stream()), "pixel_geometry_keyed")
    return xyz, mask, bev_splat_plan_keyed(coords, work, B, P, GH, GW)


def bev_splat_plan_keyed(coords, work, B, P, GH, GW) -> SplatPlan:
    """the rest of the binning plan behind a keyed pixel geometry (CSR build, record fill, per-cell sort)"""
    _lib.check(_lib.load().creste_bev_splat_plan_keyed_f32(B, P, GH, GW, coords.data_ptr(), work.data_ptr(), _stream()),
               "bev_splat_plan_keyed")
    return SplatPlan(coords, work, B, P, GH, GW)


class SplatPlan:
    """Binning plan of one batch of points (creste_bev_splat_plan_f32): bev_coords + the workspace the gather reads."""

    def __init__(self, coords, work, B, P, GH, GW):
        self.coords, self.work, self.B, self.P, self.GH, self.GW = coords, work, B, P, GH, GW


def bev_splat_plan(xyz, off_xy, vox_xy, GH, GW) -> SplatPlan:
    """xyz [B,P,3] -> SplatPlan.  Needs only the points: the model enqueues it right after `pixel_geometry`, ahead of the
    fusion conv that produces the features (reference splat_projection.py:185-187 and the index half of :293-333)."""
    lib = _lib.load()
    B, P, _ = xyz.shape
    dev = xyz.device
    coords = rows_empty((B, P, 2), torch.float32, dev)
    work = torch.empty(lib.creste_bev_splat_workspace_bytes(B, P, GH, GW), dtype=torch.uint8, device=dev)
    _lib.check(lib.creste_bev_splat_plan_f32(_chk(xyz).data_ptr(), B, P, float(off_xy[0]), float(off_xy[1]),
                                             float(vox_xy[0]), float(vox_xy[1]), GH, GW, coords.data_ptr(),
                                             work.data_ptr(), _stream()), "bev_splat_plan")
    return SplatPlan(coords, work, B, P, GH, GW)


def bev_splat_gather(plan: SplatPlan, feats: Act, min_weight=1.0, scatter_mode="mean"):
    """plan + feats Act viewed as [B,P,F] -> (bev Act [B,GH,GW,F], dens [B,GH,GW]); same stream as the plan."""
    lib = _lib.load()
    if scatter_mode not in SPLAT_MODES:
        raise Exception("Unknown splat scatter mode:", scatter_mode)
    B, P, GH, GW, F, dev = plan.B, plan.P, plan.GH, plan.GW, feats.C, plan.coords.device
    if feats.N * feats.H * feats.W != B * P:
        raise HipLibraryError(f"bev_splat_gather: {feats.N * feats.H * feats.W} feature rows for a plan of {B * P} points")
    bev = Act.empty(B, GH, GW, F, dev)
    dens = rows_empty((B, GH, GW), torch.float32, dev)
    _lib.check(lib.creste_bev_splat_gather_f32(feats.ptr, feats.cs, B, P, F, GH, GW, float(min_weight),
                                               SPLAT_MODES[scatter_mode], bev.ptr, dens.data_ptr(), plan.work.data_ptr(),
                                               _stream()), "bev_splat_gather")
    # |bev| <= max|feats|: 'mean' divides the tap-weighted sum by max(sum of weights, min_weight >= 1), 'max' takes
    # w * f with w <= 1 -- the consumer's operand bound needs no pass over the 25 MB/frame map
    if scatter_mode in ("mean", "max") and min_weight >= 1.0 and feats.amax is not None:
        bev.amax = feats.amax
    return bev, dens


def bev_splat(xyz, feats: Act, off_xy, vox_xy, GH, GW, min_weight=1.0, scatter_mode="mean"):
    """xyz [B,P,3], feats Act viewed as [B,P,F] -> (coords [B,P,2], bev Act [B,GH,GW,F], dens [B,GH,GW]).
    scatter_mode: the reference's 'mean' | 'sum' | 'max' (splat_projection.py:334-352).  Plan + gather back to back."""
    if scatter_mode not in SPLAT_MODES:
        raise Exception("Unknown splat scatter mode:", scatter_mode)
    plan = bev_splat_plan(xyz, off_xy, vox_xy, GH, GW)
    bev, dens = bev_splat_gather(plan, feats, min_weight, scatter_mode)
    return plan.coords, bev, dens


def bev_splat_bwd(coords, feats: Act, g_bev: Act, g_dens, bev: Act, dens, vox_xy, min_weight=1.0, scatter_mode="mean"):
    """cotangents of (bev, dens) -> (g_feats Act viewed as [B,P,F], g_xyz [B,P,3]); see creste_bev_splat_mode_bwd_f32."""
    lib = _lib.load()
    B, P, _ = coords.shape
    F, GH, GW = feats.C, bev.H, bev.W
    dev = coords.device
    assert g_bev.cs == F and g_bev.co == 0 and bev.cs == F and bev.co == 0, "dense [B,GH,GW,F] maps expected"
    g_feats = Act.empty(feats.N, feats.H, feats.W, F, dev)
    g_xyz = torch.empty((B, P, 3), dtype=torch.float32, device=dev)
    work = torch.empty(B * GH * GW, dtype=torch.float32, device=dev)
    _lib.check(lib.creste_bev_splat_mode_bwd_f32(_chk(coords).data_ptr(), feats.ptr, feats.cs, g_bev.ptr,
                                                 _chk(g_dens).data_ptr() if g_dens is not None else None, bev.ptr,
                                                 _chk(dens).data_ptr(), B, P, F, GH, GW, float(vox_xy[0]), float(vox_xy[1]),
                                                 float(min_weight), SPLAT_MODES[scatter_mode], g_feats.ptr, g_feats.cs,
                                                 g_xyz.data_ptr(), work.data_ptr(), _stream()), "bev_splat_bwd")
    return g_feats, g_xyz


def depth_expectation_bwd(logits: Act, bin_values, g_depth, g_logits: Act | None = None):
    """g_logits (+)= d depth / d logits * g_depth (accumulates when g_logits is given)."""
    lib = _lib.load()
    acc = g_logits is not None
    if not acc:
        g_logits = Act.empty(logits.N, logits.H, logits.W, logits.C, logits.buf.device)
    _lib.check(lib.creste_depth_expectation_bwd_f32(logits.ptr, logits.cs, logits.N * logits.H * logits.W, logits.C,
                                                    _chk(bin_values).data_ptr(), _chk(g_depth).data_ptr(), g_logits.ptr,
                                                    g_logits.cs, int(acc), _stream()), "depth_expectation_bwd")
    return g_logits


def check_vi_sweeps(sweeps: torch.Tensor) -> int:
    """The contract of `value_iteration`'s asynchronous sweep count, checked on the host (synchronises): > 0 converged
    after that many sweeps; -n: NOT converged within n sweeps (v / q / policy are the state after n sweeps: discount >= 1,
    diverging rewards, max_sweeps too small); INT32_MIN: the persistent solver's workgroups were not all resident and gave
    up waiting (another full-chip kernel or a second solve shared the device) -- the outputs are unusable."""
    n = int(sweeps.item())
    if n == -2 ** 31:
        raise HipLibraryError("value_iteration: the persistent solver could not get all of its workgroups resident "
                              "(another kernel held the device); outputs are invalid -- run the solve alone on the device "
                              "or set CRESTE_VI_MULTI=1")
    if n <= 0:
        raise HipLibraryError(f"value_iteration: no convergence within {-n} sweeps (reference vin.py:68-74 would still be "
                              "iterating): check the discount (< 1) and the reward scale")
    return n


VI_ABORTED = -2 ** 31
# per DEVICE: [(event, pinned int32[1])] of solves whose sweep count has not been looked at yet, and the ring of pinned slots
# (one shared list would hand device 1's check to whoever polls on device 0, and two threads driving two devices
# would race on the ring position)
_vi_state: dict = {}
_vi_lock = threading.Lock()
_vi_tls = threading.local()      # .chunked > 0: solves of THIS thread take the launch-per-chunk form (vi_launch_per_chunk)


def _vi_dev(device) -> dict:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    with _vi_lock:
        st = _vi_state.get(idx)
        if st is None:
            # (one pinned allocation per device: they cost ~100 us of host time each)
            st = _vi_state[idx] = {"pending": [], "ring": torch.empty(16, dtype=torch.int32).pin_memory(), "pos": 0,
                                   "lock": threading.Lock()}
    return st


def _vi_note(sweeps: torch.Tensor):
    """queue the asynchronous look at a solve's sweep count: a 4-byte copy into pinned memory behind the solve + an event"""
    if _lib._recorder is not None or torch.cuda.is_current_stream_capturing():
        return
    st = _vi_dev(sweeps.device)
    if len(st["pending"]) >= 12:                                          # never reuse a slot that is still unlooked-at
        vi_check(wait=True, device=sweeps.device)
    with st["lock"]:
        host = st["ring"][st["pos"]:st["pos"] + 1]
        st["pos"] = (st["pos"] + 1) % 16
        host.copy_(sweeps, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(sweeps.device))
        st["pending"].append((ev, host))


def vi_poll(wait: bool = True, device=None) -> list:
    """The sweep counts of the solves issued since the last poll (on `device`; None: on every device) whose result has arrived
    (all of them with wait=True: waits for the LAST SOLVE's event, not for the stream -- work queued behind the solves keeps
    the device busy; wait=False waits only while more than four solves are unlooked-at).  The caller decides what a negative
    count means for it; `vi_check()` raises."""
    out = []
    if device is None:
        with _vi_lock:
            states = list(_vi_state.values())
    else:
        states = [_vi_dev(device)]
    for st in states:
        with st["lock"]:
            pend = st["pending"]
            while pend and (wait or len(pend) > 4 or pend[0][0].query()):
                ev, host = pend.pop(0)
                ev.synchronize()
                out.append(int(host[0]))
    return out


def vi_check(wait: bool = True, device=None):
    """Raise if a solve issued since the last poll failed (see check_vi_sweeps).  Runs by itself at the head of every
    `value_iteration` (wait=False there: back-to-back solves stay asynchronous, at most four behind), waiting in IRLTrainer
    before every optimiser step and at the end of its validation / epoch hooks, and in `VIN.last_sweeps` -- call it with
    wait=True after the LAST solve of a loop of your own (nothing else looks at that one)."""
    for n in vi_poll(wait, device):
        check_vi_sweeps(torch.tensor([n], dtype=torch.int32))


def value_iteration(r: torch.Tensor, discount: float, threshold: float = 1e-3, max_sweeps: int = 100000, chunked: bool = False):
    """r [B,H,W] -> v [B,H,W], q [B,8,H,W], policy [B,8,H,W], sweeps (device int32 tensor).  The call is asynchronous:
    failures are reported in the SIGN of `sweeps` (see `check_vi_sweeps`).  Every solve's count is copied to pinned memory
    behind it and checked at the head of a later solve on the same device (`vi_check`; IRLTrainer checks before each optimiser
    step and redoes a step whose solve was aborted through the launch-per-chunk form); `VIN.last_sweeps` checks on demand;
    the last solve of a loop is only checked by an explicit `vi_check()`.  Under CRESTE_CHECK_VI=1 every call checks at once
    (one host sync per solve) and an aborted persistent solve is redone in the launch-per-chunk form before returning.
    chunked (or inside `vi_launch_per_chunk()`): the launch-per-chunk form (creste_value_iteration_chunked_f32)."""
    lib = _lib.load()
    B, H, W = r.shape
    dev = r.device
    eager = _lib._recorder is None and not torch.cuda.is_current_stream_capturing()
    if eager:
        vi_check(wait=False, device=dev)
    v = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    q = torch.empty((B, 8, H, W), dtype=torch.float32, device=dev)
    pi = torch.empty((B, 8, H, W), dtype=torch.float32, device=dev)
    sweeps = torch.zeros(1, dtype=torch.int32, device=dev)
    work = torch.empty(lib.creste_value_iteration_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)

    def solve(per_chunk: bool):
        fn = lib.creste_value_iteration_chunked_f32 if per_chunk else lib.creste_value_iteration_f32
        _lib.check(fn(_chk(r).data_ptr(), B, H, W, float(discount), float(threshold), int(max_sweeps), v.data_ptr(),
                      q.data_ptr(), pi.data_ptr(), sweeps.data_ptr(), work.data_ptr(), _stream()), "value_iteration")
    per_chunk = bool(chunked) or getattr(_vi_tls, "chunked", 0) > 0
    solve(per_chunk)
    if eager and os.environ.get("CRESTE_CHECK_VI") == "1":
        if int(sweeps.item()) == VI_ABORTED and not per_chunk:
            solve(True)
        check_vi_sweeps(sweeps)
    elif eager:
        _vi_note(sweeps)
    return v, q, pi, sweeps


class vi_launch_per_chunk:
    """`with vi_launch_per_chunk():` -- solves THIS THREAD issues inside take the launch-per-chunk form (host-synchronous, needs
    no co-residency): the retry path of a persistent solve that reported VI_ABORTED.  (A thread-local flag handed to the C call
    as its entry point -- not the process environment)"""

    def __enter__(self):
        _vi_tls.chunked = getattr(_vi_tls, "chunked", 0) + 1

    def __exit__(self, *exc):
        _vi_tls.chunked -= 1
        return False


def expected_svf(policy, expert_xy, fov_u8, T, ds, temperature, sharpen=True, zero_terminal=False):
    lib = _lib.load()
    B, A, H, W = policy.shape
    assert A == 8
    if expert_xy.dim() != 3 or expert_xy.shape[0] != B or expert_xy.shape[2] != 2:
        raise HipLibraryError(f"expected_svf: expert_xy must be [B,T_expert,2], got {tuple(expert_xy.shape)}")
    Te = expert_xy.shape[1]        # poses per expert trajectory; independent of the rollout horizon T (lfd.py:171-177)
    dev = policy.device
    sharp = torch.empty_like(policy)
    svf = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    states = torch.empty((B, T, 2), dtype=torch.int64, device=dev)
    grid = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    _lib.check(lib.creste_expected_svf_f32(_chk(policy).data_ptr(), _chk(expert_xy).data_ptr(),
                                           _chk(fov_u8, torch.uint8).data_ptr(), B, H, W, T, Te, float(ds),
                                           float(temperature), int(sharpen), int(zero_terminal),
                                           sharp.data_ptr(), svf.data_ptr(), states.data_ptr(),
                                           grid.data_ptr(), _stream()), "expected_svf")
    return svf, states, grid
