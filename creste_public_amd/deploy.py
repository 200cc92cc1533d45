"""Deployment artefact: the counterpart of the reference's scripts/runtime/compile.py:160-210, which traces
`TraversabilityModel(solve_mdp=False)` / `TerrainNet` with `torch.jit.trace(model, (inputs,))`, dry-runs the traced
module on `(rgbd, p2p)` and saves it.

Here "compiling" is capturing the model's kernel launches into ONE hipGraph (no tracing compiler): after two
eager warm-up passes (lazy kernel loads, weight packing) the forward is recorded on static input / output
buffers; a call copies the new frame into the input buffers and replays the graph -- ~400 launches become one
submission, results are bit-identical to the eager path.  The saved artefact is a weight pack (config + state_dict
+ conv operand mode + input shapes), not a serialized program: `load()` rebuilds the module from it and captures
again on the target GPU.

    cm = deploy.compile_model(model, (rgbd, p2p))      # model.eval() on the GPU
    out = cm((rgbd, p2p))                              # dict of tensors (views of the static output buffers)
    cm.save("traversability_hip.pt")
    cm = deploy.load("traversability_hip.pt", device="cuda:0")
"""
from __future__ import annotations

import torch

from . import hipnn, ops
from .config import Cfg

FORMAT = "creste_hip_deploy/1"
_TYPES = {"MaxEntIRL": "traversability", "TerrainNet": "bev_map"}


class CompiledModel:
    def __init__(self, model: torch.nn.Module, example_inputs, warmup: int = 2):
        if model.training:
            raise ValueError("compile_model: put the model in eval() mode (inference artefact)")
        self.model = model
        self.precision = hipnn.get_precision()
        self.static_in = tuple(t.detach().clone() for t in example_inputs)
        dev = self.static_in[0].device
        if dev.type != "cuda":
            raise ops.HipLibraryError("compile_model: the model and the example inputs must live on the GPU")
        with torch.no_grad():
            for _ in range(max(1, warmup)):
                model(self.static_in)
            torch.cuda.synchronize(dev)
            ops.reset_amax_pool()                 # |max| slots used inside the graph must be zeroed BY the graph
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = model(self.static_in)
        self.graph.replay()
        torch.cuda.synchronize(dev)

    def __call__(self, inputs, clone: bool = False) -> dict:
        """inputs = (rgbd [B,1,4,H,W], p2p [B,1,4,4]) with the captured shapes.  The returned tensors alias the
        static output buffers and are overwritten by the next call unless `clone=True`."""
        if len(inputs) != len(self.static_in):
            raise ValueError(f"expected {len(self.static_in)} input tensors")
        for dst, src in zip(self.static_in, inputs):
            if tuple(dst.shape) != tuple(src.shape):
                raise ValueError(f"captured for input shape {tuple(dst.shape)}, got {tuple(src.shape)}: "
                                 "compile one artefact per batch size / resolution")
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        if clone:
            return {k: v.clone() for k, v in self.static_out.items()}
        return self.static_out

    forward = __call__

    def save(self, path: str):
        name = type(self.model).__name__
        if name not in _TYPES:
            raise NotImplementedError(f"deployment artefact for {name}")
        torch.save({"format": FORMAT, "model_type": _TYPES[name], "cfg": _plain(self.model.model_cfg),
                    "state_dict": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                    "precision": self.precision,
                    "input_shapes": [tuple(t.shape) for t in self.static_in]}, path)


def _plain(cfg):
    if isinstance(cfg, dict):
        return {k: _plain(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [_plain(v) for v in cfg]
    return cfg


def compile_model(model, example_inputs, warmup: int = 2) -> CompiledModel:
    return CompiledModel(model, example_inputs, warmup)


def load(path: str, device="cuda:0", example_inputs=None) -> CompiledModel:
    """Rebuild the module from a saved artefact and capture it on `device` (zeros stand in for the example inputs
    unless given)."""
    from .creste.models.lfd import MaxEntIRL
    from .creste.models.terrainnet import TerrainNet
    ck = torch.load(path, weights_only=False, map_location="cpu")
    if ck.get("format") != FORMAT:
        raise ValueError(f"{path}: not a {FORMAT} artefact")
    hipnn.set_precision(ck["precision"])
    cls = {"traversability": MaxEntIRL, "bev_map": TerrainNet}[ck["model_type"]]
    model = cls(Cfg(ck["cfg"]))
    model.load_state_dict(ck["state_dict"], strict=True)
    dev = torch.device(device)
    model = model.to(dev).eval()
    if example_inputs is None:
        example_inputs = tuple(torch.zeros(s, dtype=torch.float32) for s in ck["input_shapes"])
        example_inputs[1][...] = torch.eye(4)              # a valid pixel->LiDAR transform for the warm-up
    return CompiledModel(model, tuple(t.to(dev) for t in example_inputs))


# ------------------------------------------------------------------------------------------------------------------
# Python-free artefact: a launch PLAN for libcreste_hip.so's creste_hip_model_load / _infer (csrc/plan_runtime.cpp).
# reference scripts/runtime/compile.py:160-210 (torch.jit.trace -> a module the C++ stack loads without Python).
PLAN_MAGIC = b"CRESTEPLAN\0\0"
_DTYPES = {torch.float32: 0, torch.int64: 1, torch.uint8: 2, torch.bool: 2}
_DESC_PTR_FIELDS = ("in_", "wpk", "bias", "res", "a_scale", "row_mask", "out", "work", "a_amax", "out_amax", "w_unscale", "up_src")


def _snapshot(device):
    """[(segment address, size, [(block address, size, allocated?)])] of the caching allocator on `device`."""
    segs = []
    for s in torch.cuda.memory_snapshot():
        if s["device"] != device.index:
            continue
        addr, blocks = s["address"], []
        for b in s["blocks"]:
            ba = b.get("address", addr)
            blocks.append((ba, b["size"], b["state"] == "active_allocated"))
            addr = ba + b["size"]
        segs.append((s["address"], s["total_size"], blocks))
    return sorted(segs)


def export_plan(model: torch.nn.Module, example_inputs, path: str, warmup: int = 2, pipelined: bool = True) -> dict:
    """Trace ONE eval-mode forward of `model` (MaxEntIRL with solve_mdp=False, or TerrainNet) on `example_inputs` =
    (rgbd [B,1,4,H,W], p2p [B,1,4,4]) and write the plan file `path`:

      * `pipelined` (default): a forward that the model runs as parts on several streams (batches of >= 12 frames:
        ops.forward_in_parts) is recorded AS SUCH -- every call carries its stream, the fork / join / buffer-ordering edges
        are recorded as event record / wait pairs, and the C runtime replays them on streams of its own: the Python-free
        artefact pipelines exactly as the Python path does (False: the forward is traced on one stream);

      * every launching C-ABI call of the forward, in order (entry-point name, scalars, conv descriptors);
      * the allocator segments those calls address (the arena) -- each pointer becomes (segment, offset);
      * the bytes of every block that was ALREADY allocated before the traced forward and is read by it (packed,
        BatchNorm-folded weights, biases, bin values, geometry constants): raw little-endian bytes, no pickle;
      * the two inputs and every tensor of the output dict (name, arena location, dtype, shape, strides).

    The forward must consist of this library's launches only (tests/test_deploy_plan_gpu.py compares the replay with
    the Python path bit for bit).  Returns a summary dict."""
    import bisect
    import ctypes as C
    import gc
    import struct

    from . import _lib
    if model.training:
        raise ValueError("export_plan: put the model in eval() mode")
    if getattr(model, "solve_mdp", False):
        raise NotImplementedError("export_plan: the MDP solve (host-checked convergence) is not part of the "
                                  "deployment graph -- the reference traces solve_mdp=False as well")
    static_in = tuple(t.detach().clone().contiguous() for t in example_inputs)
    dev = static_in[0].device
    if dev.type != "cuda":
        raise ops.HipLibraryError("export_plan: the model and the example inputs must live on the GPU")
    lib = _lib.load()
    with torch.no_grad():
        for _ in range(max(1, warmup)):
            model(static_in)
        torch.cuda.synchronize(dev)
        gc.collect()
        before = _snapshot(dev)                               # persistent state: parameters, packed weights, inputs
        ops.reset_amax_pool()                                 # |max| slot blocks are filled INSIDE the traced forward
        rec = _lib.PlanRecorder(pipelined)
        main_handle = int(torch.cuda.current_stream(dev).cuda_stream)
        _lib._recorder = rec
        try:
            out = model(static_in)
        finally:
            _lib._recorder = None
        torch.cuda.synchronize(dev)
    after = _snapshot(dev)
    seg_addr = [s[0] for s in after]
    persistent = sorted((ba, sz) for _, _, blocks in before for ba, sz, alloc in blocks if alloc)
    pers_addr = [b[0] for b in persistent]
    in_blocks = {t.data_ptr(): t for t in static_in}

    used_segments, const_blocks = {}, {}

    def locate(p: int):
        """device address -> (plan segment index, offset); registers the segment and, when the address lies in a block
        that predates the traced forward (and is not an input), that block as a constant."""
        if p == 0:
            return 0xffffffff, 0
        i = bisect.bisect_right(seg_addr, p) - 1
        if i < 0 or p >= after[i][0] + after[i][1]:
            raise ops.HipLibraryError(f"export_plan: pointer {p:#x} is not inside a caching-allocator segment")
        si = used_segments.setdefault(i, len(used_segments))
        j = bisect.bisect_right(pers_addr, p) - 1
        if j >= 0 and p < persistent[j][0] + persistent[j][1]:
            ba, sz = persistent[j]
            if not any(ba <= a < ba + sz for a in in_blocks):
                const_blocks[ba] = sz
        return si, p - after[i][0]

    calls, fn_ids = [], {}
    stream_ids = {main_handle: 0}                      # stream 0 = the stream the caller hands to creste_hip_model_infer
    for name, args, handle in rec:
        fid = fn_ids.setdefault(name, len(fn_ids))
        sid = stream_ids.setdefault(handle, len(stream_ids))
        enc = []
        for kind, v in args:
            if kind == "p":
                enc.append((4, locate(v)))
            elif kind == "desc":
                d = _lib.ConvDesc.from_buffer_copy(v)
                rel = []
                for f in _DESC_PTR_FIELDS:
                    val = getattr(d, f)
                    if val:
                        rel.append((getattr(_lib.ConvDesc, f).offset, locate(int(val))))
                enc.append((5, (v, rel)))
            else:
                enc.append(({"i": 0, "l": 1, "f": 2, "d": 3}[kind], v))
        calls.append((fid, sid, enc))

    def tensor_entry(name, t):
        if t.dtype not in _DTYPES or t.dim() > 6:
            raise ops.HipLibraryError(f"export_plan: output {name}: unsupported dtype / rank")
        seg, off = locate(t.data_ptr())
        shape = list(t.shape) + [0] * (6 - t.dim())
        stride = list(t.stride()) + [0] * (6 - t.dim())
        span = (sum((s - 1) * st for s, st in zip(t.shape, t.stride())) + 1) * t.element_size() if t.numel() else 0
        return name, seg, off, span, _DTYPES[t.dtype], t.dim(), shape, stride

    inputs = [tensor_entry(n, t) for n, t in zip(("rgbd", "p2p"), static_in)]
    outputs = [tensor_entry(k, v) for k, v in out.items() if torch.is_tensor(v)]
    # constants may have been registered by the tensor entries too (none expected); read their bytes now
    consts = []
    for ba, sz in sorted(const_blocks.items()):
        seg, off = locate(ba)
        host = (C.c_char * sz)()
        _lib.check(lib.creste_hip_memcpy_d2h(C.addressof(host), ba, sz), "memcpy_d2h")
        consts.append((seg, off, bytes(host)))

    def s_(b: bytes):
        return struct.pack("<I", len(b)) + b

    if len(stream_ids) > rec.MAX_STREAMS or rec.num_events > rec.MAX_EVENTS or len(used_segments) > 4096:
        raise ops.HipLibraryError(f"export_plan: {len(stream_ids)} streams / {rec.num_events} events / {len(used_segments)} segments exceed "
                              f"the plan loader's limits ({rec.MAX_STREAMS} / {rec.MAX_EVENTS} / 4096, csrc/plan_runtime.cpp)")

    info = (f"{type(model).__name__} precision={hipnn.get_precision()} inputs="
            f"{[tuple(t.shape) for t in static_in]} calls={len(calls)} streams={len(stream_ids)} abi={_lib.ABI_VERSION}").encode()
    seg_sizes = [None] * len(used_segments)
    for i, si in used_segments.items():
        seg_sizes[si] = after[i][1]
    blob = [PLAN_MAGIC, struct.pack("<III", 3, C.sizeof(_lib.ConvDesc), _lib.ABI_VERSION), s_(info), struct.pack("<I", len(seg_sizes))]
    blob += [struct.pack("<Q", sz) for sz in seg_sizes]
    for group in (inputs, outputs):
        blob.append(struct.pack("<I", len(group)))
        for name, seg, off, nbytes, dt, nd, shape, stride in group:
            blob.append(s_(name.encode()) + struct.pack("<IQQII", seg, off, nbytes, dt, nd) +
                        struct.pack("<6q", *shape) + struct.pack("<6q", *stride))
    blob.append(struct.pack("<I", len(consts)))
    for seg, off, data in consts:
        blob.append(struct.pack("<IQQ", seg, off, len(data)) + data)
    names = sorted(fn_ids, key=fn_ids.get)
    blob.append(struct.pack("<I", len(names)))
    blob += [s_(n.encode()) for n in names]
    blob.append(struct.pack("<I", len(calls)))
    for fid, sid, enc in calls:
        blob.append(struct.pack("<III", fid, sid, len(enc)))
        for kind, v in enc:
            blob.append(struct.pack("<I", kind))
            if kind == 0:
                blob.append(struct.pack("<i", v if v < 2 ** 31 else v - 2 ** 32))
            elif kind == 1:
                blob.append(struct.pack("<q", v))
            elif kind == 2:
                blob.append(struct.pack("<f", v))
            elif kind == 3:
                blob.append(struct.pack("<d", v))
            elif kind == 4:
                blob.append(struct.pack("<IQ", *v))
            else:
                raw, rel = v
                blob.append(raw + struct.pack("<I", len(rel)))
                blob += [struct.pack("<IIQ", foff, seg, off) for foff, (seg, off) in rel]
    with open(path, "wb") as f:
        for b in blob:
            f.write(b)
    launches = [n for n in names if not n.startswith("__")]
    return {"calls": sum(1 for c in calls if not names[c[0]].startswith("__")), "entry_points": launches,
            "streams": len(stream_ids), "events": len(rec._events), "segments": len(seg_sizes), "arena_bytes": sum(seg_sizes),
            "constant_bytes": sum(len(c[2]) for c in consts), "outputs": [o[0] for o in outputs]}


class PlanModel:
    """Thin ctypes client of creste_hip_model_load / _infer / _output: runs an exported plan WITHOUT the Python model
    (torch only provides the tensors that wrap the returned device pointers).  `graph=True` replays through a hipGraph
    captured by the C runtime."""

    def __init__(self, path: str, graph: bool = False):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.creste_hip_model_load(path.encode(), int(bool(graph)), C.byref(h)), "model_load")
        self._h = h
        self.info = self._lib.creste_hip_model_info(h).decode()
        self.num_streams = self._lib.creste_hip_model_num_streams(h)      # > 1: a pipelined plan (side streams inside the runtime)
        self.inputs = [self._describe(self._lib.creste_hip_model_input, i)
                       for i in range(self._lib.creste_hip_model_num_inputs(h))]
        self.outputs = [self._describe(self._lib.creste_hip_model_output, i)
                        for i in range(self._lib.creste_hip_model_num_outputs(h))]

    def _describe(self, fn, i):
        C = self._C
        name, ptr, dt, nd = C.c_char_p(), C.c_void_p(), C.c_int(), C.c_int()
        shape, stride = (C.c_int64 * 6)(), (C.c_int64 * 6)()
        from . import _lib
        _lib.check(fn(self._h, i, C.byref(name), C.byref(ptr), C.byref(dt), C.byref(nd), shape, stride), "model_describe")
        return dict(name=name.value.decode(), ptr=ptr.value, dtype=dt.value, shape=tuple(shape[:nd.value]),
                    stride=tuple(stride[:nd.value]))

    def __call__(self, inputs, stream=None) -> dict:
        """inputs: CUDA fp32 tensors with the plan's shapes -> dict name -> HOST torch tensor copied out of the arena
        (every output, synchronously: a convenience for tests -- a deployment calls `run` and reads the outputs it needs
        in place, `self.outputs[i]["ptr"]`)."""
        self.run(inputs, stream)
        torch.cuda.current_stream().synchronize()
        return {d["name"]: self._fetch(d) for d in self.outputs}

    def run(self, inputs, stream=None) -> None:
        """enqueue one inference on `stream` (default: torch's current stream); the outputs stay in the plan's arena"""
        C = self._C
        from . import _lib
        arr = (C.c_void_p * len(self.inputs))()
        keep = []
        for i, (t, d) in enumerate(zip(inputs, self.inputs)):
            if tuple(t.shape) != d["shape"]:
                raise ValueError(f"plan input {d['name']}: expected shape {d['shape']}, got {tuple(t.shape)}")
            t = t.contiguous().float()
            keep.append(t)
            arr[i] = t.data_ptr()
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _lib.check(self._lib.creste_hip_model_infer(self._h, arr, len(self.inputs), s), "model_infer")
        self._keep = keep                       # the staged inputs must outlive the asynchronous copy into the arena

    def _fetch(self, d):
        """copy an output out of the arena through the library's own d2h helper (no torch view of foreign memory)."""
        import numpy as np
        C = self._C
        from . import _lib
        np_dt = {0: np.float32, 1: np.int64, 2: np.uint8}[d["dtype"]]
        span = (sum((s - 1) * st for s, st in zip(d["shape"], d["stride"])) + 1) if all(d["shape"]) else 0
        host = np.empty(span, dtype=np_dt)
        if span:
            _lib.check(self._lib.creste_hip_memcpy_d2h(host.ctypes.data, d["ptr"], host.nbytes), "memcpy_d2h")
        view = np.lib.stride_tricks.as_strided(host, shape=d["shape"], strides=[st * host.itemsize for st in d["stride"]])
        return torch.from_numpy(np.ascontiguousarray(view))

    def close(self):
        if self._h:
            self._lib.creste_hip_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
