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
