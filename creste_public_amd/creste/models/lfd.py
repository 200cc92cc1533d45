"""MaxEnt / counterfactual IRL model: frozen TerrainNet + costmap head + MDP solve + expected SVF.

Mirrors /root/reference/creste/models/lfd.py (MaxEntIRL :21-392): constructor config, buffers
`dynamics` / `transition_probs`, `load_weights`, `expected_state_visitation_frequency`, forward input
tuple `(image, p2p[, expert])` and output keys.  Value iteration and the 49-step policy propagation +
greedy rollout are single HIP launches sequences with no per-step host sync (the reference does ~700
`.item()` syncs per step).  The frozen backbone always runs in eval mode on the HIP path (the reference
leaves it in train() mode unless `load_weights` ran -- an intentional, documented deviation).
"""
import os

import torch
from torch import nn

from ... import hipnn as _hipnn
from ... import ops
from ...hipnn import require_hip
from ..utils import train_utils as tu
from .blocks.conv import _cfg_get
from .blocks.vin import VIN
from .terrainnet import TerrainNet

_DYNAMICS = [[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 1], [1, -1], [1, 0], [1, 1]]


class MaxEntIRL(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        # checkpoint loaders that write through `.data` do not bump tensor versions: drop derived-weight caches
        # ... and a frozen half prefetched with the OLD weights must not be picked up by the next forward
        def _after_load(m, keys):
            _hipnn.invalidate_caches()
            m._prefetched = None
        self.register_load_state_dict_post_hook(_after_load)
        self.model_cfg = model_cfg
        self.backbone_cfg = model_cfg["vision_backbone"]
        self.traversability_head_cfg = model_cfg["traversability_head"]
        self.policy_cfg = _cfg_get(model_cfg, "policy_kwargs", {})
        self.ckpt_path = _cfg_get(model_cfg, "ckpt_path", "")
        self.weights_path = _cfg_get(model_cfg, "weights_path", "")
        self.map_size = list(_cfg_get(model_cfg, "map_size", [64, 128]))
        self.policy_method = _cfg_get(model_cfg, "policy_method", "fc")
        self.goal_cfg = _cfg_get(model_cfg, "goal_kwargs", {})
        self.action_horizon = _cfg_get(model_cfg, "action_horizon")
        self.solve_mdp = _cfg_get(model_cfg, "solve_mdp", False)
        self.zero_terminal_state = _cfg_get(model_cfg, "zero_terminal_state", False)
        if self.policy_method not in ("pp", "fc"):
            raise ValueError(f"Policy method {self.policy_method} not found.")
        self.register_buffer("dynamics", torch.tensor(_DYNAMICS, dtype=torch.long))
        H, W = self.map_size
        fov = tu.create_trapezoidal_fov_mask(H * 2, W, 70, 70, 0, 100).view(1, 1, H * 2, W)
        self.fov_mask = fov[:, :, :H, :W]
        tp = torch.zeros(8, 1, 3, 3)
        for a, (dr, dc) in enumerate(_DYNAMICS):          # the previous cell sits opposite the move
            tp[a, 0, 1 - dr, 1 - dc] = 1.0
        self.register_buffer("transition_probs", tp)

        if "TerrainNet" not in self.backbone_cfg["project_name"]:
            raise ValueError(f"Model {self.backbone_cfg['project_name']} not found.")
        if self.backbone_cfg["load_setting"] not in ("strict_freeze", "strict_unfreezesplat"):
            self.backbone_cfg["load_setting"] = "strict_freeze"       # lfd.py:80-83
        self.backbone = TerrainNet(self.backbone_cfg)
        wp = self.backbone_cfg["weights_path"]
        if wp and os.path.exists(wp):
            self.backbone.load_weights(wp)
        if self.traversability_head_cfg["value_iterator"] != "VIN":
            raise NotImplementedError(self.traversability_head_cfg["value_iterator"])
        nk = self.traversability_head_cfg["net_kwargs"]
        self.traversability_head = VIN(nk["reward_cfg"], nk["qvalue_cfg"])
        if self.policy_method == "fc":     # reference lfd.py:96-101 (the default when a config names no method)
            self.fc = nn.Linear(nk["qvalue_cfg"]["dims"][-1], 8, bias=False)
            self.sm = nn.Softmax(dim=1)
        self.freeze_backbone = _cfg_get(model_cfg, "freeze_backbone", True)
        self.freeze_head = _cfg_get(model_cfg, "freeze_head", False)
        self.load_strict = _cfg_get(model_cfg, "load_strict", True)
        for p in self.backbone.parameters():              # the perception backbone is frozen for IRL
            p.requires_grad = False
        self.backbone.eval()
        self._fov_u8 = None
        self._side_stream, self._prefetched = None, None
        self._parts_warm = False
        if self.weights_path and os.path.isfile(self.weights_path) and not os.path.isfile(self.ckpt_path or ""):
            self.load_weights(self.weights_path)

    def train(self, mode: bool = True):
        super().train(mode)
        self.backbone.eval()          # frozen + folded BN on the HIP path (see module docstring)
        if self.freeze_head and getattr(self, "_head_frozen", False):
            self.traversability_head.eval()
        return self

    def _state_to_coord(self, state, vectorized=False):
        if vectorized:
            return torch.stack([state // self.map_size[1], state % self.map_size[1]], dim=1)
        return torch.tensor([state // self.map_size[1], state % self.map_size[1]], dtype=torch.long)

    def _coord_to_state(self, coord, vectorized=False):
        return coord[:, 0] * self.map_size[1] + coord[:, 1] if vectorized else coord[0] * self.map_size[1] + coord[1]

    def load_weights(self, weights_path):
        sd = torch.load(weights_path, weights_only=False)["state_dict"]
        sd = {k.replace("model.", "", 1): v for k, v in sd.items() if k.startswith("model.")}
        self.load_state_dict(sd, strict=self.load_strict)
        if self.freeze_backbone:
            self.backbone.eval()
            for p in self.backbone.parameters():
                p.requires_grad = False
        if self.freeze_head:
            self._head_frozen = True
            self.traversability_head.eval()
            for p in self.traversability_head.parameters():
                p.requires_grad = False

    def expected_state_visitation_frequency(self, policy, expert):
        """policy [B,8,H,W], expert [B,T,3,3] -> exp_svf, state_preds_grid, state_preds (lfd.py:156-277)."""
        require_hip(policy, "expected_state_visitation_frequency")
        B, A, H, W = policy.shape
        ds = self.traversability_head_cfg["net_kwargs"]["reward_cfg"]["ds"]
        method = self.policy_cfg["method"]
        if method not in ("sharpen", "none"):
            raise ValueError(f"Policy method {method} not found.")
        if self._fov_u8 is None or self._fov_u8.device != policy.device:
            self._fov_u8 = self.fov_mask[0, 0].to(torch.uint8).to(policy.device).contiguous()
        assert tuple(self._fov_u8.shape) == (H, W), "policy grid and fov mask disagree"
        xy = expert[:, :, :2, 2].float().contiguous()
        svf, states, grid = ops.expected_svf(
            policy.contiguous().float(), xy, self._fov_u8, self.action_horizon, float(ds),
            float(_cfg_get(self.policy_cfg, "temperature", 1.0)), method == "sharpen",
            bool(self.zero_terminal_state))
        return {"exp_svf": svf, "state_preds_grid": grid, "state_preds": states}

    def iterative_policy_rollout(self, q, expert, T):
        """policy_method 'fc' (reference lfd.py:279-312): the Q vectors at the expert's cells of steps 0..T-2 go through `fc`
        + softmax in ONE batched product; only the greedy state walk is sequential (it clamps at the grid border).
        q [B,l_q,H,W], expert [B,>=T-1,2] grid cells -> policy_fc [B,T,8] (row 0 zero), state_preds [B,T,2] (long)."""
        B, lq, H, W = q.shape
        cells = expert[:, :T - 1, :2].long()
        bi = torch.arange(B, device=q.device).view(B, 1).expand(B, T - 1)
        qv = q[bi, :, cells[..., 0], cells[..., 1]]                        # [B, T-1, l_q]
        probs = self.sm(self.fc(qv.reshape(B * (T - 1), lq))).view(B, T - 1, 8)
        policy = torch.zeros(B, T, 8, dtype=torch.float32, device=q.device)
        policy[:, 1:] = probs
        with torch.no_grad():
            moves = self.dynamics[probs.argmax(dim=2)]                     # [B, T-1, 2]
            lo = torch.zeros(2, dtype=torch.long, device=q.device)
            hi = torch.tensor([H - 1, W - 1], dtype=torch.long, device=q.device)
            states = torch.zeros(B, T, 2, dtype=torch.long, device=q.device)
            states[:, 0] = expert[:, 0, :2].long()
            for t in range(1, T):
                states[:, t] = torch.minimum(torch.maximum(states[:, t - 1] + moves[:, t - 1], lo), hi)
        return {"policy_fc": policy, "state_preds": states}

    # ---- frozen half of the forward (perception backbone -> BEV predictions -> pooled / cropped reward input): depends on
    # no trainable parameter, so in IRL training the NEXT batch's frozen half can run on a second stream while this
    # batch's reward network / value iteration / SVF / loss / backward / Adam run (reference train_traversability.py:66-105
    # runs them back to back, 24 of the 46 ms of a step at BASELINE configs[2])
    @staticmethod
    def _input_key(inputs):
        """(tensor, version) of image and p2p.  The TENSORS are held (and compared by identity): a key of addresses alone
        would match a new batch that the allocator placed in a dropped batch's block."""
        return tuple((t, t._version) for t in inputs[:2])

    @staticmethod
    def _same_inputs(key, inputs):
        return len(key) == 2 and all(k[0] is t and k[1] == t._version for k, t in zip(key, inputs[:2]))

    def _frozen_half(self, image, p2p):
        head = self.traversability_head
        keys = list(head.reward_cfg["input_keys"])
        want = [f"{p}_preds" for p in self.backbone.bevclassifier.output_prefix]
        r = self.backbone.forward_act(image, p2p)
        outputs = self.backbone.pack_outputs(r, image.shape[0])
        if keys == want:
            # the shipped configs: the heads' 1x1 projections already wrote one channel-concatenated tensor (no copy)
            src = r["preds_buf"]
        else:
            # any other selection of backbone outputs (reference vin.py:104-107: torch.cat over feat_map[key], dim=1)
            missing = [k for k in keys if k not in outputs]
            if missing:
                raise KeyError(f"reward input_keys {missing} are not outputs of the backbone ({sorted(outputs)})")
            cat = torch.cat([outputs[k].float() for k in keys], dim=1).contiguous()
            if cat.shape[1] % 4:
                raise NotImplementedError(f"reward input_keys select {cat.shape[1]} channels: the HIP path reads channel "
                                          "quads (NHWC rows of 16 bytes)")
            src = ops.nchw_to_nhwc(cat)
        r["iv_hw"] = (src.H, src.W)
        with ops.shared_rows():
            view = head.input_view_act(src)
        return r, outputs, view

    # ---- pipelined inference.  The frozen half of an eval forward is 46 % matrix-bound GEMM kernels (persistent, one
    # workgroup per CU) and 40 % bandwidth-bound transform / elementwise kernels that alternate on ONE stream.  Run as
    # `inference_parts` forwards of B / parts frames on as many streams (a side stream PROBED to run beside the caller's:
    # ops.concurrent_stream), issued one after the other by this thread, part
    # k + 1 trails part k by the host's issue time and its bandwidth-bound kernels fill in beside / between the other
    # part's GEMMs (measured: batch 16, 39.8 -> 37.9 ms with two parts; four parts lose, profiles/r04_pipeline_notes.md).
    # Each part is exactly `_frozen_half` of its frames (bit-identical to calling the model on those frames); the parts
    # write their rows of shared whole-batch output buffers (ops.PartContext), nothing is concatenated.
    inference_parts = 2            # 0 / 1: off
    inference_part_rows = 6        # smallest part worth a stream of its own (measured, 1216x608: batch 8 as 2 x 4 gains nothing
                                   # and a lone step gets 1.4 ms slower; 12 as 2 x 6: -3 %, 16: -4 %, 32: -4.4 %)

    def _parts_for(self, B, device=None):
        return ops.parts_for(B, device or torch.device("cuda", torch.cuda.current_device()), self.inference_parts,
                             self.inference_part_rows)

    def _frozen_parts(self, image, p2p, parts):
        ctx, res = ops.forward_in_parts(self._frozen_half, (image, p2p), parts, owner=self)
        outputs = ops.whole_outputs(ctx, [r[1] for r in res])
        view0 = res[0][2]
        view = ctx.whole_act(view0)
        if view is None:                               # (not in the shipped configs: the input view is a shared buffer)
            bufs = [r[2].buf for r in res]
            for t in bufs[1:]:
                t.record_stream(torch.cuda.current_stream(image.device))
            view = ops.Act(torch.cat(bufs), view0.C, view0.co)
        return {"iv_hw": res[0][0]["iv_hw"], "parts": [x[0] for x in res]}, outputs, view

    def prefetch_backbone(self, inputs):
        """Enqueue the frozen half of `forward(inputs)` on a side stream; the next `forward` with the SAME input tensors
        (same storage, same version) picks the result up instead of recomputing it.  Returns immediately.  IRLTrainer calls it
        right after it has picked up the CURRENT batch's prefetched half, ahead of the reward forward and the MDP solve (round 6:
        the barrier-free solver only waits for its halo neighbours, with bounded polls, and was never seen to abort beside the
        backbone's kernels -- 0 of 276 solves; a solve that reports INT32_MIN is redone in the launch-per-chunk form)."""
        image, p2p = inputs[0], inputs[1]
        require_hip(image, "MaxEntIRL")
        main = torch.cuda.current_stream(image.device)
        if self._side_stream is None or self._side_stream.device != image.device:
            # LOWEST priority: the frozen half is throughput work (full-chip MFMA kernels); the trainable half on the
            # caller's stream is a latency chain of ~600 small kernels that must not queue behind it
            # (a stream probed to run beside the caller's -- ops.concurrent_stream; any stream keeps the results right)
            self._side_stream = ops.concurrent_stream(image.device, "prefetch") or torch.cuda.Stream(device=image.device)
        side = self._side_stream
        side.wait_stream(main)                       # the inputs are ready once the main stream gets here
        with torch.cuda.stream(side), torch.no_grad():
            r, outputs, view = self._frozen_half(image, p2p)
            done = torch.cuda.Event()
            done.record(side)
        for t in (image, p2p):                       # keep the inputs' memory until the side stream has read them
            t.record_stream(side)
        self._prefetched = (self._input_key(inputs), r, outputs, view, done)

    def _take_prefetched(self, inputs):
        pf, self._prefetched = self._prefetched, None
        if pf is None or not self._same_inputs(pf[0], inputs):
            return None
        _, r, outputs, view, done = pf
        main = torch.cuda.current_stream(inputs[0].device)
        main.wait_event(done)
        # the tensors were allocated on the side stream's pool: tell the allocator the main stream uses them too
        ops.mark_stream((r, outputs, view), main)
        return r, outputs, view

    def forward(self, inputs):
        image, p2p = inputs[0], inputs[1]
        require_hip(image, "MaxEntIRL")
        B = image.shape[0]
        got = self._take_prefetched(inputs)
        if got is None:
            parts = self._parts_for(B, image.device)
            got = self._frozen_parts(image, p2p, parts) if parts > 1 else self._frozen_half(image, p2p)
        return self._forward_trainable(inputs, got)

    def _forward_trainable(self, inputs, frozen):
        """the rest of `forward` given the frozen half's results (r, outputs, view)"""
        r, outputs, view = frozen if frozen is not None else self._frozen_half(inputs[0], inputs[1])
        outputs = dict(outputs)
        head = self.traversability_head
        Ho, Wo = r["iv_hw"]
        if not self.solve_mdp:
            outputs.update(head.forward_from_view(view, Ho, Wo, None, False))
            return outputs
        assert len(inputs) > 2, "Goal location required for MDP solver"
        expert = inputs[2]
        map_ds = Wo // self.map_size[1]
        S = expert[:, :, :2, 2].long() // map_ds
        S[:, :, 0] = S[:, :, 0].clamp(0, self.map_size[0] - 1)
        S[:, :, 1] = S[:, :, 1].clamp(0, self.map_size[1] - 1)
        if "method" in self.goal_cfg:
            raise NotImplementedError("goal maps (goal_kwargs) are not used by the shipped configs")
        outputs.update(head.forward_from_view(view, Ho, Wo, S, solve_mdp=True))
        with torch.no_grad():
            if self.policy_method == "fc":         # reference lfd.py:357-360
                outputs.update(self.iterative_policy_rollout(outputs["q_estimate"], S, self.action_horizon))
            else:
                outputs.update(self.expected_state_visitation_frequency(outputs["policy"], expert))
        return outputs
