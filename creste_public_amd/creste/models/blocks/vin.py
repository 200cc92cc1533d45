"""Costmap (reward) head + value iteration on HIP kernels.

Mirrors /root/reference/creste/models/blocks/vin.py (VIN :21-155): same constructor
(reward_cfg, qvalue_cfg), buffer `w` [8,1,3,3], sub-module `r` (MultiScaleFCN) and output keys
(`traversability_preds`, `traversability_preds_full`, `input_view`, `policy`, `q_estimate`,
`value_estimate`).  max-pool + front-half crop, the reward convs (eval), the full-size bilinear
resize and the value-iteration sweeps are HIP kernels; value iteration runs without a host
round-trip per sweep (the reference syncs with `.item()` every sweep, vin.py:73).
"""
import torch
from torch import nn

from .... import ops
from ....hipnn import Act, require_hip
from .conv import MultiScaleFCN, _cfg_get


class VIN(nn.Module):
    def __init__(self, reward_cfg, qvalue_cfg):
        super().__init__()
        from ....hipnn import hook_invalidate
        hook_invalidate(self)      # load_state_dict drops the packed / BN-folded weight caches (hipnn.invalidate_caches)
        self.reward_cfg, self.qvalue_cfg = reward_cfg, qvalue_cfg
        self.discount = _cfg_get(qvalue_cfg, "discount", 0.95)
        if reward_cfg["name"] != "MultiScaleFCN":
            raise NotImplementedError(f"reward network {reward_cfg['name']}")
        self.r = MultiScaleFCN(reward_cfg["net_kwargs"])
        assert len(qvalue_cfg["kernels"]) == 1, "Only single layer Q value network supported"
        A = qvalue_cfg["dims"][1]
        w = torch.zeros(A, 1, 3, 3)
        # 0.8 on the move's own cell, 0.1 on its two neighbours along the 3x3 ring (vin.py:36-46)
        ring = [(0, 0), (0, 1), (0, 2), (1, 2), (2, 2), (2, 1), (2, 0), (1, 0)]
        moves = [(0, 0), (0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1), (2, 2)]
        for a, c in enumerate(moves[:A]):
            k = ring.index(c)
            w[a, 0, c[0], c[1]] = 0.8
            for nb in (ring[k - 1], ring[(k + 1) % 8]):
                w[a, 0, nb[0], nb[1]] = 0.1
        self.register_buffer("w", w)

    def value_iteration_manual(self, r, goal, threshold=0.001, discount=0.95):
        """r [B,1,H,W] -> (v [B,1,H,W], policy [B,8,H,W], q [B,8,H,W]) (vin.py:48-80)."""
        require_hip(r, "value_iteration")
        if self.w.shape[0] != 8:
            raise NotImplementedError("HIP value iteration is built for the 8-connected action set")
        v, q, pi, sweeps = ops.value_iteration(r.detach()[:, 0].contiguous().float(), discount, threshold)
        self._last_sweeps = sweeps
        return v.unsqueeze(1), pi, q

    @property
    def last_sweeps(self):
        """Sweep count of the last solve (device int32 tensor).  The solve is asynchronous, so a failure cannot raise where
        it happens: it is reported in the sign (ops.value_iteration).  It is checked without anybody asking: at the head
        of the next solve and by IRLTrainer before every optimiser step (ops.vi_check: a 4-byte copy behind each solve);
        reading this property checks NOW (one host synchronisation): a negative count raises, as the reference's loop
        would have spun / the launch-per-chunk form returned CRESTE_ERR_NOCONV."""
        s = getattr(self, "_last_sweeps", None)
        if s is not None:
            ops.check_vi_sweeps(s)
        return s

    def input_view_act(self, preds: Act) -> Act:
        """cat(input_keys) -> max_pool2d(ds) -> front half rows (vin.py:104-109), one kernel pass."""
        ds = int(self.reward_cfg["ds"])
        if ds not in (1, 2, 4):
            raise NotImplementedError("HIP max-pool is built for reward_cfg.ds in {1, 2, 4}")
        return ops.maxpool2(preds, Ho=(preds.H // ds) // 2, Wo=preds.W // ds, ds=ds)

    def forward_from_view(self, view: Act, Ho, Wo, S, solve_mdp=False):
        name = self.reward_cfg["output_prefix"][0]
        B = view.N
        if self.r.training:
            iv = view.nchw().detach()
            iv.requires_grad_(True)
            r = self.r(iv)                                   # autograd path (IRL training)
        else:
            iv = view.nchw()
            r = self.r.forward_act(view).nchw()              # [B,1,h,w] (C == 1: dense)
        full = ops.fill_(torch.empty((B, Ho, Wo), dtype=torch.float32, device=view.buf.device), 0.0)
        rr = r.detach()[:, 0].contiguous()
        ops.resize_plane(rr, Ho // 2, Wo, Ho, rr.shape[1] / (Ho // 2), rr.shape[2] / Wo, full)
        outputs = {name: r, f"{name}_full": full.unsqueeze(1), "input_view": iv}
        if not solve_mdp:
            return outputs
        assert S is not None, "No expert demonstrations given but solve mdp is True"
        with torch.no_grad():
            v, policy, q = self.value_iteration_manual(r, S[:, -1, :], threshold=0.001,
                                                       discount=self.discount)
        outputs.update({"policy": policy, "q_estimate": q, "value_estimate": v})
        return outputs

    def forward(self, feat_map, S, solve_mdp=False):
        keys = self.reward_cfg["input_keys"]
        first = feat_map[keys[0]]
        require_hip(first, "VIN")
        B, _, Ho, Wo = first.shape
        C = sum(feat_map[k].shape[1] for k in keys)
        cat = Act.empty(B, Ho, Wo, C, first.device)
        co = 0
        for k in keys:                                     # NCHW dict entries -> one NHWC concat buffer
            t = feat_map[k]
            ops.nchw_to_nhwc(t.contiguous().float(), out=cat.slice(co, t.shape[1]))
            co += t.shape[1]
        return self.forward_from_view(self.input_view_act(cat), Ho, Wo, S, solve_mdp)
