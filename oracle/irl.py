"""Oracle (TEST INFRASTRUCTURE) -- CPU/PyTorch restatement of the IRL half of the path:
reward FCN -> value iteration -> policy -> expected state-visitation -> MaxEnt/CF-IRL loss.

Follows /root/reference/creste/models/blocks/vin.py:21-155, /root/reference/creste/models/
lfd.py:21-392, /root/reference/creste/utils/loss_utils.py:25-91,971-1259 (the *second*
`compute_expert_visitation`, :1054-1116, which shadows the first) and
/root/reference/creste/utils/train_utils.py:511-557,670-682,765-803.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .blocks import MultiScaleFCN, _get
from .perception import TerrainNet

# (row, col) displacement of action a; the 0.8 tap of VIN.w[a] sits at 1+DYNAMICS[a].
DYNAMICS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


# ------------------------------------------------------------------------------ helpers
def trapezoid_fov_mask(H, W, top_deg=50, bottom_deg=40, near=10, far=50):
    """Boolean HxW north-facing trapezoid (train_utils.py:511-557)."""
    y, x = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    cx, cy = W / 2, H / 2
    dist = torch.sqrt((x - cx) ** 2 + (y - cy) ** 2)
    ang = torch.atan2(x - cx, cy - y) * 180 / torch.pi
    ang[ang < -180] += 360
    top = torch.full_like(dist, top_deg / 2)
    bot = torch.full_like(dist, bottom_deg / 2)
    spread = torch.where(dist <= near, top,
                         torch.where(dist >= far, bot, top + (bot - top) * ((dist - near) / (far - near))))
    return (dist >= near) & (dist <= far) & (torch.abs(ang) <= spread)


def first_pose_in_fov(S, fov):
    """S [B,T,2] int grid poses, fov [1,1,H,W] -> [B,2]: earliest pose with fov==1, else
    (H-1, W//2) (train_utils.py:765-803)."""
    B, T, _ = S.shape
    H, W = fov.shape[-2:]
    r, c = S[:, :, 0].long(), S[:, :, 1].long()
    ok = fov.to(r.device)[0, 0, r, c] == 1
    t_idx = torch.where(ok, torch.arange(T).unsqueeze(0).expand(B, -1), torch.tensor(T))
    first = t_idx.min(dim=1).values
    none = first == T
    first = torch.where(none, torch.tensor(0), first)
    ar = torch.arange(B)
    sel = torch.stack([r[ar, first], c[ar, first]], dim=1)
    sel[none] = torch.tensor([H - 1, W // 2], dtype=sel.dtype)
    return sel


def resize_and_crop(img, new_size, crop):
    """nearest resize then crop [y1:y2, x1:x2] (train_utils.py:670-682)."""
    y1, y2, x1, x2 = crop
    return F.interpolate(img, size=tuple(new_size), mode="nearest")[:, :, y1:y2, x1:x2].clone()


# ------------------------------------------------------------------------------ VIN
class VIN(nn.Module):
    def __init__(self, reward_cfg, qvalue_cfg):
        super().__init__()
        self.reward_cfg, self.qvalue_cfg = reward_cfg, qvalue_cfg
        self.discount = _get(qvalue_cfg, "discount", 0.95)
        if reward_cfg["name"] != "MultiScaleFCN":
            raise NotImplementedError(reward_cfg["name"])
        self.r = MultiScaleFCN(reward_cfg["net_kwargs"])
        assert len(qvalue_cfg["kernels"]) == 1
        A = qvalue_cfg["dims"][1]
        w = torch.zeros(A, 1, 3, 3)
        # ring order of the 8 neighbours (clockwise from north-west); the two 0.1 taps are the
        # ring neighbours of the 0.8 tap (vin.py:36-46).
        ring = [(0, 0), (0, 1), (0, 2), (1, 2), (2, 2), (2, 1), (2, 0), (1, 0)]
        for a, (dr, dc) in enumerate(DYNAMICS[:A]):
            k = ring.index((1 + dr, 1 + dc))
            w[a, 0, 1 + dr, 1 + dc] = 0.8
            for nb in (ring[(k - 1) % 8], ring[(k + 1) % 8]):
                w[a, 0, nb[0], nb[1]] = 0.1
        self.register_buffer("w", w)

    def value_iteration(self, r, threshold=0.001, discount=0.95):
        """Jacobi sweeps with hard-max backup until the batch-global max change <= threshold,
        then one more q evaluation and a softmax policy (vin.py:48-80).  Returns the sweep
        count as a 4th value (oracle-only extra)."""
        v = torch.zeros_like(r)
        q_of = lambda vv: F.conv2d(r + vv * discount, self.w, stride=1, padding=1)
        delta, sweeps = float("inf"), 0
        while delta > threshold:
            new_v = q_of(v).max(dim=1, keepdim=True)[0]
            delta = (new_v - v).abs().max().item()
            v = new_v
            sweeps += 1
        q = q_of(v)
        e = torch.exp(q - q.max(dim=1, keepdim=True)[0])
        return v, e / e.sum(dim=1, keepdim=True), q, sweeps

    def forward(self, feat_map, S, solve_mdp=False):
        rc = self.reward_cfg
        view = torch.cat([feat_map[k] for k in rc["input_keys"]], dim=1)
        Ho, Wo = view.shape[-2:]
        view = F.max_pool2d(view, kernel_size=rc["ds"], stride=rc["ds"])
        B, C, H, W = view.shape
        view = view[:, :, :H // 2, :].detach()
        view.requires_grad_(True)
        r = self.r(view)
        with torch.no_grad():
            full = torch.zeros(B, 1, Ho, Wo)
            full[:, :, :Ho // 2, :] = F.interpolate(r, size=(Ho // 2, Wo), mode="bilinear",
                                                    align_corners=False)
        name = rc["output_prefix"][0]
        out = {name: r, f"{name}_full": full, "input_view": view}
        if not solve_mdp:
            return out
        assert S is not None
        with torch.no_grad():
            v, pol, q, sweeps = self.value_iteration(r, threshold=0.001, discount=self.discount)
        out.update({"policy": pol, "q_estimate": q, "value_estimate": v, "_vi_sweeps": sweeps})
        return out


# ------------------------------------------------------------------------------ MaxEntIRL
class MaxEntIRL(nn.Module):
    """frozen TerrainNet + VIN + policy-propagation SVF (lfd.py:21-392; 'pp' policy only)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        th = cfg["traversability_head"]
        self.head_cfg = th
        self.policy_cfg = _get(cfg, "policy_kwargs", {})
        self.map_size = list(_get(cfg, "map_size", [64, 128]))
        self.policy_method = _get(cfg, "policy_method", "fc")
        if self.policy_method not in ("pp", "fc"):
            raise ValueError(f"Policy method {self.policy_method} not found.")
        if self.policy_method == "fc":            # lfd.py:96-101: a linear read-out of the Q vector at the expert's cell
            self.fc = nn.Linear(th["net_kwargs"]["qvalue_cfg"]["dims"][-1], 8, bias=False)
        self.action_horizon = cfg["action_horizon"]
        self.solve_mdp = _get(cfg, "solve_mdp", False)
        self.zero_terminal_state = _get(cfg, "zero_terminal_state", False)
        self.register_buffer("dynamics", torch.tensor(DYNAMICS, dtype=torch.long))
        H, W = self.map_size
        fov = trapezoid_fov_mask(H * 2, W, 70, 70, 0, 100).view(1, 1, H * 2, W)
        self.fov_mask = fov[:, :, :H, :W]
        tp = torch.zeros(8, 1, 3, 3)
        for a, (dr, dc) in enumerate(DYNAMICS):       # inverse move: tap at 1 - d_a (lfd.py:59-70)
            tp[a, 0, 1 - dr, 1 - dc] = 1.0
        self.register_buffer("transition_probs", tp)
        self.backbone = TerrainNet(cfg["vision_backbone"])
        self.traversability_head = VIN(**th["net_kwargs"])

    def expected_svf(self, policy, expert):
        """policy [B,8,H,W], expert [B,T,3,3] -> exp_svf [B,H,W], state_preds [B,T,2],
        state_preds_grid [B,H,W] (lfd.py:156-277)."""
        B, A, H, W = policy.shape
        ds = self.head_cfg["net_kwargs"]["reward_cfg"]["ds"]
        T = self.action_horizon
        S = (expert[:, :, :2, 2] // ds).long()
        S[:, :, 0].clamp_(0, H - 1)
        S[:, :, 1].clamp_(0, W - 1)
        S0 = first_pose_in_fov(S, self.fov_mask)
        s0 = S0[:, 0] * W + S0[:, 1]
        s1 = S[:, -1, 0] * W + S[:, -1, 1]
        ar = torch.arange(B)
        mu = torch.zeros(B, T, H * W)
        mu[ar, 0, s0] = 1.0
        pol = policy
        if self.policy_cfg["method"] == "sharpen":
            lg = (pol - pol.max(dim=1, keepdim=True)[0]) / self.policy_cfg["temperature"]
            pol = F.softmax(lg, dim=1)
        elif self.policy_cfg["method"] != "none":
            raise ValueError(self.policy_cfg["method"])
        for t in range(1, T):
            if self.zero_terminal_state:
                mu[ar, t - 1, s1] = 0.0
            prev = mu[:, t - 1].clone().view(B, 1, H, W)
            nxt = F.conv2d(pol * prev, self.transition_probs, stride=1, padding=1, groups=A)
            mu[:, t] = nxt.sum(dim=1, keepdim=True).view(B, H * W)
        svf = mu.sum(dim=1).view(B, H, W)
        grid = torch.zeros(B, H, W)
        states = torch.zeros(B, T, 2, dtype=torch.long)
        states[:, 0] = torch.stack([s0 // W, s0 % W], dim=1)
        grid[ar, states[:, 0, 0], states[:, 0, 1]] += 1
        best = policy.view(B, A, H * W).argmax(dim=1)
        st = s0
        for t in range(1, T):
            a = best[ar, st]
            c = torch.stack([st // W, st % W], dim=1) + self.dynamics[a]
            c[:, 0].clamp_(0, H - 1)
            c[:, 1].clamp_(0, W - 1)
            states[:, t] = c
            grid[ar, c[:, 0], c[:, 1]] += 1
            st = c[:, 0] * W + c[:, 1]
        assert torch.all(svf >= 0)
        return {"exp_svf": svf, "state_preds_grid": grid, "state_preds": states}

    def iterative_policy_rollout(self, q, expert, T):
        """policy_method 'fc' (lfd.py:279-312): at step t the Q vector at the expert's cell of step t-1 goes through `fc` and a
        softmax; the greedy action moves the predicted state (clamped to the grid).  q [B,l_q,H,W], expert [B,>=T-1,2] grid
        cells -> policy_fc [B,T,8] (row 0 zero), state_preds [B,T,2]."""
        B, lq, H, W = q.shape
        states = torch.zeros(B, T, 2, dtype=torch.long)
        states[:, 0] = expert[:, 0, :2].long()
        probs = torch.zeros(B, T, 8)
        for t in range(1, T):
            for b in range(B):
                r, c = int(expert[b, t - 1, 0]), int(expert[b, t - 1, 1])
                p = F.softmax(self.fc(q[b, :, r, c].view(1, lq)), dim=1)[0]
                a = int(p.argmax())
                nxt = states[b, t - 1] + self.dynamics[a]
                states[b, t, 0] = int(nxt[0].clamp(0, H - 1))
                states[b, t, 1] = int(nxt[1].clamp(0, W - 1))
                probs[b, t] = p
        return {"policy_fc": probs, "state_preds": states}

    def forward(self, inputs):
        image, p2p = inputs[0], inputs[1]
        out = self.backbone((image, p2p))
        if not self.solve_mdp:
            out.update(self.traversability_head(out, None, False))
            return out
        assert len(inputs) > 2, "Goal location required for MDP solver"
        expert = inputs[2]
        W = out["bev_features"].shape[-1]
        map_ds = W // self.map_size[1]
        S = expert[:, :, :2, 2].long() // map_ds
        S[:, :, 0].clamp_(0, self.map_size[0] - 1)
        S[:, :, 1].clamp_(0, self.map_size[1] - 1)
        out.update(self.traversability_head(out, S, solve_mdp=True))
        with torch.no_grad():
            if self.policy_method == "fc":        # lfd.py:357-360
                out.update(self.iterative_policy_rollout(out["q_estimate"], S, self.action_horizon))
            else:
                out.update(self.expected_svf(out["policy"], expert))
        return out


# ------------------------------------------------------------------------------ loss
def rasterise_expert(gt, map_ds, map_sz):
    """Expert polyline -> binary visitation grid (loss_utils.py:1054-1116, 2nd definition).
    gt [B,T,3,3] poses or [B,T,2] xy.  Returns (points [B,N,2], counts [B,H,W])."""
    xy = gt if gt.ndim == 3 else gt[:, :, :2, 2]
    B = xy.shape[0]
    xy = xy / map_ds
    H, W = map_sz
    a, b = xy[:, :-1], xy[:, 1:]
    steps = torch.ceil(torch.norm(b - a, dim=-1)).long().max().item()
    t = torch.linspace(0, 1, steps, device=xy.device).view(1, 1, -1, 1)
    pts = (a.unsqueeze(2) + t * (b - a).unsqueeze(2)).view(B, -1, 2)
    pts = torch.cat([pts, xy[:, -1:]], dim=1)
    lin = pts[:, :, 0].clamp(0, H - 1).long() * W + pts[:, :, 1].clamp(0, W - 1).long()
    cnt = torch.zeros(B, H * W, dtype=torch.float32, device=xy.device)
    cnt.scatter_add_(1, lin, torch.ones_like(lin, dtype=torch.float32))
    cnt = cnt.view(B, H, W)
    cnt[cnt > 1] = 1
    return pts, cnt


class MaxEntIRLLoss(nn.Module):
    """loss_utils.py:971-1259 (+ the weight wrapping of Loss.forward, :34-53)."""

    def __init__(self, cfg):
        super().__init__()
        self.name = cfg["name"] + _get(cfg, "tag", "")
        self.weight = _get(cfg, "weight", 1.0)
        self.task = _get(cfg, "task", None)
        self.pred_key, self.lab_key, self.fov_key = cfg["pred_key"], cfg["lab_key"], cfg["fov_key"]
        self.map_ds = _get(cfg, "map_ds", 2)
        self.map_sz = _get(cfg, "map_sz", [64, 128])
        self.maxent_weight = _get(cfg, "maxent_weight", 1.0)
        self.reward_weight = _get(cfg, "reward_weight", 0.1)
        self.use_fov_mask = _get(cfg, "use_fov_mask", False)
        self.alpha = _get(cfg, "alpha", None)
        self.cf_key = _get(cfg, "cf_key", None)

    def loss(self, td):
        exp_svf, gt, fov = td[self.pred_key], td[self.lab_key], td[self.fov_key]
        r = td["outputs/traversability_preds"].squeeze(1)
        feats = td["outputs/input_view"]
        _, Ho, Wo = fov.shape
        _, H, W = exp_svf.shape
        fov = resize_and_crop(fov.unsqueeze(1).byte(), (Ho // 2, Wo // 2), (0, H, 0, W))
        fov = fov.squeeze(1).bool()
        _, svf = rasterise_expert(gt, self.map_ds, self.map_sz)
        if self.use_fov_mask:
            svf = svf * fov.float()
            exp_svf = exp_svf * fov.float()
        svf = svf / (svf.sum(dim=(1, 2), keepdim=True) + 1e-5)
        exp_svf = exp_svf / (exp_svf.sum(dim=(1, 2), keepdim=True) + 1e-5)
        cf_total = torch.zeros_like(svf)
        exp_total = exp_svf.clone()
        if self.cf_key is not None and self.alpha is not None:
            for i, cf in enumerate(td[self.cf_key]):
                if cf is None:
                    continue
                bad = cf["trajectories"][cf["rank"] > 0]
                if bad.shape[0] == 0:
                    continue
                _, c = rasterise_expert(torch.as_tensor(bad).to(svf.device), self.map_ds, self.map_sz)
                c = c.sum(dim=0)
                c = c / (c.sum(dim=(0, 1), keepdim=True) + 1e-5)
                exp_svf[i] = self.alpha * c + (1 - self.alpha) * exp_svf[i]
                cf_total[i] = c
        assert torch.all(exp_svf >= 0) and torch.all(svf >= 0)
        if self.use_fov_mask:
            m = torch.ones_like(r)
            m[~fov] = 0
            r = r * m
        e_exp = (exp_svf * r).sum(dim=(1, 2)).mean()
        e_svf = (svf * r).sum(dim=(1, 2)).mean()
        pen = torch.tensor(0.0, device=svf.device)
        if r.requires_grad and self.reward_weight > 0:
            g = torch.autograd.grad(r.sum(), feats, create_graph=True, retain_graph=True)[0]
            pen = ((g.norm(2, dim=1) - 1) ** 2).mean()
        loss = self.maxent_weight * (e_exp - e_svf) + self.reward_weight * pen
        with torch.no_grad():
            cf_r = (cf_total * r).sum(dim=(1, 2))
            op_r = (exp_total * r).sum(dim=(1, 2))
            ok = cf_r != 0
        meta = {"reward_penalty": self.reward_weight * pen, "mean_expected_svf_rewards": e_exp,
                "mean_svf_rewards": e_svf, "sum_cf_rewards": cf_r[ok].sum(),
                "sum_opt_rewards": op_r[ok].sum()}
        return {"maxentirl_loss": loss}, meta

    def forward(self, td):
        ld, md = self.loss(td)
        return {k: (self.weight * 1.0, v) for k, v in ld.items()}, md


class LossManager(nn.Module):
    """name-prefixed dict of (weight, value) per loss (loss_utils.py:63-91)."""

    def __init__(self, cfg):
        super().__init__()
        table = {"MaxEntIRLLoss": MaxEntIRLLoss}
        self.losses = nn.ModuleList([table[lc["name"]](lc) for lc in cfg["loss"]])

    def forward(self, td):
        loss_dict, meta = {}, {}
        for l in self.losses:
            if l.task is None or l.task == td["task"]:
                ld, md = l(td)
                meta.update({f"{l.name}/{k}": v for k, v in md.items()})
                loss_dict.update({f"{l.name}/{k}": v for k, v in ld.items()})
        return loss_dict, meta
