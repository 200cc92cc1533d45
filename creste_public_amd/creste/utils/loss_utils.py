"""Loss layer of the IRL step -- mirrors /root/reference/creste/utils/loss_utils.py: Loss (:25-60),
LossManager (:63-91) and MaxEntIRLLoss (:971-1259, with the second `compute_expert_visitation`,
:1054-1116, the one Python actually binds).  Device-agnostic torch autograd code: the objective is a
handful of reductions over [B,64,128] maps plus a double backward through the reward network, which
stays on stock autograd ops.  Inputs `exp_svf`, `traversability_preds`, `input_view` come
from the HIP path."""
import numpy as np
import torch
import torch.nn as nn

from . import train_utils as tu


class Loss(nn.Module):
    def __init__(self, name, config):
        super().__init__()
        self.config = config
        self._name = name + config.get("tag", "")
        self.weight = config.get("weight", 1.0)
        self.task = config.get("task", None)

    @property
    def name(self):
        return self._name

    def loss(self, tensor_dict):
        raise Exception("Not Implemented!")

    def forward(self, tensor_dict):
        loss_dict, meta = self.loss(tensor_dict)
        out, w = {}, 1.0
        logvar_key = self.config.get("logvar_key", None)
        if logvar_key is not None:                       # learned uncertainty weighting
            log_var = tensor_dict[logvar_key]
            w = 1.0 / (2.0 * torch.exp(log_var))
            out["log_std"] = (1.0, 0.5 * log_var)
        out.update({k: (self.weight * w, v) for k, v in loss_dict.items()})
        return out, meta


class MaxEntIRLLoss(Loss):
    def __init__(self, config):
        super().__init__(config["name"], config)
        self.pred_key, self.lab_key, self.fov_key = config["pred_key"], config["lab_key"], config["fov_key"]
        self.map_ds = config.get("map_ds", 2)
        self.map_sz = config.get("map_sz", [64, 128])
        self.bandwidth = config.get("bandwidth", 0.2)
        self.kernel = config.get("kernel", "epanechnikov")
        self.maxent_weight = config.get("maxent_weight", 1.0)
        self.reward_weight = config.get("reward_weight", 0.1)
        self.use_fov_mask = config.get("use_fov_mask", False)
        self.alpha = config.get("alpha", None)
        self.cf_key = config.get("cf_key", None)

    @staticmethod
    def compute_expert_visitation(gt, map_ds, map_sz):
        """Rasterise expert polylines: every segment is sampled at `max_steps` points (max over the
        batch of ceil(segment length)), the last pose is appended, points are clamped/truncated to
        cells and each visited cell counts once.  gt [B,T,3,3] or [B,T,2]."""
        xy = gt if gt.ndim == 3 else gt[:, :, :2, 2]
        B = xy.shape[0]
        H, W = map_sz
        xy = xy / map_ds
        seg0, seg1 = xy[:, :-1], xy[:, 1:]
        max_steps = torch.ceil(torch.norm(seg1 - seg0, dim=-1)).long().max().item()
        lam = torch.linspace(0, 1, max_steps, device=gt.device).view(1, 1, -1, 1)
        pts = (seg0.unsqueeze(2) + lam * (seg1 - seg0).unsqueeze(2)).view(B, -1, 2)
        pts = torch.cat([pts, xy[:, -1:]], dim=1)
        rows = pts[:, :, 0].clamp(0, H - 1).long()
        cols = pts[:, :, 1].clamp(0, W - 1).long()
        lin = rows * W + cols
        counts = torch.zeros(B, H * W, dtype=torch.float32, device=gt.device)
        counts.scatter_add_(1, lin, torch.ones_like(lin, dtype=torch.float32))
        counts = counts.view(B, H, W)
        counts[counts > 1] = 1
        return pts, counts

    def _visitation_host(self, exp_svf, gt, fov, tensor_dict):
        """reference loss_utils.py:1139-1186 as tensor code (CPU tensors: tests of the host logic)."""
        _, svf = self.compute_expert_visitation(gt, self.map_ds, self.map_sz)
        if self.use_fov_mask:
            svf = svf * fov.float()
            exp_svf = exp_svf * fov.float()
        svf = svf / (svf.sum(dim=(1, 2), keepdim=True) + 1e-5)
        exp_svf = exp_svf / (exp_svf.sum(dim=(1, 2), keepdim=True) + 1e-5)

        cf_total = torch.zeros_like(svf)
        policy_svf = exp_svf.clone()
        if self.cf_key is not None and self.alpha is not None:
            for i, cf in enumerate(tensor_dict[self.cf_key]):
                if cf is None:
                    continue
                worse = cf["trajectories"][cf["rank"] > 0]          # [n,T,2] sub-optimal alternatives
                if worse.shape[0] == 0:
                    continue
                worse = torch.from_numpy(np.asarray(worse)).to(svf.device)
                _, cf_svf = self.compute_expert_visitation(worse, self.map_ds, self.map_sz)
                cf_svf = cf_svf.sum(dim=0)
                cf_svf = cf_svf / (cf_svf.sum(dim=(0, 1), keepdim=True) + 1e-5)
                exp_svf[i] = self.alpha * cf_svf + (1 - self.alpha) * exp_svf[i]
                cf_total[i] = cf_svf
        return svf, exp_svf, cf_total, policy_svf

    def _visitation_hip(self, exp_svf, gt, fov, tensor_dict):
        """The same on the device: the expert polylines and every sample's sub-optimal counterfactuals are rasterised
        by ONE launch each of csrc/planner.hip (max_steps per reference call = per group), masking / normalisation /
        mixing by one launch of creste_irl_visitation_mix_f32 -- instead of ~12 tensor ops and a host sync per sample.

        Deviations from the host path (documented, both covered by tests/test_model_gpu.py): counterfactual vertices
        arrive as numpy float64 and the reference keeps float64 through the /map_ds, clamp and .long(); here they are
        rounded to float32 before the rasterisation (as the expert poses already are in the reference) -- a visited
        cell can differ only for a sub-sampled point within one float32 ulp of a cell border.  Counterfactual sets of
        different lengths in one batch are padded with their last vertex: a zero-length segment adds no sub-sampled
        point other than that vertex and does not change max_steps, so the visitation maps are unchanged."""
        from ... import _lib
        from ...ops import _stream
        lib = _lib.load()
        dev = exp_svf.device
        H, W = self.map_sz
        B = exp_svf.shape[0]
        xy = (gt if gt.ndim == 3 else gt[:, :, :2, 2]).detach().to(dev).float().contiguous()

        def raster(xy_dev, group, ngroups):
            n, T = xy_dev.shape[0], xy_dev.shape[1]
            visit = torch.empty((n, H, W), dtype=torch.float32, device=dev)
            scores = torch.empty(n, dtype=torch.float32, device=dev)
            work = torch.empty(ngroups, dtype=torch.int32, device=dev)
            _lib.check(lib.creste_trajectory_scores_grouped_f32(
                xy_dev.data_ptr(), n, T, float(self.map_ds), H, W, None, None, 0,
                group.data_ptr() if group is not None else None, ngroups, scores.data_ptr(), visit.data_ptr(), None,
                work.data_ptr(), _stream()), "trajectory_scores")
            return visit

        visit_e = raster(xy, None, 1)
        sets, ptr = [], [0] * (B + 1)
        if self.cf_key is not None and self.alpha is not None:
            for i, cf in enumerate(tensor_dict[self.cf_key]):
                n = 0
                if cf is not None:
                    worse = np.asarray(cf["trajectories"])[np.asarray(cf["rank"]) > 0]
                    n = worse.shape[0]
                    if n:
                        sets.append((i, worse.astype(np.float32)))
                ptr[i + 1] = ptr[i] + n
        visit_c = cf_ptr = None
        if sets:
            Tmax = max(w.shape[1] for _, w in sets)
            sets = [(i, w if w.shape[1] == Tmax else np.concatenate([w, np.repeat(w[:, -1:], Tmax - w.shape[1], axis=1)], axis=1))
                    for i, w in sets]
            allxy = torch.from_numpy(np.concatenate([w for _, w in sets], axis=0)).to(dev)
            group = torch.from_numpy(np.concatenate([np.full(w.shape[0], j, dtype=np.int32) for j, (_, w) in enumerate(sets)])).to(dev)
            visit_c = raster(allxy.contiguous(), group, len(sets))
            cf_ptr = torch.tensor(ptr, dtype=torch.int32).to(dev)
        f8 = fov.to(torch.uint8).contiguous() if self.use_fov_mask else None
        er = exp_svf.detach().float().contiguous()
        svf, ex, cft, pol = (torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(4))
        _lib.check(lib.creste_irl_visitation_mix_f32(
            er.data_ptr(), f8.data_ptr() if f8 is not None else None, visit_e.data_ptr(),
            visit_c.data_ptr() if visit_c is not None else None, cf_ptr.data_ptr() if cf_ptr is not None else None,
            float(self.alpha if self.alpha is not None else 0.0), B, H * W, svf.data_ptr(), ex.data_ptr(), cft.data_ptr(),
            pol.data_ptr(), _stream()), "irl_visitation_mix")
        return svf, ex, cft, pol

    def loss(self, tensor_dict):
        exp_svf = tensor_dict[self.pred_key]
        gt = tensor_dict[self.lab_key]
        fov_mask = tensor_dict[self.fov_key]
        reward = tensor_dict["outputs/traversability_preds"].squeeze(1)
        state_features = tensor_dict["outputs/input_view"]
        _, Ho, Wo = fov_mask.shape
        _, H, W = exp_svf.shape
        fov = tu.resize_and_crop(fov_mask.unsqueeze(1).byte(), (Ho // 2, Wo // 2), (0, H, 0, W))
        fov = fov.squeeze(1).bool()

        if exp_svf.is_cuda:
            svf, exp_svf, cf_total, policy_svf = self._visitation_hip(exp_svf, gt, fov, tensor_dict)
        else:
            svf, exp_svf, cf_total, policy_svf = self._visitation_host(exp_svf, gt, fov, tensor_dict)
        assert torch.all(exp_svf >= 0), "Negative expert visitation frequencies"
        assert torch.all(svf >= 0), "Negative predicted visitation frequencies"

        if self.use_fov_mask:
            keep = torch.ones_like(reward, dtype=torch.float32)
            keep[~fov] = 0
            reward = reward * keep
        mean_exp = (exp_svf * reward).sum(dim=(1, 2)).mean()
        mean_svf = (svf * reward).sum(dim=(1, 2)).mean()
        visitation_loss = mean_exp - mean_svf

        penalty = torch.tensor(0.0, device=svf.device)
        if reward.requires_grad and self.reward_weight > 0:         # SMODICE-style gradient penalty
            grad = torch.autograd.grad(outputs=reward.sum(), inputs=state_features, create_graph=True,
                                       retain_graph=True, only_inputs=True)[0]
            penalty = ((grad.norm(2, dim=1) - 1) ** 2).mean()
        total = self.maxent_weight * visitation_loss + self.reward_weight * penalty

        with torch.no_grad():
            cf_r = (cf_total * reward).sum(dim=(1, 2))
            opt_r = (policy_svf * reward).sum(dim=(1, 2))
            has_cf = cf_r != 0
            cf_r, opt_r = cf_r[has_cf].sum(), opt_r[has_cf].sum()
        meta = {"reward_penalty": self.reward_weight * penalty, "mean_expected_svf_rewards": mean_exp,
                "mean_svf_rewards": mean_svf, "sum_cf_rewards": cf_r, "sum_opt_rewards": opt_r}
        return {"maxentirl_loss": total}, meta


def _bin_depths_ud(depth, depth_min, depth_max, num_bins):
    """target bins of reference depth_utils.bin_depths (mode UD, target=True)."""
    # tensor / tensor: a true IEEE division on every device (tensor / python-scalar becomes a multiplication by the
    # reciprocal on the GPU, which moves labels that sit exactly on a bin edge, e.g. depth == depth_max)
    bin_size = torch.tensor((depth_max - depth_min) / num_bins, dtype=depth.dtype, device=depth.device)
    idx = (depth - depth_min) / bin_size
    bad = (idx < 0) | (idx > num_bins) | (~torch.isfinite(idx))
    idx = idx.masked_fill(bad, num_bins)
    return idx.to(torch.int64)


def _match_depth_label(pred_hw, gt):
    """[B,S,H,W] label -> [B*S,h,w] at the prediction's resolution (nearest, as the reference's workaround)."""
    B, S, H, W = gt.shape
    if tuple(pred_hw) != (H, W):
        gt = torch.nn.functional.interpolate(gt, tuple(pred_hw), mode="nearest").detach()
    return gt.reshape(B * S, *gt.shape[-2:])


class CrossEntropyDepth(Loss):
    """depth as classification (reference loss_utils.py:477-527) -- fused HIP kernel (loss_ops.DepthCEFn)."""

    def __init__(self, config):
        super().__init__(config["name"], config)

    def loss(self, tensor_dict):
        from ...loss_ops import DepthCEFn
        pred = tensor_dict[self.config["pred_key"]]
        gt = tensor_dict[self.config["lab_key"]]
        if pred.shape[0] != gt.shape[0] * gt.shape[1]:
            raise NotImplementedError("multi-frame depth prediction is not configured by the shipped models")
        d = self.config["discretize"]
        if d["mode"] != "UD":
            raise NotImplementedError("HIP depth cross-entropy bins uniformly (mode 'UD')")
        gt = _match_depth_label(pred.shape[-2:], gt)
        loss, stats = DepthCEFn.apply(pred, gt, d["num_bins"], d["depth_min"], d["depth_max"])
        return {"depth/cls_loss": loss}, {"depth/acc": stats[1]}


class SmoothL1Depth(Loss):
    """reference loss_utils.py:530-573: smooth-L1 between the predicted depth and the label in metres over the
    pixels whose label falls into a bin.  The distillation config feeds it `depth_preds_bins` (integer class
    indices: no gradient, a logged scalar); the SSC config feeds it `depth_preds_metric`."""

    def __init__(self, config):
        super().__init__(config["name"], config)
        self.pred_key, self.lab_key = config["pred_key"], config["lab_key"]
        self.smoothl1_loss = torch.nn.SmoothL1Loss(beta=config["beta"], reduction="mean")

    def loss(self, tensor_dict):
        pred = tensor_dict[self.pred_key]
        gt = tensor_dict[self.lab_key]
        if pred.shape[0] != gt.shape[0] * gt.shape[1]:
            raise NotImplementedError("multi-frame depth prediction is not configured by the shipped models")
        # `depth_preds_metric` (the SSC config) is differentiable: its cotangent re-enters the HIP path through the
        # softmax-expectation backward (BackboneFn); `depth_preds_bins` (the distillation config) is an integer map
        d = self.config["discretize"]
        gt = _match_depth_label(pred.shape[-2:], gt)
        if pred.is_cuda and d["mode"] == "UD":              # fused HIP kernel: mask, loss and gradient in two passes
            from ...loss_ops import SmoothL1Fn
            return {"depth/reg_loss": SmoothL1Fn.apply(1, pred.float(), gt, False, self.config["beta"], d["depth_min"],
                                                       d["depth_max"], d["num_bins"])}, {}
        valid = _bin_depths_ud(gt, d["depth_min"], d["depth_max"], d["num_bins"]) != d["num_bins"]
        return {"depth/reg_loss": self.smoothl1_loss(pred[valid].float(), (gt / 1000.0)[valid].float())}, {}


def remap_labels_in_batch(gt, ignore_idx=0):
    """labels of different samples -> disjoint ascending classes, `ignore_idx` kept (reference utils/utils.py:59-77)."""
    B = gt.shape[0]
    out = torch.ones_like(gt) * ignore_idx
    offset = 0
    for b in range(B):
        # the running index counts the ignore label too (it is enumerated, then skipped), so the first real label
        # of a sample never collides with ignore_idx = 0
        remap = {l: i + offset for i, l in enumerate(torch.unique(gt[b])) if l != ignore_idx}
        for l, new in remap.items():
            out[b, gt[b] == l] = new
        offset += len(remap)
    return out


def extract_max_per_class(tensor, max_per_class=100):
    """indices of at most `max_per_class` random elements of every class, classes ascending
    (reference train_utils.py:324-352; the permutation comes from the host generator, as there)."""
    picked = torch.LongTensor().to(tensor.device)
    for cls in torch.unique(tensor):
        idx = (tensor == cls).nonzero(as_tuple=False).reshape(-1)
        if idx.size(0) > max_per_class:
            idx = idx[torch.randperm(idx.size(0))[:max_per_class]]
        picked = torch.cat((picked, idx))
    return picked


class MultiPosConLoss(nn.Module):
    """multi-positive contrastive loss over [N,D] features (reference models/losses/supcon_loss.py:56-115): features
    L2-normalised, all-gathered across ranks (with gradient) when a process group exists, positives = same label
    (self excluded), cross-entropy between the positive distribution and the softmax of the similarities / T.
    The reference rebuilds its label mask only when the local batch size changes (:87-99); that is reproduced
    (`_last_n`) because it changes the loss on consecutive batches of equal size."""

    def __init__(self, temperature=0.1, class_weights=None):
        super().__init__()
        self.temperature, self.class_weights = temperature, class_weights
        self.logits_mask = self.mask = self._last_n = None

    def forward(self, outputs):
        import torch.distributed as dist
        feats, labels = outputs["feats"], outputs["labels"]
        if self.class_weights is not None:
            self.class_weights = self.class_weights.to(feats.device)
        feats = torch.nn.functional.normalize(feats, dim=-1, p=2)
        n = feats.size(0)
        # every rank contributes however many rows its data yields (padded all-gather, features with gradient);
        # `row0` = position of this rank's first row in the gathered set (the reference's n * rank when counts agree)
        from ...dist_utils import gather_varlen
        all_feats, all_labels, row0 = gather_varlen(feats, labels)
        if feats.is_cuda and feats.shape[1] in (8, 16, 32, 64):
            # fused HIP path (no N x M tensors).  The reference's stale-mask behaviour is kept by remembering the LABELS
            # the mask was built from: positives follow those, the class weights follow the current labels.
            from ...loss_ops import MultiPosConFn
            if (n, all_feats.size(0)) != self._last_n:
                self._mask_labels, self._mask_all_labels, self._last_n = labels.clone(), all_labels.clone(), (n, all_feats.size(0))
            rw = self.class_weights[labels] if self.class_weights is not None else None
            loss = MultiPosConFn.apply(feats, all_feats, self._mask_labels, self._mask_all_labels, rw, row0,
                                       self.temperature)
            return {"loss": loss, "image_loss": loss}
        if (n, all_feats.size(0)) != self._last_n:
            mask = torch.eq(labels.view(-1, 1), all_labels.contiguous().view(1, -1)).float()
            self.logits_mask = torch.scatter(torch.ones_like(mask), 1,
                                             torch.arange(n, device=feats.device).view(-1, 1) + row0, 0)
            self._last_n = (n, all_feats.size(0))
            self.mask = mask * self.logits_mask
        mask = self.mask
        logits = torch.matmul(feats, all_feats.T) / self.temperature
        logits = logits - (1 - self.logits_mask) * 1e9
        logits = logits - logits.max(dim=-1, keepdim=True)[0].detach()
        p = mask / mask.sum(1, keepdim=True).clamp(min=1.0)
        loss = torch.sum(p * torch.log_softmax(logits, dim=-1), dim=-1)
        if self.class_weights is not None:
            loss = loss * self.class_weights[labels]
        loss = -loss.mean()
        return {"loss": loss, "image_loss": loss}


def _class_weights(config, eps=1e-5):
    """1 / log(frequency + eps) from the text file the config names (reference loss_utils.py:383-391)."""
    if "class_weights" not in config:
        return None
    freq = np.loadtxt(config["class_weights"]) if isinstance(config["class_weights"], str) else np.asarray(config["class_weights"])
    return torch.from_numpy(1 / np.log(freq + eps)).float()


class SupPixelConLoss(Loss):
    """supervised pixel-contrastive loss on BEV embeddings (reference loss_utils.py:203-286)."""

    def __init__(self, config):
        super().__init__(config["name"], config)
        self.views = config.get("views", 1)
        self.temperature = config.get("temperature", 0.1)
        cw = _class_weights(config)
        if cw is not None:
            self.register_buffer("class_weights", cw)
            assert config["num_class"] == len(cw)
        else:
            self.class_weights = None
        self.supcon_loss = MultiPosConLoss(temperature=self.temperature, class_weights=self.class_weights)
        self.ignore_index = config.get("ignore_index", -1)
        self.mask_key = config.get("mask_key", "inputs/fov_mask")
        self.pred_key = config.get("pred_key", "outputs/inpainting_preds")
        self.lab_key = config.get("lab_key", "inputs/sem_label")
        self.lab_suffix_key = self.lab_key.split("/")[-1]
        self.task = config.get("task", "3d_ssc")

    def loss(self, tensor_dict):
        preds, gt_prob, fov_mask = tensor_dict[self.pred_key], tensor_dict[self.lab_key], tensor_dict[self.mask_key]
        C = gt_prob.shape[1]
        BV, Z, H, W = preds.shape
        B = BV // self.views
        gt_label = torch.argmax(gt_prob, dim=1) if C > 1 else gt_prob.squeeze(1)
        if preds.is_cuda and self.views == 1 and gt_label.dtype == torch.int64:
            # device path (label_ops / csrc/labels.hip): remap, class-wise grouping of the valid cells and the row gather
            # are kernels; the host draws the per-class permutations with the reference's generator, in its order
            from ... import label_ops
            if self.lab_key == "inputs/3d_sam_label":
                gt_label, nclass = label_ops.remap_labels_in_batch(gt_label.contiguous(), ignore_idx=0)
                K = int(nclass.item())
            else:
                K = int(gt_label.max().item()) + 1
            cell, sel_labels = label_ops.sample_cells_per_class(gt_label, fov_mask, K, self.ignore_index)
            feats = label_ops.RowsFn.apply(preds, cell)
            out = self.supcon_loss({"feats": feats, "labels": sel_labels})
            k = f"{self.task}/{self.lab_suffix_key}/supcon"
            return {f"{k}/sem_loss": out["loss"], f"{k}/img_loss": out["image_loss"]}, {}
        if self.lab_key == "inputs/3d_sam_label":
            gt_label = remap_labels_in_batch(gt_label, ignore_idx=0)
        valid = (gt_label != self.ignore_index) & fov_mask
        valid = valid.view(B, self.views, H, W).permute(0, 2, 3, 1)[:, :, :, 0]
        preds = preds.permute(0, 2, 3, 1).view(B, self.views, H, W, Z).permute(0, 2, 3, 1, 4)
        preds = preds[valid][:, 0]
        gt_label = gt_label.view(B, self.views, H, W).permute(0, 2, 3, 1)[valid][:, 0]
        counts = torch.bincount(gt_label)
        nz = counts[counts.nonzero(as_tuple=True)].float()
        median = min(nz.median().int(), 1000)
        sel = extract_max_per_class(gt_label, median)
        out = self.supcon_loss({"feats": preds[sel, :], "labels": gt_label[sel]})
        k = f"{self.task}/{self.lab_suffix_key}/supcon"
        return {f"{k}/sem_loss": out["loss"], f"{k}/img_loss": out["image_loss"]}, {}


class CrossEntropy(Loss):
    """BEV classification with optional class weights / ignore index (reference loss_utils.py:379-474)."""

    def __init__(self, config):
        super().__init__(config["name"], config)
        self.num_class = config["num_class"]
        self.epsilon_w = 1e-5
        cw = _class_weights(config, self.epsilon_w)
        if cw is not None:
            self.register_buffer("class_weights", cw)
            assert self.num_class == len(cw)
        else:
            self.class_weights = None
        self.mask_key = config.get("mask_key", "inputs/fov_mask")
        self.pred_key = config.get("pred_key", "outputs/inpainting_preds")
        self.lab_key = config.get("lab_key", "inputs/sem_label")
        self.ignore_index = config.get("ignore_index", None)
        self.task = config.get("task", "3d_ssc")
        self.class_dim = config.get("class_dim", -1)
        kw = dict(reduction="mean", weight=self.class_weights)
        if self.ignore_index is not None:
            kw["ignore_index"] = self.ignore_index
        self.cs_loss = torch.nn.CrossEntropyLoss(**kw)

    def loss(self, tensor_dict):
        pred, gt, fov = tensor_dict[self.pred_key], tensor_dict[self.lab_key], tensor_dict[self.mask_key]
        if pred.is_cuda and pred.shape[1] <= 64:            # fused HIP kernel (mask, label, loss, metric, gradient)
            from ...loss_ops import BevCEFn
            loss, stats = BevCEFn.apply(pred, gt, fov, self.class_weights, self.class_dim, self.ignore_index, self.epsilon_w)
            return {f"{self.task}/cls_loss": loss}, {f"{self.task}/mIoU": stats[1]}
        if self.class_dim < 0:
            gt_mode = torch.argmax(gt / (torch.sum(gt, dim=1, keepdim=True) + self.epsilon_w), dim=1)
        else:
            gt_mode = gt[:, self.class_dim, :, :].long()
        pred = pred.permute(0, 2, 3, 1)[fov, :]
        gt_mode = gt_mode[fov]
        loss = self.cs_loss(pred, gt_mode)
        with torch.no_grad():
            mode = torch.argmax(torch.softmax(pred, dim=1), dim=1)
            lab = gt_mode != 0                          # class 0 is taken to be the ignore label in the metric
            acc = torch.sum(mode[lab] == gt_mode[lab]) / (torch.numel(gt_mode[lab]) + self.epsilon_w)
        return {f"{self.task}/cls_loss": loss}, {f"{self.task}/mIoU": acc}


class SmoothL1(Loss):
    """elevation regression (reference loss_utils.py:576-603): channel 1 relative to channel 0 unless `absolute`,
    optional spatial-gradient form, nan / inf labels masked.  (The reference rewrites the label tensor in place;
    the same values are used here without mutating the caller's tensor.)"""

    def __init__(self, config):
        super().__init__(config["name"], config)
        self.pred_key, self.lab_key = config["pred_key"], config["lab_key"]
        self.absolute = config.get("absolute", False)
        self.take_grad = config.get("take_grad", False)
        self.smoothl1_loss = torch.nn.SmoothL1Loss(beta=config["beta"])

    def loss(self, tensor_dict):
        pred, gt = tensor_dict[self.pred_key], tensor_dict[self.lab_key]
        if pred.is_cuda and not self.take_grad and pred.dim() == 4 and pred.shape[1] == 2:
            from ...loss_ops import SmoothL1Fn
            return {"val": SmoothL1Fn.apply(0, pred, gt, self.absolute, self.config["beta"], 0.0, 1.0, 0)}, {}
        if not self.absolute:
            gt = gt.clone()
            gt[:, 1, :, :] = gt[:, 1, :, :] - gt[:, 0, :, :]
        if self.take_grad:
            assert pred.dim() == 4
            pred = torch.cat(torch.gradient(pred, dim=[2, 3]), dim=1)
            gt = torch.cat(torch.gradient(gt, dim=[2, 3]), dim=1)
        valid = ~torch.isnan(gt) & ~torch.isinf(gt)
        return {"val": self.smoothl1_loss(pred[valid], gt[valid])}, {}


class MSELoss(Loss):
    """single-view DINO feature match (reference loss_utils.py:606-647, overlap_only=False) -- fused HIP kernel."""

    def __init__(self, config):
        super().__init__(config["name"], config)
        self.pred_key, self.lab_key = config["pred_key"], config["lab_key"]
        if config.get("overlap_only", False):
            raise NotImplementedError("MSELoss(overlap_only=True) is a BEV-stage option, not on the HIP path yet")

    def loss(self, tensor_dict):
        from ...loss_ops import MSEFn
        pred, gt = tensor_dict[self.pred_key], tensor_dict[self.lab_key]
        B, V, Z, H, W = pred.shape
        return {"loss": MSEFn.apply(pred.reshape(B * V, Z, H, W), gt.reshape(B * V, Z, H, W))}, {}


_LOSSES = {"MaxEntIRLLoss": MaxEntIRLLoss, "CrossEntropyDepth": CrossEntropyDepth, "SmoothL1Depth": SmoothL1Depth,
           "MSELoss": MSELoss, "SupPixelConLoss": SupPixelConLoss, "CrossEntropy": CrossEntropy, "SmoothL1": SmoothL1}


class LossManager(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.losses = nn.ModuleList([self.get_loss(lc) for lc in config["loss"]])

    def get_loss(self, config):
        try:
            return _LOSSES[config["name"]](config)
        except KeyError:
            raise NotImplementedError(
                f"loss {config['name']}: only the IRL objective is in this round's scope "
                "(SSC / distillation losses belong to the backbone-training rows, SURVEY.md A12)")

    def forward(self, tensor_dict):
        loss_dict, meta = {}, {}
        for l in self.losses:
            if l.task is None or l.task == tensor_dict["task"]:
                ld, md = l(tensor_dict)
                meta.update({f"{l.name}/{k}": v for k, v in md.items()})
                loss_dict.update({f"{l.name}/{k}": v for k, v in ld.items()})
        return loss_dict, meta
