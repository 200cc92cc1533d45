"""Training-step semantics of the IRL stage without Lightning -- the counterpart of the reference's
`MaxEntIRLModel` (/root/reference/creste/train_traversability.py:34-330):

* manual optimisation, per task: `zero_grad -> forward((image, p2p, expert)) -> LossManager(merged dict)
  -> sum(weight * value) -> backward -> step` (:66-96);
* Adam(lr, betas) over the parameters that require grad, ExponentialLR(gamma) stepped once per epoch
  (:313-330); seed 1337 (`pl.seed_everything`, :401-413);
* data parallel = one process per GPU; the only exchange is the reward network's gradient (102,866 fp32,
  0.41 MB): ONE flat RCCL all-reduce per step (`dist_utils.allreduce_mean_grads`) instead of DDP buckets;
  BatchNorm statistics stay per GPU, as in the reference (no SyncBN);
* checkpoints use Lightning's layout `{'state_dict': {'model.<name>': tensor}, 'epoch': ..}` so they
  load through the mirrored `load_weights` of MaxEntIRL / TerrainNet and through the reference itself.
"""
from __future__ import annotations

import random

import numpy as np
import os

import torch

from . import dist_utils, ops
from .creste.utils import train_utils as tu


def seed_everything(seed: int = 1337):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class IRLTrainer:
    def __init__(self, model: torch.nn.Module, loss_manager: torch.nn.Module, model_cfg, graphs: bool = False,
                 priority_stream: bool = False):
        """graphs=True: the reward network's launch sequences are captured into hipGraphs after one eager step
        and replayed (train_ops._Phases) -- same results, static shapes required.
        priority_stream=True: with a look-ahead batch (training_step(batch, next_batch)) the trainable half runs on a
        high-priority stream.  Measured at batch 8 of 1216x608, bf16x6 backbone: 64x128 MDP grid 38.9 serial -> 35.7
        pipelined -> 32.4 with the priority stream; 256x256 MDP grid 58.7 -> 53.2 -> 61.3 (its reward-net / weight-gradient
        kernels fill the chip themselves and then only delay the backbone), so it is opt-in."""
        self.model, self.loss, self.cfg = model, loss_manager, model_cfg
        self.priority_stream = priority_stream
        if graphs:
            from .creste.models.blocks.conv import MultiScaleFCN
            for m in model.modules():
                if isinstance(m, MultiScaleFCN):
                    m.train_graphs = True
        oc = model_cfg["optimizer"]
        if oc["name"] != "Adam":
            raise NotImplementedError(oc["name"])
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.optimizer = torch.optim.Adam(self.params, betas=(oc["beta1"], oc["beta2"]), lr=oc["lr"])
        sc = model_cfg["lr_scheduler"]
        if sc["name"] != "ExponentialLR":
            raise NotImplementedError(sc["name"])
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=sc["gamma"])
        self.epoch, self.global_step = 0, 0

    def training_step(self, batch: dict, next_batch: dict | None = None) -> dict:
        """batch: {task: {'image','p2p','traversability_label', 'fov_mask', 'counterfactuals_label', ...}}

        next_batch (optional, the loader's look-ahead): its frozen-backbone forward is enqueued on a side stream right
        after this batch's has been picked up, and overlaps this batch's trainable half (MaxEntIRL.prefetch_backbone);
        results are identical to the serial order -- the frozen half depends on no trainable parameter."""
        self.model.train()
        pipelined = next_batch is not None and hasattr(self.model, "prefetch_backbone")
        if not (pipelined and self.priority_stream):
            return self._training_step(batch, next_batch if pipelined else None)
        # the trainable half is a latency chain of ~600 small kernels: on a HIGH-priority stream its workgroups are
        # dispatched ahead of the side stream's full-chip backbone kernels instead of queueing behind them
        main = torch.cuda.current_stream()
        if getattr(self, "_hp_stream", None) is None:
            self._hp_stream = torch.cuda.Stream(priority=min(torch.cuda.Stream.priority_range()))
        self._hp_stream.wait_stream(main)
        with torch.cuda.stream(self._hp_stream):
            logs = self._training_step(batch, next_batch)
        main.wait_stream(self._hp_stream)
        return logs

    def _training_step(self, batch: dict, next_batch) -> dict:
        logs, total = {}, 0.0
        tasks = list(batch.items())
        for ti, (task, data) in enumerate(tasks):
            self.optimizer.zero_grad()
            inputs = (data["image"], data["p2p"], data["traversability_label"])
            nxt = tasks[ti + 1][1] if ti + 1 < len(tasks) else (next(iter(next_batch.values())) if next_batch else None)
            if nxt is not None and hasattr(self.model, "prefetch_backbone"):
                if getattr(self.model, "_prefetched", None) is None:          # first step: nothing in flight yet
                    self.model.prefetch_backbone(inputs)
                pf = self.model._take_prefetched(inputs)                      # (waits for the side stream's event)
                # Where the next frozen half is enqueued.  Round 3 put it BEHIND the value iteration: its solver met at a
                # device-scope rendezvous and, next to a stream of full-chip conv kernels, its unscheduled workgroups starved
                # while the resident ones spun (a 57 ms step took minutes).  The barrier-free solver of round 4 only waits for
                # its halo neighbours, with bounded polls, and gets its CUs as the backbone's workgroups retire.  Round 6
                # (scripts/irl_early.py): enqueued right HERE, before the reward forward and the solve, the backbone also
                # runs under those and the step becomes the backbone's own time -- reference grid 21.4 -> 20.45 ms, cf-IRL
                # 512^2 32.25 -> 31.3, 256^2 MDP grid 34.6 -> 34.4, with or without the priority stream, 0 aborted solves of
                # 276.  A solve that does report INT32_MIN is redone below in the launch-per-chunk form, which needs no
                # co-residency.  CRESTE_IRL_LATE_PREFETCH=1 restores the old order.
                early = os.environ.get("CRESTE_IRL_LATE_PREFETCH") != "1"
                if early:
                    self.model.prefetch_backbone((nxt["image"], nxt["p2p"]))
                outputs = self.model._forward_trainable(inputs, pf)           # reward forward, value iteration, SVF
                if not early:
                    self.model.prefetch_backbone((nxt["image"], nxt["p2p"]))
            else:
                outputs = self.model(inputs)
            loss_dict, meta, loss = self._loss_and_backward(task, data, outputs)
            # the MDP solve of this step reports failure in its sweep count, asynchronously: look at it before the
            # gradients are used (waits for the solve's event only; loss / backward are still queued behind it).  A solve
            # that was ABORTED (its workgroups were not all resident: another stream's kernels held the device) is redone
            # in the launch-per-chunk form; no convergence raises.
            if any(n == ops.VI_ABORTED for n in self._vi_counts()):
                self.optimizer.zero_grad()
                # the retry re-runs the trainable forward in training mode: its BatchNorms must not fold this batch into
                # their running statistics a second time -- momentum 0 leaves them bit-unchanged
                bns = [m for m in self.model.modules()
                       if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training and m.momentum]
                saved = [m.momentum for m in bns]
                for m in bns:
                    m.momentum = 0.0
                try:
                    with ops.vi_launch_per_chunk():
                        outputs = self.model._forward_trainable(inputs, None) if hasattr(self.model, "_forward_trainable") \
                            else self.model(inputs)
                finally:
                    for m, mom in zip(bns, saved):
                        m.momentum = mom
                        if m.num_batches_tracked is not None:
                            m.num_batches_tracked -= 1
                loss_dict, meta, loss = self._loss_and_backward(task, data, outputs)
                if any(n == ops.VI_ABORTED for n in self._vi_counts()):       # (no convergence already raised in there)
                    raise ops.HipLibraryError("IRLTrainer: the MDP solve was aborted again in its launch-per-chunk retry; "
                                              "the gradients of this step are invalid")
                self.vi_retries += 1
            dist_utils.allreduce_mean_grads(self.params)          # one flat all-reduce (no-op on 1 GPU)
            self.optimizer.step()
            total = total + loss.detach()
            logs.update({f"train/{k}": (w * v.detach()) for k, (w, v) in loss_dict.items()})
            logs.update({f"train/{k}": v.detach() for k, v in meta.items()})
        logs["train/loss"] = total
        self.global_step += 1
        return logs

    vi_retries = 0

    def _loss_and_backward(self, task, data, outputs):
        with torch.no_grad():
            merged = tu.merge_dict(("inputs", data), ("outputs", outputs))
            merged["task"] = task
        # tensors that carry the autograd graph must not be detached by the merge above
        merged["outputs/traversability_preds"] = outputs["traversability_preds"]
        merged["outputs/input_view"] = outputs["input_view"]
        loss_dict, meta = self.loss(merged)
        loss = sum(w * v for w, v in loss_dict.values())
        loss.backward()
        return loss_dict, meta, loss

    @staticmethod
    def _vi_counts() -> list:
        """sweep counts of the solves issued since the last look; raises on a solve that did not converge"""
        counts = ops.vi_poll(wait=True)
        for n in counts:
            if n != ops.VI_ABORTED:
                ops.check_vi_sweeps(torch.tensor([n], dtype=torch.int32))
        return counts

    def on_train_epoch_end(self):
        ops.vi_check(wait=True)            # the epoch's last solve: nothing later would look at its sweep count
        self.scheduler.step()
        self.epoch += 1

    def on_validation_epoch_end(self):
        """call at the end of an evaluation loop (Lightning's hook name): the LAST solve of a loop is checked by nobody else"""
        ops.vi_check(wait=True)

    # ---- Lightning-layout checkpoints
    def checkpoint(self) -> dict:
        return {"state_dict": {f"model.{k}": v.detach().cpu() for k, v in self.model.state_dict().items()},
                "epoch": self.epoch, "global_step": self.global_step,
                "optimizer_states": [self.optimizer.state_dict()],
                "lr_schedulers": [self.scheduler.state_dict()]}

    def save_checkpoint(self, path: str):
        if not dist_utils.is_dist() or torch.distributed.get_rank() == 0:
            torch.save(self.checkpoint(), path)

    def load_checkpoint(self, path: str, strict: bool = True):
        ck = torch.load(path, weights_only=False, map_location="cpu")
        sd = {k.replace("model.", "", 1): v for k, v in ck["state_dict"].items() if k.startswith("model.")}
        self.model.load_state_dict(sd, strict=strict)
        if "optimizer_states" in ck:
            self.optimizer.load_state_dict(ck["optimizer_states"][0])
        if "lr_schedulers" in ck:
            self.scheduler.load_state_dict(ck["lr_schedulers"][0])
        self.epoch, self.global_step = ck.get("epoch", 0), ck.get("global_step", 0)


def distillation_cfg(image_size=(512, 612)):
    """Stage-1 distillation hyper-parameters (reference configs/model/distillation/effnet_ds2_dinov2_128.yaml:
    optimiser :63-71, losses :72-88) on top of the backbone config."""
    from .config import terrainnet_cfg
    cfg = terrainnet_cfg(image_size)
    disc = dict(cfg["discretize"])
    cfg["optimizer"] = dict(name="Adam", beta1=0.9, beta2=0.999, lr=0.0005, eps=1e-7)
    cfg["lr_scheduler"] = dict(name="ExponentialLR", gamma=0.98)
    cfg["loss"] = [
        dict(name="CrossEntropyDepth", weight=0.5, pred_key="outputs/depth_preds_logits",
             lab_key="inputs/depth_label", discretize=disc),
        dict(name="SmoothL1Depth", weight=0.1, pred_key="outputs/depth_preds_bins", lab_key="inputs/depth_label",
             beta=0.5, discretize=disc),
        dict(name="MSELoss", weight=1.0, pred_key="outputs/dino_pe_feats", lab_key="inputs/fimg_label",
             overlap_only=False)]
    return cfg


class DistillTrainer(IRLTrainer):
    """Stage-1 distillation step (reference train_pefree.py:71-99, Lightning automatic optimisation + DDP):
    zero_grad -> DistillationBackbone(rgbd) in training mode -> LossManager -> backward -> optimizer.step.
    Data parallel: the backbone's gradients are produced into one flat buffer in backward-completion order and
    all-reduced bucket by bucket while the backward runs (dist_utils.GradArena) -- the ~102 MB exchange of
    SURVEY section 8e; BatchNorm statistics stay per rank, as in the reference (no SyncBN)."""

    def __init__(self, model, loss_manager, model_cfg, bucket_mb: int = 32):
        super().__init__(model, loss_manager, model_cfg)
        oc = model_cfg["optimizer"]
        self.optimizer = torch.optim.Adam(self.params, betas=(oc["beta1"], oc["beta2"]), lr=oc["lr"])
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=model_cfg["lr_scheduler"]["gamma"])
        self.bucket_bytes = bucket_mb << 20

    def training_step(self, batch: dict) -> dict:
        """batch: {'image' [B,1,4,H,W], 'depth_label' [B,1,Hs,Ws] mm, 'fimg_label' [B,1,Z,Hs,Ws], ...}"""
        self.model.train()
        self.optimizer.zero_grad()
        outputs = self.model(batch["image"])
        eng = getattr(self.model, "_train_engine", None)
        if eng is not None:                               # HIP engine: gradients in a flat arena, comm overlapped
            eng.arena, eng.bucket_bytes = True, self.bucket_bytes
        with torch.no_grad():
            merged = tu.merge_dict(("inputs", batch), ("outputs", outputs))
            merged["task"] = None
        for k, v in outputs.items():                      # keep the autograd graph of the differentiable outputs
            merged[f"outputs/{k}"] = v
        loss_dict, meta = self.loss(merged)
        loss = sum(w * v for w, v in loss_dict.values())
        loss.backward()
        if eng is None:                                   # stand-in models (CPU tests): one flat all-reduce afterwards
            dist_utils.allreduce_mean_grads(self.params)
        self.optimizer.step()
        logs = {f"train/{k}": (w * v.detach()) for k, (w, v) in loss_dict.items()}
        logs.update({f"train/{k}": v.detach() for k, v in meta.items()})
        logs["train/loss"] = loss.detach()
        self.global_step += 1
        return logs


def ssc_cfg(image_size=(512, 612), class_weights=None, freeze_backbone_epochs=0):
    """BEV-SSC hyper-parameters (reference configs/model/ssc_sam/terrainnet_supcon_sam2dynelev_jointdinopretrain.yaml:
    optimiser / scheduler :80-89, losses :93-135).  `class_weights`: path of the 6-class frequency file the reference
    reads (data/creste/class_weights_3d_sam_dynamic_6.txt) or a list of frequencies; None = unweighted."""
    from .config import terrainnet_cfg
    cfg = terrainnet_cfg(image_size)
    disc = dict(cfg["discretize"])
    cfg["optimizer"] = dict(name="Adam", beta1=0.9, beta2=0.999, lr=0.0005, eps=1e-7)
    cfg["lr_scheduler"] = dict(name="ExponentialLR", gamma=0.98)
    cfg["freeze_backbone_epochs"] = freeze_backbone_epochs
    ce = dict(name="CrossEntropy", weight=2.0, pred_key="outputs/inpainting_sam_dynamic_preds",
              lab_key="inputs/3d_sam_dynamic_label", num_class=6, class_dim=1, task="joint")
    if class_weights is not None:
        ce["class_weights"] = class_weights
    cfg["loss"] = [
        dict(name="SupPixelConLoss", views=1, weight=1.0, pred_key="outputs/inpainting_sam_preds",
             lab_key="inputs/3d_sam_label", ignore_index=0, temperature=0.1, task="joint", contrast_mode="batch_all"),
        ce,
        dict(name="MSELoss", weight=2.0, pred_key="outputs/dino_pe_feats", lab_key="inputs/fimg_label", overlap_only=False),
        dict(name="CrossEntropyDepth", weight=0.5, pred_key="outputs/depth_preds_logits", lab_key="inputs/depth_label",
             discretize=disc),
        dict(name="SmoothL1Depth", weight=0.1, pred_key="outputs/depth_preds_metric", lab_key="inputs/depth_label",
             beta=0.5, discretize=disc),
        dict(name="SmoothL1", weight=3.0, beta=0.2, pred_key="outputs/elevation_preds", lab_key="inputs/elevation_label",
             absolute=False, task="joint")]
    return cfg


class SSCTrainer(DistillTrainer):
    """BEV-SSC step (reference train_ssc.py:62-129, Lightning automatic optimisation + DDP): for every task of the batch
    TerrainNet((image, p2p)) in training mode and the task's losses; the summed loss is backpropagated once, gradients are
    averaged over ranks, Adam steps.  `on_train_epoch_start` freezes / unfreezes `model.depthcomp` by epoch (:71-80)."""

    def __init__(self, model, loss_manager, model_cfg, bucket_mb: int = 32):
        super().__init__(model, loss_manager, model_cfg, bucket_mb)
        self.freeze_backbone_epochs = model_cfg.get("freeze_backbone_epochs", 0)
        self.backbone_frozen = False
        self.on_train_epoch_start()

    def _rebuild_optimizer(self):
        oc = self.cfg["optimizer"]
        lr = self.optimizer.param_groups[0]["lr"]
        self.params = [p for p in self.model.parameters() if p.requires_grad]
        self.optimizer = torch.optim.Adam(self.params, betas=(oc["beta1"], oc["beta2"]), lr=lr)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=self.cfg["lr_scheduler"]["gamma"])
        self._rebuild_arena()

    def _rebuild_arena(self):
        """Gradient exchange of the step.  The encoder's engine (train_backbone.BackboneFn) owns a GradArena of its
        own: its ~64 MB of gradients are all-reduced bucket by bucket INSIDE its backward.  Everything else (BEV heads,
        splat stage: ~38 MB, finished FIRST by the backward) lives in a HookedArena: the autograd hooks send those
        buckets while the splat and encoder backward are still running."""
        if getattr(self, "arena", None) is not None:
            self.arena.close()
        dc = getattr(self.model, "depthcomp", None)
        self._engine_owned = {id(p) for p in dc.parameters()} if (dc is not None and self._hip_model()) else set()
        rest = [p for p in self.params if id(p) not in self._engine_owned]
        self.arena = dist_utils.HookedArena(rest, self.bucket_bytes) if rest else None

    def _hip_model(self) -> bool:
        return type(self.model).__module__.startswith("creste_public_amd.creste.")

    def on_train_epoch_start(self):
        if self.epoch >= self.freeze_backbone_epochs and self.backbone_frozen:
            self.model.depthcomp.unfreeze_backbone()
            self.backbone_frozen = False
            self._rebuild_optimizer()
        elif self.epoch < self.freeze_backbone_epochs and not self.backbone_frozen:
            for p in self.model.depthcomp.parameters():
                p.requires_grad = False
            self.backbone_frozen = True
            self._rebuild_optimizer()

    def load_checkpoint(self, path: str, strict: bool = True):
        # the optimiser's parameter group depends on the freeze schedule: restore the epoch first
        self.epoch = torch.load(path, weights_only=False, map_location="cpu").get("epoch", 0)
        self.on_train_epoch_start()
        super().load_checkpoint(path, strict)

    def training_step(self, batch: dict) -> dict:
        """batch: {task: {'image', 'p2p', labels ...}}"""
        self.model.train()
        self.optimizer.zero_grad()
        if not hasattr(self, "_engine_owned"):
            self._rebuild_arena()
        if self.arena is not None:
            self.arena.begin()                            # .grad of the non-encoder parameters = views of one buffer
        total, logs = 0.0, {}
        for task, data in batch.items():
            outputs = self.model((data["image"], data["p2p"], data.get("immovable_depth_label", None)))
            eng = getattr(getattr(self.model, "depthcomp", None), "_train_engine", None)
            if eng is not None and self._engine_owned:    # encoder gradients: flat arena, comm overlapped in its backward
                eng.arena, eng.bucket_bytes = True, self.bucket_bytes
            with torch.no_grad():
                merged = tu.merge_dict(("inputs", data), ("outputs", outputs))
                merged["task"] = task
            for k, v in outputs.items():
                merged[f"outputs/{k}"] = v
            loss_dict, meta = self.loss(merged)
            total = total + sum(w * v for w, v in loss_dict.values())
            logs.update({f"train/{k}": (w * v.detach()) for k, (w, v) in loss_dict.items()})
            logs.update({f"train/{k}": v.detach() for k, v in meta.items()})
        total.backward()
        if self.arena is not None:
            self.arena.finish()                           # tail bucket, wait, average
        self.optimizer.step()
        logs["train/loss"] = total.detach()
        self.global_step += 1
        return logs
