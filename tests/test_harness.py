"""IRL training harness: step order, optimiser/scheduler settings, Lightning-layout
checkpoints -- CPU test with a stand-in model for the host logic, GPU test with the real HIP model."""
import numpy as np
import pytest
import torch

from creste_public_amd import synth
from creste_public_amd.config import maxent_irl_cfg
from oracle.blocks import MultiScaleFCN            # CPU stand-in: the product net trains on HIP kernels only
from creste_public_amd.creste.utils.loss_utils import LossManager
from creste_public_amd.harness import IRLTrainer, seed_everything


class _Stub(torch.nn.Module):
    """frozen 'backbone' + trainable reward net producing the output keys the loss consumes"""

    def __init__(self, cfg):
        super().__init__()
        self.backbone = torch.nn.Linear(3, 3)
        for p in self.backbone.parameters():
            p.requires_grad = False
        self.r = MultiScaleFCN(cfg["traversability_head"]["net_kwargs"]["reward_cfg"]["net_kwargs"])

    def forward(self, inputs):
        image = inputs[0]
        iv = image.detach().clone().requires_grad_(True)
        return {"traversability_preds": self.r(iv), "input_view": iv,
                "exp_svf": torch.rand(image.shape[0], 64, 128, generator=torch.Generator().manual_seed(0))}


def _batch(B, device="cpu"):
    g = torch.Generator().manual_seed(1)
    return {"irl": {"image": torch.randn(B, 40, 64, 128, generator=g).to(device), "p2p": torch.eye(4),
                    "traversability_label": synth.make_experts(B, 50, 256, seed=2).to(device),
                    "fov_mask": torch.ones(B, 256, 256, dtype=torch.bool, device=device),
                    "counterfactuals_label": [None] * B}}


def test_trainer_semantics_and_checkpoint_roundtrip(tmp_path):
    cfg = maxent_irl_cfg()
    seed_everything(1337)
    model = _Stub(cfg)
    tr = IRLTrainer(model, LossManager(cfg), cfg)
    assert len(tr.params) == sum(1 for p in model.r.parameters())          # frozen backbone excluded
    g0 = tr.optimizer.param_groups[0]
    assert g0["lr"] == 5e-4 and g0["betas"] == (0.9, 0.999)
    w0 = {k: v.clone() for k, v in model.r.state_dict().items()}
    logs = tr.training_step(_batch(2))
    assert "train/loss" in logs and "train/MaxEntIRLLoss/maxentirl_loss" in logs
    assert "train/MaxEntIRLLoss/reward_penalty" in logs and torch.isfinite(logs["train/loss"])
    assert any(not torch.equal(w0[k], v) for k, v in model.r.state_dict().items() if "weight" in k)
    tr.on_train_epoch_end()
    assert abs(tr.optimizer.param_groups[0]["lr"] - 5e-4 * 0.96) < 1e-12 and tr.epoch == 1
    path = str(tmp_path / "Adam-epoch=00.ckpt")
    tr.save_checkpoint(path)
    ck = torch.load(path, weights_only=False)
    assert all(k.startswith("model.") for k in ck["state_dict"])
    assert "model.r.prepool.0.conv.weight" in ck["state_dict"]
    seed_everything(7)
    tr2 = IRLTrainer(_Stub(cfg), LossManager(cfg), cfg)
    tr2.load_checkpoint(path)
    for k, v in model.state_dict().items():
        assert torch.equal(tr2.model.state_dict()[k], v)
    assert tr2.epoch == 1 and abs(tr2.optimizer.param_groups[0]["lr"] - 5e-4 * 0.96) < 1e-12
    # identical next step from the restored state
    a = tr.training_step(_batch(2))["train/loss"]
    b = tr2.training_step(_batch(2))["train/loss"]
    torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-8)


@pytest.mark.gpu
def test_trainer_on_hip_model(tmp_path):
    import creste_public_amd
    from creste_public_amd import MaxEntIRL
    creste_public_amd.set_precision("bf16x6")
    try:
        H, W, B = 64, 96, 2
        cfg = maxent_irl_cfg((H, W), solve_mdp=True)
        seed_everything(1337)
        model = MaxEntIRL(cfg)
        synth.randomize_bn(model, seed=1)
        with torch.no_grad():
            model.backbone.depthcomp.depthcomp.vision_backbone.model.trunk._bn0.running_var.fill_(1e7)
            model.traversability_head.r.postpool[0].norm.weight.mul_(0.01)
        model = model.cuda()
        tr = IRLTrainer(model, LossManager(cfg).cuda(), cfg)
        assert all(n.startswith("traversability_head.r.") for n, p in model.named_parameters() if p.requires_grad)
        rgbd, p2p = synth.make_frames(B, H, W, seed=5)
        rng = np.random.RandomState(0)
        cf = [dict(trajectories=(np.array([[100.0, 128.0]]) + np.linspace(0, 1, 20)[None, :, None] *
                                 rng.uniform(-80, 80, size=(2, 1, 2))).astype(np.float32), rank=np.array([0, 1])), None]
        batch = {"irl": {"image": rgbd.cuda(), "p2p": p2p.cuda(),
                         "traversability_label": synth.make_experts(B, 50, 256, seed=3).cuda(),
                         "fov_mask": torch.ones(B, 256, 256, dtype=torch.bool, device="cuda"),
                         "counterfactuals_label": cf}}
        l0 = tr.training_step(batch)
        l1 = tr.training_step(batch)
        assert torch.isfinite(l0["train/loss"]) and torch.isfinite(l1["train/loss"])
        assert not model.backbone.training and model.traversability_head.r.training
        path = str(tmp_path / "irl.ckpt")
        tr.save_checkpoint(path)
        fresh = MaxEntIRL(maxent_irl_cfg((H, W), solve_mdp=False))
        fresh.weights_path = path
        fresh.load_weights(path)                       # the mirrored reference loader (strips 'model.')
        fresh = fresh.cuda().eval()
        model.eval()
        model.solve_mdp = False
        with torch.no_grad():
            a = model((rgbd.cuda(), p2p.cuda()))["traversability_preds"]
            b = fresh((rgbd.cuda(), p2p.cuda()))["traversability_preds"]
        assert torch.equal(a, b)
    finally:
        creste_public_amd.set_precision("f32")


@pytest.mark.gpu
def test_irl_trainer_backbone_prefetch_equals_serial_steps():
    """IRLTrainer.training_step(batch, next_batch): the next batch's frozen-backbone forward runs on a side stream under
    this batch's trainable half.  The frozen half depends on no trainable parameter, so losses and parameters after three
    steps over DIFFERENT batches must equal the serial order bit for bit (reference train_traversability.py:66-105)."""
    import creste_public_amd
    from creste_public_amd import MaxEntIRL
    creste_public_amd.set_precision("bf16x6")
    try:
        H, W, B = 64, 96, 2
        cfg = maxent_irl_cfg((H, W), solve_mdp=True)

        def build():
            seed_everything(1337)
            m = MaxEntIRL(cfg)
            synth.randomize_bn(m, seed=1)
            with torch.no_grad():
                m.backbone.depthcomp.depthcomp.vision_backbone.model.trunk._bn0.running_var.fill_(1e7)
                m.traversability_head.r.postpool[0].norm.weight.mul_(0.01)
            m = m.cuda()
            return m, IRLTrainer(m, LossManager(cfg).cuda(), cfg)

        def batch(seed):
            rgbd, p2p = synth.make_frames(B, H, W, seed=seed)
            return {"irl": {"image": rgbd.cuda(), "p2p": p2p.cuda(),
                            "traversability_label": synth.make_experts(B, 50, 256, seed=seed + 1).cuda(),
                            "fov_mask": torch.ones(B, 256, 256, dtype=torch.bool, device="cuda"),
                            "counterfactuals_label": [None] * B}}
        batches = [batch(s) for s in (5, 15, 25)]
        m0, t0 = build()
        serial = [float(t0.training_step(b)["train/loss"]) for b in batches]
        m1, t1 = build()
        piped = [float(t1.training_step(b, next_batch=batches[i + 1] if i + 1 < len(batches) else None)["train/loss"])
                 for i, b in enumerate(batches)]
        torch.cuda.synchronize()
        assert m1._side_stream is not None, "the prefetch path did not run"
        assert serial == piped, (serial, piped)
        for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
            assert torch.equal(a, b), k
    finally:
        creste_public_amd.set_precision("f32")
