"""GPU stress tests of everything that runs on MORE THAN ONE STREAM: many back-to-back
steps with no host synchronisation between them, allocator churn on the caller's stream, each compared bit for bit with its
one-stream twin.

* pipelined inference (ops.forward_in_parts): >= 60 full-size steps, with the shared output buffers preallocated before the
  fork (the default) and allocated in mid-forward behind an event (first call / changed shapes);
* the IRL step with the next batch's frozen half prefetched on a side stream (MaxEntIRL.prefetch_backbone): >= 60 steps;
* the BEV-SSC step with conv weight gradients on the side stream (ops.wgrad_stream): >= 30 steps.

What these guard: (a) the cross-stream reuse of allocator blocks (a shared buffer that part 0 allocates in mid-forward may be
a block that kernels still queued on part 0's stream are using; the other part's stream must not write it before that point
of part 0's stream -- r04 wrote it after the fork only); (b) the packed-fp32 corruption of r04 (creste_public_amd/build.py
NO_PK): VALU kernels of one stream sharing CUs with MFMA kernels of another."""
import pytest
import torch

import creste_public_amd
from creste_public_amd import MaxEntIRL, ops, synth
from creste_public_amd.config import maxent_irl_cfg

pytestmark = pytest.mark.gpu

KEYS = ("depth_preds_feats", "depth_preds_logits", "bev_features", "bev_densities", "elevation_features",
        "elevation_preds", "traversability_preds")


def _ssc_batch(B, H, W, seed=0, G=256):
    rgbd, p2p = synth.make_frames(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    Hs, Ws = H // 4, W // 4
    blocks = torch.randint(0, 5, (B, 1, G // 16, G // 16), generator=g)        # piecewise-constant SAM segments
    data = {"image": rgbd, "p2p": p2p,
            "depth_label": torch.rand(B, 1, Hs, Ws, generator=g) * 26000.0,
            "fimg_label": torch.randn(B, 1, 128, Hs, Ws, generator=g),
            "3d_sam_label": blocks.repeat_interleave(16, 2).repeat_interleave(16, 3),
            "3d_sam_dynamic_label": torch.stack([torch.zeros(B, G, G),
                                                 torch.randint(0, 6, (B, G // 8, G // 8), generator=g).float()
                                                 .repeat_interleave(8, 1).repeat_interleave(8, 2)], dim=1),
            "fov_mask": torch.rand(B, G, G, generator=g) > 0.5,
            "elevation_label": torch.randn(B, 2, G, G, generator=g)}
    return {"joint": {k: v.cuda() for k, v in data.items()}}


def _churn(gen, held):
    """allocate / free blocks of odd sizes on the caller's stream between steps: the caching allocator then hands the next
    step recycled blocks whose previous users may still be queued"""
    for _ in range(4):
        n = int(torch.randint(1 << 18, 1 << 26, (1,), generator=gen))
        held.append(torch.empty(n, dtype=torch.float32, device="cuda").fill_(1.0))
    while len(held) > 6:
        held.pop(int(torch.randint(0, len(held), (1,), generator=gen)))


@pytest.mark.parametrize("prealloc", [True, False])
def test_sixty_pipelined_full_size_steps_equal_the_one_stream_twin(prealloc, monkeypatch):
    Hf, Wf, B, STEPS = 608, 1216, 16, 60
    monkeypatch.setattr(ops, "PREALLOCATE_SHARED", prealloc)
    torch.manual_seed(3)
    creste_public_amd.set_precision("bf16x6")
    model = MaxEntIRL(maxent_irl_cfg((Hf, Wf), solve_mdp=False))
    synth.randomize_bn(model, seed=3)
    model = model.cuda().eval()
    batches = []
    for seed in (21, 22):
        rgbd, p2p = synth.make_frames(B, Hf, Wf, seed=seed)
        batches.append((rgbd.cuda(), p2p.cuda()))
    synth.calibrate_bn_hip(model, batches[0][0][:2], batches[0][1][:2])
    n = B // 2
    refs = []
    with torch.no_grad():
        model.inference_parts = 0                       # the twin: the two halves as plain forwards on ONE stream
        for rgbd, p2p in batches:
            halves = [model((rgbd[i * n:(i + 1) * n].contiguous(), p2p[i * n:(i + 1) * n].contiguous())) for i in range(2)]
            refs.append({k: torch.cat([h[k] for h in halves]) for k in KEYS})
        del model.inference_parts
        assert model._parts_for(B) == 2
        torch.cuda.synchronize()
        gen, held, flags = torch.Generator().manual_seed(0), [], []
        for s in range(STEPS):                          # no host synchronisation in here
            rgbd, p2p = batches[s % 2]
            out = model((rgbd, p2p))
            flags.append(torch.stack([(out[k] != refs[s % 2][k]).reshape(B, -1).any(1) for k in KEYS]))
            del out
            _churn(gen, held)
        torch.cuda.synchronize()
    bad = [(s, KEYS[j], torch.nonzero(f[j]).flatten().tolist()) for s, f in enumerate(torch.stack(flags).cpu()) for j in range(len(KEYS))
           if f[j].any()]
    assert not bad, f"{len(bad)} (step, key, frames) differ from the one-stream twin: {bad[:6]}"


def test_sixty_irl_steps_with_prefetched_backbone_equal_serial_steps():
    """IRLTrainer.training_step(batch, next_batch) x 60 over three alternating batches at 256 x 384: losses, the reward
    network's parameters and Adam's moments equal the serial trainer's bit for bit (the persistent MDP solver, the reward
    net's graph replays and the loss run beside the side stream's full-chip backbone kernels)."""
    from creste_public_amd.creste.utils.loss_utils import LossManager
    from creste_public_amd.harness import IRLTrainer, seed_everything
    creste_public_amd.set_precision("bf16x6")
    try:
        H, W, B, STEPS = 256, 384, 4, 60
        cfg = maxent_irl_cfg((H, W), solve_mdp=True)

        def build():
            seed_everything(1337)
            m = MaxEntIRL(cfg)
            synth.randomize_bn(m, seed=1)
            with torch.no_grad():
                m.traversability_head.r.postpool[0].norm.weight.mul_(0.01)
                m.traversability_head.r.postpool[0].norm.bias.mul_(0.01)
            m = m.cuda()
            return m, IRLTrainer(m, LossManager(cfg).cuda(), cfg, graphs=True)

        def batch(seed):
            rgbd, p2p = synth.make_frames(B, H, W, seed=seed)
            return {"irl": {"image": rgbd.cuda(), "p2p": p2p.cuda(),
                            "traversability_label": synth.make_experts(B, 50, 256, seed=seed + 1).cuda(),
                            "fov_mask": torch.ones(B, 256, 256, dtype=torch.bool, device="cuda"),
                            "counterfactuals_label": [None] * B}}
        batches = [batch(s) for s in (5, 15, 25)]
        seq = [batches[i % 3] for i in range(STEPS)]
        m0, t0 = build()
        synth.calibrate_bn_hip(m0, batches[0]["irl"]["image"][:2], batches[0]["irl"]["p2p"][:2])
        serial = torch.stack([t0.training_step(b)["train/loss"] for b in seq])
        m1, t1 = build()
        synth.calibrate_bn_hip(m1, batches[0]["irl"]["image"][:2], batches[0]["irl"]["p2p"][:2])
        piped = torch.stack([t1.training_step(b, next_batch=seq[i + 1] if i + 1 < STEPS else None)["train/loss"]
                             for i, b in enumerate(seq)])
        torch.cuda.synchronize()
        assert m1._side_stream is not None, "the prefetch path did not run"
        assert t0.vi_retries == 0 and t1.vi_retries == 0
        assert torch.isfinite(serial).all()
        assert torch.equal(serial, piped), torch.nonzero(serial != piped).flatten().tolist()
        for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
            assert torch.equal(a, b), k
    finally:
        creste_public_amd.set_precision("f32")


def test_thirty_ssc_steps_with_side_stream_weight_gradients_equal_one_stream_steps(monkeypatch):
    """SSCTrainer x 30 on 256 x 384 frames (unfrozen backbone, all six losses, Adam): every parameter after the last step
    equals the same steps with the weight gradients on the backward's own stream."""
    from creste_public_amd import harness, train_backbone
    from creste_public_amd.creste.models.terrainnet import TerrainNet
    from creste_public_amd.creste.utils.loss_utils import LossManager
    creste_public_amd.set_precision("bf16x6")
    try:
        H, W, B, STEPS = 256, 384, 4, 30
        cfg = harness.ssc_cfg((H, W), class_weights=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05], freeze_backbone_epochs=0)
        batches = [_ssc_batch(B, H, W, seed=s) for s in (0, 7)]

        def run(side):
            monkeypatch.setattr(ops, "WGRAD_STREAM", side)
            monkeypatch.setattr(train_backbone, "WGRAD_STREAM_MIN", 0)
            harness.seed_everything(5)
            model = TerrainNet(cfg).cuda()
            synth.randomize_bn(model, seed=2)
            tr = harness.SSCTrainer(model, LossManager(cfg).cuda(), cfg)
            tr.on_train_epoch_start()
            losses = torch.stack([tr.training_step(batches[i % 2])["train/loss"] for i in range(STEPS)])
            torch.cuda.synchronize()
            return losses, {k: v.clone() for k, v in model.state_dict().items()}

        l0, sd0 = run(False)
        l1, sd1 = run(True)
        assert any(s and s[0] is not None for s in ops._wgrad_streams.values()), "the side stream was never used"
        assert torch.isfinite(l0).all()
        assert torch.equal(l0, l1), torch.nonzero(l0 != l1).flatten().tolist()
        bad = [k for k in sd0 if not torch.equal(sd0[k], sd1[k])]
        assert not bad, bad[:10]
    finally:
        creste_public_amd.set_precision("f32")


def test_ssc_step_in_f16x3_with_side_stream_weight_gradients(monkeypatch):
    """f16x3 operands + weight gradients on the side stream: the |max| bounds of x / gy are made on the backward's stream
    before the side stream is entered -- only for the convs whose weight gradient reads them (the 2- and 6-class
    projections have gradients with Cout % 4 != 0 and take the exact-fp32 kernel: no bound, no aligned slice to scan)."""
    from creste_public_amd import harness, train_backbone
    from creste_public_amd.creste.models.terrainnet import TerrainNet
    from creste_public_amd.creste.utils.loss_utils import LossManager
    creste_public_amd.set_precision("f16x3")
    try:
        H, W, B = 128, 192, 2
        cfg = harness.ssc_cfg((H, W), class_weights=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05], freeze_backbone_epochs=0)
        batch = _ssc_batch(B, H, W, seed=3)

        def run(side):
            monkeypatch.setattr(ops, "WGRAD_STREAM", side)
            monkeypatch.setattr(train_backbone, "WGRAD_STREAM_MIN", 0)
            harness.seed_everything(5)
            model = TerrainNet(cfg).cuda()
            synth.randomize_bn(model, seed=2)
            tr = harness.SSCTrainer(model, LossManager(cfg).cuda(), cfg)
            losses = torch.stack([tr.training_step(batch)["train/loss"] for _ in range(3)])
            torch.cuda.synchronize()
            return losses, {k: v.clone() for k, v in model.state_dict().items()}

        l0, sd0 = run(False)
        l1, sd1 = run(True)
        assert torch.isfinite(l0).all() and torch.equal(l0, l1)
        assert not [k for k in sd0 if not torch.equal(sd0[k], sd1[k])]
    finally:
        creste_public_amd.set_precision("f32")
