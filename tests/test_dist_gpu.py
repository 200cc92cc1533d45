"""GPU, world_size 2 on ONE device over gloo (CUDA tensors): the data-parallel training steps with the REAL HIP engines
under a process group -- the encoder's GradArena (bucketed all-reduce inside BackboneFn.backward), the hook-driven
HookedArena of the BEV-SSC step and the variable-count contrastive all-gather with CUDA tensors.  (RCCL cannot put two
ranks on one device; the driver's multi-GPU run exercises the nccl backend.)  Each rank trains on its own frames; after
the step every replica must hold the same parameters, and the averaged gradient must equal the mean of the gradients
the ranks compute on their own (checked through the first Adam step: identical moments -> identical update)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
H, W = 64, 96


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ssc_batch(B, seed):
    from creste_public_amd import synth
    rgbd, p2p = synth.make_frames(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    G, Hs, Ws = 256, H // 4, W // 4
    nlab = 3 + seed % 3
    data = {"image": rgbd, "p2p": p2p, "depth_label": torch.rand(B, 1, Hs, Ws, generator=g) * 26000.0,
            "fimg_label": torch.randn(B, 1, 128, Hs, Ws, generator=g),
            "3d_sam_label": torch.randint(0, nlab, (B, 1, G // 32, G // 32), generator=g).repeat_interleave(32, 2).repeat_interleave(32, 3),
            "3d_sam_dynamic_label": torch.stack([torch.zeros(B, G, G), torch.randint(0, 6, (B, G // 8, G // 8), generator=g)
                                                 .float().repeat_interleave(8, 1).repeat_interleave(8, 2)], dim=1),
            "fov_mask": torch.rand(B, G, G, generator=g) > (0.3 + 0.2 * (seed % 2)), "elevation_label": torch.randn(B, 2, G, G, generator=g)}
    return {k: v.cuda() for k, v in data.items()}


def _worker(rank, world, port, kind, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    try:
        import creste_public_amd
        from creste_public_amd import harness, synth
        from creste_public_amd.creste.utils.loss_utils import LossManager
        creste_public_amd.set_precision("f32")

        def build():
            harness.seed_everything(3)                      # identical replicas
            if kind == "distill":
                from creste_public_amd.creste.models.distillation import DistillationBackbone
                cfg = harness.distillation_cfg((H, W))
                m = DistillationBackbone(cfg).cuda()
                synth.randomize_bn(m, seed=4)
                return m, cfg, harness.DistillTrainer(m, LossManager(cfg), cfg, bucket_mb=1)
            from creste_public_amd.creste.models.terrainnet import TerrainNet
            cfg = harness.ssc_cfg((H, W), class_weights=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05])
            m = TerrainNet(cfg).cuda()
            synth.randomize_bn(m, seed=4)
            synth.peak_depth_head(m)
            return m, cfg, harness.SSCTrainer(m, LossManager(cfg).cuda(), cfg, bucket_mb=1)

        def batch_of(r):
            b = _ssc_batch(1 + r, seed=10 + r)               # rank 0: one frame, rank 1: two frames
            if kind == "distill":
                return {k: b[k] for k in ("image", "depth_label", "fimg_label")}
            return {"joint": b}

        def flat_params(m):
            return torch.cat([p.detach().flatten().float().cpu() for p in m.parameters() if p.requires_grad])

        # 1. every rank alone (no process group yet): the gradient of its own batch, read back through Adam's first step
        #    is awkward -- take the gradients directly
        m0, _, tr0 = build()
        torch.manual_seed(50 + rank)
        tr0.optimizer.step = lambda *a, **k: None           # keep the parameters; only the gradients are wanted
        tr0.training_step(batch_of(rank))
        own = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().flatten().float().cpu()
                         for p in m0.parameters() if p.requires_grad])
        del m0, tr0
        torch.cuda.empty_cache()
        dist.init_process_group("gloo", rank=rank, world_size=world)
        m1, _, tr1 = build()
        torch.manual_seed(50 + rank)
        real_step = tr1.optimizer.step
        grads = {}

        def spy_step(*a, **k):
            grads["g"] = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().flatten().float().cpu()
                                    for p in m1.parameters() if p.requires_grad])
            return real_step(*a, **k)
        tr1.optimizer.step = spy_step
        logs = tr1.training_step(batch_of(rank))
        gathered = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(gathered, own)
        expect = torch.stack(gathered).mean(0)
        after = flat_params(m1)
        pa = [torch.zeros_like(after) for _ in range(world)]
        dist.all_gather(pa, after)
        scale = float(expect.abs().max())
        err = float((grads["g"] - expect).abs().max())
        launched = getattr(getattr(tr1, "arena", None), "launched", -1)
        q.put((rank, kind, err, scale, all(torch.equal(pa[0], x) for x in pa), float(logs["train/loss"]), launched))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["distill", "ssc"])
def test_two_ranks_one_device_data_parallel_step(kind):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, k, err, scale, same, loss, launched in res:
        assert same, f"{k}: replicas diverged after one data-parallel step"
        assert loss == loss
        if k == "distill":
            # (the contrastive loss of the SSC step couples the ranks: its gradient is NOT the mean of stand-alone runs)
            assert err <= 1e-5 * max(scale, 1e-6), (k, rank, err, scale)
        else:
            assert launched >= 1, "the heads' bucket must leave during the backward"
