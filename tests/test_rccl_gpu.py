"""GPU: every collective call site of DESIGN section 6 on the REAL `nccl` backend (= RCCL on ROCm), world size 1.

The GPU box has one device, so the scaling curve is the driver's to measure; what a one-rank group does prove is
that RCCL initialises in this image, that the async bucket all-reduces of `GradArena` / `HookedArena`, the
differentiable variable-count all-gather of the contrastive loss (`gather_varlen`; reference
creste/models/losses/supcon_loss.py:43-53,85-86), the flat gradient all-reduce of the IRL step
(`allreduce_mean_grads`; reference train_traversability.py:400-416 DDP) and the bench's max-over-ranks reduction run
on communicator streams against the HIP engines' buffers, and that a full `DistillTrainer` / `SSCTrainer` step
(reference train_pefree.py:261-288, train_ssc.py:342-358) completes under it with the same result as without a
process group.  The child writes a log that `profiles/` keeps (gpurun_out/rccl_1rank.log)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
H, W = 64, 96


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _child(port, q):
    import io
    import torch.distributed as dist
    log = io.StringIO()

    def say(*a):
        print(*a, file=log, flush=True)

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        torch.cuda.set_device(0)
        import creste_public_amd
        from creste_public_amd import dist_utils, harness, synth
        from creste_public_amd.creste.utils.loss_utils import LossManager
        creste_public_amd.set_precision("f32")

        def distill(seed):
            from creste_public_amd.creste.models.distillation import DistillationBackbone
            harness.seed_everything(seed)
            cfg = harness.distillation_cfg((H, W))
            m = DistillationBackbone(cfg).cuda()
            synth.randomize_bn(m, seed=4)
            return m, harness.DistillTrainer(m, LossManager(cfg), cfg, bucket_mb=1)

        def ssc(seed):
            from creste_public_amd.creste.models.terrainnet import TerrainNet
            harness.seed_everything(seed)
            cfg = harness.ssc_cfg((H, W), class_weights=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05])
            m = TerrainNet(cfg).cuda()
            synth.randomize_bn(m, seed=4)
            synth.peak_depth_head(m)
            return m, harness.SSCTrainer(m, LossManager(cfg).cuda(), cfg, bucket_mb=1)

        from test_dist_gpu import _ssc_batch
        b = _ssc_batch(2, seed=10)
        batches = {"distill": {k: b[k] for k in ("image", "depth_label", "fimg_label")}, "ssc": {"joint": b}}

        def flat(m):
            return torch.cat([p.detach().flatten().float() for p in m.parameters() if p.requires_grad])

        # --- stand-alone reference results (no process group yet)
        alone = {}
        for kind, mk in (("distill", distill), ("ssc", ssc)):
            m, tr = mk(3)
            torch.manual_seed(50)
            logs = tr.training_step(batches[kind])
            alone[kind] = (flat(m).cpu(), float(logs["train/loss"]))
            del m, tr
        torch.cuda.empty_cache()

        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        say(f"backend {dist.get_backend()} world {dist.get_world_size()} torch {torch.__version__} "
            f"hip {torch.version.hip} nccl(RCCL) {'.'.join(map(str, torch.cuda.nccl.version()))} "
            f"device {torch.cuda.get_device_name(0)}")
        res = {}

        # --- GradArena: buckets leave as async all-reduces while `done` is still being called
        ps = [torch.nn.Parameter(torch.randn(300_000, device="cuda")) for _ in range(4)]
        arena = dist_utils.GradArena(ps, bucket_bytes=1 << 20)
        want = []
        for i, p in enumerate(ps):
            g = torch.full_like(p, float(i + 1))
            arena.view(p).copy_(g)
            want.append(g)
            arena.done([p])
        early = len(arena.handles)
        arena.finish()
        torch.cuda.synchronize()
        res["grad_arena_async_buckets"] = early
        res["grad_arena_ok"] = all(torch.equal(arena[id(p)], g) for p, g in zip(ps, want))
        say(f"GradArena: {len(arena.handles)} all-reduces ({early} before finish()), 4 x 300k fp32, mean preserved: "
            f"{res['grad_arena_ok']}")

        # --- HookedArena driven by torch autograd
        lin = torch.nn.Sequential(torch.nn.Linear(512, 512), torch.nn.Linear(512, 512), torch.nn.Linear(512, 512)).cuda()
        hk = dist_utils.HookedArena(lin.parameters(), bucket_bytes=1 << 20)
        x = torch.randn(8, 512, device="cuda")
        lin(x).square().sum().backward()
        hk.finish()
        torch.cuda.synchronize()
        gref = torch.autograd.grad(lin(x).square().sum(), list(lin.parameters()))
        res["hooked_arena_async_buckets"] = hk.launched
        res["hooked_arena_ok"] = all(torch.allclose(p.grad, g, rtol=1e-5, atol=1e-5) for p, g in zip(lin.parameters(), gref))
        hk.close()
        say(f"HookedArena: {hk.launched} all-reduces during the backward, gradients equal plain autograd: "
            f"{res['hooked_arena_ok']}")

        # --- differentiable variable-count all-gather (contrastive loss) + its backward through RCCL
        f = torch.randn(37, 32, device="cuda", requires_grad=True)
        lab = torch.arange(37, device="cuda")
        af, al, off = dist_utils.gather_varlen(f, lab)
        (af * 2.0).sum().backward()
        res["gather_varlen_ok"] = bool(torch.equal(af.detach(), f.detach()) and torch.equal(al, lab) and off == 0 and
                                       torch.equal(f.grad, torch.full_like(f, 2.0)))
        say(f"gather_varlen: [37,32] rows gathered with gradient: {res['gather_varlen_ok']}")

        # --- flat gradient all-reduce of the IRL step + the bench's reductions
        qs = [torch.nn.Parameter(torch.randn(1000, device="cuda")) for _ in range(3)]
        for i, p in enumerate(qs[:2]):
            p.grad = torch.full_like(p, float(i + 1))
        n = dist_utils.allreduce_mean_grads(qs)
        res["flat_allreduce_ok"] = bool(n == 3000 and torch.equal(qs[0].grad, torch.full_like(qs[0], 1.0)) and
                                        torch.equal(qs[2].grad, torch.zeros_like(qs[2])))
        res["max_over_ranks"] = dist_utils.max_over_ranks(1.25, torch.device("cuda", 0))
        dist.barrier()
        say(f"allreduce_mean_grads: {n} elements in one call: {res['flat_allreduce_ok']}; max_over_ranks(1.25) = "
            f"{res['max_over_ranks']}")

        # --- full trainer steps on the HIP engines under the nccl group == stand-alone
        for kind, mk in (("distill", distill), ("ssc", ssc)):
            m, tr = mk(3)
            torch.manual_seed(50)
            logs = tr.training_step(batches[kind])
            torch.cuda.synchronize()
            got = flat(m).cpu()
            ref, ref_loss = alone[kind]
            launched = getattr(getattr(tr, "arena", None), "launched", None)
            res[kind] = (float((got - ref).abs().max()), float(ref.abs().max()), float(logs["train/loss"]), ref_loss)
            say(f"{kind} step under nccl: loss {float(logs['train/loss']):.6f} (stand-alone {ref_loss:.6f}), max |param "
                f"difference| {res[kind][0]:.3e}" + (f", {launched} bucket all-reduces during the backward" if launched is not None else ""))
            del m, tr
        # --- bench.py's data-parallel leg bookkeeping on a real step under the group (dist_utils.measure_dp_step)
        m, tr = distill(3)
        torch.manual_seed(50)
        rec = dist_utils.measure_dp_step(lambda: tr.training_step(batches["distill"]), steps=2, frames_per_rank=2,
                                         device=torch.device("cuda", 0), warmup=1)
        nparam = sum(p.numel() for p in m.parameters())
        res["dp_leg"] = rec
        res["dp_leg_ok"] = bool(rec["world"] == 1 and 0 < rec["allreduce_bytes"] <= 4 * nparam and rec["collective_calls"] >= 1 and
                                rec["step_ms"] > 0 and rec["step_ms_no_collective"] > 0 and dist_utils.is_dist())
        say(f"measure_dp_step (distillation step): {rec}")
        del m, tr
        dist.barrier()
        dist.destroy_process_group()
        say("process group destroyed cleanly")
        q.put(("ok", res, log.getvalue()))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(("error", repr(e) + "\n" + traceback.format_exc(), log.getvalue()))


def test_rccl_one_rank_runs_every_collective_call_site():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_child, args=(_free_port(), q))
    p.start()
    status, res, log = q.get(timeout=900)
    p.join(timeout=120)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "rccl_1rank.log"), "w") as f:
        f.write(log + ("" if status == "ok" else "\nFAILED: " + str(res)))
    assert status == "ok", res
    assert p.exitcode == 0
    assert res["dp_leg_ok"], res["dp_leg"]
    assert res["grad_arena_ok"] and res["grad_arena_async_buckets"] >= 1
    assert res["hooked_arena_ok"] and res["hooked_arena_async_buckets"] >= 1
    assert res["gather_varlen_ok"] and res["flat_allreduce_ok"] and res["max_over_ranks"] == 1.25
    for kind in ("distill", "ssc"):
        diff, scale, loss, ref_loss = res[kind]
        assert diff <= 1e-6 * max(scale, 1.0), (kind, diff, scale)       # world 1: sum / 1 changes nothing
        assert abs(loss - ref_loss) <= 1e-6 * max(abs(ref_loss), 1.0)
