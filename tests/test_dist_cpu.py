"""CPU, world_size 2 over gloo: the N>1 paths -- frame sharding + max-over-ranks timing used by
bench.py, and the flat gradient all-reduce of the IRL step (reward net only, backbone frozen)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from creste_public_amd import dist_utils as du


def test_shard_range_partitions_frames():
    for total in (0, 1, 16, 17, 127):
        for world in (1, 2, 3, 8):
            spans = [du.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from creste_public_amd.config import maxent_irl_cfg
        from oracle.blocks import MultiScaleFCN            # CPU stand-in: the product net trains on HIP kernels only
        from creste_public_amd.creste.utils.loss_utils import LossManager
        cfg = maxent_irl_cfg()
        torch.manual_seed(0)                                  # identical replicas
        net = MultiScaleFCN(cfg["traversability_head"]["net_kwargs"]["reward_cfg"]["net_kwargs"]).train()
        frozen = torch.nn.Linear(4, 4)                        # stands in for the frozen backbone
        for p in frozen.parameters():
            p.requires_grad = False
        lm = LossManager(cfg)
        B, H, W = 4, 64, 128                                  # global batch 4 -> 2 samples per rank
        g = torch.Generator().manual_seed(1)
        feats = torch.randn(B, 40, H, W, generator=g)
        svf = torch.rand(B, H, W, generator=g)
        t = torch.linspace(0, 1, 50).view(1, 50, 1)
        xy = torch.tensor([[120.0, 128.0]]) + t * torch.tensor([[-100.0, 30.0]])
        expert = torch.eye(3).repeat(B, 50, 1, 1)
        expert[:, :, :2, 2] = xy
        fov = torch.ones(B, 256, 256, dtype=torch.bool)

        def grads(lo, hi):
            net.zero_grad()
            iv = feats[lo:hi].clone().requires_grad_(True)
            r = net(iv)
            td = {"outputs/exp_svf": svf[lo:hi].clone(), "outputs/traversability_preds": r,
                  "outputs/input_view": iv, "inputs/traversability_label": expert[lo:hi],
                  "inputs/fov_mask": fov[lo:hi], "inputs/counterfactuals_label": [None] * (hi - lo), "task": "x"}
            ld, _ = lm(td)
            sum(w * v for w, v in ld.values()).backward()

        lo, hi = du.shard_range(B, rank, world)
        grads(lo, hi)
        n = du.allreduce_mean_grads(list(net.parameters()) + list(frozen.parameters()))
        mine = torch.cat([p.grad.flatten() for p in net.parameters()])
        # every rank must hold the same averaged gradient, equal to the mean of the per-shard gradients
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        same = all(torch.equal(gathered[0], x) for x in gathered)
        shard_grads = []
        for r_ in range(world):
            a, b = du.shard_range(B, r_, world)
            grads(a, b)
            shard_grads.append(torch.cat([p.grad.flatten() for p in net.parameters()]))
        expect = torch.stack(shard_grads).mean(0)
        t_max = du.max_over_ranks(1.0 + rank)
        frames = du.sum_over_ranks(hi - lo)
        q.put((rank, n, same, float((mine - expect).abs().max()), float(expect.abs().max()), t_max, frames))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_irl_grad_allreduce_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n, same, err, scale, t_max, frames in res:
        assert n == 102866                      # the reward net's parameter count
        assert same
        assert err <= 1e-6 * max(scale, 1.0)
        assert t_max == 2.0 and frames == 4.0   # max-over-ranks time, whole-job frame count


def _arena_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(s)) for s in [(10,), (3, 4), (7,), (2, 2, 2), (5,), (33,)]]
        order = list(reversed(params))                     # the backward finishes the last layer first
        arena = du.GradArena(order, bucket_bytes=16 * 4)   # 16-element buckets -> several async all-reduces
        g = torch.Generator().manual_seed(100 + rank)
        mine = {}
        for i, p in enumerate(order):
            if i == 2 and rank == 1:                       # a parameter that received no gradient on this rank
                arena.done([p])
                mine[id(p)] = torch.zeros_like(p)
                continue
            v = torch.randn(p.shape, generator=g)
            arena.view(p).copy_(v)
            mine[id(p)] = v
            arena.done([p])
        n_async = len(arena.handles)
        arena.finish()
        # expected: mean over ranks of what every rank wrote
        flat_mine = torch.cat([mine[id(p)].flatten() for p in order])
        gathered = [torch.zeros_like(flat_mine) for _ in range(world)]
        dist.all_gather(gathered, flat_mine)
        expect = torch.stack(gathered).mean(0)
        got = torch.cat([arena[id(p)].flatten() for p in order])
        q.put((rank, float((got - expect).abs().max()), n_async, all(arena[id(p)].shape == p.shape for p in params),
               arena.flat.data_ptr() == arena[id(order[0])].data_ptr()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_grad_arena_bucketed_overlap():
    """dist_utils.GradArena: gradients written in backward order into one flat buffer, buckets all-reduced while
    later gradients are still being produced, averaged result identical on every rank."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_arena_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, n_async, shapes_ok, zero_copy in res:
        assert err < 1e-6 and shapes_ok and zero_copy
        assert n_async >= 2                    # buckets left before finish(): the exchange overlaps the backward


def test_grad_arena_without_process_group():
    params = [torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(2, 3))]
    a = du.GradArena(params)
    a.view(params[0]).fill_(2.0)
    a.done([params[0]])
    a.finish()                                 # params[1] never written -> zeros
    assert torch.equal(a[id(params[0])], torch.full((4,), 2.0)) and float(a[id(params[1])].abs().sum()) == 0.0


# ------------------------------------------------------------------------------------------------------------------
# SSC step, 2 ranks, UNEQUAL contrastive sample counts: padded all-gather (dist_utils.gather_varlen) + hook-driven
# bucketed all-reduce (dist_utils.HookedArena) + Adam, through harness.SSCTrainer and the product's LossManager.
class _StandInTerrainNet(torch.nn.Module):
    """CPU stand-in with TerrainNet's output contract for the SupPixelConLoss / CrossEntropy / SmoothL1 keys (the
    product network trains on HIP kernels only)."""

    def __init__(self):
        super().__init__()
        self.enc = torch.nn.Conv2d(4, 8, 3, padding=1)
        self.sam = torch.nn.Conv2d(8, 16, 1)
        self.dyn = torch.nn.Conv2d(8, 6, 1)
        self.elev = torch.nn.Conv2d(8, 2, 1)

    def forward(self, x):
        image = x[0]                      # (image, p2p, immovable mask | None), as train_ssc.py:103 passes it
        h = torch.relu(self.enc(image[:, 0]))
        return {"inpainting_sam_preds": self.sam(h), "inpainting_sam_dynamic_preds": self.dyn(h),
                "elevation_preds": self.elev(h)}


def _ssc_cfg():
    from creste_public_amd.config import Cfg
    return Cfg(optimizer=dict(name="Adam", beta1=0.9, beta2=0.999, lr=1e-2), lr_scheduler=dict(name="ExponentialLR", gamma=0.98),
               freeze_backbone_epochs=0,
               loss=[dict(name="SupPixelConLoss", views=1, weight=1.0, pred_key="outputs/inpainting_sam_preds",
                          lab_key="inputs/3d_sam_label", ignore_index=0, temperature=0.1, task="joint"),
                     dict(name="CrossEntropy", weight=2.0, pred_key="outputs/inpainting_sam_dynamic_preds",
                          lab_key="inputs/3d_sam_dynamic_label", num_class=6, class_dim=1, task="joint"),
                     dict(name="SmoothL1", weight=3.0, beta=0.2, pred_key="outputs/elevation_preds",
                          lab_key="inputs/elevation_label", absolute=False, task="joint")])


def _ssc_batch(rank):
    """rank 0: 3 frames with few labelled cells, rank 1: 2 frames with many -> different contrastive row counts"""
    g = torch.Generator().manual_seed(10 + rank)
    B, G = (3, 16) if rank == 0 else (2, 16)
    nlab = 3 if rank == 0 else 6
    sam = torch.randint(0, nlab, (B, 1, G // 4, G // 4), generator=g).repeat_interleave(4, 2).repeat_interleave(4, 3)
    dyn = torch.stack([torch.zeros(B, G, G), torch.randint(0, 6, (B, G, G), generator=g).float()], dim=1)
    fov = torch.rand(B, G, G, generator=g) > (0.6 if rank == 0 else 0.2)
    return {"joint": {"image": torch.randn(B, 1, 4, G, G, generator=g), "p2p": torch.eye(4).repeat(B, 1, 1, 1),
                      "3d_sam_label": sam, "3d_sam_dynamic_label": dyn, "fov_mask": fov,
                      "elevation_label": torch.randn(B, 2, G, G, generator=g)}}


def _ssc_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from creste_public_amd import harness
        from creste_public_amd.creste.utils.loss_utils import LossManager
        torch.manual_seed(0)
        model = _StandInTerrainNet()
        cfg = _ssc_cfg()
        tr = harness.SSCTrainer(model, LossManager(cfg), cfg, bucket_mb=0)      # 1-element buckets: many async sends
        tr.bucket_bytes = 256
        tr._rebuild_arena()
        rows = []
        import creste_public_amd.dist_utils as du2
        orig = du2.gather_varlen

        def spy(f, l):
            out = orig(f, l)
            rows.append((f.shape[0], out[0].shape[0], out[2]))
            return out
        du2.gather_varlen = spy
        torch.manual_seed(100 + rank)                        # the per-class sampling draws from the host RNG
        logs = tr.training_step(_ssc_batch(rank))
        logs2 = tr.training_step(_ssc_batch(rank))
        flat = torch.cat([p.detach().flatten() for p in model.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        q.put((rank, rows, all(torch.equal(gathered[0], x) for x in gathered), tr.arena.launched,
               float(logs["train/loss"]), float(logs2["train/loss"]),
               all(p.grad.data_ptr() == tr.arena.flat.data_ptr() + 4 * tr.arena.span[id(p)][0] for p in tr.arena.order)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_ssc_step_with_unequal_contrastive_counts():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ssc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, rows0, same0, launched0, l0a, l0b, views0), (r1, rows1, same1, launched1, l1a, l1b, views1) = res
    assert rows0[0][0] != rows1[0][0], "the two ranks must contribute different row counts for this test to bite"
    assert rows0[0][1] == rows1[0][1] == rows0[0][0] + rows1[0][0]          # everybody sees all rows
    assert rows0[0][2] == 0 and rows1[0][2] == rows0[0][0]                  # rank 1's rows follow rank 0's
    assert same0 and same1                                                  # replicas stay identical after two Adam steps
    assert launched0 >= 2 and launched1 >= 2                                # buckets left DURING the backward
    assert views0 and views1                                                # .grad are views of the flat buffer
    assert all(map(lambda v: v == v and abs(v) < 1e6, (l0a, l0b, l1a, l1b)))


def _varlen_grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from creste_public_amd.creste.utils.loss_utils import MultiPosConLoss
        g = torch.Generator().manual_seed(5)
        feats_all = torch.randn(11, 8, generator=g)
        labels_all = torch.randint(0, 3, (11,), generator=g)
        lo, hi = (0, 4) if rank == 0 else (4, 11)
        f = feats_all[lo:hi].clone().requires_grad_(True)
        loss = MultiPosConLoss(temperature=0.1)({"feats": f, "labels": labels_all[lo:hi]})["loss"]
        loss.backward()
        # single-process reference on the concatenated rows: this rank's rows of the [11 x 11] problem
        fr = feats_all.clone().requires_grad_(True)
        fn = torch.nn.functional.normalize(fr, dim=-1, p=2)
        logits = fn[lo:hi] @ fn.T / 0.1
        lm = torch.ones(hi - lo, 11)
        lm[torch.arange(hi - lo), torch.arange(lo, hi)] = 0
        mask = (labels_all[lo:hi].view(-1, 1) == labels_all.view(1, -1)).float() * lm
        logits = logits - (1 - lm) * 1e9
        logits = logits - logits.max(dim=-1, keepdim=True)[0].detach()
        pmat = mask / mask.sum(1, keepdim=True).clamp(min=1.0)
        ref = -(pmat * torch.log_softmax(logits, dim=-1)).sum(-1).mean()
        q.put((rank, float(loss), float(ref)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_varlen_contrastive_loss_value():
    """4 rows on rank 0, 7 on rank 1: each rank's loss equals its rows of the single-process 11 x 11 problem (the
    reference's own formula, supcon_loss.py:56-115, with the row offset generalised from n * rank)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_varlen_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, loss, ref in res:
        assert abs(loss - ref) < 1e-5 * max(1.0, abs(ref)), (rank, loss, ref)


# ------------------------------------------------------------------------------------------------------------------
# bench.py's data-parallel training legs: dist_utils.measure_dp_step (with / without the exchange, max over ranks,
# payload accounting) on the SSC stand-in, 2 ranks.
def _dp_measure_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        from creste_public_amd import harness
        from creste_public_amd.creste.utils.loss_utils import LossManager
        torch.manual_seed(0)
        model = _StandInTerrainNet()
        cfg = _ssc_cfg()
        tr = harness.SSCTrainer(model, LossManager(cfg), cfg)
        batch = _ssc_batch(rank)
        nparam = sum(p.numel() for p in model.parameters())
        calls = []

        def step():
            torch.manual_seed(100 + rank)
            calls.append(du.is_dist())
            if rank == 1:
                time.sleep(0.02)                      # the slower rank sets the job's step time
            tr.training_step(batch)
        res = du.measure_dp_step(step, steps=3, frames_per_rank=batch["joint"]["image"].shape[0], warmup=1)
        # after the measurement the collectives are back on and a step re-synchronises nothing by itself: the replicas
        # diverged under collectives(False) (documented) -- but the group still works
        assert du.is_dist()
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        q.put((rank, res, nparam, calls, float(t)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_measure_dp_step_bookkeeping():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_measure_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, r0, nparam, calls0, s0), (_, r1, _, calls1, s1) = res
    assert s0 == s1 == 1.0
    # 1 warm-up + 3 timed steps with the exchange, then 1 + 3 without
    assert calls0 == calls1 == [True] * 4 + [False] * 4
    for r in (r0, r1):
        assert r["world"] == 2 and r["step_ms"] >= 20.0 and r["step_ms_no_collective"] >= 20.0      # max over ranks: rank 1 sleeps
        assert abs(r["allreduce_exposed_ms"] - (r["step_ms"] - r["step_ms_no_collective"])) < 2e-3
        assert r["frames_per_s"] == round(2 * r["frames_per_rank"] / r["step_ms"] * 1e3, 2)
        # payload per step: every gradient once (fp32) + the contrastive loss's padded feature / label all-gather
        assert r["allreduce_bytes"] >= 4 * nparam and r["collective_calls"] >= 3
    assert r0["step_ms"] == r1["step_ms"] and r0["step_ms_no_collective"] == r1["step_ms_no_collective"]
    # one process: the same function is a plain timer
    one = du.measure_dp_step(lambda: None, steps=2, frames_per_rank=8)
    assert one["world"] == 1 and one["allreduce_bytes"] == 0 and one["collective_calls"] == 0


def test_bench_launches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus 2` with no RANK in the environment (how the driver calls it) must become two ranks under
    torch.distributed.run and print ONE JSON line from rank 0.  `--dry-run`: gloo, a step is a sleep --
    the launcher, rendezvous, barrier / max-over-ranks bookkeeping and exit code, nothing else."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["launched_by"] == "self" and line["data"] == "dry-run"
    # the slowest rank's step (2 ms x 2) bounds the step time: max over ranks, not rank 0's own clock
    assert line["ms_per_step"] >= 4.0
    assert abs(line["value"] - 16 * 2 * 4 / (line["ms_per_step"] * 4e-3)) < 0.05 * line["value"]
    # a failing rank must fail the command (WORLD_SIZE / --gpus disagreement is checked by every rank)
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


def test_bench_dead_rank_yields_a_line_with_an_error_field():
    """a rank that dies must not leave the job hanging at a barrier or the caller without a line.
    (a) self-launched: rank 1 exits before its first barrier -> torch.distributed.run tears the job down, and whichever of
    rank 0's SIGTERM handler / the launching parent gets there first prints ONE JSON line with `error`, rc != 0;
    (b) the in-rank deadline alone (the driver's own torch.distributed.run at N > 1, no parent of ours): rank 0 waits at a
    barrier its peer never reaches and prints the error line when CRESTE_BENCH_BARRIER_TIMEOUT_S passes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CRESTE_BENCH_TEST_DIE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and lines[0]["value"] is None and lines[0]["error"] and lines[0]["n_gpus"] == 2, r.stdout + r.stderr[-1500:]
    # (b) rank 0 alone in a 2-rank world whose rank 1 never shows up at the barrier: the rendezvous itself needs both, so rank 1
    # joins and then sleeps (CRESTE_BENCH_TEST_DIE=sleep:1)
    env["CRESTE_BENCH_TEST_DIE"] = "sleep:1"
    env["CRESTE_BENCH_BARRIER_TIMEOUT_S"] = "5"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--dry-run"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "did not complete within 5 s" in lines[0]["error"], r.stdout + r.stderr[-1500:]
