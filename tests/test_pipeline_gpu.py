"""GPU: the pipelined inference forward (MaxEntIRL._frozen_parts: one batch as `inference_parts` forwards on as many
streams, the parts writing their rows of shared whole-batch output buffers -- ops.PartContext).

The contract: every output key of the pipelined forward of a batch is BIT-IDENTICAL to the concatenation of the plain
forwards of its parts (each part IS the plain forward of its frames; nothing is concatenated or re-computed), whatever the
streams do.  (A part's forward differs from the same rows of a whole-batch forward at float-noise level already without
any pipelining: the squeeze-excite pooling partitions its partial sums by the launch's batch size.)"""
import pytest
import torch

import creste_public_amd
from creste_public_amd import MaxEntIRL, ops, synth
from creste_public_amd.config import maxent_irl_cfg

pytestmark = pytest.mark.gpu

H, W = 128, 192


def _model(solve_mdp=False, seed=7):
    torch.manual_seed(seed)
    creste_public_amd.set_precision("bf16x6")
    model = MaxEntIRL(maxent_irl_cfg((H, W), solve_mdp=solve_mdp))
    synth.randomize_bn(model, seed=seed)
    model = model.cuda().eval()
    rgbd, p2p = synth.make_frames(2, H, W, seed=seed + 1)
    synth.calibrate_bn_hip(model, rgbd.cuda(), p2p.cuda())
    model.inference_part_rows = 4          # (the default, 6, is a speed threshold measured at 1216x608: these are correctness tests)
    return model


def _plain(model, inputs):
    model.inference_parts = 0
    try:
        with torch.no_grad():
            return {k: v.clone() for k, v in model(inputs).items()}
    finally:
        del model.inference_parts


@pytest.mark.parametrize("B,parts", [(8, 2), (12, 3), (16, 2)])
def test_pipelined_forward_is_the_parts_forwards(B, parts):
    model = _model()
    rgbd, p2p = synth.make_frames(B, H, W, seed=1234)
    rgbd, p2p = rgbd.cuda(), p2p.cuda()
    n = B // parts
    want = [_plain(model, (rgbd[i * n:(i + 1) * n].contiguous(), p2p[i * n:(i + 1) * n].contiguous())) for i in range(parts)]
    model.inference_parts = parts
    with torch.no_grad():
        assert model._parts_for(B) == parts
    for rep in range(3):                                   # repeated: the allocator hands the parts recycled blocks
        with torch.no_grad():
            got = model((rgbd, p2p))
        torch.cuda.synchronize()
        assert set(got) == set(want[0])
        for k, v in got.items():
            ref = torch.cat([w[k] for w in want])
            assert v.shape == ref.shape and v.dtype == ref.dtype, k
            assert torch.equal(v, ref), f"{k} (repeat {rep})"
    assert float((got["bev_densities"] > 0).float().mean()) > 0.01
    # the big outputs are views of the shared buffers (no concatenation happened): one storage for all rows
    for k in ("depth_preds_logits", "depth_preds_feats", "bev_features", "elevation_features", "bev_coords"):
        assert got[k].untyped_storage().nbytes() >= got[k].numel() * got[k].element_size(), k
        assert got[k][:n].untyped_storage().data_ptr() == got[k][n:].untyped_storage().data_ptr(), k


def test_first_pipelined_forward_builds_the_caches_for_every_part():
    """packed weights / folded BatchNorm are (re)built lazily by launches on the stream of the first forward that misses them --
    part 0's: the other parts' streams must not read them before those launches ran."""
    model = _model()
    rgbd, p2p = synth.make_frames(8, H, W, seed=5)
    rgbd, p2p = rgbd.cuda(), p2p.cuda()
    for seed in (11, 12):
        synth.randomize_bn(model, seed=seed)                     # new BatchNorm statistics: every folded cache is stale
        with torch.no_grad():
            assert model._parts_for(8) == 2
            got = {k: v.clone() for k, v in model((rgbd, p2p)).items()}
        want = [_plain(model, (rgbd[i * 4:(i + 1) * 4].contiguous(), p2p[i * 4:(i + 1) * 4].contiguous())) for i in range(2)]
        for k, v in got.items():
            assert torch.equal(v, torch.cat([w[k] for w in want])), k


def test_pipelining_is_off_where_it_must_be():
    model = _model()
    assert model._parts_for(2) == 1 and model._parts_for(7) == 1           # too small / not divisible
    del model.inference_part_rows
    with torch.no_grad():
        assert model._parts_for(8) == 1 and model._parts_for(12) == 2      # the shipped threshold: parts of >= 6 frames
    model.inference_part_rows = 4
    with torch.enable_grad():
        assert model._parts_for(16) == 1                                   # autograd is recording
    with torch.no_grad():
        assert model._parts_for(16) == 2
        from creste_public_amd import _lib
        _lib._recorder = _lib.PlanRecorder()
        try:
            assert model._parts_for(16) == 1                               # deploy.export_plan traces one stream
        finally:
            _lib._recorder = None


def test_pipelined_frozen_half_feeds_the_mdp_solve():
    """solve_mdp=True under no_grad: the frozen half runs in parts, the reward network / value iteration / SVF on the whole
    batch behind the join."""
    model = _model(solve_mdp=True)
    B = 8
    rgbd, p2p = synth.make_frames(B, H, W, seed=99)
    expert = synth.make_experts(B, 50, 256, seed=3)
    inputs = (rgbd.cuda(), p2p.cuda(), expert.cuda())
    halves = [_plain(model, tuple(t[i * 4:(i + 1) * 4].contiguous() for t in inputs)) for i in range(2)]
    with torch.no_grad():
        assert model._parts_for(B) == 2
        got = model(inputs)
    torch.cuda.synchronize()
    # the frozen half's outputs are the parts' (exactly); the trainable half ran on the whole batch behind the join: its
    # reward network is per-sample arithmetic on identical inputs, the MDP solve converges batch-wide (sweep count of the
    # slowest sample), so its outputs agree with the parts' solves to the solver's threshold
    for k in ("bev_features", "elevation_preds", "input_view", "traversability_preds"):
        assert torch.equal(got[k], torch.cat([h[k] for h in halves])), k
    for k in ("value_estimate", "policy", "exp_svf"):
        assert torch.isfinite(got[k].float()).all(), k
    v = torch.cat([h["value_estimate"] for h in halves])
    assert (got["value_estimate"] - v).abs().max() <= 5e-2 * max(1.0, float(v.abs().max()))
    assert int(model.traversability_head.last_sweeps) > 0      # (raises on a failed solve)


def test_back_to_back_pipelined_steps_are_reproducible_at_full_size():
    """batch 16 of 1216x608, six pipelined steps with no host synchronisation in between: every output key of every step
    equals the first step's, bit for bit.  (With packed-fp32 VALU in the build -- v_pk_fma_f32 in the fused MBConv kernels --
    single frames came out wrong in 3-8 of 12 such steps: a part's VALU kernels then share CUs with the other part's MFMA
    kernels; creste_public_amd/build.py NO_PK, scripts/concurrency_bisect.py.)"""
    Hf, Wf, B = 608, 1216, 16
    torch.manual_seed(3)
    creste_public_amd.set_precision("bf16x6")
    model = MaxEntIRL(maxent_irl_cfg((Hf, Wf), solve_mdp=False))
    synth.randomize_bn(model, seed=3)
    model = model.cuda().eval()
    rgbd, p2p = synth.make_frames(B, Hf, Wf, seed=21)
    rgbd, p2p = rgbd.cuda(), p2p.cuda()
    synth.calibrate_bn_hip(model, rgbd[:2], p2p[:2])
    with torch.no_grad():
        assert model._parts_for(B) == 2
        model((rgbd, p2p))
        ref = {k: v.clone() for k, v in model((rgbd, p2p)).items()}
        torch.cuda.synchronize()
        outs = [model((rgbd, p2p)) for _ in range(6)]
        torch.cuda.synchronize()
    for i, o in enumerate(outs):
        for k, v in o.items():
            assert torch.equal(v, ref[k]), f"step {i}: {k}"
    assert float((ref["bev_densities"] > 0).float().mean()) > 0.05


def test_side_stream_probe_tells_one_queue_from_two():
    """ops.concurrent_stream hands out only streams that were MEASURED to run beside the caller's (creste_spin_us): a stream
    probed against itself is the one-queue case and must fail; whatever the probe accepts must pass it again, and the roles'
    streams are distinct."""
    dev = torch.device("cuda", 0)
    main = torch.cuda.current_stream(dev)
    assert not ops._runs_beside(main, main)                       # one queue: the small kernel waits for the large grid
    got = {role: ops.concurrent_stream(dev, role) for role in ("parts", "wgrad", "prefetch")}
    assert all(s is None or isinstance(s, torch.cuda.Stream) for s in got.values())
    live = [s for s in got.values() if s is not None]
    assert len({s.cuda_stream for s in live}) == len(live) and all(s.cuda_stream != main.cuda_stream for s in live)
    for s in live:
        assert ops._runs_beside(main, s)
    assert ops.concurrent_stream(dev, "parts") is got["parts"]    # cached per (device, role, caller's stream)


def test_terrainnet_eval_forward_is_pipelined_the_same_way():
    """TerrainNet.forward in eval mode (SSC validation, the deployment script's model) runs batches of >= 12 frames as two
    forwards on two streams too: equal to the plain forwards of its halves, bit for bit, outputs in shared buffers."""
    from creste_public_amd import TerrainNet, terrainnet_cfg
    torch.manual_seed(5)
    creste_public_amd.set_precision("bf16x6")
    net = TerrainNet(terrainnet_cfg((H, W)))
    synth.randomize_bn(net, seed=5)
    net = net.cuda().eval()
    B = 12
    rgbd, p2p = synth.make_frames(B, H, W, seed=77)
    rgbd, p2p = rgbd.cuda(), p2p.cuda()
    with torch.no_grad():
        net.inference_parts = 0
        want = [{k: v.clone() for k, v in net((rgbd[i * 6:(i + 1) * 6].contiguous(), p2p[i * 6:(i + 1) * 6].contiguous())).items()}
                for i in range(2)]
        del net.inference_parts
        assert ops.parts_for(B, rgbd.device, net.inference_parts, net.inference_part_rows) == 2
        for rep in range(2):
            got = net((rgbd, p2p))
            torch.cuda.synchronize()
            assert set(got) == set(want[0])
            for k, v in got.items():
                assert torch.equal(v, torch.cat([w[k] for w in want])), f"{k} (repeat {rep})"
    assert got["bev_features"][:6].untyped_storage().data_ptr() == got["bev_features"][6:].untyped_storage().data_ptr()
