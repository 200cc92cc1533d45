"""GPU: the Python-free deployment entry (reference scripts/runtime/compile.py:160-210).
`deploy.export_plan` records one forward as a plan file; `creste_hip_model_load/_infer` (csrc/plan_runtime.cpp) replay
it from C.  Every output of the replay equals the Python host path BIT FOR BIT -- through the ctypes client, through a
hipGraph inside the C runtime, on fresh inputs, and from a C program that never starts Python."""
import os
import subprocess

import numpy as np
import pytest
import torch

import creste_public_amd
from creste_public_amd import MaxEntIRL, deploy, maxent_irl_cfg, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
H, W, B = 128, 192, 2


@pytest.fixture(scope="module", params=["f16x3", "f32"])
def exported(request, tmp_path_factory):
    creste_public_amd.set_precision(request.param)
    torch.manual_seed(3)
    model = MaxEntIRL(maxent_irl_cfg((H, W), solve_mdp=False))
    synth.randomize_bn(model, seed=4)
    model = model.cuda().eval()
    rgbd, p2p = synth.make_frames(B, H, W, seed=5)
    rgbd, p2p = rgbd.cuda(), p2p.cuda()
    synth.calibrate_bn_hip(model, rgbd, p2p)
    path = str(tmp_path_factory.mktemp("plan") / f"irl_{request.param}.plan")
    summary = deploy.export_plan(model, (rgbd, p2p), path)
    yield model, path, summary, (rgbd, p2p)
    creste_public_amd.set_precision("f32")


def _eager(model, inputs):
    with torch.no_grad():
        out = model(inputs)
    torch.cuda.synchronize()
    return {k: v.detach().cpu().contiguous() for k, v in out.items()}


def test_plan_is_self_contained_and_small(exported):
    model, path, summary, _ = exported
    assert summary["calls"] > 100 and set(summary["outputs"]) == set(_eager(model, exported[3]))
    assert all(n.startswith("creste_") for n in summary["entry_points"])
    n_params = sum(p.numel() for p in model.parameters()) * 4
    assert summary["constant_bytes"] < 4 * n_params         # packed / split weights, not activations
    assert os.path.getsize(path) < summary["constant_bytes"] + (4 << 20)
    blob = open(path, "rb").read(64)
    assert blob.startswith(b"CRESTEPLAN") and b"torch" not in blob and b"pickle" not in blob


@pytest.mark.parametrize("graph", [False, True])
def test_replay_matches_python_path_bit_for_bit(exported, graph):
    model, path, _, inputs = exported
    pm = deploy.PlanModel(path, graph=graph)
    try:
        assert [d["name"] for d in pm.inputs] == ["rgbd", "p2p"]
        ref = _eager(model, inputs)
        got = pm(inputs)
        assert set(got) == set(ref)
        for k in ref:
            assert got[k].shape == ref[k].shape, k
            assert torch.equal(got[k], ref[k].to(got[k].dtype)), k
        # fresh data through the SAME plan (|max| slots, splat workspaces ... must not carry state between runs)
        rgbd2, p2p2 = synth.make_frames(B, H, W, seed=99)
        rgbd2, p2p2 = rgbd2.cuda() * 0.5, p2p2.cuda()
        ref2 = _eager(model, (rgbd2, p2p2))
        for _ in range(2):
            got2 = pm((rgbd2, p2p2))
        for k in ref2:
            assert torch.equal(got2[k], ref2[k].to(got2[k].dtype)), k
        assert not torch.equal(ref2["traversability_preds"], ref["traversability_preds"])
    finally:
        pm.close()


def test_c_program_runs_the_plan_without_python(exported, tmp_path):
    model, path, _, inputs = exported
    exe = str(tmp_path / "creste_infer_main")
    lib_dir = os.path.join(ROOT, "creste_public_amd", "lib")
    subprocess.run(["g++", "-O1", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    os.path.join(ROOT, "tests", "c", "creste_infer_main.c"), "-L" + lib_dir, "-lcreste_hip",
                    "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir + ":/opt/rocm/lib", "-o", exe], check=True)
    for t, name in zip(inputs, ("rgbd", "p2p")):
        t.detach().cpu().contiguous().numpy().tofile(str(tmp_path / f"{name}.f32"))
    outdir = tmp_path / "out"
    outdir.mkdir()
    r = subprocess.run([exe, path, str(tmp_path / "rgbd.f32"), str(tmp_path / "p2p.f32"), str(outdir)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    ref = _eager(model, inputs)
    pm = deploy.PlanModel(path)          # only for the output layouts
    try:
        for d in pm.outputs:
            np_dt = {0: np.float32, 1: np.int64, 2: np.uint8}[d["dtype"]]
            raw = np.fromfile(str(outdir / f"{d['name']}.bin"), dtype=np_dt)
            view = np.lib.stride_tricks.as_strided(raw, shape=d["shape"], strides=[s * raw.itemsize for s in d["stride"]])
            got = torch.from_numpy(np.ascontiguousarray(view))
            assert torch.equal(got, ref[d["name"]].to(got.dtype)), d["name"]
    finally:
        pm.close()


# ---- the pipelined plan: a forward that the Python path runs as two half-batch forwards on two
# streams is exported with its streams and stream-order edges, and the C runtime replays it on streams of its own
@pytest.fixture(scope="module")
def exported_pipelined(tmp_path_factory):
    creste_public_amd.set_precision("bf16x6")
    torch.manual_seed(8)
    model = MaxEntIRL(maxent_irl_cfg((H, W), solve_mdp=False))
    synth.randomize_bn(model, seed=9)
    model = model.cuda().eval()
    model.inference_part_rows = 4                    # 8 frames = 2 parts of 4 (the shipped threshold is a speed threshold)
    rgbd, p2p = synth.make_frames(8, H, W, seed=6)
    rgbd, p2p = rgbd.cuda(), p2p.cuda()
    synth.calibrate_bn_hip(model, rgbd[:2], p2p[:2])
    with torch.no_grad():
        assert model._parts_for(8) == 2, "no probed side stream on this box: the pipelined plan cannot be traced"
    d = tmp_path_factory.mktemp("plan_pipe")
    path, path1 = str(d / "irl_pipe.plan"), str(d / "irl_one_stream.plan")
    summary = deploy.export_plan(model, (rgbd, p2p), path)
    summary1 = deploy.export_plan(model, (rgbd, p2p), path1, pipelined=False)
    yield model, path, summary, (rgbd, p2p), path1, summary1
    creste_public_amd.set_precision("f32")


@pytest.mark.parametrize("graph", [False, True])
def test_pipelined_plan_replays_the_two_stream_forward_bit_for_bit(exported_pipelined, graph):
    model, path, summary, inputs, path1, summary1 = exported_pipelined
    assert summary["streams"] == 2 and summary["events"] >= 2          # fork + join (+ buffer-ordering edges)
    assert summary1["streams"] == 1 and summary1["events"] == 0
    assert summary["calls"] > summary1["calls"]                         # two half-batch forwards launch more kernels
    ref = _eager(model, inputs)                                          # the Python path: pipelined as well
    pm = deploy.PlanModel(path, graph=graph)
    try:
        assert pm.num_streams == 2
        for rep in range(3):
            got = pm(inputs)
            for k in ref:
                assert torch.equal(got[k], ref[k].to(got[k].dtype)), f"{k} (replay {rep})"
        rgbd2, p2p2 = synth.make_frames(8, H, W, seed=123)
        ref2 = _eager(model, (rgbd2.cuda(), p2p2.cuda()))
        got2 = pm((rgbd2.cuda(), p2p2.cuda()))
        for k in ref2:
            assert torch.equal(got2[k], ref2[k].to(got2[k].dtype)), k
    finally:
        pm.close()
    # the one-stream plan of the same model computes the WHOLE batch in one forward: equal to the Python path with
    # pipelining off (a half-batch forward differs from the whole-batch one at float-noise level: the SE pooling's partition)
    model.inference_parts = 0
    try:
        ref1 = _eager(model, inputs)
    finally:
        del model.inference_parts
    pm1 = deploy.PlanModel(path1)
    try:
        assert pm1.num_streams == 1
        got1 = pm1(inputs)
        for k in ref1:
            assert torch.equal(got1[k], ref1[k].to(got1[k].dtype)), k
    finally:
        pm1.close()
