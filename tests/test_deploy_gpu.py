"""GPU: the hipGraph deployment artefact (creste_public_amd/deploy.py; reference scripts/runtime/compile.py:160-210):
captured replay == eager launches bit for bit, new frames through the captured graph, save / load round trip."""
import pytest
import torch

from creste_public_amd import maxent_irl_cfg, synth

pytestmark = pytest.mark.gpu

H, W = 128, 192


def _model(precision):
    import creste_public_amd
    from creste_public_amd import MaxEntIRL
    creste_public_amd.set_precision(precision)
    torch.manual_seed(3)
    model = MaxEntIRL(maxent_irl_cfg((H, W), solve_mdp=False))          # compile.py: "don't solve mdp for inference"
    synth.randomize_bn(model, seed=1)
    with torch.no_grad():
        model.backbone.depthcomp.depthcomp.vision_backbone.model.trunk._bn0.running_var.fill_(1.0e7)
    return model.cuda().eval()


@pytest.mark.parametrize("precision,B", [("f16x3", 1), ("f32", 2)])
def test_compiled_model_matches_eager_and_roundtrips(tmp_path, precision, B):
    import creste_public_amd
    from creste_public_amd import deploy
    try:
        model = _model(precision)
        frames = [tuple(t.cuda() for t in synth.make_frames(B, H, W, seed=s)) for s in (1, 2, 3)]
        with torch.no_grad():
            eager = [{k: v.clone() for k, v in model(f).items()} for f in frames]
        cm = deploy.compile_model(model, frames[0])
        for f, ref in zip(frames[1:] + frames[:1], eager[1:] + eager[:1]):     # other frames than the captured one
            out = cm(f)
            torch.cuda.synchronize()
            assert set(out) == set(ref)
            for k in ref:
                assert torch.equal(out[k], ref[k]), k
        path = tmp_path / "traversability_hip.pt"
        cm.save(str(path))
        creste_public_amd.set_precision("f32")                                 # load() restores the artefact's mode
        cm2 = deploy.load(str(path), device="cuda:0")
        assert creste_public_amd.get_precision() == precision
        out = cm2(frames[1], clone=True)
        for k in eager[1]:
            assert torch.equal(out[k], eager[1][k]), k
        with pytest.raises(ValueError):
            cm2((frames[0][0][:, :, :, : H // 2], frames[0][1]))
    finally:
        creste_public_amd.set_precision("f32")


def test_compile_refuses_training_mode_and_cpu():
    from creste_public_amd import deploy
    from creste_public_amd.ops import HipLibraryError
    model = _model("f32")
    f = tuple(t.cuda() for t in synth.make_frames(1, H, W, seed=1))
    with pytest.raises(ValueError):
        deploy.compile_model(model.train(), f)
    with pytest.raises(HipLibraryError):
        deploy.compile_model(model.eval(), tuple(t.cpu() for t in f))
