"""Independent cross-check of the two THIRD-PARTY sub-graphs of the path (test infrastructure).

The reference builds its RGB-D trunk with `efficientnet_pytorch.EfficientNet` (call sites
/root/reference/creste/models/blocks/effnet.py:37-45,83) and its BEV trunk with torchvision's `resnet18`
(/root/reference/creste/models/blocks/inpainting.py:80-90).  Neither package is in this image and the reference holds
no vectors at those boundaries, so `oracle/blocks.py` restates the published architectures.  `transformers` (in the
image) ships independently written implementations of both: `EfficientNetModel` (a port of the Keras EfficientNet) and
`ResNetModel(layer_type="basic")`.  This module copies the oracle's seeded weights into them so that
  * oracle vs HF pins MBConv structure, static padding sides, SE width (0.25 x block INPUT filters), swish, BN eps,
    endpoint selection / BasicBlock order, shortcut placement, stride placement;
  * the committed HF outputs (tests/golden/trunk_hf.npz, made by tests/golden/make_trunk_hf.py) pin the HIP trunk on
    the GPU box.
Weights and inputs are regenerated from seeds (the trunk has 4 M parameters); the fixture stores a float64 checksum of
them, so RNG drift across torch builds is detected instead of producing a bogus mismatch.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EFFNET_CASES = {"224": (224, 224), "odd": (160, 204)}     # 204 -> 102 -> 51 (odd) -> 26 -> 13 (odd) -> 7
RESNET_HW = (64, 64)


def _randomize(module, seed):
    """Random BatchNorm affine + running statistics (gamma ~ U[0.5,1.5] so that zero_init_residual blocks are not
    identities), deterministic."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    return module


def checksum(module) -> float:
    return float(sum(v.double().abs().sum() for v in module.state_dict().values() if v.dtype.is_floating_point))


def seeded_trunk(image_size, seed=11):
    from oracle.blocks import EfficientNetB0Trunk
    torch.manual_seed(seed)
    return _randomize(EfficientNetB0Trunk(4, image_size), seed + 1).eval()


def seeded_bev(seed=21):
    from oracle.blocks import InpaintingResNet18MultiHead
    torch.manual_seed(seed)
    return _randomize(InpaintingResNet18MultiHead(96, [32, 6, 2], "bev_features",
                                                  ["inpainting_sam", "inpainting_sam_dynamic", "elevation"]),
                      seed + 1).eval()


def trunk_input(hw, seed=31):
    return torch.rand(1, 4, *hw, generator=torch.Generator().manual_seed(seed)) * 2 - 0.5


def bev_input(seed=41):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 96, *RESNET_HW, generator=g)
    return x * (torch.rand(1, 1, *RESNET_HW, generator=g) > 0.6)        # sparse, like a splatted map


def _copy_bn(dst, src):
    for n in ("weight", "bias", "running_mean", "running_var"):
        getattr(dst, n).data.copy_(getattr(src, n).data)


def hf_effnet_from(trunk):
    """transformers.EfficientNetModel (B0, 4 input channels) carrying the oracle trunk's weights."""
    from transformers import EfficientNetConfig, EfficientNetModel
    cfg = EfficientNetConfig(width_coefficient=1.0, depth_coefficient=1.0, hidden_dim=1280, num_channels=4,
                             image_size=224, batch_norm_eps=1e-3)
    hf = EfficientNetModel(cfg).eval()
    with torch.no_grad():
        hf.embeddings.convolution.weight.copy_(trunk._conv_stem.weight)
        _copy_bn(hf.embeddings.batchnorm, trunk._bn0)
        assert len(hf.encoder.blocks) == len(trunk._blocks)
        for hb, ob in zip(hf.encoder.blocks, trunk._blocks):
            if ob.expand:
                hb.expansion.expand_conv.weight.copy_(ob._expand_conv.weight)
                _copy_bn(hb.expansion.expand_bn, ob._bn0)
            hb.depthwise_conv.depthwise_conv.weight.copy_(ob._depthwise_conv.weight)
            _copy_bn(hb.depthwise_conv.depthwise_norm, ob._bn1)
            hb.squeeze_excite.reduce.weight.copy_(ob._se_reduce.weight)
            hb.squeeze_excite.reduce.bias.copy_(ob._se_reduce.bias)
            hb.squeeze_excite.expand.weight.copy_(ob._se_expand.weight)
            hb.squeeze_excite.expand.bias.copy_(ob._se_expand.bias)
            hb.projection.project_conv.weight.copy_(ob._project_conv.weight)
            _copy_bn(hb.projection.project_bn, ob._bn2)
    return hf


def hf_effnet_endpoints(hf, x):
    """The five endpoints efficientnet_pytorch.extract_endpoints would return, taken from HF's per-block hidden
    states: the block output BEFORE each resolution drop, plus the last block."""
    with torch.no_grad():
        emb = hf.embeddings(x)
        hs = hf.encoder(emb, output_hidden_states=True).hidden_states      # (stem, block0, ..., block15)
    eps = []
    for a, b in zip(hs[:-1], hs[1:]):
        if a.shape[2] > b.shape[2]:
            eps.append(a)
    eps.append(hs[-1])
    assert len(eps) == 5
    return {f"reduction_{i + 1}": e for i, e in enumerate(eps)}


def hf_resnet_from(bev):
    """transformers.ResNetModel (basic blocks, 64-128-256) carrying the oracle BEV trunk's weights."""
    from transformers import ResNetConfig, ResNetModel
    hf = ResNetModel(ResNetConfig(num_channels=96, embedding_size=64, hidden_sizes=[64, 128, 256], depths=[2, 2, 2],
                                  layer_type="basic", downsample_in_first_stage=False)).eval()
    with torch.no_grad():
        hf.embedder.embedder.convolution.weight.copy_(bev.conv1.weight)
        _copy_bn(hf.embedder.embedder.normalization, bev.bn1)
        for stage, layer in zip(hf.encoder.stages, (bev.layer1, bev.layer2, bev.layer3)):
            for hl, ob in zip(stage.layers, layer):
                hl.layer[0].convolution.weight.copy_(ob.conv1.weight)
                _copy_bn(hl.layer[0].normalization, ob.bn1)
                hl.layer[1].convolution.weight.copy_(ob.conv2.weight)
                _copy_bn(hl.layer[1].normalization, ob.bn2)
                if ob.downsample is not None:
                    hl.shortcut.convolution.weight.copy_(ob.downsample[0].weight)
                    _copy_bn(hl.shortcut.normalization, ob.downsample[1])
    return hf


def hf_resnet_stages(hf, x):
    """(x1 = layer1 output, x3 = layer3 output) -- the reference's trunk has NO max-pool after the stem
    (inpainting.py:96-101), so HF's embedder conv layer is applied without its pooler."""
    with torch.no_grad():
        h = hf.embedder.embedder(x)
        x1 = hf.encoder.stages[0](h)
        x3 = hf.encoder.stages[2](hf.encoder.stages[1](x1))
    return x1, x3


def oracle_bev_stages(bev, x):
    with torch.no_grad():
        h = bev.relu(bev.bn1(bev.conv1(x)))
        x1 = bev.layer1(h)
        return x1, bev.layer3(bev.layer2(x1))


def load_fixture():
    p = os.path.join(ROOT, "tests", "golden", "trunk_hf.npz")
    return {k: v for k, v in np.load(p).items()}
