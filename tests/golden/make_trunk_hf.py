#!/usr/bin/env python3
"""Generate tests/golden/trunk_hf.npz: outputs of `transformers`' independently written EfficientNet-B0 and basic-block
ResNet carrying the oracle's seeded weights (see tests/hf_crosscheck.py).  Needs `transformers` (in the image); does
not need /root/reference -- the third-party packages the reference calls (`efficientnet_pytorch`, `torchvision`) are
absent from it anyway (call sites: reference creste/models/blocks/effnet.py:37-45,83, inpainting.py:80-90).

    python tests/golden/make_trunk_hf.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hf_crosscheck as hc  # noqa: E402


def main():
    torch.set_num_threads(8)
    out = {}
    for name, hw in hc.EFFNET_CASES.items():
        trunk = hc.seeded_trunk(hw)
        hf = hc.hf_effnet_from(trunk)
        x = hc.trunk_input(hw)
        eps = hc.hf_effnet_endpoints(hf, x)
        with torch.no_grad():
            ref = trunk.extract_endpoints(x)
        out[f"effnet_{name}_wsum"] = np.float64(hc.checksum(trunk))
        out[f"effnet_{name}_xsum"] = np.float64(float(x.double().abs().sum()))
        for k, v in eps.items():
            out[f"effnet_{name}_{k}"] = v.numpy()
            d = float((v - ref[k]).abs().max() / v.abs().max())
            print(f"effnet {name} {k} {tuple(v.shape)}: oracle vs HF max rel {d:.2e}")
            assert d < 2e-5, k
    bev = hc.seeded_bev()
    hfr = hc.hf_resnet_from(bev)
    x = hc.bev_input()
    x1, x3 = hc.hf_resnet_stages(hfr, x)
    o1, o3 = hc.oracle_bev_stages(bev, x)
    out["resnet_wsum"] = np.float64(hc.checksum(bev))
    out["resnet_xsum"] = np.float64(float(x.double().abs().sum()))
    out["resnet_x1"], out["resnet_x3"] = x1.numpy(), x3.numpy()
    for n, a, b in (("x1", x1, o1), ("x3", x3, o3)):
        d = float((a - b).abs().max() / a.abs().max())
        print(f"resnet {n} {tuple(a.shape)}: oracle vs HF max rel {d:.2e}")
        assert d < 2e-5
    np.savez_compressed(os.path.join(HERE, "trunk_hf.npz"), **out)
    print("wrote trunk_hf.npz", os.path.getsize(os.path.join(HERE, "trunk_hf.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
