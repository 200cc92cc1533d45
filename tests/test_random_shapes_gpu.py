"""GPU: the fused / tiled kernels of the encoder trunk on RANDOM small shapes (maps smaller than a tile, widths of 1-3 pixels,
bands and strips that end mid-tile) against float64 torch -- the fixed-size cases live in test_kernels_gpu.py."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from creste_public_amd import ops as o
    o._lib.load()
    return o


def dev(t):
    return t.cuda().contiguous()


def swish(t):
    return t * torch.sigmoid(t)


def _same_pad(K, s):
    return ((K - 1) // 2, (K - 1) - (K - 1) // 2) if s == 1 else ((K - 2) // 2, (K - 2) - (K - 2) // 2)


def test_depthwise_tile_and_register_kernels_on_random_shapes(ops):
    rng = random.Random(7)
    g = torch.Generator().manual_seed(7)
    for _ in range(24):
        K, s = rng.choice([3, 5]), rng.choice([1, 2])
        C = 4 * rng.randint(4, 70)
        N, H, W = rng.randint(1, 3), rng.randint(1, 40), rng.randint(1, 45)
        pt, pb = _same_pad(K, s); pl, pr = _same_pad(K, s)
        if (H + pt + pb - K) < 0 or (W + pl + pr - K) < 0:
            continue
        x = torch.randn(N, C, H, W, generator=g)
        w, b = torch.randn(C, 1, K, K, generator=g) / K, torch.randn(C, generator=g)
        Cse = max(1, C // 16)
        w1, b1 = torch.randn(Cse, C, generator=g) / C ** 0.5, torch.randn(Cse, generator=g)
        w2, b2 = torch.randn(C, Cse, generator=g) / Cse ** 0.5, torch.randn(C, generator=g)
        y = swish(F.conv2d(F.pad(x.double(), (pl, pr, pt, pb)), w.double(), b.double(), stride=s, groups=C))
        h = y.mean(dim=(2, 3)) @ w1.double().t() + b1.double()
        gate_ref = torch.sigmoid(swish(h) @ w2.double().t() + b2.double())
        out, gate = ops.dwconv2d_se(ops.nchw_to_nhwc(dev(x)), dev(w.view(C, K * K).t().contiguous()), dev(b), K, s,
                                    (pt, pb, pl, pr), ops.ACT_SWISH, dev(w1), dev(b1), dev(w2), dev(b2))
        tag = f"C={C} K={K} s={s} {N}x{H}x{W}"
        torch.testing.assert_close(out.nchw().cpu().double(), y, rtol=2e-5, atol=2e-5, msg=lambda m: f"{tag}: {m}")
        torch.testing.assert_close(gate.cpu().double(), gate_ref, rtol=2e-5, atol=2e-6, msg=lambda m: f"{tag} gate: {m}")
        plain = ops.dwconv2d(ops.nchw_to_nhwc(dev(x)), dev(w.view(C, K * K).t().contiguous()), None, K, s, (pt, pb, pl, pr), ops.ACT_NONE)
        ref0 = F.conv2d(F.pad(x.double(), (pl, pr, pt, pb)), w.double(), None, stride=s, groups=C)
        torch.testing.assert_close(plain.nchw().cpu().double(), ref0, rtol=2e-5, atol=2e-5, msg=lambda m: f"{tag} plain: {m}")


def test_mbconv_front_half_on_random_shapes(ops):
    rng = random.Random(11)
    g = torch.Generator().manual_seed(11)
    for _ in range(18):
        K, s, Cin = rng.choice([3, 5]), rng.choice([1, 2]), rng.choice([16, 24, 40])
        Cexp = 4 * rng.randint(4, 64)
        N, H, W = rng.randint(1, 3), rng.randint(2, 50), rng.randint(2, 70)
        pt, pb = _same_pad(K, s); pl, pr = _same_pad(K, s)
        x = torch.randn(N, Cin, H, W, generator=g)
        we, be = torch.randn(Cexp, Cin, generator=g) / Cin ** 0.5, torch.randn(Cexp, generator=g)
        wd, bd = torch.randn(Cexp, 1, K, K, generator=g) / K, torch.randn(Cexp, generator=g)
        Cse = max(1, Cin // 4)
        w1, b1 = torch.randn(Cse, Cexp, generator=g) / Cexp ** 0.5, torch.randn(Cse, generator=g)
        w2, b2 = torch.randn(Cexp, Cse, generator=g) / Cse ** 0.5, torch.randn(Cexp, generator=g)
        e = swish(F.conv2d(x.double(), we.double().view(Cexp, Cin, 1, 1), be.double()))
        y = swish(F.conv2d(F.pad(e, (pl, pr, pt, pb)), wd.double(), bd.double(), stride=s, groups=Cexp))
        h = y.mean(dim=(2, 3)) @ w1.double().t() + b1.double()
        gate_ref = torch.sigmoid(swish(h) @ w2.double().t() + b2.double())
        out, gate = ops.mbconv_expand_dw_se(ops.nchw_to_nhwc(dev(x)), dev(we.t().contiguous()), dev(be),
                                            dev(wd.view(Cexp, K * K).t().contiguous()), dev(bd), K, s, (pt, pb, pl, pr),
                                            dev(w1), dev(b1), dev(w2), dev(b2))
        tag = f"Cin={Cin} Cexp={Cexp} K={K} s={s} {N}x{H}x{W}"
        torch.testing.assert_close(out.nchw().cpu().double(), y, rtol=3e-5, atol=3e-5, msg=lambda m: f"{tag}: {m}")
        torch.testing.assert_close(gate.cpu().double(), gate_ref, rtol=2e-5, atol=3e-6, msg=lambda m: f"{tag} gate: {m}")


def test_upsample2x_and_stem_on_random_shapes(ops):
    rng = random.Random(13)
    g = torch.Generator().manual_seed(13)
    for _ in range(16):
        N, C1, C2 = rng.randint(1, 3), 4 * rng.randint(1, 80), 4 * rng.randint(0, 8)
        H1, W1 = rng.randint(2, 70), rng.randint(2, 150)
        x1 = torch.randn(N, C1, H1, W1, generator=g)
        skip = torch.randn(N, C2, 2 * H1, 2 * W1, generator=g) if C2 else None
        up = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=False)
        ref = torch.cat([skip, up], dim=1) if C2 else up
        out = ops.upsample_concat(ops.nchw_to_nhwc(dev(x1)), ops.nchw_to_nhwc(dev(skip)) if C2 else None, 2 * H1, 2 * W1, 0.5, 0.5)
        torch.testing.assert_close(out.nchw().cpu(), ref, rtol=1e-6, atol=1e-6, msg=lambda m: f"up2x C1={C1} C2={C2} {N}x{H1}x{W1}: {m}")
    for _ in range(8):
        N, H, W = rng.randint(1, 3), rng.randint(3, 90), rng.randint(3, 150)
        pad = (rng.randint(0, 1), 1, rng.randint(0, 1), 1)
        x = torch.randn(N, 4, H, W, generator=g)
        ws, bs = torch.randn(32, 4, 3, 3, generator=g) / 6, torch.randn(32, generator=g) * 0.5
        wd, bd = torch.randn(32, 1, 3, 3, generator=g) / 3, torch.randn(32, generator=g)
        w1, b1 = torch.randn(8, 32, generator=g) / 6, torch.randn(8, generator=g)
        w2, b2 = torch.randn(32, 8, generator=g) / 3, torch.randn(32, generator=g)
        e = swish(F.conv2d(F.pad(x.double(), (pad[2], pad[3], pad[0], pad[1])), ws.double(), bs.double(), stride=2))
        y = swish(F.conv2d(e, wd.double(), bd.double(), padding=1, groups=32))
        out, gate = ops.stem_dw_se(ops.nchw_to_nhwc(dev(x)), dev(ws.permute(2, 3, 1, 0).reshape(36, 32)), dev(bs), pad,
                                   dev(wd.view(32, 9).t().contiguous()), dev(bd), (1, 1, 1, 1), dev(w1), dev(b1), dev(w2), dev(b2))
        torch.testing.assert_close(out.nchw().cpu().double(), y, rtol=3e-5, atol=3e-5, msg=lambda m: f"stem {N}x{H}x{W} pad {pad}: {m}")
