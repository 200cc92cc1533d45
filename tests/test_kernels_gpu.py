"""GPU: each HIP kernel, called through the C ABI, against a plain PyTorch-CPU fp32 statement of the
same op (dense ops) or against the reference-generated golden vectors (splat / VI / SVF)."""
import numpy as np
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


def to_act(ops, x_nchw):
    return ops.nchw_to_nhwc(dev(x_nchw))


def from_act(a):
    return a.nchw().contiguous().cpu()


CONV_CASES = [
    # N, Cin, H, W, Cout, K, stride, pad(t,b,l,r), act, bias, bn, res
    (2, 4, 19, 23, 32, 3, 2, (0, 1, 0, 1), 2, False, True, False),     # stem-like, asymmetric pad, swish
    (1, 496, 9, 11, 496, 3, 1, (1, 1, 1, 1), 1, False, True, False),   # up3-like channels (not /32)
    (2, 24, 12, 10, 144, 1, 1, (0, 0, 0, 0), 2, False, True, False),   # expand 1x1
    (2, 144, 12, 10, 24, 1, 1, (0, 0, 0, 0), 0, False, True, True),    # project 1x1 + residual
    (1, 96, 20, 20, 64, 7, 2, (3, 3, 3, 3), 1, False, True, False),    # BEV stem 7x7/2
    (3, 40, 8, 16, 64, 5, 1, (2, 2, 2, 2), 1, False, True, False),     # reward prepool 5x5
    (1, 128, 16, 16, 6, 1, 1, (0, 0, 0, 0), 0, True, False, False),    # proj with bias, tiny Cout
    (1, 48, 8, 8, 1, 1, 1, (0, 0, 0, 0), 1, False, True, False),       # postpool Cout=1
    (2, 64, 17, 13, 128, 3, 2, (1, 1, 1, 1), 1, False, True, False),   # resnet stride-2
    (1, 256, 130, 3, 128, 3, 1, (1, 1, 1, 1), 1, False, True, False),  # M not a multiple of 128
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm(ops, case):
    N, Cin, H, W, Cout, K, s, pad, act, use_bias, use_bn, use_res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g) if use_bias else None
    bn = None
    if use_bn:
        bn = (torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1,
              torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5, 1e-3)
    xp = F.pad(x, (pad[2], pad[3], pad[0], pad[1]))
    ref = F.conv2d(xp.double(), w.double(), None if b is None else b.double(), stride=s)
    if bn is not None:
        ref = F.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, bn[4])
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + res.double()
    ref = {0: lambda t: t, 1: F.relu, 2: lambda t: t * torch.sigmoid(t)}[act](ref)

    pc = ops.pack_conv(dev(w), None if b is None else dev(b),
                       None if bn is None else tuple(dev(t) if isinstance(t, torch.Tensor) else t for t in bn),
                       s, pad, act)
    out = ops.conv2d(to_act(ops, x), pc, res=None if res is None else to_act(ops, res))
    got = from_act(out)
    assert got.shape == ref.shape
    torch.testing.assert_close(got.double(), ref, rtol=2e-5, atol=2e-5)


PATCH_CASES = [c for c in CONV_CASES if c[5] in (1, 3) and c[6] == 1] + [
    (2, 64, 37, 70, 96, 3, 1, (1, 1, 1, 1), 1, False, True, True),     # partial tiles in x and y, residual
    (1, 20, 9, 33, 200, 1, 1, (0, 0, 0, 0), 2, True, True, False),     # Cin not /16, two channel tiles
    (1, 36, 12, 40, 40, 3, 1, (1, 1, 1, 1), 0, False, False, False),   # ragged chunk, BN=64 variant
]


@pytest.mark.parametrize("prec,tol", [("bf16x6", 2e-6), ("f16x3", 2e-6), ("bf16x3", 6e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("case", PATCH_CASES)
def test_conv_patch_bf16(ops, case, prec, tol):
    """bf16-MFMA patch engine: relative rms error vs a float64 conv.  bf16x3 must stay fp32-grade."""
    N, Cin, H, W, Cout, K, s, pad, act, use_bias, use_bn, use_res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g) if use_bias else None
    bn = (torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1,
          torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5, 1e-3) if use_bn else None
    ref = F.conv2d(F.pad(x, (pad[2], pad[3], pad[0], pad[1])).double(), w.double(),
                   None if b is None else b.double(), stride=s)
    if bn is not None:
        ref = F.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, bn[4])
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + res.double()
    ref = {0: lambda t: t, 1: F.relu, 2: lambda t: t * torch.sigmoid(t)}[act](ref)
    code = {"bf16x6": ops.PREC_BF16X6, "bf16x3": ops.PREC_BF16X3, "bf16": ops.PREC_BF16,
            "f16x3": ops.PREC_F16X3}[prec]
    assert ops.conv_supported(code, K, s)
    pc = ops.pack_conv(dev(w), None if b is None else dev(b),
                       None if bn is None else tuple(dev(t) if isinstance(t, torch.Tensor) else t for t in bn),
                       s, pad, act, code, algo=ops.ALGO_DIRECT)      # the DIRECT kernels (Winograd: test_conv_winograd)
    got = from_act(ops.conv2d(to_act(ops, x), pc, res=None if res is None else to_act(ops, res))).double()
    assert got.shape == ref.shape
    rel = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-12))
    assert rel < tol, f"{prec}: relative rms error {rel:.2e}"
    assert float((got - ref).abs().max()) < 60 * tol * float(ref.abs().max().clamp_min(1.0))


WINO_CASES = [
    # N, Cin, H, W, Cout, act, bias, bn, res, mask, out_slice
    (1, 496, 9, 11, 496, 1, False, True, False, False, False),      # up3-like channels (two 256-cout tiles, ragged chunk)
    (2, 64, 37, 70, 96, 1, False, True, True, False, False),        # odd extents (half tiles at the right / bottom), residual
    (1, 36, 12, 40, 40, 0, False, False, False, False, False),      # Cin not /16, Cout < 64
    (2, 256, 16, 16, 128, 1, True, True, False, True, True),        # 128-cout tile, bias, row mask, output channel slice
    (3, 128, 33, 34, 320, 2, False, True, True, False, False),      # three images: tile blocks cross image borders
    (1, 132, 128, 153, 132, 1, False, True, False, False, False),   # the reference's 128 x 153 map (odd width)
    (5, 32, 64, 60, 64, 1, False, True, False, False, False),       # F(4x4): five tile blocks x one cout tile (ragged panel groups)
    (9, 16, 128, 128, 272, 0, False, False, False, False, False),   # F(4x4): 36 tile blocks (XCDs with 4 and 5), two cout tiles
]


@pytest.mark.parametrize("variant,prec,tol", [("f2", "bf16x6", 2e-6), ("f2", "bf16x3", 8e-5),
                                              ("f4", "bf16x6", 6e-6), ("f4", "bf16x3", 1.2e-4)])
@pytest.mark.parametrize("case", WINO_CASES)
def test_conv_winograd(ops, case, variant, prec, tol):
    """Winograd paths of the stride-1 3x3 convs -- F(2x2,3x3) (csrc/conv_wino.hip, same bound as the direct split-operand
    kernels) and F(4x4,3x3) (csrc/conv_wino4.hip: fp32 transforms with coefficients up to 8, bound 3x wider) -- against
    a float64 conv, and against the DIRECT kernel of the same mode."""
    wino = ops.ALGO_WINOGRAD if variant == "f2" else ops.ALGO_WINOGRAD4
    N, Cin, H, W, Cout, act, use_bias, use_bn, use_res, use_mask, use_slice = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) if use_bias else None
    bn = (torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1,
          torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5, 1e-3) if use_bn else None
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    if bn is not None:
        ref = F.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, bn[4])
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + res.double()
    ref = {0: lambda t: t, 1: F.relu, 2: lambda t: t * torch.sigmoid(t)}[act](ref)
    mask = (torch.rand(N * H * W, generator=g) > 0.3).float() if use_mask else None
    if mask is not None:
        ref = ref * mask.view(N, 1, H, W).double()
    code = {"bf16x6": ops.PREC_BF16X6, "bf16x3": ops.PREC_BF16X3}[prec]
    bn_d = None if bn is None else tuple(dev(t) if isinstance(t, torch.Tensor) else t for t in bn)
    outs = {}
    for algo in (wino, ops.ALGO_DIRECT):
        pc = ops.pack_conv(dev(w), None if b is None else dev(b), bn_d, 1, 1, act, code, algo=algo)
        assert pc.algo == algo
        out = None
        if use_slice:
            buf = ops.Act.empty(N, H, W, Cout, "cuda", cs=Cout + 24)
            ops.fill_(buf.buf, 7.0)
            out = buf.slice(8, Cout)
        y = ops.conv2d(to_act(ops, x), pc, out=out, res=None if res is None else to_act(ops, res),
                       row_mask=None if mask is None else dev(mask))
        if use_slice:      # nothing outside the slice is touched
            assert float(y.buf[..., :8].min()) == 7.0 and float(y.buf[..., 8 + Cout:].max()) == 7.0
        outs[algo] = from_act(y).double()
    got = outs[wino]
    assert got.shape == ref.shape
    rel = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-12))
    rel_direct = float((outs[ops.ALGO_DIRECT] - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-12))
    assert rel < tol, f"{prec} winograd: relative rms error {rel:.2e} (direct kernel: {rel_direct:.2e})"
    assert float((got - ref).abs().max()) < 60 * tol * float(ref.abs().max().clamp_min(1.0))


S2_CASES = [c for c in CONV_CASES if c[6] == 2] + [
    (2, 96, 64, 70, 64, 7, 2, (3, 3, 3, 3), 1, False, True, False),    # BEV stem, several tiles, partial in x
    (1, 64, 33, 65, 128, 3, 2, (1, 1, 1, 1), 1, False, True, True),    # odd extents, residual, two 64-channel units
    (2, 64, 16, 40, 128, 1, 2, (0, 0, 0, 0), 0, False, True, False),   # ResNet 1x1/2 downsample
    (1, 20, 21, 35, 200, 3, 2, (0, 1, 0, 1), 2, True, False, False),   # ragged chunk, bias, static 'same' padding
    # stride-1 5x5 / 7x7 on the same row-at-a-time kernel (reward network, input gradient of the 7x7/2 stem)
    (3, 40, 8, 16, 64, 5, 1, (2, 2, 2, 2), 1, False, True, False),
    (1, 64, 41, 70, 96, 7, 1, (3, 4, 3, 4), 0, False, False, False),   # zero-inserted cotangent, one-sided extra pad
    (2, 48, 19, 37, 40, 5, 1, (2, 2, 2, 2), 1, True, True, True),
]


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("case", S2_CASES)
def test_conv_patch_stride2_f16x3(ops, case, mode):
    """stride-2 convs (K = 1, 3, 7) and stride-1 5x5 / 7x7 convs on the row kernel vs a float64 conv, fp32-grade: f16x3
    (two fp16 pieces) and bf16x6 (three bf16 pieces; the 7x7/2 stem, whose halo patch does not fit the LDS in three pieces,
    in two row-parity passes per channel chunk)"""
    N, Cin, H, W, Cout, K, s, pad, act, use_bias, use_bn, use_res = case
    prec = ops.PREC_F16X3 if mode == "f16x3" else ops.PREC_BF16X6
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g) if use_bias else None
    bn = (torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1,
          torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5, 1e-3) if use_bn else None
    ref = F.conv2d(F.pad(x, (pad[2], pad[3], pad[0], pad[1])).double(), w.double(),
                   None if b is None else b.double(), stride=s)
    if bn is not None:
        ref = F.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, bn[4])
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + res.double()
    ref = {0: lambda t: t, 1: F.relu, 2: lambda t: t * torch.sigmoid(t)}[act](ref)
    assert ops.conv_supported(prec, K, s)
    pc = ops.pack_conv(dev(w), None if b is None else dev(b),
                       None if bn is None else tuple(dev(t) if isinstance(t, torch.Tensor) else t for t in bn),
                       s, pad, act, prec)
    assert pc.prec == prec
    out = ops.conv2d(to_act(ops, x), pc, res=None if res is None else to_act(ops, res))
    got = out.nchw().cpu().double()
    assert got.shape == ref.shape
    err = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30))
    assert err < 2e-6, err
    if mode == "f16x3":
        assert float(out.amax.cpu()) >= float(got.abs().max()) * (1 - 1e-6)


@pytest.mark.parametrize("xs,ws", [(1e-20, 1.0), (1e-6, 1e3), (1.0, 1e-12), (3e4, 1.0), (1e15, 1e15), ("outlier", 1.0),
                                   ("zero", 1.0)])
@pytest.mark.parametrize("K", [1, 3])
def test_conv_f16x3_dynamic_range(ops, xs, ws, K):
    """fp16 has 5 exponent bits: the f16x3 engine must stay fp32-grade for ANY operand magnitude (exact
    power-of-two rescaling from the tracked |max| and per-channel weight scales), including a tensor whose
    |max| is an outlier 10^4 above the bulk, and must report max|out| of what it wrote."""
    g = torch.Generator().manual_seed(7 + K)
    N, Cin, H, W, Cout = 2, 48, 19, 45, 72
    x = torch.randn(N, Cin, H, W, generator=g)
    if xs == "outlier":
        x[1, 5, 7, 9] = 1e4
    elif xs == "zero":
        x.zero_()
    else:
        x = x * xs
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5 * ws
    w = w * torch.logspace(-3, 3, Cout).view(-1, 1, 1, 1)            # per-channel spread of BN-folded weights
    b = torch.randn(Cout, generator=g) * float(w.abs().mean() * x.abs().mean() + 1e-30)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=K // 2)
    pc = ops.pack_conv(dev(w), dev(b), None, 1, K // 2, 0, ops.PREC_F16X3)
    xa = to_act(ops, x)
    out = ops.conv2d(xa, pc)
    got = from_act(out).double()
    assert torch.isfinite(got).all()
    assert float(xa.amax.item()) == float(x.abs().max())
    assert float(out.amax.item()) == float(got.abs().max())          # tracked bound = exact max of the output
    if xs == "zero":
        assert torch.equal(got, b.double().view(1, -1, 1, 1).expand_as(got).contiguous())
        return
    # per output channel (the weight scale differs by 10^6 across channels)
    err = (got - ref).pow(2).mean(dim=(0, 2, 3)).sqrt() / ref.pow(2).mean(dim=(0, 2, 3)).sqrt()
    assert float(err.max()) < (2e-5 if xs == "outlier" else 2e-6), f"relative rms error {float(err.max()):.2e}"


def test_conv_slices_gate_and_rowmask(ops):
    """channel-slice input/output (zero-copy concat), SE gate on the A operand, row mask epilogue."""
    g = torch.Generator().manual_seed(5)
    N, H, W = 2, 7, 9
    x = torch.randn(N, 40, H, W, generator=g)
    gate = torch.rand(N, 24, generator=g)
    mask = (torch.rand(N * H * W, generator=g) > 0.3).float()
    w = torch.randn(16, 24, 1, 1, generator=g) / 5
    ref = F.conv2d(x[:, 8:32] * gate.view(N, 24, 1, 1), w) * mask.view(N, 1, H, W)
    xa = to_act(ops, x).slice(8, 24)
    outbuf = ops.Act.empty(N, H, W, 16, "cuda", cs=48)
    outbuf.buf.fill_(-7.0)
    pc = ops.pack_conv(dev(w), None, None, 1, 0, 0)
    ops.conv2d(xa, pc, out=outbuf.slice(20, 16), a_scale=dev(gate), row_mask=dev(mask))
    full = outbuf.buf.cpu()
    torch.testing.assert_close(full[..., 20:36].permute(0, 3, 1, 2), ref, rtol=2e-5, atol=2e-5)
    assert (full[..., :20] == -7).all() and (full[..., 36:] == -7).all()


@pytest.mark.parametrize("prec,tol", [("bf16x6", 1e-5), ("f16x3", 1e-5)])
@pytest.mark.parametrize("N,H,W,Cin,Cout,gated", [(4, 19, 40, 1152, 192, True), (16, 19, 38, 192, 1152, False),
                                                  (8, 38, 76, 112, 72, True), (2, 19, 38, 64, 64, True)])
def test_conv1x1_flat_retiling(ops, prec, tol, N, H, W, Cin, Cout, gated):
    """1x1 convs on maps that the 8 x 32 pixel tiles cover badly are re-tiled as one flat image of width 32 (when
    N*H*W is a multiple of 32; the last case is not and keeps the 2-D tiles): squeeze-excite gate per image (tiles that
    straddle two images), residual, bias, running |max|"""
    g = torch.Generator().manual_seed(N * H + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    gate = torch.rand(N, Cin, generator=g) + 0.25 if gated else None
    res = torch.randn(N, Cout, H, W, generator=g)
    xin = x.double() * (gate.double().view(N, Cin, 1, 1) if gated else 1.0)
    ref = F.conv2d(xin, w.double(), b.double()) + res.double()
    xa = to_act(ops, x)
    xa.amax = dev(x.abs().max().reshape(1))
    pc = ops.pack_conv(dev(w), dev(b), None, 1, 0, 0, {"bf16x6": ops.PREC_BF16X6, "f16x3": ops.PREC_F16X3}[prec])
    old = ops.TRACK_AMAX
    ops.TRACK_AMAX = True
    try:
        out = ops.conv2d(xa, pc, res=to_act(ops, res), a_scale=dev(gate) if gated else None)
    finally:
        ops.TRACK_AMAX = old
    got = from_act(out).double()
    rel = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert rel < tol, f"{prec}: relative rms error {rel:.2e}"
    assert float((got - ref).abs().max()) < 50 * tol * float(ref.abs().max())
    assert float(out.amax) >= float(ref.abs().max()) * (1 - 1e-5)


@pytest.mark.parametrize("prec", ["bf16x6", "bf16x3", "bf16"])
@pytest.mark.parametrize("N,H,W,Cin,Cout,gated", [(8, 19, 38, 1152, 192, True), (3, 19, 38, 192, 1152, False), (1, 38, 76, 672, 112, True),
                                                  (2, 5, 7, 72, 30, True), (2, 9, 13, 100, 66, False), (5, 19, 38, 64, 130, True)])
def test_conv1x1_deep_prefetch_kernel_is_bit_identical(ops, monkeypatch, prec, N, H, W, Cin, Cout, gated):
    """conv1x1_deep_kernel (flat 128-pixel tiles, a 4-chunk register ring of A / gate / weight tiles; csrc/conv_patch.hip) forms the
    same pieces and adds them in the same order as conv_patch_kernel<1, ...>: outputs are equal bit for bit -- pixel counts that
    are no multiple of 128 or 32, Cin no multiple of 16 (a partial last chunk), Cout no multiple of 4 (scalar epilogue) or of 64,
    squeeze-excite gates of tiles that straddle images, residual, bias, activation; both tile widths."""
    g = torch.Generator().manual_seed(N * H + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    gate = torch.rand(N, Cin, generator=g) + 0.25 if gated else None
    res = torch.randn(N, Cout, H, W, generator=g)
    P = {"bf16x6": ops.PREC_BF16X6, "bf16x3": ops.PREC_BF16X3, "bf16": ops.PREC_BF16}[prec]
    pc = ops.pack_conv(dev(w), dev(b), None, 1, 0, ops.ACT_SWISH, P)
    xa, ra = to_act(ops, x), to_act(ops, res)
    outs = {}
    for mode, bn in (("0", "0"), ("2", "64"), ("2", "128"), ("1", "0")):
        monkeypatch.setenv("CRESTE_CONV1X1_DEEP", mode)
        monkeypatch.setenv("CRESTE_CONV1X1_DEEP_BN", bn)
        outs[mode, bn] = ops.conv2d(xa, pc, res=ra, a_scale=dev(gate) if gated else None).buf.clone()
    ref = outs["0", "0"]
    for k, v in outs.items():
        assert torch.equal(v, ref), k
    want = F.silu(F.conv2d(x.double() * (gate.double().view(N, Cin, 1, 1) if gated else 1.0), w.double(), b.double()) + res.double())
    got = from_act(ops.Act(ref, Cout, 0)).double()
    tol = {"bf16x6": 1e-5, "bf16x3": 3e-4, "bf16": 2e-2}[prec]
    assert float((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()) < tol


def test_conv_winograd4_narrow_tiles_on_small_maps_are_bit_identical(ops, monkeypatch):
    """F(4x4,3x3) GEMM: where the 256-wide tiling leaves <= 512 items (the BEV trunk's 256 -> 256 convs at 32 x 32), 128-wide tiles
    double the workgroups (CRESTE_W4_SMALL_ITEMS); per output the K order is unchanged: same bits."""
    g = torch.Generator().manual_seed(5)
    x = to_act(ops, torch.randn(2, 256, 32, 32, generator=g))
    w = torch.randn(256, 256, 3, 3, generator=g) / (256 * 9) ** 0.5
    pc = ops.pack_conv(dev(w), None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
    outs = []
    for items in ("0", "512"):
        monkeypatch.setenv("CRESTE_W4_SMALL_ITEMS", items)
        outs.append(ops.conv2d(x, pc).buf.clone())
    assert float(outs[0].abs().max()) > 0.1
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("K,s,pad", [(3, 1, (1, 1, 1, 1)), (3, 2, (0, 1, 0, 1)), (5, 2, (1, 2, 1, 2)),
                                     (5, 1, (2, 2, 2, 2))])
def test_dwconv(ops, K, s, pad):
    g = torch.Generator().manual_seed(K * 10 + s)
    N, Cc, H, W = 2, 96, 21, 17
    x = torch.randn(N, Cc, H, W, generator=g)
    w = torch.randn(Cc, 1, K, K, generator=g) / K
    b = torch.randn(Cc, generator=g)
    ref = F.conv2d(F.pad(x, (pad[2], pad[3], pad[0], pad[1])), w, b, stride=s, groups=Cc)
    ref = ref * torch.sigmoid(ref)
    wt = w.view(Cc, K * K).t().contiguous()
    out = ops.dwconv2d(to_act(ops, x), dev(wt), dev(b), K, s, pad, ops.ACT_SWISH)
    torch.testing.assert_close(from_act(out), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("Cc,K,s,pad", [(96, 3, 2, (0, 1, 0, 1)), (240, 5, 1, (2, 2, 2, 2)), (1152, 3, 1, (1, 1, 1, 1)),
                                        (672, 5, 2, (1, 2, 1, 2)), (40, 5, 1, (2, 2, 2, 2)), (168, 3, 1, (1, 1, 1, 1))])
def test_dwconv_se_fused(ops, Cc, K, s, pad):
    g = torch.Generator().manual_seed(Cc + K)
    N, H, W, Cse = 2, 45, 31, 12
    x = torch.randn(N, Cc, H, W, generator=g)
    w = torch.randn(Cc, 1, K, K, generator=g) / K
    b = torch.randn(Cc, generator=g)
    w1, b1 = torch.randn(Cse, Cc, generator=g) / Cc ** 0.5, torch.randn(Cse, generator=g)
    w2, b2 = torch.randn(Cc, Cse, generator=g) / Cse ** 0.5, torch.randn(Cc, generator=g)
    y = F.conv2d(F.pad(x, (pad[2], pad[3], pad[0], pad[1])), w, b, stride=s, groups=Cc)
    y = y * torch.sigmoid(y)
    m = y.double().mean(dim=(2, 3))
    h = m @ w1.double().t() + b1.double()
    h = h * torch.sigmoid(h)
    gate_ref = torch.sigmoid(h @ w2.double().t() + b2.double())
    out, gate = ops.dwconv2d_se(to_act(ops, x), dev(w.view(Cc, K * K).t().contiguous()), dev(b), K, s, pad,
                                ops.ACT_SWISH, dev(w1), dev(b1), dev(w2), dev(b2))
    torch.testing.assert_close(from_act(out), y, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gate.cpu().double(), gate_ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("Cin,K,s,pad,HW", [(16, 3, 2, (0, 1, 0, 1), (45, 71)), (24, 3, 1, (1, 1, 1, 1), (37, 50)),
                                            (24, 5, 2, (1, 2, 1, 2), (40, 33)), (40, 5, 1, (2, 2, 2, 2), (23, 19)),
                                            (40, 3, 2, (0, 1, 0, 1), (30, 41)), (16, 5, 1, (2, 2, 2, 2), (9, 7))])
def test_mbconv_expand_depthwise_fused(ops, Cin, K, s, pad, HW):
    """expand 1x1 + swish -> depthwise KxK + swish -> squeeze-excite gate in one pass (csrc/mbconv.hip) against float64
    torch: ragged strips / bands (sizes that are no multiple of the tile), an input SLICE of a wider buffer, the
    zero padding of the EXPANDED map (bias + swish must not leak into the border), the running |max|."""
    g = torch.Generator().manual_seed(Cin * 100 + K * 10 + s)
    N, (H, W), Cexp, Cse = 3, HW, 6 * Cin, max(1, Cin // 4)
    x = torch.randn(N, Cin, H, W, generator=g)
    we, be = torch.randn(Cexp, Cin, generator=g) / Cin ** 0.5, torch.randn(Cexp, generator=g) + 0.5
    wd, bd = torch.randn(Cexp, 1, K, K, generator=g) / K, torch.randn(Cexp, generator=g)
    w1, b1 = torch.randn(Cse, Cexp, generator=g) / Cexp ** 0.5, torch.randn(Cse, generator=g)
    w2, b2 = torch.randn(Cexp, Cse, generator=g) / Cse ** 0.5, torch.randn(Cexp, generator=g)
    e = F.conv2d(x.double(), we.double().view(Cexp, Cin, 1, 1), be.double())
    e = e * torch.sigmoid(e)
    y = F.conv2d(F.pad(e, (pad[2], pad[3], pad[0], pad[1])), wd.double(), bd.double(), stride=s, groups=Cexp)
    y = y * torch.sigmoid(y)
    h = y.mean(dim=(2, 3)) @ w1.double().t() + b1.double()
    gate_ref = torch.sigmoid((h * torch.sigmoid(h)) @ w2.double().t() + b2.double())
    wide = torch.randn(N, H, W, Cin + 8, generator=g)                      # the block's input is a channel slice
    wide[..., 4:4 + Cin] = x.permute(0, 2, 3, 1)
    xa = ops.Act(dev(wide), Cin, 4)
    old = ops.TRACK_AMAX
    ops.TRACK_AMAX = True
    try:
        out, gate = ops.mbconv_expand_dw_se(xa, dev(we.t().contiguous()), dev(be), dev(wd.view(Cexp, K * K).t().contiguous()),
                                            dev(bd), K, s, pad, dev(w1), dev(b1), dev(w2), dev(b2))
    finally:
        ops.TRACK_AMAX = old
    torch.testing.assert_close(from_act(out).double(), y, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(gate.cpu().double(), gate_ref, rtol=1e-5, atol=2e-6)
    assert abs(float(out.amax) - float(y.abs().max())) <= 1e-5 * float(y.abs().max())


@pytest.mark.parametrize("HW,pad", [((64, 130), (0, 1, 0, 1)), ((45, 37), (1, 1, 1, 1)), ((17, 9), (0, 1, 1, 1))])
def test_stem_depthwise_fused(ops, HW, pad):
    """3x3/2 stem conv of the RGB-D image + swish -> depthwise 3x3 + swish -> SE gate in one pass (csrc/mbconv.hip)
    against float64 torch, on sizes that are no multiple of the strip / band tiling, with asymmetric 'same' padding."""
    g = torch.Generator().manual_seed(HW[0])
    N, (H, W), C1, Cse = 3, HW, 32, 8
    x = torch.randn(N, 4, H, W, generator=g)
    ws, bs = torch.randn(C1, 4, 3, 3, generator=g) / 6, torch.randn(C1, generator=g) * 0.5
    wd, bd = torch.randn(C1, 1, 3, 3, generator=g) / 3, torch.randn(C1, generator=g)
    w1, b1 = torch.randn(Cse, C1, generator=g) / C1 ** 0.5, torch.randn(Cse, generator=g)
    w2, b2 = torch.randn(C1, Cse, generator=g) / Cse ** 0.5, torch.randn(C1, generator=g)
    e = F.conv2d(F.pad(x.double(), (pad[2], pad[3], pad[0], pad[1])), ws.double(), bs.double(), stride=2)
    e = e * torch.sigmoid(e)
    y = F.conv2d(e, wd.double(), bd.double(), padding=1, groups=C1)
    y = y * torch.sigmoid(y)
    h = y.mean(dim=(2, 3)) @ w1.double().t() + b1.double()
    gate_ref = torch.sigmoid((h * torch.sigmoid(h)) @ w2.double().t() + b2.double())
    xa = to_act(ops, x)
    assert xa.cs == 4
    out, gate = ops.stem_dw_se(xa, dev(ws.permute(2, 3, 1, 0).reshape(36, C1)), dev(bs), pad,
                               dev(wd.view(C1, 9).t().contiguous()), dev(bd), (1, 1, 1, 1), dev(w1), dev(b1), dev(w2), dev(b2))
    assert (out.H, out.W) == tuple(y.shape[-2:])
    torch.testing.assert_close(from_act(out).double(), y, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(gate.cpu().double(), gate_ref, rtol=1e-5, atol=2e-6)


def test_mbconv_fused_rejects_what_it_is_not_built_for(ops):
    lib = ops._lib.load()
    assert lib.creste_mbconv_partial_count(2, 10, 10, 32, 192, 3, 1) < 0          # 32 input channels
    assert lib.creste_mbconv_partial_count(2, 10, 10, 16, 96, 7, 1) < 0           # 7x7
    assert lib.creste_mbconv_partial_count(2, 10, 10, 40, 480, 3, 1) < 0          # more than 256 expanded channels
    x = ops.Act(torch.zeros(1, 8, 8, 32, device="cuda"), 32, 0)
    w = torch.zeros(32, 192, device="cuda")
    with pytest.raises(ops.HipLibraryError):
        ops.mbconv_expand_dw_se(x, w, w[0], torch.zeros(9, 192, device="cuda"), w[0], 3, 1, (1, 1, 1, 1), w[:8], w[0, :8],
                                w[:, :8].contiguous(), w[0])


@pytest.mark.parametrize("Cc,Cse,HW", [(32, 8, (40, 52)), (96, 4, (64, 70)), (1152, 48, (5, 7))])
def test_se_gate(ops, Cc, Cse, HW):
    g = torch.Generator().manual_seed(Cc)
    N = 3
    x = torch.randn(N, Cc, *HW, generator=g)
    w1, b1 = torch.randn(Cse, Cc, generator=g) / Cc ** 0.5, torch.randn(Cse, generator=g)
    w2, b2 = torch.randn(Cc, Cse, generator=g) / Cse ** 0.5, torch.randn(Cc, generator=g)
    m = x.double().mean(dim=(2, 3))
    h = m @ w1.double().t() + b1.double()
    h = h * torch.sigmoid(h)
    ref = torch.sigmoid(h @ w2.double().t() + b2.double())
    got = ops.se_gate(to_act(ops, x), dev(w1), dev(b1), dev(w2), dev(b2)).cpu()
    torch.testing.assert_close(got.double(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("sf", [2, 4, (128 / 64, 153 / 76)])
def test_upsample_concat(ops, sf):
    g = torch.Generator().manual_seed(3)
    H1, W1 = (64, 76) if isinstance(sf, tuple) else (9, 11)
    x1 = torch.randn(2, 8, H1, W1, generator=g)
    up = torch.nn.Upsample(scale_factor=sf, mode="bilinear", align_corners=False)(x1)
    Ho, Wo = up.shape[-2:]
    skip = torch.randn(2, 4, Ho, Wo, generator=g)
    ref = torch.cat([skip, up], dim=1)
    sfh, sfw = sf if isinstance(sf, tuple) else (sf, sf)
    rh, rw = np.float32(1.0 / sfh), np.float32(1.0 / sfw)     # torch: scale = (float)(1.0 / scale_factor)
    out = ops.upsample_concat(to_act(ops, x1), to_act(ops, skip), Ho, Wo, rh, rw)
    torch.testing.assert_close(from_act(out), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("N,C1,C2,H1,W1", [(2, 40, 8, 96, 100), (1, 472, 24, 40, 410), (3, 8, 0, 5, 7), (1, 256, 0, 130, 128)])
def test_upsample2x_concat_ring_and_register_kernels(ops, N, C1, C2, H1, W1):
    """exact 2x: the LDS-ring kernel (wide maps; ragged strips and bands, clamped borders) and the register kernel
    (small maps) against torch, plus the running |max|"""
    g = torch.Generator().manual_seed(H1 * W1)
    x1 = torch.randn(N, C1, H1, W1, generator=g)
    up = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=False)
    skip = torch.randn(N, C2, 2 * H1, 2 * W1, generator=g) * 3 if C2 else None
    ref = torch.cat([skip, up], dim=1) if C2 else up
    old = ops.TRACK_AMAX
    ops.TRACK_AMAX = True
    try:
        out = ops.upsample_concat(to_act(ops, x1), to_act(ops, skip) if C2 else None, 2 * H1, 2 * W1, 0.5, 0.5)
    finally:
        ops.TRACK_AMAX = old
    torch.testing.assert_close(from_act(out), ref, rtol=1e-6, atol=1e-6)
    if out.amax is not None:
        assert float(out.amax) >= float(ref.abs().max()) * (1 - 1e-6)


def test_maxpool_affine_resize(ops):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 40, 16, 24, generator=g)
    ref = F.max_pool2d(x, 2, 2)[:, :, :4]
    out = ops.maxpool2(to_act(ops, x), Ho=4, Wo=12)
    assert torch.equal(from_act(out), ref)
    sc, sh = torch.rand(40, generator=g) + 0.5, torch.randn(40, generator=g)
    ref2 = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    torch.testing.assert_close(from_act(ops.affine_act(to_act(ops, x), dev(sc), dev(sh), ops.ACT_RELU)), ref2,
                               rtol=1e-6, atol=1e-6)
    r = torch.rand(3, 1, 8, 16, generator=g)
    full = torch.zeros(3, 1, 32, 32)
    full[:, :, :16] = F.interpolate(r, size=(16, 32), mode="bilinear", align_corners=False)
    o = torch.zeros(3, 32, 32, device="cuda")
    ops.resize_plane(dev(r[:, 0]), 16, 32, 32, 0.5, 0.5, o)
    torch.testing.assert_close(o.cpu(), full[:, 0], rtol=1e-6, atol=1e-6)


def test_depth_expectation(ops, golden):
    g = golden("utils.npz")
    logits = g.t("logits")                                     # [2,128,6,7]
    bins_v = torch.linspace(300, 25600, 128)
    d, b = ops.depth_expectation(to_act(ops, logits), dev(bins_v))
    torch.testing.assert_close(d.cpu() * 1000, g.t("metric_depth_mm"), rtol=1e-5, atol=1e-3)
    assert torch.equal(b.cpu(), logits.argmax(dim=1))


def _splat_inputs(golden, name):
    g = golden(name)
    sd = golden("splat_small.npz").sd()
    return g, sd


@pytest.mark.parametrize("name", ["splat_small.npz", "splat_wide.npz"])
def test_pixel_geometry_and_splat(ops, golden, name):
    g, sd = _splat_inputs(golden, name)
    depth, p2p = g.t("depth"), g.t("p2p")
    B, _, Hs, Ws = depth.shape
    bounds = torch.cat([sd["min_bound"].view(-1), sd["max_bound"].view(-1)])
    zbuf = ops.Act.empty(B, Hs, Ws, 32, "cuda", cs=288)
    xyz, mask = ops.pixel_geometry(dev(depth[:, 0]), dev(p2p[:, 0]), dev(bounds),
                                   dev(sd["z_proj.0.weight"].view(-1)), dev(sd["z_proj.0.bias"]),
                                   dev(sd["z_proj.2.weight"]), dev(sd["z_proj.2.bias"]), zbuf.slice(256, 32))
    ref_xyz = g.t("xyz")[:, 0].permute(0, 2, 3, 1).reshape(B, Hs * Ws, 3)
    assert torch.equal(xyz.cpu(), ref_xyz)                     # bit-exact fma chain
    assert torch.equal(mask.cpu().bool(), g.t("mask").view(B, -1))
    # z features vs the reference MLP
    z = ref_xyz[..., 2:3]
    zf = F.relu(F.linear(F.relu(F.linear(z, sd["z_proj.0.weight"], sd["z_proj.0.bias"])),
                         sd["z_proj.2.weight"], sd["z_proj.2.bias"]))
    torch.testing.assert_close(zbuf.buf[..., 256:].cpu().view(B, -1, 32), zf, rtol=1e-5, atol=1e-5)

    # splat of the reference's own fused features -> same coords (bit-exact), same sums (sorted order)
    fused = (g.t("fused") * g.t("mask"))[:, 0]                 # [B,96,Hs,Ws]
    fa = to_act(ops, fused)
    coords, bev, dens = ops.bev_splat(xyz, fa, (12.8, 12.8), (np.float32(0.1), np.float32(0.1)), 256, 256)
    assert torch.equal(coords.cpu(), g.t("bev_coords"))
    assert torch.equal(coords.cpu().floor().long(), g.t("bev_coords").floor().long())
    assert torch.equal(dens.cpu().view(B, 1, 256, 256), g.t("bev_densities"))
    idx = g.t("touched_idx")
    got = bev.buf.cpu()[idx[:, 0], idx[:, 1], idx[:, 2]]
    assert torch.equal(got, g.t("touched_feats"))              # reference summation order reproduced
    torch.testing.assert_close(bev.buf.abs().sum().cpu(), g.t("bev_features_abs_sum"), rtol=1e-5, atol=0)


def test_splat_plan_is_reusable_and_independent_of_the_features(ops, golden):
    """creste_bev_splat_plan_f32 needs only the points (the model enqueues it ahead of the fusion conv): ONE plan serves
    any number of gathers, each equal to the one-call splat of the same features, bit for bit, in every scatter mode."""
    g, sd = _splat_inputs(golden, "splat_wide.npz")
    xyz = dev(g.t("xyz")[:, 0].permute(0, 2, 3, 1).reshape(g.t("xyz").shape[0], -1, 3).contiguous())
    off, vox = (12.8, 12.8), (np.float32(0.1), np.float32(0.1))
    plan = ops.bev_splat_plan(xyz, off, vox, 256, 256)
    assert torch.equal(plan.coords.cpu(), g.t("bev_coords"))
    gen = torch.Generator().manual_seed(9)
    fa = to_act(ops, (g.t("fused") * g.t("mask"))[:, 0])
    fb = to_act(ops, torch.randn(g.t("fused")[:, 0].shape, generator=gen))
    for feats in (fa, fb, fa):
        for mode in ("mean", "sum", "max"):
            bev, dens = ops.bev_splat_gather(plan, feats, 1.0, mode)
            c1, b1, d1 = ops.bev_splat(xyz, feats, off, vox, 256, 256, 1.0, mode)
            assert torch.equal(bev.buf, b1.buf) and torch.equal(dens, d1) and torch.equal(plan.coords, c1)
    with pytest.raises(Exception):
        ops.bev_splat_gather(plan, to_act(ops, torch.randn(1, 96, 3, 5)), 1.0, "mean")      # rows != points of the plan


@pytest.mark.parametrize("name", ["vi_a.npz", "vi_b.npz"])
def test_value_iteration(ops, golden, name):
    g = golden(name)
    r = g.t("r")[:, 0]
    v, q, pi, sweeps = ops.value_iteration(dev(r), float(g["discount"]), float(g["threshold"]))
    n = int(sweeps.item())
    # EXACTLY the reference loop's count (vin.py:68-74) on the reference-generated reward maps: 683 / 688 sweeps.  (The count is an
    # integer the rounding can move: the same loop in float64 stops after 682 / 686, scripts/vi_golden_sweeps.py -- which is why the
    # end-to-end model test, whose reward map itself differs from the oracle's at float-noise level, accepts +-1.)
    assert n == int(g["sweeps"]), (n, int(g["sweeps"]))
    torch.testing.assert_close(v.cpu(), g.t("v")[:, 0], rtol=2e-5, atol=2e-3)
    torch.testing.assert_close(q.cpu(), g.t("q"), rtol=2e-5, atol=2e-3)
    torch.testing.assert_close(pi.cpu(), g.t("policy"), rtol=0, atol=1e-4)
    torch.testing.assert_close(pi.sum(dim=1).cpu(), torch.ones_like(r), rtol=0, atol=1e-6)


@pytest.mark.parametrize("name,zts", [("svf.npz", False), ("svf_zts.npz", True)])
def test_expected_svf(ops, golden, name, zts):
    g = golden(name)
    pol = golden("vi_b.npz").t("policy")
    expert = g.t("expert")[:, :, :2, 2].contiguous()
    fov = g.t("fov_mask")[0, 0].to(torch.uint8)
    svf, states, grid = ops.expected_svf(dev(pol), dev(expert), dev(fov), 50, 2.0, 0.005, True, zts)
    assert torch.equal(states.cpu(), g.t("state_preds"))
    assert torch.equal(grid.cpu(), g.t("state_preds_grid"))
    torch.testing.assert_close(svf.cpu(), g.t("exp_svf"), rtol=1e-5, atol=1e-6)


def test_conv_winograd_large_batch_is_sliced(ops, monkeypatch):
    """inputs of >= 2^30 elements go through the Winograd kernels in batch slices (32-bit byte offsets in the loader);
    exercised here by lowering the host-side bound."""
    import builtins
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 32, 12, 16, generator=g)
    w = torch.randn(32, 32, 3, 3, generator=g) / 17.0
    pc = ops.pack_conv(dev(w), None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD)
    whole = from_act(ops.conv2d(to_act(ops, x), pc))
    src = open(ops.__file__).read().replace("(1 << 30)", "(12 * 16 * 32 * 2 + 1)")      # two images per slice
    ns = {"__name__": ops.__name__, "__package__": ops.__package__, "__file__": ops.__file__}
    exec(compile(src, ops.__file__, "exec"), ns)
    sliced = from_act(ns["conv2d"](ns["nchw_to_nhwc"](dev(x)), ns["PackedConv"](**pc.__dict__)))
    assert torch.equal(whole, sliced)
    ref = F.relu(F.conv2d(x.double(), w.double(), padding=1))
    assert float((whole.double() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("N,C1,C2,H1,W1,Cout", [
    (2, 256, 0, 16, 24, 128),      # DeconvHead.up2: no skip connection
    (1, 320, 176, 19, 38, 256),    # Up block: skip + upsampled channels, pair chunks straddling the boundary (176 = 5.5 x 32)
    (3, 256, 64, 9, 11, 132),      # odd low-resolution extents: partial tiles, borders everywhere
])
def test_conv_winograd4_fused_upsample_concat(ops, N, C1, C2, H1, W1, Cout):
    """F(4x4,3x3) conv over cat([skip, up2x(x1)]) formed inside the input transform (ops.LazyUpCat) == the same conv over
    the materialised tensor, bit for bit (reference Up.forward effnet.py:16-23 / DeconvHead.up2 inpainting.py:81)."""
    g = torch.Generator().manual_seed(N * 1000 + C1)
    x1 = to_act(ops, torch.randn(N, C1, H1, W1, generator=g))
    skip = to_act(ops, torch.randn(N, C2, 2 * H1, 2 * W1, generator=g)) if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, generator=g) / ((C1 + C2) * 9) ** 0.5
    pc = ops.pack_conv(dev(w), None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
    lazy = ops.upsample_concat_lazy(x1, skip, 2 * H1, 2 * W1, 0.5, 0.5)
    assert lazy.exact2x
    fused = ops.conv2d(lazy, pc)
    assert lazy._mat is None                      # never materialised
    plain = ops.conv2d(ops.upsample_concat(x1, skip, 2 * H1, 2 * W1, 0.5, 0.5), pc)
    assert torch.equal(fused.buf, plain.buf)
    # the policy's other branches materialise once and cache
    pcd = ops.pack_conv(dev(w), None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_DIRECT)
    y = ops.conv2d(lazy, pcd)
    assert lazy._mat is not None and ops.conv2d(lazy, pcd).buf.shape == y.buf.shape


@pytest.mark.parametrize("N,Cin,H,W,prec", [
    (2, 256, 19, 38, "bf16x6"),            # the distillation head's widths (reference distillation.py:179), ragged last workgroup
    (1, 96, 16, 16, "bf16x6"),             # a narrow first layer: three steps (a partial group of the four-deep input prefetch)
    (3, 256, 8, 32, "bf16x3"),
    (16, 256, 152, 304, "bf16x6"),         # BASELINE configs[1]: the distillation head at batch 16 of 608 x 1216 (152 x 304 maps)
])
def test_conv1x1_chain_of_three_layers_in_one_kernel(ops, N, Cin, H, W, prec):
    """MultiLayerConv(kernels [1,1,1], dims [Cin,128,128,128]): three 1x1 conv(+bias) + BatchNorm(eval) + ReLU layers as ONE launch whose
    hidden activations stay in the accumulator registers (creste_conv1x1_chain3_f32) -- against float64, against the three launches
    of the 1x1 engine it replaces (same piece products, another summation order), and into a channel slice."""
    import torch.nn.functional as F
    P = getattr(ops, "PREC_" + prec.upper())
    g = torch.Generator().manual_seed(Cin + H)
    xt = torch.randn(N, Cin, H, W, generator=g)
    dims = [Cin, 128, 128, 128]
    layers, ref = [], xt.double()
    for i in range(3):
        w = torch.randn(dims[i + 1], dims[i], 1, 1, generator=g) / dims[i] ** 0.5
        b = torch.randn(128, generator=g) * 0.2
        gamma, beta = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.3
        mean, var = torch.randn(128, generator=g) * 0.2, torch.rand(128, generator=g) + 0.5
        layers.append((dev(w), dev(b), (dev(gamma), dev(beta), dev(mean), dev(var), 1e-5)))
        scale = gamma.double() / torch.sqrt(var.double() + 1e-5)
        ref = torch.relu((F.conv2d(ref, w.double(), b.double()) - mean.double()[None, :, None, None]) * scale[None, :, None, None]
                         + beta.double()[None, :, None, None])
    x = to_act(ops, xt)
    assert ops.conv1x1_chain3_supported(P, dims)
    pk = ops.pack_conv1x1_chain3(layers, P)
    y = ops.conv1x1_chain3(x, pk)
    got = y.nchw().double().cpu()
    tol = 1e-5 if prec == "bf16x6" else 2e-3
    assert float((got - ref).abs().max()) < tol * float(ref.abs().max()), float((got - ref).abs().max())
    z = x
    for (w, b, bn) in layers:
        z = ops.conv2d(z, ops.pack_conv(w, b, bn, 1, 0, ops.ACT_RELU, P))
    assert float((z.buf - y.buf).abs().max()) < 2 * tol * float(ref.abs().max())
    wide = ops.Act(torch.full((N, H, W, 140), 5.0, device="cuda"), 128, 8)
    ops.conv1x1_chain3(x, pk, out=wide)
    assert torch.equal(wide.buf[..., 8:136], y.buf) and bool((wide.buf[..., :8] == 5).all()) and bool((wide.buf[..., 136:] == 5).all())
    # the input as a channel slice of a wider buffer (the encoder's 288-channel fusion buffer holds the head's input)
    xw = ops.Act(torch.randn(N, H, W, Cin + 32, device="cuda"), Cin, 16)
    xw.buf[..., 16:16 + Cin] = x.buf
    assert torch.equal(ops.conv1x1_chain3(xw, pk).buf, y.buf)


@pytest.mark.parametrize("N,Cin,Cout,H,W,prec", [
    (2, 256, 128, 16, 24, "bf16x6"),       # DeconvHead.up2's channel counts, whole tiles
    (1, 64, 8, 5, 7, "bf16x6"),            # odd extents: partial tiles, every ring pixel near a corner, Cout of two quads
    (3, 128, 132, 33, 18, "bf16x6"),       # Cout = 33 quads per phase (528 phase channels), sides longer than one 32-pixel chunk
    (1, 64, 16, 2, 2, "bf16x6"),           # the smallest map: the 4 x 4 output is ring + corners only... plus 2 x 2 interior
    (2, 256, 128, 32, 32, "bf16x3"),
    (16, 256, 128, 128, 128, "bf16x6"),    # BASELINE configs[1]: the three BEV heads' up2 at batch 16 (128 x 128 -> 256 x 256)
])
def test_upsample_conv3x3_as_phase_convolutions(ops, N, Cin, Cout, H, W, prec):
    """`nn.Upsample(scale_factor=2, bilinear, align_corners=False) -> nn.Conv2d(3, padding=1) -> BatchNorm (eval) -> ReLU` (reference
    DeconvHead.up2, inpainting.py:56-60) as four phase convolutions on the LOW-resolution map + the border-ring correction
    (ops.upconv2x: CRESTE_CONV_REPLICATE_PAD | CRESTE_CONV_PHASE2X, creste_upconv2x_ring_fix_f32) against float64 PyTorch on the
    CPU, at the F(4x4) engine's tolerance, ring and corners included; also into a channel slice of a wider buffer, and against
    the conv over the upsampled map the other modes keep using."""
    import torch.nn.functional as F
    P = getattr(ops, "PREC_" + prec.upper())
    g = torch.Generator().manual_seed(N * 100 + Cin + Cout)
    xt = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
    mean, var = torch.randn(Cout, generator=g) * 0.2, torch.rand(Cout, generator=g) + 0.5
    up = F.interpolate(xt.double(), scale_factor=2, mode="bilinear", align_corners=False)
    scale = gamma.double() / torch.sqrt(var.double() + 1e-5)
    ref = torch.relu(F.conv2d(up, w.double(), padding=1) * scale[None, :, None, None] + (beta.double() - mean.double() * scale)[None, :, None, None])
    x = to_act(ops, xt)
    assert ops.upconv2x_supported(P, Cin, Cout)
    pu = ops.pack_upconv2x(dev(w), None, (dev(gamma), dev(beta), dev(mean), dev(var), 1e-5), ops.ACT_RELU, P)
    y = ops.upconv2x(x, pu)
    assert (y.N, y.H, y.W, y.C) == (N, 2 * H, 2 * W, Cout)
    got = y.nchw().double().cpu()
    tol = 3e-5 if prec == "bf16x6" else 2e-3
    err = (got - ref).abs()
    assert float(err.max()) < tol * float(ref.abs().max()), (float(err.max()), float(ref.abs().max()))
    ring = torch.ones(2 * H, 2 * W, dtype=torch.bool); ring[1:-1, 1:-1] = False
    assert float(err[:, :, ring].max()) < tol * float(ref.abs().max())
    rms = float((err ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    assert rms < (1e-5 if prec == "bf16x6" else 5e-4), rms
    # into a channel slice of a wider buffer (the neighbours stay untouched), twice: same bits
    wide = ops.Act(torch.full((N, 2 * H, 2 * W, Cout + 8), 7.0, device="cuda"), Cout, 4)
    ops.upconv2x(x, pu, out=wide)
    assert torch.equal(wide.buf[..., 4:4 + Cout], y.buf) and bool((wide.buf[..., :4] == 7).all()) and bool((wide.buf[..., 4 + Cout:] == 7).all())
    assert torch.equal(ops.upconv2x(x, pu).buf, y.buf)
    # the conv over the (lazily) upsampled map: the same operator to the engine's tolerance
    pc = ops.pack_conv(dev(w), None, (dev(gamma), dev(beta), dev(mean), dev(var), 1e-5), 1, 1, ops.ACT_RELU, P, algo=ops.ALGO_WINOGRAD4)
    z = ops.conv2d(ops.upsample_concat_lazy(x, None, 2 * H, 2 * W, 0.5, 0.5), pc)
    assert float((z.buf - y.buf).abs().max()) < 2 * tol * float(ref.abs().max())


@pytest.mark.parametrize("case", [
    ("wino4", 2, 128, 256, 20, 28),        # F(4x4) output transform: ragged tile groups
    ("wino4", 1, 144, 132, 37, 41),        # partial tiles in both directions, Cout = 33 quads (4 + 1/8 cout groups)
    ("1x1", 2, 96, 24, 40, 72),            # 64-channel tile, partial in x and y
    ("1x1", 3, 40, 240, 19, 38),           # flat re-tiled map, two 128-channel tiles
    ("1x1", 1, 496, 256, 24, 64),          # 256-wide tiles are narrowed to 128 for the statistics epilogue
])
def test_conv_output_statistics_for_batchnorm(ops, case):
    """creste_conv_desc.out_stats (ops.conv2d want_stats): the per-workgroup channel sums a conv leaves of its output give the
    training-mode BatchNorm's batch mean / variance without another pass (reference: nn.Conv2d -> nn.BatchNorm2d in train mode,
    train_pefree.py / train_ssc.py); the conv's own output is unchanged, bit for bit."""
    kind, N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = to_act(ops, torch.randn(N, Cin, H, W, generator=g) + 0.3)
    K = 3 if kind == "wino4" else 1
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    pc = ops.pack_conv(dev(w), dev(b), None, 1, K // 2, ops.ACT_NONE, ops.PREC_BF16X6,
                       algo=ops.ALGO_WINOGRAD4 if kind == "wino4" else ops.ALGO_DIRECT)
    plain = ops.conv2d(x, pc)
    assert plain.stats is None
    y = ops.conv2d(x, pc, want_stats=True)
    assert y.stats is not None and torch.equal(y.buf, plain.buf)
    part, rows = y.stats
    assert part.shape == (rows, 2, Cout)
    yd = y.buf.double().reshape(-1, Cout)
    s = part.double().sum(0).cpu()
    assert torch.allclose(s[0], yd.sum(0).cpu(), rtol=1e-6, atol=1e-4 * yd.shape[0] ** 0.5)
    assert torch.allclose(s[1], (yd * yd).sum(0).cpu(), rtol=2e-6)
    # two runs leave the same partial sums (one row per workgroup, fixed order: no atomics)
    y2 = ops.conv2d(x, pc, want_stats=True)
    assert torch.equal(y2.stats[0], part)


@pytest.mark.parametrize("N,Cin,Cmid,Cout,H,W,up", [
    (2, 128, 256, 128, 20, 28, False),     # one partial block in both directions (5 x 7 tiles)
    (1, 320, 496, 496, 37, 41, False),     # partial tiles (37 x 41 pixels), 31 chunks, two cout tiles
    (3, 144, 136, 132, 70, 132, False),    # several blocks per image (18 x 33 tiles), Cmid = 8.5 chunks: the padded channel quads
    (2, 256, 128, 256, 32, 64, True),      # exactly one block; conv 1 with the fused upsample input
    (1, 128, 256, 128, 152, 304, False),   # the up3 map: 38 x 76 tiles, 5 x 5 blocks with ragged right / bottom blocks
])
def test_conv_pair_fused_output_input_transform_is_bit_identical(ops, monkeypatch, N, Cin, Cmid, Cout, H, W, up):
    """conv3x3+ReLU -> conv3x3+ReLU (reference Up.conv, effnet.py:15-28) with conv 1's output transform writing conv 2's
    transformed input (CRESTE_CONV_EMIT_NEXT_V / V_VALID, csrc/conv_wino4.hip: wino4_outin_kernel) == the two separate
    calls, bit for bit: the intermediate tensor never exists, its values are formed with the same expressions."""
    g = torch.Generator().manual_seed(Cin + Cmid + H)
    w1 = torch.randn(Cmid, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    w2 = torch.randn(Cout, Cmid, 3, 3, generator=g) / (Cmid * 9) ** 0.5
    b1, b2 = torch.randn(Cmid, generator=g), torch.randn(Cout, generator=g)
    pc1 = ops.pack_conv(dev(w1), dev(b1), None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
    pc2 = ops.pack_conv(dev(w2), dev(b2), None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
    if up:
        x1 = to_act(ops, torch.randn(N, Cin - 64, H // 2, W // 2, generator=g))
        skip = to_act(ops, torch.randn(N, 64, H, W, generator=g))
        make = lambda: ops.upsample_concat_lazy(x1, skip, H, W, 0.5, 0.5)
    else:
        x = to_act(ops, torch.randn(N, Cin, H, W, generator=g))
        make = lambda: x
    assert ops.conv_pair_fusable(pc1, pc2)
    monkeypatch.setattr(ops, "FUSE_PAIR_MIN_FILL", 0.0)        # every case through the fused kernel, however ragged
    fused = ops.conv2d_pair(make(), pc1, pc2)
    monkeypatch.setattr(ops, "FUSE_CONV_PAIRS", False)
    assert not ops.conv_pair_fusable(pc1, pc2)
    plain = ops.conv2d_pair(make(), pc1, pc2)
    assert torch.equal(fused.buf, plain.buf)
    # and into a channel slice of a wider buffer, as the encoder's last Up block writes (effnet.py: the 288-channel fusion buffer)
    monkeypatch.setattr(ops, "FUSE_CONV_PAIRS", True)
    wide = ops.Act.empty(N, H, W, Cout + 32, "cuda")
    wide.buf.zero_()
    sl = ops.Act(wide.buf, Cout, 16)
    ops.conv2d_pair(make(), pc1, pc2, out=sl)
    assert torch.equal(wide.buf[..., 16:16 + Cout], plain.buf) and float(wide.buf[..., :16].abs().max()) == 0.0


@pytest.mark.parametrize("prec", ["bf16x6", "bf16x3"])
@pytest.mark.parametrize("N,Cin,Cout,H,W,up", [
    (2, 128, 256, 20, 28, False),      # 8 chunks, one ragged tile block
    (1, 496, 496, 37, 41, False),      # 31 chunks, partial tiles in both directions, two cout tiles
    (3, 144, 132, 16, 16, False),      # nchunk = 9 (odd), Cout tile of 128 (TN = 2), padded couts
    (2, 256, 128, 24, 40, True),       # fused upsample form of the input transform
])
def test_conv_winograd4_fp32_transformed_input_is_bit_identical(ops, monkeypatch, prec, N, Cin, Cout, H, W, up):
    """The default F(4x4,3x3) path keeps the transformed input V as fp32 and splits it into bf16 pieces inside the GEMM
    (csrc/conv_wino4.hip: wino4_gemm32_kernel, the continuous pipeline; CRESTE_W4_F32V=2: the per-item kernel).  Both must
    equal the pre-split form (V written as bf16 pieces by the input transform, CRESTE_W4_F32V=0) bit for bit: the same
    conversions and subtractions produce the same pieces, the same MFMAs in the same order the same products."""
    code = {"bf16x6": ops.PREC_BF16X6, "bf16x3": ops.PREC_BF16X3}[prec]
    g = torch.Generator().manual_seed(Cin + H)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    pc = ops.pack_conv(dev(w), dev(b), None, 1, 1, ops.ACT_RELU, code, algo=ops.ALGO_WINOGRAD4)
    if up:
        x1 = to_act(ops, torch.randn(N, Cin - 64, H // 2, W // 2, generator=g))
        skip = to_act(ops, torch.randn(N, 64, H, W, generator=g))
        make = lambda: ops.upsample_concat_lazy(x1, skip, H, W, 0.5, 0.5)
    else:
        x = to_act(ops, torch.randn(N, Cin, H, W, generator=g))
        make = lambda: x
    outs = {}
    for mode in ("0", "2", "1"):
        monkeypatch.setenv("CRESTE_W4_F32V", mode)
        for order in ("0", "3"):                       # workgroup orders of the transform kernels: placement only
            monkeypatch.setenv("CRESTE_W4_ORDER", order)
            for rs in ("0", "1"):                      # wave layout of the streaming GEMM (row split): same products, same order
                monkeypatch.setenv("CRESTE_W4_RS", rs)
                outs[mode, order, rs] = ops.conv2d(make(), pc).buf.clone()
    ref = outs["0", "0", "0"]
    assert float(ref.abs().max()) > 0.1
    for k, v in outs.items():
        assert torch.equal(v, ref), k


def test_conv_winograd4_shared_input_transform(ops):
    """three F(4x4,3x3) convs over ONE materialised upsample + concat (the BEV heads' first conv, inpainting.py:141-146):
    the input transform runs once (CRESTE_CONV_V_VALID) and every output equals the stand-alone conv bit for bit."""
    g = torch.Generator().manual_seed(11)
    x1 = to_act(ops, torch.randn(2, 256, 8, 12, generator=g))
    skip = to_act(ops, torch.randn(2, 64, 32, 48, generator=g))
    lazy = ops.upsample_concat_lazy(x1, skip, 32, 48, 0.25, 0.25)
    assert not lazy.exact2x
    outs, refs = [], []
    for Cout in (256, 256, 128):
        w = torch.randn(Cout, 320, 3, 3, generator=g) / (320 * 9) ** 0.5
        pc = ops.pack_conv(dev(w), None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
        outs.append(ops.conv2d(lazy, pc).buf.clone())
        refs.append(ops.conv2d(ops.upsample_concat(x1, skip, 32, 48, 0.25, 0.25), pc).buf.clone())
    assert lazy._w4 is not None
    for a, b in zip(outs, refs):
        assert torch.equal(a, b)
