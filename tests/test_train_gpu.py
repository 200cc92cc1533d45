"""GPU: the HIP training path of the IRL reward network (creste_public_amd/train_ops.py, csrc/train.hip)
against torch autograd on the CPU oracle in float64 -- primal output, first-order input gradient
(`autograd.grad(create_graph=True)`), and the parameter gradients of the full IRL-shaped objective
<D, r> + lambda * mean((||d sum(r)/dx||_2 - 1)^2) (second-order through the gradient penalty, training-mode
BatchNorm), plus the individual primitives (wgrad, BatchNorm forward/tangent/backward, pool, upsample)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from creste_public_amd.config import maxent_irl_cfg
from oracle import blocks as ob

pytestmark = pytest.mark.gpu


def _cfg():
    return maxent_irl_cfg()["traversability_head"]["net_kwargs"]["reward_cfg"]["net_kwargs"]


def _objective(net, x, D, lam):
    x = x.clone().requires_grad_(True)
    r = net(x)
    g = torch.autograd.grad(outputs=r.sum(), inputs=x, create_graph=True, retain_graph=True, only_inputs=True)[0]
    gp = ((g.norm(2, dim=1) - 1) ** 2).mean()
    loss = (D * r).sum() + lam * gp
    loss.backward()
    return r.detach(), g.detach(), loss.detach()


@pytest.mark.parametrize("shape,lam", [((2, 40, 32, 48), 0.1), ((3, 40, 20, 36), 0.0), ((8, 40, 64, 128), 0.01)])
def test_reward_net_training_step(shape, lam):
    from creste_public_amd.creste.models.blocks.conv import MultiScaleFCN
    torch.manual_seed(shape[2])
    net = MultiScaleFCN(_cfg())
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    ref = ob.MultiScaleFCN(_cfg()).double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    net = net.cuda().train()
    ref.train()
    x = torch.rand(shape) * 2
    D = torch.randn(shape[0], 1, shape[2], shape[3]) / (shape[2] * shape[3])

    r0, g0, l0 = _objective(ref, x.double(), D.double(), lam)
    r1, g1, l1 = _objective(net, x.cuda(), D.cuda(), lam)
    torch.cuda.synchronize()

    def close(a, b, tol, what):
        """95th percentile of |error| relative to the largest entry (< tol) and a loose bound on the rms error.
        One ReLU / max-pool decision that flips between fp32 and the float64 oracle (a pre-activation within 1e-7
        of zero somewhere among 10^5..10^6 values: observed) changes the gradient inside that unit's receptive
        field by O(1); a wrong formula or kernel is off everywhere, which is what the percentile detects."""
        a, b = a.double().cpu().flatten(), b.double().flatten()
        rms = float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))
        k = max(1, int(0.95 * a.numel()))
        p95 = float((a - b).abs().kthvalue(k).values / b.abs().max().clamp_min(1e-30))
        assert p95 < tol and rms < 1e-2, f"{what}: p95 |err| / max {p95:.2e}, rel rms {rms:.2e}"

    close(r1, r0, 2e-5, "reward")
    close(g1, g0, 1e-4, "d sum(r) / d x")
    close(l1, l0, 1e-4, "loss")
    ref_p = dict(ref.named_parameters())
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        # a flipped unit moves these sums by up to ~1e-3 (float32 oracle vs float64 oracle: 1.5e-4 on the
        # 8x64x128 case); the primitives are held to 2e-5..2e-4 in the tests below
        close(p.grad, ref_p[name].grad, 2e-3, f"grad of {name}")
    ref_b = dict(ref.named_buffers())
    for name, b in net.named_buffers():                      # running statistics follow nn.BatchNorm2d
        if b.dtype.is_floating_point:
            close(b, ref_b[name], 1e-5, name)
        else:
            assert int(b) == int(ref_b[name]), name


def test_graph_replay_matches_eager():
    """hipGraph replay of the four launch sequences: bit-identical parameters after 4 Adam steps."""
    from creste_public_amd.creste.models.blocks.conv import MultiScaleFCN
    torch.manual_seed(5)
    base = MultiScaleFCN(_cfg())
    xs = [torch.rand(4, 40, 32, 64) * 2 for _ in range(4)]
    Ds = [torch.randn(4, 1, 32, 64) / 2048 for _ in range(4)]
    results = []
    for graphs in (False, True):
        net = copy.deepcopy(base).cuda().train()
        net.train_graphs = graphs
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        losses = []
        for x, D in zip(xs, Ds):
            opt.zero_grad()
            _, _, loss = _objective(net, x.cuda(), D.cuda(), 0.1)
            opt.step()
            losses.append(float(loss))
        torch.cuda.synchronize()
        results.append((losses, {k: v.clone() for k, v in net.state_dict().items()}))
    assert results[0][0] == results[1][0], (results[0][0], results[1][0])
    for k, v in results[0][1].items():
        assert torch.equal(v, results[1][1][k]), k
    assert int(results[1][1]["prepool.0.norm.num_batches_tracked"]) == 4


def test_stale_backward_is_refused():
    from creste_public_amd.creste.models.blocks.conv import MultiScaleFCN
    net = MultiScaleFCN(_cfg()).cuda().train()
    x = torch.rand(1, 40, 16, 16, device="cuda", requires_grad=True)
    r_old = net(x)
    net(x)
    with pytest.raises(RuntimeError, match="stale forward"):
        r_old.sum().backward()


@pytest.mark.parametrize("N,H,W,Cin,Cout,K", [(2, 13, 17, 40, 64, 5), (1, 8, 9, 32, 16, 1), (3, 16, 20, 48, 1, 1),
                                              (2, 11, 7, 64, 32, 3),
                                              # row-walk wgrad (csrc/train.hip): several column segments with a ragged
                                              # last one, partial 16-channel tiles, every tap-split instantiation
                                              (1, 70, 150, 32, 32, 3), (2, 40, 64, 24, 40, 5), (2, 33, 130, 64, 64, 3),
                                              (1, 37, 66, 16, 24, 5), (3, 9, 200, 40, 64, 5), (1, 19, 65, 48, 8, 3), (2, 30, 70, 32, 32, 1),
                                              (1, 5, 129, 64, 40, 1), (2, 21, 33, 48, 3, 1), (1, 7, 5, 8, 4, 1), (2, 16, 16, 128, 6, 1)])
def test_conv_wgrad_and_dgrad(N, H, W, Cin, Cout, K):
    from creste_public_amd import train_ops as T
    from creste_public_amd.ops import Act
    g = torch.Generator().manual_seed(K + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g).double().requires_grad_(True)
    gy = torch.randn(N, Cout, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    y = F.conv2d(xr, w, padding=K // 2)
    y.backward(gy.double())
    conv = torch.nn.Conv2d(Cin, Cout, K, padding=K // 2, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(w.detach().float())
    op = T.ConvT(conv)
    ya = op.fwd(T.as_act(x.cuda()))
    torch.testing.assert_close(ya.nchw().cpu().double(), y.detach(), rtol=1e-4, atol=1e-4)
    grads = {}
    gx, _ = op.bwd(T.as_act(gy.cuda()), None, grads)
    torch.testing.assert_close(gx.nchw().cpu().double(), xr.grad, rtol=1e-4, atol=1e-4)
    gw = grads[id(conv.weight)].cpu().double()
    assert float((gw - w.grad).abs().max() / w.grad.abs().max()) < 2e-5
    op.bwd(T.as_act(gy.cuda()), None, grads, need_input=False)          # accumulate path
    assert float((grads[id(conv.weight)].cpu().double() - 2 * w.grad).abs().max() / w.grad.abs().max()) < 4e-5


@pytest.mark.parametrize("C,relu", [(1, True), (16, False), (40, True), (64, True)])
def test_batchnorm_train_forward_tangent_backward(C, relu):
    """forward / JVP / joint backward of training-mode BatchNorm against autograd (double backward)."""
    from creste_public_amd import train_ops as T
    g = torch.Generator().manual_seed(C)
    N, H, W = 3, 9, 14
    x = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
    xd = torch.randn(N, C, H, W, generator=g)
    gy = torch.randn(N, C, H, W, generator=g)
    gyd = torch.randn(N, C, H, W, generator=g)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
    ref = copy.deepcopy(bn).double().train()

    def f(xx):
        # BatchNorm from elementary ops: ATen's fused batch_norm double-backward treats the saved mean / invstd
        # as constants when differentiated once more, so jvp-then-backward through nn.BatchNorm2d is NOT the
        # true derivative (checked: first order, jvp and gxd agree with this form, gx does not)
        m = lambda z: z.mean(dim=(0, 2, 3), keepdim=True)      # noqa: E731
        mu = m(xx)
        var = m((xx - mu) ** 2)
        y = ref.weight.view(1, -1, 1, 1) * (xx - mu) * (var + ref.eps) ** -0.5 + ref.bias.view(1, -1, 1, 1)
        return F.relu(y) if relu else y
    xr2 = x.double().requires_grad_(True)
    xdr = xd.double().requires_grad_(True)
    yy, yyd = torch.autograd.functional.jvp(f, (xr2,), (xdr,), create_graph=True)
    ((gy.double() * yy).sum() + (gyd.double() * yyd).sum()).backward()

    bn = bn.cuda().train()
    op = T.BNT(bn, relu)
    ya = op.fwd(T.as_act(x.cuda()))
    yda = op.tan(T.as_act(xd.cuda()))
    torch.testing.assert_close(ya.nchw().cpu().double(), yy.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(yda.nchw().cpu().double(), yyd.detach(), rtol=1e-4, atol=1e-5)
    grads = {}
    gx, gxd = op.bwd(T.as_act(gy.cuda()), T.as_act(gyd.cuda()), grads)
    torch.testing.assert_close(gx.nchw().cpu().double(), xr2.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(gxd.nchw().cpu().double(), xdr.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(grads[id(bn.weight)].cpu().double(), ref.weight.grad, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(grads[id(bn.bias)].cpu().double(), ref.bias.grad, rtol=2e-4, atol=2e-4)
    ref(x.double())                                          # running statistics as nn.BatchNorm2d keeps them
    torch.testing.assert_close(bn.running_mean.cpu().double(), ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var.cpu().double(), ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("C", [3, 16, 144])
def test_batchnorm_swish_forward_backward(C):
    """BatchNorm + swish as one op (activation inside the BatchNorm kernels, swish'(z) recomputed from x in the backward):
    the backbone's BatchNorm2d -> MemoryEfficientSwish pairs, against float64 autograd"""
    from creste_public_amd import train_ops as T
    g = torch.Generator().manual_seed(100 + C)
    N, H, W = 2, 11, 13
    x = torch.randn(N, C, H, W, generator=g) * 1.5 - 0.3
    gy = torch.randn(N, C, H, W, generator=g)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 2.0); bn.bias.normal_(0, 0.5)
    ref = copy.deepcopy(bn).double().train()
    xr = x.double().requires_grad_(True)
    z = ref(xr)
    y = z * torch.sigmoid(z)
    y.backward(gy.double())
    bn = bn.cuda().train()
    op = T.BNT(bn, 2)
    ya = op.fwd(T.as_act(x.cuda()))
    torch.testing.assert_close(ya.nchw().cpu().double(), y.detach(), rtol=1e-4, atol=1e-5)
    grads = {}
    gx, gxd = op.bwd(T.as_act(gy.cuda()), None, grads)
    assert gxd is None
    torch.testing.assert_close(gx.nchw().cpu().double(), xr.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(grads[id(bn.weight)].cpu().double(), ref.weight.grad, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(grads[id(bn.bias)].cpu().double(), ref.bias.grad, rtol=2e-4, atol=2e-4)
    with pytest.raises(NotImplementedError):
        op.tan(T.as_act(x.cuda()))


def test_batchnorm_one_pass_statistics_on_hard_data():
    """the training-mode statistics are ONE read of the tensor (pivot-shifted moments, csrc/train.hip): channels with a
    mean far larger than their spread, a constant channel, a channel whose first pixel is an outlier, a big tensor"""
    from creste_public_amd import train_ops as T
    g = torch.Generator().manual_seed(0)
    N, C, H, W = 4, 8, 96, 130
    x = torch.randn(N, C, H, W, generator=g)
    x[:, 0] = x[:, 0] * 1e-2 + 100.0                          # mean / std = 1e4
    x[:, 1] = 3.25                                            # zero variance
    x[:, 2] = x[:, 2] * 0.1 - 7.0
    x[0, 2, 0, 0] = 500.0                                     # the first pixel (a pivot sample) is an outlier
    x[:, 3] *= 1e-3
    x[:, 4] = x[:, 4] * 50.0 + 1e3
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    op = T.BNT(bn, False)
    y = op.fwd(T.as_act(x.cuda())).nchw().cpu().double()
    xd = x.double()
    mu = xd.mean(dim=(0, 2, 3), keepdim=True)
    var = ((xd - mu) ** 2).mean(dim=(0, 2, 3), keepdim=True)
    ref = (xd - mu) / torch.sqrt(var + bn.eps)
    # tolerance: fp32 input resolution relative to the channel's spread (channel 0: 100 * 2^-24 / 0.01 = 6e-4)
    err = (y - ref).abs().amax(dim=(0, 2, 3))
    lim = torch.tensor([3e-3, 1e-6, 2e-3, 1e-4, 1e-3, 1e-4, 1e-4, 1e-4], dtype=torch.float64)
    assert (err <= lim).all(), err
    n = N * H * W
    torch.testing.assert_close(bn.running_mean.cpu().double(), 0.1 * mu.view(-1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var.cpu().double(), 0.9 + 0.1 * var.view(-1) * n / (n - 1), rtol=2e-4, atol=1e-7)


def test_pool_and_upsample_transposes():
    from creste_public_amd import train_ops as T
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 10, 14, generator=g)
    x[0, :, :4] = 0.0                                         # ties: the first maximum must win, as in ATen
    xr = x.double().requires_grad_(True)
    y = F.max_pool2d(xr, 2, 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.double())
    p = T.PoolT()
    ya = p.fwd(T.as_act(x.cuda()))
    assert torch.equal(ya.nchw().cpu(), y.detach().float())
    gx, _ = p.bwd(T.as_act(gy.cuda()), None, None)
    assert torch.equal(gx.nchw().cpu(), xr.grad.float())
    xd = torch.randn(x.shape, generator=g)
    yd = p.tan(T.as_act(xd.cuda()))
    ref_yd = torch.autograd.functional.jvp(lambda t: F.max_pool2d(t, 2, 2), (x.double(),), (xd.double(),))[1]
    assert torch.equal(yd.nchw().cpu(), ref_yd.float())

    for shape in [(2, 8, 5, 7), (1, 4, 1, 3), (1, 4, 16, 16)]:
        x = torch.randn(shape, generator=g)
        up = torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)
        xr = x.double().requires_grad_(True)
        y = up(xr)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy.double())
        u = T.UpT(up)
        ya = u.fwd(T.as_act(x.cuda()))
        torch.testing.assert_close(ya.nchw().cpu().double(), y.detach(), rtol=1e-5, atol=1e-6)
        gx, _ = u.bwd(T.as_act(gy.cuda()), None, None)
        torch.testing.assert_close(gx.nchw().cpu().double(), xr.grad, rtol=1e-5, atol=1e-6)
