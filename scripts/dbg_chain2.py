import torch, sys, os, copy
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import train_ops as T
from creste_public_amd.config import maxent_irl_cfg
from creste_public_amd.creste.models.blocks.conv import MultiScaleFCN
cfg = maxent_irl_cfg()["traversability_head"]["net_kwargs"]["reward_cfg"]["net_kwargs"]
shape = (3, 40, 20, 36)
torch.manual_seed(shape[2])
net = MultiScaleFCN(cfg)
with torch.no_grad():
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
ref = copy.deepcopy(net).double().train()
net = net.cuda().train()
rel = lambda u, v: float((u - v).pow(2).mean().sqrt() / v.pow(2).mean().sqrt().clamp_min(1e-30))
B, _, H, W = shape
chain = T.Chain(T._ops_of(net.prepool))
flat = []
for m in ref.prepool:
    flat += list(m)
x = torch.rand(B, 40, H, W) * 2
xr = x.double().requires_grad_(True)
y = xr; outs = [xr]
for m in flat:
    y = torch.relu(y) if isinstance(m, torch.nn.ReLU) else m(y)
    y.retain_grad(); outs.append(y)
gy = torch.randn(y.shape)
y.backward(gy.double())
print([type(m).__name__ for m in flat], [type(o).__name__ for o in chain.ops])
ya = chain.fwd(T.as_act(x.cuda()))
g = T.as_act(gy.cuda())
# reference grads at: outs = [x, conv1, bn1, relu1, conv2, bn2, relu2]; chain ops = [conv, bn+relu, conv, bn+relu]
ref_at = {3: outs[4].grad, 2: outs[3].grad, 1: outs[1].grad, 0: outs[0].grad}
for i in range(len(chain.ops) - 1, -1, -1):
    g, _ = chain.ops[i].bwd(g, None, None)
    d = (g.nchw().cpu().double() - ref_at[i]).abs()
    print("   bad elems", int((d > 1e-4 * ref_at[i].abs().max()).sum()), "of", d.numel(), "worst at", (d == d.max()).nonzero()[0].tolist(), float(d.max()), float(ref_at[i].abs().max()))
    print("after bwd of op", i, type(chain.ops[i]).__name__, "rel err", rel(g.nchw().cpu().double(), ref_at[i]), "shape", tuple(g.nchw().shape))
