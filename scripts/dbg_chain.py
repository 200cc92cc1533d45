import torch, sys, os, copy
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import train_ops as T
from creste_public_amd.config import maxent_irl_cfg
from creste_public_amd.creste.models.blocks.conv import MultiScaleFCN
cfg = maxent_irl_cfg()["traversability_head"]["net_kwargs"]["reward_cfg"]["net_kwargs"]
shape = tuple(int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (3, 40, 20, 36)
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
for name, cin, hw in [("prepool", 40, (H, W)), ("skip", 32, (H, W)), ("trunk", 32, (H, W)), ("postpool", 48, (H, W))]:
    chain = T.Chain(T._ops_of(getattr(net, name)))
    mods = getattr(ref, name)
    x = torch.rand(B, cin, *hw) * 2
    xr = x.double().requires_grad_(True)
    # step through the reference op by op
    flat = []
    for m in mods:
        flat += list(m) if isinstance(m, torch.nn.Sequential) else [m]
    y = xr
    outs = []
    for m in flat:
        y = m(y) if not isinstance(m, torch.nn.ReLU) else torch.relu(y)
        outs.append(y)
    gy = torch.randn(y.shape)
    y.backward(gy.double())
    ya = chain.fwd(T.as_act(x.cuda()))
    gx, _ = chain.bwd(T.as_act(gy.cuda(), pad_to4=False) if gy.shape[1] % 4 == 0 else T.Act(T.as_act(gy.cuda(), pad_to4=True).buf, gy.shape[1], 0), None, None)
    print(name, "y", rel(ya.nchw().cpu().double(), y.detach()), "gx", rel(gx.nchw().cpu().double(), xr.grad))
