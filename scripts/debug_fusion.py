import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT)
import creste_public_amd
from creste_public_amd import MaxEntIRL, synth, ops
from creste_public_amd.config import maxent_irl_cfg
H, W, B = 608, 1216, 2
torch.manual_seed(0)
m = MaxEntIRL(maxent_irl_cfg((H, W), solve_mdp=False))
synth.randomize_bn(m, seed=1)
m = m.cuda().eval()
rgbd, p2p = synth.make_frames(B, H, W, seed=77)
rgbd, p2p = rgbd.cuda(), p2p.cuda()
synth.calibrate_bn_hip(m, rgbd, p2p)
tn = m.backbone
res = {}
for mode in ("f32", "bf16x6", "f16x3"):
    creste_public_amd.set_precision(mode)
    with torch.no_grad():
        r = tn.forward_act(rgbd, p2p)
    fused = r["fused"]
    fb = None
    res[mode] = dict(fused=fused.nchw().double().clone(), bev=r["bev"].nchw().double().clone(), mask=r["mask"].clone(),
                     feats=r["feats"].nchw().double().clone())
    # fp64 re-evaluation of the fusion conv from THIS run's own inputs
    conv, bn = tn.cam2map.vision_fusion.convs[0], tn.cam2map.vision_fusion.convs[1]
    # rebuild the 288-ch input: feats (256) + z feats (32) live in one buffer; find it through r
    xin = r["fused_in"].nchw().double() if "fused_in" in r else None
    res[mode]["xin"] = xin
    if xin is not None:
        y = torch.nn.functional.conv2d(xin, conv.weight.double(), conv.bias.double())
        y = (y - bn.running_mean.double().view(1, -1, 1, 1)) / torch.sqrt(bn.running_var.double().view(1, -1, 1, 1) + bn.eps) * bn.weight.double().view(1, -1, 1, 1) + bn.bias.double().view(1, -1, 1, 1)
        y = torch.relu(y) * r["mask"].view(B, 1, H // 4, W // 4).double()
        e = (fused.nchw().double() - y)
        print(mode, "fusion conv vs fp64 of own inputs: rel rms", float(e.pow(2).mean().sqrt() / y.pow(2).mean().sqrt()), "max", float(e.abs().max()), "ymax", float(y.abs().max()), "xin absmax", float(xin.abs().max()), "z absmax", float(xin[:, 256:].abs().max()))
for mode in ("bf16x6", "f16x3"):
    for k in ("feats", "fused", "bev"):
        a, b = res[mode][k], res["f32"][k]
        print(mode, k, "vs f32 rel rms", float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()))
