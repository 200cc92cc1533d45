import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
from creste_public_amd import synth
creste_public_amd.set_precision("f16x3")
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
rgbd, p2p = synth.make_frames(4, bench.IMG_H, bench.IMG_W, seed=1337)
rgbd, p2p = rgbd.to(dev), p2p.to(dev)
bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
mom = [m.momentum for m in bns]
for m in bns:
    m.momentum = 1.0
with torch.no_grad():
    model.backbone.train(); model.backbone((rgbd[:2], p2p[:2]))
    model.train(); model((rgbd[:2], p2p[:2]))
for m, mo in zip(bns, mom):
    m.momentum = mo
model.eval()
head = model.backbone.depthcomp.depthcomp.depth_head.model
bn = [m for m in head if isinstance(m, torch.nn.BatchNorm2d)][-1]
base = bn.weight.detach().clone()
for gain in (1.0, 4.0, 8.0, 16.0):
    with torch.no_grad():
        bn.weight.copy_(base * gain)
        out = model((rgbd, p2p))
    d = out["depth_preds_metric"]
    P = d[0].numel()
    print(f"gain x{gain}: depth mean {float(d.mean()):.2f} std {float(d.std()):.2f} min {float(d.min()):.2f} max {float(d.max()):.2f}; "
          f"tap mass/point {float(out['bev_densities'].sum()) / (4 * P):.3f}; occupied cells {float((out['bev_densities'] > 0).float().mean()):.3f}; "
          f"costmap mean {float(out['traversability_preds'].mean()):.3f} max {float(out['traversability_preds'].max()):.2f}; bev feat rms {float(out['bev_features'].pow(2).mean().sqrt()):.3f}")
