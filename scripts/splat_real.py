"""The BEV splat gather on the BENCH's own data (the calibrated network's predicted depths and fused features at batch 16): plan and
features are captured from one forward, the gather is timed alone, with the per-cell list statistics its tail depends on.
usage: splat_real.py [precision]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
from creste_public_amd import ops, synth
creste_public_amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x6")
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
B = 16
gen = torch.Generator().manual_seed(1337)
rgbd = torch.zeros(B, 1, 4, bench.IMG_H, bench.IMG_W, device=dev)
rgbd[:, 0, :3] = torch.rand(B, 3, bench.IMG_H, bench.IMG_W, generator=gen).to(dev)
scan = synth.lidar_scan(B, gen).to(dev); l2c = synth.lidar2camrect(B, bench.IMG_H, bench.IMG_W).to(dev)
p2p = synth.make_p2p(B, bench.IMG_H, bench.IMG_W).to(dev)
from creste_public_amd.creste.utils.projection import lidar_depth_images
lidar_depth_images(scan, l2c, bench.IMG_H, bench.IMG_W, out=rgbd[:, 0, 3], scale=1000.0, depth_priority="max")
cap = {}
opk, og = ops.bev_splat_plan_keyed, ops.bev_splat_gather
def spy_plan(*a, **k):
    cap["plan"] = opk(*a, **k); return cap["plan"]
def spy_gather(plan, feats, min_weight=1.0, scatter_mode="mean"):
    cap["gather"] = (feats, min_weight, scatter_mode)
    return og(plan, feats, min_weight, scatter_mode)
ops.bev_splat_plan_keyed, ops.bev_splat_gather = spy_plan, spy_gather
model.inference_parts = 0
with torch.no_grad():
    model((rgbd, p2p))
ops.bev_splat_plan_keyed, ops.bev_splat_gather = opk, og
plan, (feats, mw, mode) = cap["plan"], cap["gather"]
G = plan.GH
for _ in range(3):
    bev, dens = og(plan, feats, mw, mode)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    og(plan, feats, mw, mode)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
coords = plan.coords
cell = coords.floor().long()
ok = (cell >= -1).all(-1) & (cell[..., 0] <= G - 1) & (cell[..., 1] <= G - 1)
key = ((cell[..., 1] + 1) * (G + 1) + cell[..., 0] + 1 + torch.arange(B, device="cuda").view(B, 1) * (G + 1) ** 2)[ok]
cnt = torch.bincount(key, minlength=B * (G + 1) ** 2).view(B, G + 1, G + 1)
# entries a BEV cell (Y, X) visits = the lists of its four base cells
T = cnt[:, 1:, 1:] + cnt[:, :-1, 1:] + cnt[:, 1:, :-1] + cnt[:, :-1, :-1]
occ = T[T > 0].float()
print(f"{int(ok.sum())} of {B * plan.P} points own in-grid taps; BEV cells with entries {100 * occ.numel() / T.numel():.1f} %; entries per such cell: "
      f"mean {occ.mean():.1f} p99 {occ.quantile(0.99):.0f} p99.9 {occ.quantile(0.999):.0f} max {int(occ.max())}")
rows = T.sum(-1).float()
print(f"entries per BEV row (= workgroup): mean {rows.mean():.0f} p99 {rows.flatten().quantile(0.99):.0f} max {int(rows.max())}; "
      f"per wave of a row (64 cells round-robin): max {int(T.view(B, G, 8, 32).sum(2).max())} per lane group")
print(f"gather alone, warm caches: {us:.1f} us")
