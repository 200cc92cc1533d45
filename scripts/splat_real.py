"""The BEV splat on the BENCH's own data (the calibrated network's predicted depths and fused features at batch 16): the
call is captured from one forward and timed alone, with the cell-list statistics the gather's tail depends on.
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
orig = ops.bev_splat
def spy(*a, **k):
    cap["a"], cap["k"] = a, k
    return orig(*a, **k)
ops.bev_splat = spy
with torch.no_grad():
    model((rgbd, p2p))
ops.bev_splat = orig
a, k = cap["a"], cap["k"]
xyz, feats = a[0], a[1]
G = a[4]
for _ in range(3):
    coords, bev, dens = orig(*a, **k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = orig(*a, **k)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
P, F = xyz.shape[1], feats.C
alg = 4.0 * (F * P + 2 * P + F * G * G + G * G) * B
cell = coords.floor().long()
ok = (cell >= -1).all(-1) & (cell[..., 0] <= G - 1) & (cell[..., 1] <= G - 1)
key = ((cell[..., 1] + 1) * (G + 1) + cell[..., 0] + 1 + torch.arange(B, device="cuda").view(B, 1) * (G + 1) ** 2)[ok]
cnt = torch.bincount(key, minlength=B * (G + 1) ** 2).view(B, G + 1, G + 1)
occ = cnt[cnt > 0].float()
rows = cnt.sum(-1).float()       # entries per extended row
print(f"real: {int(ok.sum())} of {B * P} points own in-grid taps; occupied base cells {occ.numel()}, points per occupied cell "
      f"mean {occ.mean():.1f} p99 {occ.quantile(0.99):.0f} max {int(occ.max())}; per extended row: mean {rows.mean():.0f} "
      f"p99 {rows.flatten().quantile(0.99):.0f} max {int(rows.max())}; occupied BEV cells {100 * float((dens > 0).float().mean()):.1f} %")
hist = torch.bincount(cnt.flatten().clamp(max=256))
print("cells by list length: " + ", ".join(f"{lo}-{hi}: {int(hist[lo:hi + 1].sum())}" for lo, hi in ((1, 4), (5, 16), (17, 64), (65, 128), (129, 256))))
print(f"bev_splat B={B} P={P} F={F}: {ms * 1e3:.1f} us  algorithmic {alg / 1e6:.1f} MB -> {alg / ms / 1e9:.2f} TB/s = {alg / ms / 1e9 / 8 * 100:.1f}% of 8 TB/s")
torch.save({"xyz": xyz.cpu(), "depth_note": "bench bev_splat inputs"}, "gpurun_out/splat_real_xyz.pt") if os.environ.get("SAVE_XYZ") else None
