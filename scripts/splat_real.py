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
op, og = ops.bev_splat_plan, ops.bev_splat_gather
def spy_plan(xyz, off, vox, GH, GW):
    cap["plan"] = (xyz, off, vox, GH, GW)
    return op(xyz, off, vox, GH, GW)
def spy_gather(plan, feats, min_weight=1.0, scatter_mode="mean"):
    cap["gather"] = (feats, min_weight, scatter_mode)
    return og(plan, feats, min_weight, scatter_mode)
ops.bev_splat_plan, ops.bev_splat_gather = spy_plan, spy_gather
model.inference_parts = 0
with torch.no_grad():
    model((rgbd, p2p))
ops.bev_splat_plan, ops.bev_splat_gather = op, og
a = (cap["plan"][0], cap["gather"][0], cap["plan"][1], cap["plan"][2], cap["plan"][3], cap["plan"][4], cap["gather"][1], cap["gather"][2])
k = {}
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

# ---- why is the gather slower inside the step (143 us) than alone (~120)?  Three conditions, gather only (one plan):
#   clean    : the features were written long ago (they come from HBM, nothing of theirs is dirty in the caches)
#   rewritten: a copy kernel has just REWRITTEN the feature buffer (as the fusion conv does in the step): its dirty lines
#              sit in the L2 / infinity cache and are written back while the gather streams its 403 MB output over them
#   after-mfma: a matrix-bound conv on OTHER memory ran just before (clock / power state), features clean
plan = ops.bev_splat_plan(xyz, a[2], a[3], a[4], a[5])
fbuf2 = feats.buf.clone()
wgt = torch.randn(256, 256, 3, 3, device=dev) / 48
pcw = ops.pack_conv(wgt, None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
big = ops.Act(torch.randn(8, 152, 304, 256, device=dev), 256)
mode = k.get("scatter_mode", a[7] if len(a) > 7 else "mean")
minw = k.get("min_weight", a[6] if len(a) > 6 else 1.0)


def timed(prep):
    ts = []
    for _ in range(12):
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.bev_splat_gather(plan, feats, minw, mode)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def clean():
    torch.cuda.synchronize()
    junk = torch.empty(1 << 28, device=dev).fill_(1.0)        # 1 GiB streamed through the caches: nothing of the features stays
    torch.cuda.synchronize()


def rewritten():
    clean()
    feats.buf.copy_(fbuf2)


def after_mfma():
    clean()
    ops.conv2d(big, pcw)


print(f"gather alone: features clean {timed(clean):.1f} us | just rewritten {timed(rewritten):.1f} us | after a matrix-bound conv {timed(after_mfma):.1f} us")

