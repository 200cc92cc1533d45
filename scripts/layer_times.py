"""Per-layer conv timing of one inference step (GPU box): HIP events around every conv launch, grouped by shape.
usage: layer_times.py [precision]"""
import os, sys, collections, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
from creste_public_amd import synth
from creste_public_amd.creste.utils.projection import lidar_depth_images
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
creste_public_amd.set_precision(prec)
dev = torch.device("cuda")
model = bench.build_model(dev)
B, H, W = (int(sys.argv[2]) if len(sys.argv) > 2 else 16), bench.IMG_H, bench.IMG_W
gen = torch.Generator().manual_seed(1337)
rgbd = torch.zeros(B, 1, 4, H, W, device=dev); rgbd[:, 0, :3] = torch.rand(B, 3, H, W, generator=gen).to(dev)
scan = synth.lidar_scan(B, gen).to(dev); l2c = synth.lidar2camrect(B, H, W).to(dev); p2p = synth.make_p2p(B, H, W).to(dev)
def step():
    with torch.no_grad():
        lidar_depth_images(scan, l2c, H, W, out=rgbd[:, 0, 3], scale=1000.0, depth_priority="max")
        return model((rgbd, p2p))
for _ in range(3): step()
prof = bench.ConvProfiler(); prof.install()
for _ in range(3): step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for e0, e1, fl, (pn, kn), shape in prof.records:
    d = agg.setdefault((shape, kn), [0.0, 0.0, 0]); d[0] += e0.elapsed_time(e1); d[1] += fl; d[2] += 1
tot = sum(v[0] for v in agg.values()) / 3
print(f"conv time per step {tot:.2f} ms ({prec})")
print(f"{'Cin':>5} {'Cout':>5} k {'H':>4} {'W':>4} {'n/step':>6} {'ms/step':>8} {'TF':>7}  kernel")
for (shape, kn), (ms, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{shape[0]:5d} {shape[1]:5d} {shape[2]} {shape[3]:4d} {shape[4]:4d} {n / 3:6.1f} {ms / 3:8.3f} {fl / ms / 1e9:7.1f}  {kn}")
