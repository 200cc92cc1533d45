"""Are back-to-back pipelined steps (no host synchronisation in between) reproducible?  Per output key."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
import creste_public_amd
from creste_public_amd import synth
from creste_public_amd.creste.utils.projection import lidar_depth_images
B, H, W = 16, bench.IMG_H, bench.IMG_W
device = torch.device("cuda", 0)
creste_public_amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x6")
model = bench.build_model(device)
gen = torch.Generator().manual_seed(1337)
rgbd = torch.zeros(B, 1, 4, H, W, device=device)
rgbd[:, 0, :3] = torch.rand(B, 3, H, W, generator=gen).to(device)
scan = synth.lidar_scan(B, gen).to(device)
l2c = synth.lidar2camrect(B, H, W).to(device)
p2p = synth.make_p2p(B, H, W).to(device)


def step():
    with torch.no_grad():
        lidar_depth_images(scan, l2c, H, W, out=rgbd[:, 0, 3], scale=1000.0, depth_priority="max")
        return model((rgbd, p2p))


for parts in (0, 2):
    model.inference_parts = parts
    step(); torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in step().items()}
    torch.cuda.synchronize()
    outs = [step() for _ in range(6)]          # no synchronisation between the steps
    torch.cuda.synchronize()
    bad = {}
    for i, o in enumerate(outs):
        for k, v in o.items():
            if not torch.equal(v, ref[k]):
                d = (v.double() - ref[k].double()).abs()
                rows = sorted(set(torch.nonzero(d.reshape(B, -1).amax(1) > 0).flatten().tolist()))
                bad.setdefault(k, []).append((i, float(d.max()), rows))
    print(f"parts {parts}: {'reproducible' if not bad else 'DIFFERS'}")
    for k, v in bad.items():
        print("   ", k, v[:3])
