"""Whole-step experiment behind scripts/wino_overlap_micro.py: the bench's inference step on a batch of 16 (a) as one
forward on one stream, (b) as PARTS forwards of 16 / PARTS frames, each on its own stream, issued one after the other by
one host thread (part k+1's launches trail part k's by the host's issue time, so the parts run out of phase and the
bandwidth-bound kernels of one can fill the CUs the other's persistent GEMM leaves free).
usage: pipeline_micro.py [steps] [parts]"""
import os, sys, time, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
import creste_public_amd
from creste_public_amd import synth
from creste_public_amd.creste.utils.projection import lidar_depth_images

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B, H, W = 16, bench.IMG_H, bench.IMG_W
device = torch.device("cuda", 0)
creste_public_amd.set_precision("bf16x6")
model = bench.build_model(device)
gen = torch.Generator().manual_seed(1337)
rgbd = torch.zeros(B, 1, 4, H, W, device=device)
rgbd[:, 0, :3] = torch.rand(B, 3, H, W, generator=gen).to(device)
scan = synth.lidar_scan(B, gen).to(device)
l2c = synth.lidar2camrect(B, H, W).to(device)
p2p = synth.make_p2p(B, H, W).to(device)
streams = [torch.cuda.Stream(device=device) for _ in range(parts)]
h = B // parts


def whole():
    with torch.no_grad():
        lidar_depth_images(scan, l2c, H, W, out=rgbd[:, 0, 3], scale=1000.0, depth_priority="max")
        return model((rgbd, p2p))


def split():
    main = torch.cuda.current_stream()
    outs = []
    with torch.no_grad():
        for i, st in enumerate(streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                sl = slice(i * h, (i + 1) * h)
                lidar_depth_images(scan[sl], l2c[sl], H, W, out=rgbd[sl, 0, 3], scale=1000.0, depth_priority="max")
                outs.append(model((rgbd[sl], p2p[sl])))
        for st in streams:
            main.wait_stream(st)
        return {k: torch.cat([o[k] for o in outs]) for k in ("traversability_preds",)}


def timeit(fn):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


t, ref = timeit(whole)
print(f"whole batch, one stream: {t:.3f} ms / step")
for wgs in (0, 28, 24):
    for chain in (0, 1):
        os.environ["CRESTE_W4_GEMM_WGS"] = str(wgs)
        os.environ["CRESTE_W4_CHAIN"] = str(chain)
        t, out = timeit(split)
        same = torch.equal(out["traversability_preds"], ref["traversability_preds"])
        print(f"{parts} parts on {parts} streams, GEMM workgroups / XCD {wgs or 32}, chain {chain}: {t:.3f} ms / step  costmap bit-identical {same}")
os.environ["CRESTE_W4_CHAIN"] = "0"
os.environ["CRESTE_W4_GEMM_WGS"] = "28"
t, _ = timeit(whole)
print(f"whole batch, one stream, GEMM workgroups / XCD 28: {t:.3f} ms / step")
